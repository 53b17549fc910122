"""The multi-GPU orchestration (fastqdedup_amd/sharded.py) on 2 and 3 CPU ranks
over gloo: geometry agreement, all-to-all by key owner, all-gather of the unique
table and of the edge shards, global read ids. The arithmetic is a numpy stand-in
(tests only; production uses HipBackend); the answer is checked against the CPU
oracle run on the concatenation of every rank's reads."""
import os
import socket
import zlib

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class NumpyBackend:
    """Same interface as fastqdedup_amd.sharded.HipBackend, on CPU tensors."""

    def __init__(self, oracle):
        self.O = oracle
        self.device = torch.device("cpu")

    # -- helpers -----------------------------------------------------------
    @staticmethod
    def _split(keys, offsets, key_len):
        keys = np.asarray(keys, dtype=np.uint8)
        if offsets is None:
            n = keys.size // key_len if key_len else 0
            return [bytes(keys[i * key_len:(i + 1) * key_len]) for i in range(n)]
        off = np.asarray(offsets, dtype=np.uint64)
        return [bytes(keys[int(off[i]):int(off[i + 1])]) for i in range(len(off) - 1)]

    def _decode(self, recs, lens):
        if lens is None:        # not ragged: the orchestration does not move lengths
            lens = [self.max_len] * recs.shape[0]
        raw = recs.numpy().view(np.uint8).reshape(recs.shape[0], -1)
        return [bytes(raw[i, :int(lens[i])]) for i in range(recs.shape[0])]

    # -- interface -----------------------------------------------------------
    def scan(self, keys, offsets, key_len):
        ks = self._split(keys, offsets, key_len)
        present = np.zeros(128, dtype=np.uint8)
        for k in ks:
            present[list(set(k))] = 1
        lens = [len(k) for k in ks]
        return present, (max(lens) if lens else 0), (len(set(lens)) > 1)

    def configure(self, present, max_len, ragged):
        self.ragged = bool(ragged)
        self.max_len = int(max_len)
        self.stride = max(1, (self.max_len + 3) // 4)

    def pack_by_owner(self, keys, offsets, key_len, n_parts, id0, weights):
        ks = self._split(keys, offsets, key_len)
        owner = np.array([zlib.crc32(k) % n_parts for k in ks], dtype=np.int64)
        order = np.argsort(owner, kind="stable")
        recs = np.zeros((len(ks), self.stride * 4), dtype=np.uint8)
        for row, i in enumerate(order):
            recs[row, :len(ks[i])] = np.frombuffer(ks[i], dtype=np.uint8)
        lens = np.array([len(ks[i]) for i in order], dtype=np.int32)
        ids = torch.from_numpy((id0 + order).astype(np.int64))
        w = None if weights is None else torch.from_numpy(np.asarray(weights, dtype=np.int32)[order].copy())
        counts = np.bincount(owner, minlength=n_parts).tolist()
        return (torch.from_numpy(recs.view(np.int32).reshape(len(ks), self.stride).copy()),
                torch.from_numpy(lens) if self.ragged else None, ids, w, counts)

    def collapse_packed(self, recs, lens, weights, read_ids):
        ks = self._decode(recs, lens)
        first, count = {}, {}
        wl = [1] * len(ks) if weights is None else weights.tolist()
        for k, w, i in zip(ks, wl, read_ids.tolist()):
            first[k] = min(first.get(k, i), i)
            count[k] = count.get(k, 0) + w
        uniq = [k for k in first if count[k] > 0]
        idx = {k: j for j, k in enumerate(ks)}
        rows = [idx[k] for k in uniq]
        sel = torch.tensor(rows, dtype=torch.long)
        return (recs[sel].contiguous(), None if lens is None else lens[sel].contiguous(),
                torch.tensor([count[k] for k in uniq], dtype=torch.int32),
                torch.tensor([first[k] for k in uniq], dtype=torch.int64))

    def find_edges(self, urecs, ulens, ucounts, ufirst, max_distance, metric, shard, n_shards):
        self.keys = [k.decode("latin-1") for k in self._decode(urecs, ulens)]
        self.counts, self.first = ucounts.tolist(), ufirst.tolist()
        edges = []
        for u in range(len(self.keys)):
            for v in range(u + 1, len(self.keys)):
                if (u + v) % n_shards == shard and self.O.within_distance(
                        self.keys[u], self.keys[v], max_distance, bool(metric)):
                    edges.append((u, v))
        self.d, self.edit = max_distance, bool(metric)
        return torch.tensor(edges, dtype=torch.int32).reshape(-1, 2)

    def finish(self, edges, method, id_lo, id_hi):
        parent = list(range(len(self.keys)))

        def find(x):
            while parent[x] != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x
        for u, v in edges.tolist():
            ru, rv = find(u), find(v)
            if ru != rv:
                parent[max(ru, rv)] = min(ru, rv)
        comps = {}
        for i in range(len(self.keys)):
            comps.setdefault(find(i), []).append(i)
        name = {0: "highest_count", 1: "adjacency", 2: "directional"}[method]
        fn = self.O.CLUSTER_DISSECTION_METHODS[name]
        first_of = {k: f for k, f in zip(self.keys, self.first)}
        kept = []
        for members in comps.values():
            cluster = [(self.counts[i], self.keys[i]) for i in members]
            kept.extend(first_of[k] for k in fn(cluster, self.d, self.edit))
        mine = sorted(i for i in kept if id_lo <= i < id_hi)
        return torch.tensor(mine, dtype=torch.int64), len(comps), len(kept)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, shards, d, method, weights, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fastqdedup_amd.sharded import cluster_keys_sharded
        from oracle import oracle as O
        keys = shards[rank]
        raw = np.frombuffer(b"".join(keys) or b"", dtype=np.uint8)
        off = np.concatenate([[0], np.cumsum([len(k) for k in keys])]).astype(np.uint64)
        w = None if weights is None else np.asarray(weights[rank], dtype=np.int32)
        res = cluster_keys_sharded(NumpyBackend(O), raw, off, 0, w, max_distance=d, method=method)
        q.put((rank, res.kept_read_ids.tolist(), res.n_clusters, res.n_unique, res.n_reads, res.n_kept))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,d,method,ragged", [(2, 1, "directional", True), (2, 2, "adjacency", False),
                                                   (3, 1, "highest_count", True)])
def test_sharded_job_equals_single_job(oracle, world, d, method, ragged):
    from fastqdedup_amd.synth import synth_keys
    n, L = 240, 12
    allk = [bytes(r) for r in synth_keys(n, L, 4, 5 + world, sub_rate=0.02, n_rate=0.01)]
    if ragged:
        allk[7] = allk[7][:-2]        # ragged on one rank only
        allk[200] = b"ACGTacgt"       # a symbol set the other ranks do not have
    else:
        allk[200] = b"ACGTacgtACGT"
    rng = np.random.default_rng(world)
    weights = rng.integers(0, 3, size=n).tolist()
    cuts = [0, 90, 240] if world == 2 else [0, 90, 90, 240]     # rank 1 of 3 is empty
    shards = [allk[cuts[r]:cuts[r + 1]] for r in range(world)]
    wshards = [weights[cuts[r]:cuts[r + 1]] for r in range(world)]

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shards, d, method, wshards, q))
             for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    raw = np.frombuffer(b"".join(allk), dtype=np.uint8)
    off = np.concatenate([[0], np.cumsum([len(k) for k in allk])]).astype(np.uint64)
    want = oracle.dedup(raw, off, np.array(weights, dtype=np.uint32), max_distance=d, method=method)
    want_ids = want["kept_read_ids"].tolist()
    union = []
    for rank, kept, n_clusters, n_unique, n_reads, n_kept in got:
        lo, hi = cuts[rank], cuts[rank + 1]
        assert kept == [i for i in want_ids if lo <= i < hi], rank      # each rank lists its own reads
        union += kept
        assert n_kept == len(want_ids)
        assert n_clusters == want["n_clusters"]
        assert n_unique == want["n_unique"]
        assert n_reads == n
    assert sorted(union) == want_ids
