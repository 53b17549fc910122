"""The multi-GPU orchestration (fastqdedup_amd/sharded.py) on 2 and 3 CPU ranks
over gloo, both plans: geometry agreement, all-to-all by owner, segment-routed search
passes / all-gather of the unique table, edge all-gather, clusters dissected away from
their owners, verdicts and kept ids sent home, global read ids. The arithmetic is a numpy stand-in
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

    @staticmethod
    def _segment(key, s, nseg):
        return key[s * len(key) // nseg:(s + 1) * len(key) // nseg]

    def _seg_owner(self, key, s, nseg, n_parts):
        return zlib.crc32(bytes([len(key) & 255, s]) + self._segment(key, s, nseg)) % n_parts

    def _first_agreeing(self, a, b, nseg):
        if len(a) != len(b):
            return None
        for s in range(nseg):
            if self._segment(a, s, nseg) == self._segment(b, s, nseg):
                return s
        return None

    def _rows_of(self, ks):
        recs = np.zeros((len(ks), self.stride * 4), dtype=np.uint8)
        for row, k in enumerate(ks):
            recs[row, :len(k)] = np.frombuffer(k, dtype=np.uint8)
        lens = torch.tensor([len(k) for k in ks], dtype=torch.int32)
        return (torch.from_numpy(recs.view(np.int32).reshape(len(ks), self.stride).copy()),
                lens if self.ragged else None)

    def pack_by_owner(self, keys, offsets, key_len, n_parts, id0, weights, n_segments=0):
        ks = self._split(keys, offsets, key_len)
        if n_segments:
            self.max_distance = n_segments - 1
            owner = np.array([self._seg_owner(k, 0, n_segments, n_parts) for k in ks], dtype=np.int64)
        else:
            owner = np.array([zlib.crc32(k) % n_parts for k in ks], dtype=np.int64)
        order = np.argsort(owner, kind="stable")
        recs, lens = self._rows_of([ks[i] for i in order])
        ids = torch.from_numpy((id0 + order).astype(np.int64))
        w = None if weights is None else torch.from_numpy(np.asarray(weights, dtype=np.int32)[order].copy())
        counts = np.bincount(owner, minlength=n_parts).tolist()
        return recs, lens, ids, w, counts

    # -- segment-routed plan ---------------------------------------------------
    def collapse_resident(self, recs, lens, weights, read_ids, seg_rows=None, seg_id0=None, id_limit=None):
        urecs, ulens, ucounts, ufirst = self.collapse_packed(recs, lens, weights, read_ids)
        self.table = self._decode(urecs, ulens)
        self.table_counts, self.table_first = ucounts.tolist(), ufirst.tolist()
        return len(self.table)

    def _pass_edges(self, ks, d, s):
        out = []
        for u in range(len(ks)):
            for v in range(u + 1, len(ks)):
                if self._first_agreeing(ks[u], ks[v], d + 1) == s and self.O.within_distance(
                        ks[u].decode("latin-1"), ks[v].decode("latin-1"), d, False):
                    out.append((u, v))
        return torch.tensor(out, dtype=torch.int32).reshape(-1, 2)

    def local_edges(self, d, seg_lo, seg_hi):
        parts = [self._pass_edges(self.table, d, s) for s in range(seg_lo, seg_hi)]
        return torch.cat(parts) if parts else torch.empty((0, 2), dtype=torch.int32)

    def unique_by_segment(self, n_parts, n_segments, segment, uid_base):
        owner = np.array([self._seg_owner(k, segment, n_segments, n_parts) for k in self.table], dtype=np.int64)
        order = np.argsort(owner, kind="stable")
        recs, lens = self._rows_of([self.table[i] for i in order])
        uids = torch.from_numpy((uid_base + order).astype(np.int32))
        return recs, lens, uids, np.bincount(owner, minlength=n_parts).tolist()

    def routed_edges(self, recs, lens, uids, d, segment):
        e = self._pass_edges(self._decode(recs, lens), d, segment)
        return uids[e.long()] if e.shape[0] else e

    def edge_labels(self, edges, n_nodes):
        parent = list(range(n_nodes))

        def find(x):
            while parent[x] != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x
        hooks = 0
        for u, v in edges.tolist():
            ru, rv = find(u), find(v)
            if ru != rv:
                parent[max(ru, rv)] = min(ru, rv)
                hooks += 1
        return torch.tensor([find(u) for u, _ in edges.tolist()], dtype=torch.int32), n_nodes - hooks

    def local_labels(self, edges):
        """Union-find over this rank's table and edges between its rows: (smallest row of every row's component, components)."""
        n = len(self.table)
        parent = list(range(n))

        def find(x):
            while parent[x] != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x
        hooks = 0
        for u, v in edges.tolist():
            ru, rv = find(u), find(v)
            if ru != rv:
                parent[max(ru, rv)] = min(ru, rv)
                hooks += 1
        return torch.tensor([find(i) for i in range(n)], dtype=torch.int32), n - hooks

    def gather_unique(self, rows):
        sel = rows.tolist()
        recs, lens = self._rows_of([self.table[i] for i in sel])
        return recs, lens, torch.tensor([self.table_counts[i] for i in sel], dtype=torch.int32)

    def dissect_subgraph(self, recs, lens, counts, edges, method):
        self.keys = [k.decode("latin-1") for k in self._decode(recs, lens)]
        self.counts, self.first = counts.tolist(), list(range(len(self.keys)))
        self.d, self.edit = self.max_distance, False
        kept, _, _ = self.finish(edges, method, 0, len(self.keys))
        flags = torch.zeros(len(self.keys), dtype=torch.uint8)
        flags[kept] = 1
        return flags

    def finish_owner(self, dropped_rows, id_hi):
        gone = set(dropped_rows.tolist())
        kept = sorted(f for i, f in enumerate(self.table_first) if i not in gone)
        return torch.tensor(kept, dtype=torch.int64), len(kept)

    def finish_owner_home(self, home_edges, method, dropped_rows, id_hi):
        """Home clusters (edges over rows of this rank's table) dissected in place; rows dropped elsewhere go too."""
        self.keys = [k.decode("latin-1") for k in self.table]
        self.counts, self.first = list(self.table_counts), list(range(len(self.table)))
        self.d, self.edit = self.max_distance, False
        kept_rows, self.home_components, _ = self.finish(home_edges, method, 0, len(self.table))
        gone = set(dropped_rows.tolist())
        assert not (gone & set(home_edges.reshape(-1).tolist())), "a dropped key has an edge at home"
        kept = sorted(self.table_first[i] for i in kept_rows.tolist() if i not in gone)
        return torch.tensor(kept, dtype=torch.int64), len(kept)

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


def _worker(rank, world, port, shards, d, method, weights, plan, q):
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
        res = cluster_keys_sharded(NumpyBackend(O), raw, off, 0, w, max_distance=d, method=method, plan=plan)
        assert res.plan == plan
        q.put((rank, res.kept_read_ids.tolist(), res.n_clusters, res.n_unique, res.n_reads, res.n_kept))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("plan", ["segment-routed", "gathered"])
@pytest.mark.parametrize("world,d,method,ragged", [(2, 1, "directional", True), (2, 2, "adjacency", False),
                                                   (3, 1, "highest_count", True)])
@pytest.mark.parametrize("home", [True, False])
def test_sharded_job_equals_single_job(oracle, monkeypatch, world, d, method, ragged, plan, home):
    """(home: clusters that live on one rank are dissected there in place -- sharded.py step 4; False: every cluster
    is dealt out by its root and its keys fetched, the way before)"""
    from fastqdedup_amd.synth import synth_keys
    if not home:
        if plan == "gathered":
            pytest.skip("the gathered plan has no home clusters")
        monkeypatch.setenv("FQD_NO_HOME_CLUSTERS", "1")      # (inherited by the spawned ranks)
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
    procs = [ctx.Process(target=_worker, args=(r, world, port, shards, d, method, wshards, plan, q))
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
