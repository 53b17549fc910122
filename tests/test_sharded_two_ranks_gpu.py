"""Two ranks of the production multi-GPU path on ONE MI355X: both processes run the
arithmetic on GPU 0 through HipBackend (every C-ABI export/import, owner grouping with 2
parts, 2-shard bucket search, id windows) while the collectives go over gloo on host copies
(RCCL refuses two ranks on one device). Checked against the CPU oracle. GPU only."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fastqdedup_amd as F
        from fastqdedup_amd.sharded import HipBackend, cluster_keys_sharded

        gpu = torch.device("cuda", 0)

        class HostComm(HipBackend):
            """HipBackend whose results are handed to the orchestrator as host tensors."""

            def __init__(self, ctx):
                super().__init__(ctx, gpu)
                self.device = torch.device("cpu")

            @staticmethod
            def _up(x):
                return None if x is None else (x.to(gpu) if torch.is_tensor(x) else x)

            @staticmethod
            def _down(xs):
                return tuple(x.cpu() if torch.is_tensor(x) else x for x in xs)

            def pack_by_owner(self, keys, offsets, key_len, n_parts, id0, weights):
                self.device = gpu
                try:
                    out = super().pack_by_owner(self._up(keys), self._up(offsets), key_len, n_parts, id0,
                                                self._up(weights))
                finally:
                    self.device = torch.device("cpu")
                return self._down(out)

            def scan(self, keys, offsets, key_len):
                return super().scan(self._up(keys), self._up(offsets), key_len)

            def collapse_packed(self, recs, lens, weights, read_ids):
                self.device = gpu
                try:
                    out = super().collapse_packed(self._up(recs), self._up(lens), self._up(weights),
                                                  self._up(read_ids))
                finally:
                    self.device = torch.device("cpu")
                return self._down(out)

            def find_edges(self, urecs, ulens, ucounts, ufirst, d, metric, shard, n_shards):
                self.device = gpu
                try:
                    out = super().find_edges(self._up(urecs), self._up(ulens), self._up(ucounts),
                                             self._up(ufirst), d, metric, shard, n_shards)
                finally:
                    self.device = torch.device("cpu")
                return out.cpu()

            def finish(self, edges, method, id_lo, id_hi):
                self.device = gpu
                try:
                    kept, ncl, nk = super().finish(self._up(edges), method, id_lo, id_hi)
                finally:
                    self.device = torch.device("cpu")
                return kept.cpu(), ncl, nk

        keys, offsets, key_len, weights, d, edit, method = case[rank]
        k = torch.from_numpy(keys)
        o = None if offsets is None else torch.from_numpy(offsets.astype(np.int64))
        w = None if weights is None else torch.from_numpy(weights.astype(np.int32))
        res = cluster_keys_sharded(HostComm(F.Context(0)), k, o, key_len, w, max_distance=d,
                                   use_edit_distance=edit, method=method)
        q.put((rank, res.kept_read_ids.tolist(), res.n_clusters, res.n_unique, res.n_kept, res.n_reads))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape", ["fixed32", "fixed100_weights", "ragged_edit"])
def test_two_ranks_on_one_gpu(oracle, shape):
    from fastqdedup_amd.synth import fixed_offsets, synth_keys
    world = 2
    if shape == "fixed32":
        n, L, d, edit, method = 120_000, 32, 1, False, "directional"
        allk = synth_keys(n, L, L, 5, sub_rate=3e-3, n_rate=3e-4)
        cut = 70_000
        case = [(allk[:cut].reshape(-1), None, L, None, d, edit, method),
                (allk[cut:].reshape(-1), None, L, None, d, edit, method)]
        raw, off, w = allk.reshape(-1), fixed_offsets(n, L), None
        cuts = [0, cut, n]
    elif shape == "fixed100_weights":
        n, L, d, edit, method = 60_000, 100, 2, False, "adjacency"
        allk = synth_keys(n, L, 12, 6, sub_rate=3e-3, n_rate=3e-4)
        w = np.random.default_rng(1).choice(np.array([0, 1, 1, 2], dtype=np.uint32), size=n)
        cut = 25_000
        case = [(allk[:cut].reshape(-1), None, L, w[:cut], d, edit, method),
                (allk[cut:].reshape(-1), None, L, w[cut:], d, edit, method)]
        raw, off = allk.reshape(-1), fixed_offsets(n, L)
        cuts = [0, cut, n]
    else:
        import random
        rng = random.Random(4)
        mols = ["".join(rng.choice("ACGT") for _ in range(rng.randint(16, 26))) for _ in range(300)]
        strs = []
        for _ in range(4000):
            s = list(rng.choice(mols))
            if rng.random() < 0.4:
                pos = rng.randrange(len(s))
                s[pos:pos + 1] = rng.choice([[], [rng.choice("ACGT")], [s[pos], rng.choice("ACGTN")]])
            strs.append("".join(s))
        d, edit, method, w = 1, True, "directional", None
        enc = [s.encode() for s in strs]
        cut = 1500

        def pk(part):
            r = np.frombuffer(b"".join(part), dtype=np.uint8).copy()
            o = np.concatenate([[0], np.cumsum([len(e) for e in part])]).astype(np.uint64)
            return r, o
        r0, o0 = pk(enc[:cut])
        r1, o1 = pk(enc[cut:])
        case = [(r0, o0, 0, None, d, edit, method), (r1, o1, 0, None, d, edit, method)]
        raw, off = pk(enc)
        cuts = [0, cut, len(enc)]
        n = len(enc)

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = oracle.dedup(raw, off, w, max_distance=d, use_edit_distance=edit, method=method)
    want_ids = want["kept_read_ids"].tolist()
    union = []
    for rank, kept, n_clusters, n_unique, n_kept, n_reads in got:
        lo, hi = cuts[rank], cuts[rank + 1]
        assert kept == [i for i in want_ids if lo <= i < hi], (shape, rank)
        union += kept
        assert (n_clusters, n_unique, n_kept, n_reads) == (want["n_clusters"], want["n_unique"],
                                                           len(want_ids), n)
    assert sorted(union) == want_ids
