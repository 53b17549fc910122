"""Parity at the sizes bench.py times (BASELINE.json configs 2 and 3 in full): the HIP path with
its DEFAULT switches -- so the fused pack -> collapse route, the 8192 level-1 slabs, the 2^16
level-2 cursors and the id-bin kept list engage on their own -- against the CPU oracle on the very
same bytes. The oracle needs about as long as the reference (~2 min for
50 M x 32 nt), so both oracle runs start in threads (ctypes drops the GIL) before any GPU work.
GPU only."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# (name, reads, key length, umi, seed, d, method): bench.py's WORKLOADS, unchanged
CONFIGS = {
    "config3": (50_000_000, 32, 32, 1003, 1, "directional"),
    "config2": (10_000_000, 100, 12, 1002, 1, "directional"),
}


class _OracleRun:
    def __init__(self, oracle, host, n, L, d, method):
        from fastqdedup_amd.synth import fixed_offsets
        self.out, self.err = None, None

        def work():
            try:
                self.out = oracle.dedup(host, fixed_offsets(n, L), max_distance=d, method=method)
            except BaseException as exc:  # surfaced by result()
                self.err = exc
        self.thread = threading.Thread(target=work, daemon=True)
        self.thread.start()

    def result(self):
        self.thread.join()
        if self.err is not None:
            raise self.err
        return self.out


@pytest.fixture(scope="module")
def full_size(oracle):
    """Device keys of both configurations + their oracle answers, running in the background."""
    import torch
    import fastqdedup_amd as F
    ctx = F.Context(0)
    jobs = {}
    for name, (n, L, umi, seed, d, method) in CONFIGS.items():
        dev = torch.empty(n * L, dtype=torch.uint8, device="cuda:0")
        ctx.synth_keys(dev, n, 0, n, L, umi, seed)
        host = dev.cpu().numpy()
        jobs[name] = (dev, _OracleRun(oracle, host, n, L, d, method))
    yield F, ctx, jobs
    jobs.clear()


@pytest.mark.parametrize("name", ["config2", "config3"])
def test_timed_configuration_matches_oracle_at_full_size(full_size, name, monkeypatch):
    F, ctx, jobs = full_size
    for var in ("FQD_COLLAPSE", "FQD_NO_FUSED_PACK", "FQD_FUSED_MIN_READS", "FQD_EDGES", "FQD_LDS_NO_SLABS",
                "FQD_GROUP_NO_SLABS", "FQD_KEPT_BY_MAP", "FQD_KEPT_BY_SORT", "FQD_DIRECTIONAL_ROUNDS",
                "FQD_NO_COMPACT_RECORDS"):
        monkeypatch.delenv(var, raising=False)       # the switches bench.py runs with: none
    n, L, umi, seed, d, method = CONFIGS[name]
    dev, run = jobs[name]
    got = F.cluster_keys(dev, key_len=L, max_distance=d, method=method, context=ctx)
    again = F.cluster_keys(dev, key_len=L, max_distance=d, method=method, context=ctx)
    assert np.array_equal(got.kept_read_ids, again.kept_read_ids)      # a warm context answers the same
    times = ctx.kernel_times(reset=True)
    if name == "config3":
        # the route bench.py times: pack fused with level 1 (no level-1 scatter launch), 12-byte records through
        # level 2 and the LDS dedupe (the keys with an N -- 160 K of 50 M reads -- through the side path)
        assert times["pack_kernel"][1] and not times["part_scatter_kernel<1>"][1], times
        assert times["part_scatter12_kernel"][1] and times["bucket_dedupe12_kernel"][1], times
        assert not times["part_scatter_kernel<2>"][1] and not times["bucket_dedupe_kernel"][1], times
    want = run.result()
    assert got.n_reads == n
    assert got.n_unique == want["n_unique"]
    assert got.n_clusters == want["n_clusters"]
    assert got.n_kept == len(want["kept_read_ids"])
    assert np.array_equal(got.kept_read_ids, want["kept_read_ids"])
