"""Randomised parity of fqd_cluster_keys' fused way in (pack -> level 1 -> level 2 -> dedupe -> compaction with search
pass 0, the spill list, the side path) against the CPU oracle: key lengths, N rates, hot keys, Zipf copies, keys sharing
segment 0, weights, distances and dissection methods drawn from a FIXED seed set, two jobs per context (the second
starts in whatever mode the first one left the context in: spill list, routing off, ...). The loop that
tools/fuzz_fused.py runs open-ended, bounded so that the driver's `-m gpu` run carries it. GPU only.

What it checks follows the reference: `Trie.add_sequence` counts (`_triemodule.c:222-288`), `pop_cluster`'s components
(`:778-897`) and the three dissections (`__init__.py:60-122`) through the oracle's restatement."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEEDS = [1, 2, 3, 5, 8, 13]
CASES_PER_SEED = 4


def _case(rng):
    from fastqdedup_amd.synth import synth_keys
    n = int(rng.integers(210_000, 500_000))
    L = int(rng.choice([16, 20, 24, 28, 31, 32]))
    d = int(rng.choice([1, 1, 2]))
    method = str(rng.choice(["directional", "adjacency", "highest_count"]))
    n_rate = float(rng.choice([0.0, 1e-4, 1e-3, 5e-3]))
    keys = synth_keys(n, L, min(L, 12), int(rng.integers(1 << 30)), sub_rate=3e-3, n_rate=n_rate)
    what = str(rng.choice(["plain", "hot", "zipf", "lowc", "hot+lowc"]))
    if "hot" in what:
        rows = rng.choice(n, size=int(n * rng.choice([0.02, 0.1, 0.2])), replace=False)
        keys[rows] = keys[rows[0]]
        near = rows[: max(1, len(rows) // 40)]
        keys[near, rng.integers(0, L, size=len(near))] = ord("C")
    if what == "zipf":
        src = rng.choice(n, size=3000, replace=False)
        p = 1.0 / np.arange(1, 3001)
        rows = rng.choice(n, size=n // 5, replace=False)
        keys[rows] = keys[src[rng.choice(3000, size=len(rows), p=p / p.sum())]]
    if "lowc" in what:
        rows = rng.choice(n, size=int(n * 0.02), replace=False)
        keys[rows, : L // (d + 1)] = ord("A")
    weights = rng.integers(0, 3, size=n).astype(np.uint32) if rng.random() < 0.3 else None
    return n, L, d, method, what, n_rate, np.ascontiguousarray(keys).reshape(-1), weights


@pytest.mark.parametrize("seed", SEEDS)
def test_fused_way_in_on_random_inputs_matches_oracle(oracle, seed, monkeypatch):
    import fastqdedup_amd as F
    from fastqdedup_amd.synth import fixed_offsets
    monkeypatch.setenv("FQD_FUSED_MIN_READS", "100000")
    rng = np.random.default_rng(seed)
    for case in range(CASES_PER_SEED):
        n, L, d, method, what, n_rate, raw, weights = _case(rng)
        want = oracle.dedup(raw, fixed_offsets(n, L), weights, max_distance=d, method=method)
        ctx = F.Context(0)
        for job in range(2):
            got = F.cluster_keys(raw, key_len=L, weights=weights, max_distance=d, method=method, context=ctx)
            where = (f"seed {seed} case {case} job {job}: n={n} L={L} d={d} {method} n_rate={n_rate} {what} "
                     f"weights={weights is not None} route={[k for k, v in got.route.items() if v]}")
            assert got.n_unique == want["n_unique"], where
            assert got.n_clusters == want["n_clusters"], where
            assert np.array_equal(got.kept_read_ids, want["kept_read_ids"]), where
        del ctx
