#!/usr/bin/env python3
"""Randomised parity run of fqd_cluster_keys' fused way in against the oracle: key lengths, N rates, hot keys, Zipf
copies, weights, distances and dissection methods drawn at random; each case runs twice on one context (the second
job starts in whatever mode the first one left).   tools/fuzz_fused.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FQD_FUSED_MIN_READS", "100000")
import numpy as np
import fastqdedup_amd as F
from fastqdedup_amd.synth import fixed_offsets, synth_keys
from oracle import oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
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
    raw = np.ascontiguousarray(keys).reshape(-1)
    want = O.dedup(raw, fixed_offsets(n, L), weights, max_distance=d, method=method)
    ctx = F.Context(0)
    for job in range(2):
        got = F.cluster_keys(raw, key_len=L, weights=weights, max_distance=d, method=method, context=ctx)
        ok = (got.n_unique == want["n_unique"] and got.n_clusters == want["n_clusters"]
              and np.array_equal(got.kept_read_ids, want["kept_read_ids"]))
        bad += not ok
        print(f"case {case} job {job}: n={n} L={L} d={d} {method} n_rate={n_rate} {what} weights={weights is not None} "
              f"-> {'ok' if ok else 'MISMATCH'}  route={[k for k, v in got.route.items() if v]}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
