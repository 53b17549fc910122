#!/usr/bin/env python3
"""The one-kernel dedupe + compaction (bucket_collapse12_kernel) against the two kernels it replaces, on the same keys:
kept read ids, unique keys, edges and clusters must be identical (the unique table's ORDER differs -- rows of a team in
arrival order -- so only order-free results are compared).   tools/check_one_kernel.py [reads] [L] [d]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fastqdedup_amd as F

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
d = int(sys.argv[3]) if len(sys.argv) > 3 else 1
keys = torch.empty(n * L, dtype=torch.uint8, device="cuda:0")
ctx = F.Context(0)
ctx.synth_keys(keys, n, 0, n, L, L, 1003)
out = {}
for mode in ("two", "one", "one"):
    if mode == "two":
        os.environ.pop("FQD_ONE_KERNEL_COLLAPSE", None)
    else:
        os.environ["FQD_ONE_KERNEL_COLLAPSE"] = "1"
    for method in ("directional", "adjacency"):
        t0 = time.perf_counter()
        r = F.cluster_keys(keys, key_len=L, max_distance=d, method=method, context=ctx)
        dt = time.perf_counter() - t0
        route = [k for k, v in r.route.items() if v]
        print(mode, method, f"{dt*1e3:.2f} ms", r.n_unique, r.n_edges, r.n_clusters, r.n_kept, route, flush=True)
        key = (method,)
        val = (r.n_unique, r.n_edges, r.n_clusters, r.n_kept, np.asarray(r.kept_read_ids).copy())
        if mode == "two":
            out[key] = val
        else:
            assert "one_kernel_collapse" in route, route
            ref = out[key]
            assert val[:4] == ref[:4], (val[:4], ref[:4])
            assert np.array_equal(val[4], ref[4]), "kept read ids differ"
print("one kernel == two kernels")
