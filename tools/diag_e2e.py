#!/usr/bin/env python3
"""Diagnostic: where the PCIe-inclusive step (host keys -> host kept ids) spends its time -- pageable or pinned keys,
a fresh pageable or a pinned buffer for the ids."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastqdedup_amd as F
n, L = 50_000_000, 32
dev = torch.device("cuda", 0)
ctx = F.Context(0)
keys = torch.empty(n * L, dtype=torch.uint8, device=dev)
ctx.synth_keys(keys, n, 0, n, L, L, 1003)
pageable = keys.cpu().numpy()
pinned_t = torch.empty(n * L, dtype=torch.uint8).pin_memory(); pinned_t.copy_(keys); torch.cuda.synchronize()
pinned = pinned_t.numpy()
ids_pinned_t = torch.empty(n, dtype=torch.int64).pin_memory()
ids_pinned = ids_pinned_t.numpy().view(np.uint64)
def run(name, k, out):
    best = 1e9
    for i in range(4):
        ctx.synchronize(); torch.cuda.synchronize(); t = time.perf_counter()
        r = F.cluster_keys(k, key_len=L, max_distance=1, method="directional", context=ctx,
                           kept_out=None if out is None else out)
        dt = (time.perf_counter() - t) * 1e3
        best = min(best, dt)
    print(f"{name:46s} {best:8.3f} ms   kept={r.n_kept}", flush=True)
run("device keys -> fresh pageable ids", keys, None)
run("device keys -> pinned ids", keys, ids_pinned)
run("pageable keys -> fresh pageable ids", pageable, None)
run("pageable keys -> pinned ids", pageable, ids_pinned)
run("pinned keys -> fresh pageable ids", pinned, None)
run("pinned keys -> pinned ids", pinned, ids_pinned)
