#!/usr/bin/env python3
"""Diagnostic: the one-rank sharded step on long keys (config 4's shape), its phases, the route the owner's collapse took
and the kernels it ran.   tools/diag_sharded_long.py [reads] [L] [d]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FQD_KERNEL_TIMERS", "1")
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29593")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import fastqdedup_amd as F
from fastqdedup_amd.sharded import HipBackend, cluster_keys_sharded
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
d = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ctx = F.Context(0)
keys = torch.empty(n * L, dtype=torch.uint8, device=dev)
ctx.synth_keys(keys, n, 0, n, L, L, 1004)
be = HipBackend(ctx, dev)
for i in range(3):
    ctx.kernel_times(reset=True)
    t = time.perf_counter()
    r = cluster_keys_sharded(be, keys, None, L, max_distance=d, method="directional", timing=(i == 2))
    ctx.synchronize(); torch.cuda.synchronize(dev)
    print(f"step {i}: {(time.perf_counter() - t) * 1e3:.2f} ms  unique={r.n_unique} edges={r.n_edges} clusters={r.n_clusters}", flush=True)
print("phases", r.phases_ms)
print("main ctx route", [k for k, v in ctx.route().items() if v])
print("main ctx kernels", {k: (round(ms, 3), c) for k, (ms, c) in ctx.kernel_times().items() if c})
plain = F.cluster_keys(keys, key_len=L, max_distance=d, method="directional", context=F.Context(0))
print("plain: unique", plain.n_unique, "clusters", plain.n_clusters)
dist.destroy_process_group()
