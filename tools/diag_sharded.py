#!/usr/bin/env python3
"""Diagnostic: per-step wall time of the one-rank sharded step, with and without phase syncs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29591")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import fastqdedup_amd as F
from fastqdedup_amd.sharded import HipBackend, cluster_keys_sharded
n, L = 50_000_000, 32
ctx = F.Context(0)
keys = torch.empty(n * L, dtype=torch.uint8, device=dev)
ctx.synth_keys(keys, n, 0, n, L, L, 1003)
be = HipBackend(ctx, dev)
def fence():
    ctx.synchronize(); torch.cuda.synchronize(dev)
for timing in (False, True, False):
    for i in range(6):
        fence(); t = time.perf_counter()
        r = cluster_keys_sharded(be, keys, None, L, max_distance=1, method="directional", timing=timing)
        fence(); dt = (time.perf_counter() - t) * 1e3
        st = torch.cuda.memory_stats(dev)
        print(f"timing={timing} step {i}: {dt:.3f} ms  device_allocs={st.get('num_device_alloc')} frees={st.get('num_device_free')} reserved={st.get('reserved_bytes.all.current',0)>>20} MiB", flush=True)
    if timing: print(r.phases_ms)
dist.destroy_process_group()
