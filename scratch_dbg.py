import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import fastqdedup_amd as F
from fastqdedup_amd.synth import synth_keys
n, L, umi, seed = 50000, 100, 12, 1002
host = synth_keys(n, L, umi, seed)
ctx = F.Context(0)
def pk(keys):
    raw = np.frombuffer("".join(keys).encode(), dtype=np.uint8)
    off = np.concatenate([[0], np.cumsum([len(k) for k in keys])]).astype(np.uint64)
    return raw, off
for prelude in (["AC", "GT", "TT", "ACG"], ["ACGT"] * 3 + ["ACGA"], []):
    if prelude:
        raw, off = pk(prelude)
        big = F.cluster_keys(raw, off, max_distance=5, method="highest_count", context=ctx)
        print("prelude", prelude, big.n_clusters, big.kept_read_ids)
    b = F.cluster_keys(host.reshape(-1), key_len=L, context=ctx)
    print("n_kept", b.n_kept, "listed", ctx.kept_count(), "sorted", bool(np.all(np.diff(b.kept_read_ids.astype(np.int64)) > 0)))
