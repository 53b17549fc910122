import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastqdedup_amd as F
from fastqdedup_amd.synth import synth_keys
os.environ["FQD_FUSED_MIN_READS"] = "100000"
os.environ["FQD_DEBUG"] = "1"
n, L = 300_000, 32
keys = np.ascontiguousarray(synth_keys(n, L, L, 91, sub_rate=3e-3, n_rate=2e-4)).reshape(-1)
for env in ({}, {"FQD_NO_COMPACT_RECORDS": "1"}, {"FQD_NO_FUSED_PACK": "1"}):
    for k, v in env.items():
        os.environ[k] = v
    ctx = F.Context(0)
    r = F.cluster_keys(keys, key_len=L, max_distance=1, method="directional", context=ctx)
    kt = ctx.kernel_times(reset=True)
    print(env, r.n_unique, r.n_clusters, r.n_kept, {k: v[1] for k, v in kt.items() if v[1]}, flush=True)
    for k in env:
        del os.environ[k]
