#!/usr/bin/env python3
"""bench.py -- reads/s of the clustering hot path on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config3]

A step = one pass of the hot path (pack -> collapse -> bucket pair search ->
components -> dissection -> kept read ids) over one batch of synthetic keys that
already sit in HBM. Prints ONE JSON line (rank 0). See DESIGN.md "measurement".
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# BASELINE.json configs. n = reads PER GPU (weak scaling), key = what enters the trie.
WORKLOADS = {
    # configs[2]: the configuration BASELINE.json's metric ("Hamming<=1, 150 bp") and target are quoted on
    "config3": dict(n=50_000_000, L=32, umi=32, d=1, edit=False, method="directional", seed=1003,
                    name="50M paired 2x150-bp reads, --check-lengths 16,16 (key = R1[:16]+R2[:16], 32 nt), "
                         "Hamming d=1, directional"),
    "config2": dict(n=10_000_000, L=100, umi=12, d=1, edit=False, method="directional", seed=1002,
                    name="10M single-end 100-bp reads, UMI=12, Hamming d=1, directional"),
    "config1": dict(n=10_000, L=50, umi=8, d=1, edit=False, method="directional", seed=1001,
                    name="10k single-end 50-bp reads, UMI=8, Hamming d=1, directional"),
    "config4": dict(n=25_000_000, L=300, umi=300, d=2, edit=False, method="directional", seed=1004,
                    name="200M paired 2x150-bp reads over 8 GPUs (25M per GPU), key = R1+R2 (300 nt), "
                         "Hamming d=2, directional"),
    "config5": dict(n=50_000_000, L=300, umi=300, d=1, edit=True, method="adjacency", seed=1005,
                    name="50M paired 2x150-bp reads, key = R1+R2 (300 nt), --edit d=1, adjacency"),
    # config 3 under the SKEWED model of fastqdedup_amd/synth.py (what real libraries look like): one key
    # with a million copies, heavy-tailed molecule abundance, 1 % of the molecules poly-A in segment 0 of the key, and
    # a ladder of 65 536 keys that form ONE connected component
    "config3_skew": dict(n=50_000_000, L=32, umi=32, d=1, edit=False, method="directional", seed=1003, skew=True,
                         name="config 3's shape under the skewed model (a key with 1 M copies, heavy-tailed abundance, "
                              "1 % low-complexity keys sharing a segment, a 65 536-key component), Hamming d=1, directional"),
    # ... and dissected by adjacency (the 65 536-key component through host-checked rounds; reference __init__.py:105-122)
    "config3_skew_adj": dict(n=50_000_000, L=32, umi=32, d=1, edit=False, method="adjacency", seed=1003, skew=True,
                             name="config 3's shape under the skewed model, Hamming d=1, adjacency"),
    # config 4's shape (300-nt keys, d = 2) under the skewed model: crowded segment values matched on 120 finer items per key
    # (round 4, first version: without the model's LADDER -- all 4^8 values of eight adjacent bases lie inside ONE of the 16
    # fine pieces of a 300-nt key, the keys stayed one group of the refinement and the search took the quadratic sort path,
    # 19 s; crowded buckets now go all pairs in tiles, group.hip "the last resort", and the ladder is back in)
    "config4_skew": dict(n=25_000_000, L=300, umi=300, d=2, edit=False, method="directional", seed=1004, skew=True,
                         name="config 4's shape (25M per GPU, 300-nt keys) under the skewed model (a key with 500 K "
                              "copies, heavy-tailed abundance, 1 % of the molecules poly-A in their first half, 1 % of "
                              "the reads on a ladder of 4^8 keys), Hamming d=2, directional"),
    # SURVEY.md 8d's variant of configs[4]: 1 % of the reads are one base short or long, so the keys
    # have three lengths and the Levenshtein search proper runs (equal lengths at d=1 reduce to Hamming)
    "config5v": dict(n=50_000_000, L=300, umi=300, d=1, edit=True, method="adjacency", seed=1005, indel_rate=0.01,
                     name="50M paired 2x150-bp reads, 1% with a 149- or 151-nt mate (keys of 299/300/301 nt), "
                          "--edit d=1, adjacency"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(ctx, wl, sample_reads: int):
    """The reference CPU path (oracle/_ref: the reference's own trie + distance
    extensions, driven as deduplicate_cluster drives them) on a bounded sample of
    the same workload; falls back to the oracle's C port when _ref is absent.

    Parity gate: after the CPU clock has stopped, the HIP path clusters the
    very same sample and its kept read ids must equal the CPU path's kept keys mapped through
    their first holders (reference __init__.py:201-206), its counters the CPU path's counters.
    `parity` reports the outcome; main() fails the run when it is false."""
    import numpy as np
    import fastqdedup_amd as F
    from oracle import oracle as O
    n = min(sample_reads, wl["n"])
    dev_off = host_off = None
    if wl.get("indel_rate"):
        dev, dev_off = ctx.synth_indel_keys(n, 0, n, wl["L"], wl["umi"], wl["seed"], indel_rate=wl["indel_rate"])
        host_off = dev_off.cpu().numpy().astype(np.uint64)
    else:
        from fastqdedup_amd.synth import SKEW
        dev = torch.empty(n * wl["L"], dtype=torch.uint8, device="cuda:0")
        ctx.synth_keys(dev, n, 0, n, wl["L"], wl["umi"], wl["seed"], skew=(wl["skew"] if isinstance(wl.get("skew"), dict) else SKEW) if wl.get("skew") else None)
    host = dev.cpu().numpy()
    sample = (f"first {n} reads of the same generator (n_total={n}, L={wl['L']}, umi={wl['umi']}, "
              f"seed={wl['seed']}{', indel tail ' + str(wl['indel_rate']) if wl.get('indel_rate') else ''}), "
              f"d={wl['d']}, {'edit' if wl['edit'] else 'hamming'}, {wl['method']}")
    cores = 1
    # (the skewed model's 65 536-key component: the reference's dissection loops are quadratic PYTHON loops over a
    # cluster, __init__.py:60-122 -- hours; the oracle's C restatement of the same loops is timed instead)
    if O.reference_available() and not wl.get("skew"):
        from oracle import ref_driver
        if host_off is not None:
            raw = host.tobytes()
            strs = [raw[int(host_off[i]):int(host_off[i + 1])].decode() for i in range(n)]
            del raw
        else:
            strs = [s.decode() for s in host.view(f"S{wl['L']}")]
        out = ref_driver.run_reference_path(strs, wl["d"], wl["edit"], wl["method"])
        secs = out["seconds"]
        first = {}
        for i, s in enumerate(strs):
            first.setdefault(s, i)
        cpu_kept = np.sort(np.fromiter((first[k] for k in out["kept_keys"]), dtype=np.uint64,
                                       count=len(out["kept_keys"])))
        del strs, first
        base = {"value": n / secs["total"], "unit": "reads/s", "cores": cores, "kind": "reference",
                "sample": sample, "seconds": {k: round(v, 3) for k, v in secs.items()},
                "n_unique": out["n_unique"], "n_clusters": out["n_clusters"], "n_kept": len(out["kept_keys"]),
                "cpu_count": os.cpu_count(),
                "note": "reference C extensions (Trie, within_distance) built from the reference sources; "
                        "its Python dissection loops restated in oracle/ref_driver.py"}
    else:
        from fastqdedup_amd.synth import fixed_offsets
        out = O.dedup(host, host_off if host_off is not None else fixed_offsets(n, wl["L"]), max_distance=wl["d"],
                      use_edit_distance=wl["edit"], method=wl["method"])
        total = sum(out["stage_seconds"].values())
        cpu_kept = out["kept_read_ids"]
        base = {"value": n / total, "unit": "reads/s", "cores": cores, "kind": "port", "sample": sample,
                "seconds": {k: round(v, 3) for k, v in out["stage_seconds"].items()},
                "n_unique": out["n_unique"], "n_clusters": out["n_clusters"],
                "n_kept": int(len(out["kept_read_ids"])), "cpu_count": os.cpu_count()}
    # ---- parity gate: the HIP path on the same sample, after the CPU clock stopped ----
    got = F.cluster_keys(dev, dev_off, 0 if dev_off is not None else wl["L"], max_distance=wl["d"],
                         use_edit_distance=wl["edit"], method=wl["method"], context=ctx)
    base["parity"] = bool(got.n_unique == base["n_unique"] and got.n_clusters == base["n_clusters"]
                          and got.n_kept == base["n_kept"]
                          and np.array_equal(got.kept_read_ids.astype(np.uint64), cpu_kept))
    base["parity_checked"] = "kept read-id set, n_unique, n_clusters, n_kept of the HIP path on this sample"
    del dev
    return base


SURVEY = 3      # steps with event pairs around every kernel launch, before the timed ones (main)


def pmc_traffic(kernel_substr: str, workload: str, reads_per_gpu: int):
    """HBM bytes per launch of one kernel, and of ALL kernels of one step, from the PMC counters,
    collected the way MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE
    rocprofv3 passes (kernel-trace only), KB -> bytes, and FETCH_SIZE doubled (on gfx950 it reports
    half the bytes of a wide coalesced stream; exact for 16-B/lane reads like the pack
    kernel's, an over-estimate for gathers). The child runs 1 first + SURVEY survey + 1 warm-up + 1 timed
    step: the step total is the sum over every dispatch except the input generator's, over that many."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, None, "rocprofv3 not found"
    vals, job = {}, {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix=f"fqd_pmc_{counter}_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1",
               "--workload", workload, "--no-cpu-baseline", "--no-pmc", "--no-host-input", "--no-copy-peak"]
        if reads_per_gpu:
            cmd += ["--reads-per-gpu", str(reads_per_gpu)]
        env = dict(os.environ, TMPDIR="/tmp")
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, timeout=240, check=True, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            rows = [(r["Kernel_Name"], float(r["Counter_Value"])) for f in files for r in csv.DictReader(open(f))
                    if r["Counter_Name"] == counter]
            got = [v for k, v in rows if kernel_substr in k]
            if not got:
                return None, None, f"no {counter} rows for {kernel_substr}"
            vals[counter] = sum(got) / len(got)
            job[counter] = sum(v for k, v in rows if "synth_kernel" not in k) / (1 + SURVEY + 1 + 1)
        except Exception as exc:
            return None, None, f"{counter} pass failed: {type(exc).__name__}"
        finally:
            shutil.rmtree(out, ignore_errors=True)
    traffic = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    job_traffic = (2.0 * job["FETCH_SIZE"] + job["WRITE_SIZE"]) * 1024.0
    return traffic, job_traffic, {"FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"],
                                  "job_FETCH_SIZE_KB": job["FETCH_SIZE"], "job_WRITE_SIZE_KB": job["WRITE_SIZE"],
                                  "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024"}


def copy_peak_gbs(ctx, device, nbytes: int = 2 << 30, reps: int = 5):
    """Achievable HBM bandwidth in this run, on this GPU: a device-to-device copy of `nbytes`
    (read + write counted), best of `reps`, by the library's 16-byte-per-lane copy kernel and by
    torch's copy -- the better of the two ('achievable peak')."""
    a = torch.empty(nbytes, dtype=torch.uint8, device=device)
    b = torch.empty_like(a)
    a.zero_()
    b.copy_(a)
    best = 0.0
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        e1.synchronize()
        best = max(best, 2.0 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    torch.cuda.synchronize(device)
    best = max(best, ctx.copy_bandwidth(a, b, reps))
    del a, b
    return round(best, 1)


def pinned_h2d_gbs(device, nbytes: int = 1 << 30):
    """Host-to-device rate out of PINNED memory, GB/s: what the link gives in this run."""
    try:
        src = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        dst = torch.empty(nbytes, dtype=torch.uint8, device=device)
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        del src, dst
        return round(nbytes / dt / 1e9, 1)
    except Exception:
        return None


def launch_ranks(n_ranks: int, script: str = None, argv=None) -> int:
    """`python bench.py --gpus N` without torch.distributed.run: start N fresh child processes of this
    script (one per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, rendezvous on 127.0.0.1) BEFORE this
    process has touched a GPU, pass their output through (rank 0 prints the JSON line) and return the
    worst exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks),
                   LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__),
                                       *(sys.argv[1:] if argv is None else argv)], env=env))
    codes = [None] * n_ranks
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        if any(c not in (None, 0) for c in codes):      # a rank died: the others would wait forever
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.terminate()
                    codes[i] = p.wait()
            break
        time.sleep(0.05)
    return max((abs(c) for c in codes), default=0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="config3", choices=sorted(WORKLOADS))
    ap.add_argument("--reads-per-gpu", type=int, default=0, help="override the workload's n (testing)")
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="reads of the CPU baseline's sample (default: 4 M for keys up to 100 nt, 1 M above)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-GPU code path even with one rank (diagnostic)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two child rocprofv3 --pmc passes (traffic=null)")
    ap.add_argument("--no-host-input", action="store_true", help="skip the PCIe-inclusive extra step")
    ap.add_argument("--no-copy-peak", action="store_true", help="skip the device-copy bandwidth measurement")
    ap.add_argument("--kernel-timers", default="dominant", choices=["all", "dominant", "none"],
                    help="HIP-event pairs around the hand-written kernels during the timed steps (diagnostic A/B)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))      # no external launcher: start the ranks ourselves
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import fastqdedup_amd as F
    from fastqdedup_amd.sharded import HipBackend, cluster_keys_sharded

    wl = dict(WORKLOADS[args.workload])
    if args.reads_per_gpu:
        wl["n"] = args.reads_per_gpu
    n, L = wl["n"], wl["L"]
    n_total = n * world
    t_ctx0 = time.perf_counter()
    ctx = F.Context(local_rank)
    t_context_create_ms = (time.perf_counter() - t_ctx0) * 1e3

    # synthetic input, generated in HBM before the timed region
    key_offsets = None
    if wl.get("indel_rate"):
        keys, key_offsets = ctx.synth_indel_keys(n_total, rank * n, n, L, wl["umi"], wl["seed"],
                                                 indel_rate=wl["indel_rate"])
    else:
        from fastqdedup_amd.synth import SKEW
        keys = torch.empty(n * L, dtype=torch.uint8, device=device)
        ctx.synth_keys(keys, n_total, rank * n, n, L, wl["umi"], wl["seed"], skew=(wl["skew"] if isinstance(wl.get("skew"), dict) else SKEW) if wl.get("skew") else None)
    kept_buf = torch.empty(n, dtype=torch.int64, device=device)
    backend = HipBackend(ctx, device) if sharded else None

    def step():
        if not sharded:
            return F.cluster_keys(keys, key_offsets, 0 if key_offsets is not None else L, max_distance=wl["d"],
                                  use_edit_distance=wl["edit"], method=wl["method"], context=ctx, kept_out=kept_buf,
                                  stage_times=False)
        return cluster_keys_sharded(backend, keys, key_offsets, 0 if key_offsets is not None else L,
                                    max_distance=wl["d"], use_edit_distance=wl["edit"], method=wl["method"])

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize(device)
        if sharded:
            dist.barrier()
            torch.cuda.synchronize(device)

    # ---- which kernel is the dominant one, and the per-kernel table: SURVEY steps, untimed by the clock, with HIP
    # event pairs around EVERY hand-written kernel launch (those records stop the stream: ~0.12 ms of a config-3
    # step, so the timed steps below carry the pair of the dominant kernel only)
    kern_names = list(ctx.KERNELS)
    # the COLD call: the first job of a fresh context (its workspace is allocated on the way -- hipMalloc of every
    # table, the pinned read-back words -- and the kernels' code is touched for the first time); what a one-shot CLI
    # user pays once per process, never `value`
    fence()
    t_cold0 = time.perf_counter()
    step()
    fence()
    t_first_call_ms = (time.perf_counter() - t_cold0) * 1e3
    ctx.set_timing(True, None)
    ctx.kernel_times(reset=True)
    for _ in range(SURVEY):
        res = step()
    fence()
    kern = {k: [kms, kl] for k, (kms, kl) in ctx.kernel_times(reset=True).items()}
    per_step = {k: max(1, round(v[1] / SURVEY)) for k, v in kern.items()}      # launches of each kernel in one step
    stage_sum = {k: v * args.steps for k, v in ctx.stage_times()[0].items()}      # (of the last survey step)
    dominant = max((k for k in kern if kern[k][1]), key=lambda k: kern[k][0] / kern[k][1], default=kern_names[0])
    timed_kernels = {"all": None, "dominant": (dominant,), "none": ()}[args.kernel_timers]
    ctx.set_timing(False, timed_kernels)
    for _ in range(args.warmup):
        step()
    fence()
    ctx.kernel_times(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    fence()
    elapsed = time.perf_counter() - t0
    # the dominant kernel's event pairs of the TIMED steps (the pool holds 512 pairs; later launches are not
    # timed -- see launches_timed); every other kernel keeps its survey numbers
    live = ctx.kernel_times(reset=True)
    dominant_live = live.get(dominant, (0.0, 0))
    if dominant_live[1]:
        kern[dominant] = [dominant_live[0], dominant_live[1]]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = n_total * args.steps / elapsed
    # one more, UNTIMED step with a device sync after every phase: where a sharded step spends its time
    phases = None
    if sharded:
        phases = cluster_keys_sharded(backend, keys, None, L, max_distance=wl["d"], use_edit_distance=wl["edit"],
                                      method=wl["method"], timing=True).phases_ms
        fence()

    # ---- roofline of the dominant hand-written kernel (DESIGN.md "kernels") --------
    # Every hand-written kernel of the step is timed live with HIP events on the context's stream
    # (fqd_kernel_times). `roofline` is the one with the longest average launch; `kernels` lists all.
    sh = ctx.shape()
    st = ctx.edge_stats()          # of the last step: all d+1 launches
    nseg = wl["d"] + 1
    b_key = sh.planes * sh.words * 4
    U, E = res.n_unique, res.n_edges
    n_in = n                # reads this rank collapses (multi-GPU: about n again after the all-to-all)
    U_own = U // world      # unique keys this rank's collapse produces (multi-GPU: its owner share)
    alg = {  # algorithmic bytes of ONE launch
        "pack_kernel": n * (L + b_key + 4),                          # key bytes in, record + hash out
        "part_hist_kernel<1>": n_in * 4,                             # hash
        "part_scatter_kernel<1>": n_in * (16 + 16),                  # record in (hash recomputed), record out
        "part_hist_kernel<2>": n_in * 16,
        "part_scatter_kernel<2>": n_in * (16 + 16),
        "bucket_dedupe_kernel": n_in * 16 + U_own * 24,                  # reads in, unique (record, count, first) out
        "bucket_compact_kernel": U_own * (24 + 28),
        # compact records (csrc/collapse_lds.hip): level 2 turns the 16-byte records into 12 bytes (two key words
        # + read index), the dedupe reads those
        "part_scatter12_kernel": n_in * (16 + 12),
        "bucket_dedupe12_kernel": n_in * 12 + U_own * 16,                # reads in, one uint4 row per unique key out
        "head_flags_kernel": n_in * (4 + 4 + b_key + 4),             # (hash, id), the record once, a flag
        "write_unique_kernel": U_own * (2 * b_key + 16),
        "segment_hashes_kernel": U * (b_key + 4 * nseg),
        # (bucket hash, uid) of every unique key, the record of every key in a bucket >= 2, 8 B per edge
        "bucket_pairs_kernel": U * 8 + st["keys_gathered"] / nseg * b_key + st["edges"] / nseg * 8,
        "gp_hist_kernel": U * 6,                                     # level 1 reads the hash, level 2 the (hash, uid) item
        "gp_scatter_kernel": U * 14,                                 # 4 + 8 at level 1, 8 + 8 at level 2
        "verify_candidates_kernel": st["pairs_compared"] / nseg * (8 + 2 * b_key) + st["edges"] / nseg * 8,
        # kept_bin + kept_emit (one timing slot): verdict arrays (count, state, parent, taint, best) and the
        # first id in, a flag out; per kept id 4 B out + 4 B in (bin lists) and 8 B out (the list)
        "kept_flags_kernel": U * (4 + 1 + 4 + 1 + 4 + 8 + 1) + res.n_kept * 16,
        "uf_union_kernel": E * 8,
        "uf_flatten_kernel": U * 8,
        "dissect_round_kernel": E * 8,
    }
    rocprof_name = {"part_hist_kernel<1>": "part_hist_kernel<true>", "part_hist_kernel<2>": "part_hist_kernel<false>",
                    "part_scatter_kernel<1>": "part_scatter_kernel<true",
                    "part_scatter_kernel<2>": "part_scatter_kernel<false",
                    "part_scatter12_kernel": "part_scatter12_kernel", "bucket_dedupe12_kernel": "bucket_dedupe12_kernel",
                    "dissect_round_kernel": "_round_kernel" if wl["method"] == "directional" else "adjacency_edges"}
    compact_records = bool(kern.get("part_scatter12_kernel", (0, 0))[1])
    if kern.get("pack_kernel", (0, 0))[1] and not kern.get("part_scatter_kernel<1>", (0, 0))[1] \
            and (kern.get("part_scatter_kernel<2>", (0, 0))[1] or compact_records):
        # fqd_cluster_keys took the fused way in: the pack kernel wrote its records straight into
        # level 1 of the collapse (key bytes in, 16-byte record out; no hash array)
        kern["pack_kernel (fused with level 1)"] = kern.pop("pack_kernel")
        alg["pack_kernel (fused with level 1)"] = n * (L + 16)
        rocprof_name["pack_kernel (fused with level 1)"] = "pack_kernel"
    if compact_records:
        alg["bucket_compact_kernel"] = U_own * (16 + 28)
    if kern.get("bucket_dedupe_kernel", (0, 0))[1] and not kern.get("part_scatter_kernel<2>", (0, 0))[1] \
            and not kern.get("part_scatter_kernel<1>", (0, 0))[1]:
        # records longer than one uint4: the collapse worked on (hash, position) pairs and compared the
        # records where they lie (collapse_pairs.hip)
        rec = sh.stride_words * 4
        alg["bucket_dedupe_kernel"] = n_in * 8 + (n_in - U_own) * 2 * rec + U_own * 12
        alg["bucket_compact_kernel"] = U_own * (12 + rec + rec + 12)
    if kern.get("bucket_compact_kernel", (0, 0))[1] and not kern.get("segment_hashes_kernel", (0, 0))[1]:
        alg["bucket_compact_kernel"] += U_own * 4 * nseg      # the compaction wrote the search's segment hashes too
    if kern.get("gp_scatter_kernel", (0, 0))[1]:
        # the sort-free search pass ran: the FQD_K_PAIRS slot timed grouped_candidates_kernel
        # ((hash, uid) items in, candidate pairs out), not bucket_pairs_kernel
        kern["grouped_candidates_kernel"] = kern.pop("bucket_pairs_kernel")
        alg["grouped_candidates_kernel"] = U * 8 + st["pairs_compared"] / nseg * 8
        rocprof_name["grouped_candidates_kernel"] = "grouped_candidates_kernel"
    # (names may have been re-keyed above: "pack_kernel (fused with level 1)", "grouped_candidates_kernel")
    renamed = {"pack_kernel": "pack_kernel (fused with level 1)", "bucket_pairs_kernel": "grouped_candidates_kernel"}
    dominant_name = renamed[dominant] if dominant in renamed and renamed[dominant] in kern else dominant
    per_step_name = {renamed.get(k, k) if renamed.get(k, k) in kern else k: v for k, v in per_step.items()}
    table = []
    for name, (kms, kl) in kern.items():
        if not kl:
            continue
        avg = kms / kl
        gbs = alg[name] / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
        table.append({"kernel": name, "bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS,
                      "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5), "traffic": None,
                      "alg_bytes_per_launch": int(alg[name]), "avg_launch_ms": round(avg, 4),
                      "launches_timed": kl, "ms_per_step": round(avg * per_step_name.get(name, 1), 4),
                      "timed_in": "timed steps" if name == dominant_name and dominant_live[1] else "survey steps"})
    table.sort(key=lambda r: -r["avg_launch_ms"])
    roofline = dict(table[0])
    kernels = [{k: r[k] for k in ("kernel", "avg_launch_ms", "ms_per_step", "achieved", "frac")} for r in table]
    job_traffic = None
    if rank == 0 and world == 1 and not args.no_pmc:
        traffic, job_traffic, how = pmc_traffic(rocprof_name.get(roofline["kernel"], roofline["kernel"]),
                                                args.workload, args.reads_per_gpu)
        roofline["traffic"] = None if traffic is None else int(traffic)
        roofline["traffic_source"] = how
    if not args.no_copy_peak:
        roofline["achievable_peak_gbs"] = copy_peak_gbs(ctx, device)

    # the same step with the keys in (pageable) host memory: PCIe-inclusive, never `value`
    pcie = None
    if rank == 0 and world == 1 and not args.no_host_input:
        host_keys = keys.cpu().numpy()
        host_offsets = None if key_offsets is None else key_offsets.cpu().numpy().astype("uint64")
        host_len = 0 if key_offsets is not None else L
        F.cluster_keys(host_keys, host_offsets, host_len, max_distance=wl["d"], use_edit_distance=wl["edit"],
                       method=wl["method"], context=ctx)
        t1 = time.perf_counter()
        F.cluster_keys(host_keys, host_offsets, host_len, max_distance=wl["d"], use_edit_distance=wl["edit"],
                       method=wl["method"], context=ctx)
        dt = time.perf_counter() - t1
        in_bytes = int(host_keys.nbytes)
        pcie = {"reads_per_s": round(n / dt, 1), "ms": round(dt * 1e3, 3), "input_bytes": in_bytes,
                "input_gb_per_s_incl_compute": round(in_bytes / dt / 1e9, 1),
                "input_gb_per_s_excl_compute": round(in_bytes / max(dt - ms_per_step * 1e-3, 1e-9) / 1e9, 1),
                "pinned_h2d_gb_per_s": pinned_h2d_gbs(device),
                "note": "keys start in pageable host memory, kept ids end in host memory (a page-locked array out of "
                        "torch's caching host allocator: into a fresh pageable array the 100 MB of ids took 9.8 ms, "
                        "tools/diag_e2e.py). pinned_h2d_gb_per_s: the link's rate in this run (a 1 GiB copy out of pinned "
                        "memory); the driver's own pageable path moves the keys at that rate too (staging them through "
                        "pinned buffers filled by host threads only added time). The keys travel in 8 pieces with the "
                        "pack kernel of a piece under the copy of the next (FQD_NO_CHUNKED_UPLOAD=1: one copy)"}
        del host_keys
        # ... and with the keys in PINNED host memory (what a reader that decodes into page-locked buffers hands over):
        # the copy then runs at the link's rate and the step behind it
        if in_bytes <= (4 << 30) and pcie["pinned_h2d_gb_per_s"]:
            try:
                pinned = torch.empty(in_bytes, dtype=torch.uint8).pin_memory()
                pinned.copy_(keys.reshape(-1)[:in_bytes])
                torch.cuda.synchronize(device)
                pk = pinned.numpy()
                F.cluster_keys(pk, host_offsets, host_len, max_distance=wl["d"], use_edit_distance=wl["edit"],
                               method=wl["method"], context=ctx)
                t1 = time.perf_counter()
                F.cluster_keys(pk, host_offsets, host_len, max_distance=wl["d"], use_edit_distance=wl["edit"],
                               method=wl["method"], context=ctx)
                dtp = time.perf_counter() - t1
                link_ms = in_bytes / (pcie["pinned_h2d_gb_per_s"] * 1e9) * 1e3
                pcie["pinned_input"] = {"ms": round(dtp * 1e3, 3), "reads_per_s": round(n / dtp, 1),
                                        "bytes_over_link_rate_ms": round(link_ms, 3),
                                        "ratio_to_link_time": round(dtp * 1e3 / link_ms, 3)}
                del pinned, pk
            except Exception as exc:          # (no page-locked memory to be had: the pageable figure stands alone)
                pcie["pinned_input"] = {"error": str(exc)[:200]}

    # whole-job algorithmic traffic as SURVEY.md section 8d defines it (ALG_BYTES_V1): pack + collapse
    # + (d+1) search passes + edges + dissection, with N, U, E of this run
    b_key_v1 = 8 * ((L + 31) // 32 + (L + 63) // 64)
    alg_v1 = (n_total * (L + 2 * b_key_v1 + 24) + U * (b_key_v1 + 8) + nseg * U * (2 * b_key_v1 + 24) + 16 * E
              + U * (b_key_v1 + 13))
    job_gbs = alg_v1 / (ms_per_step * 1e-3) / 1e9
    job_roofline = {"alg_bytes": int(alg_v1), "formula": "ALG_BYTES_V1 (SURVEY.md 8d)", "achieved": round(job_gbs, 1),
                    "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": round(job_gbs / (HBM_PEAK_GBS * world), 5),
                    "traffic": None if job_traffic is None else int(job_traffic),
                    "traffic_note": "PMC bytes of every kernel of one step, (2*FETCH_SIZE + WRITE_SIZE)*1024"}
    # the whole-job fraction SURVEY.md 8d defines, next to the dominant kernel's
    roofline["job"] = {k: job_roofline[k] for k in ("alg_bytes", "achieved", "frac", "traffic")}

    out = {
        "metric": "reads/sec clustered (Hamming<=1, 150 bp)", "value": round(value, 1), "unit": "reads/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
        "timed_region": "t_dev: key bytes resident in HBM -> ascending kept read ids resident in HBM "
                        "(value = reads / t_dev); t_e2e (pageable host keys -> host ids, PCIe-inclusive) is "
                        "reported as t_e2e_ms / host_input and is never `value`",
        "t_dev_ms": round(ms_per_step, 3), "t_e2e_ms": None if pcie is None else pcie["ms"],
        "t_first_call_ms": round(t_first_call_ms, 3), "t_context_create_ms": round(t_context_create_ms, 3),
        "t_first_call_note": "the first job of the fresh context, keys resident in HBM: workspace allocations and first "
                             "use of every kernel included (fqd_create itself: t_context_create_ms); later jobs of "
                             "the context reuse the workspace -- t_dev_ms",
        "t_e2e_pinned_ms": None if pcie is None else pcie.get("pinned_input", {}).get("ms"),
        "data": "synthetic keys generated in HBM (fqd_synth_keys, fastqdedup_amd/synth.py)",
        "config": {"workload": wl["name"], "reads_per_gpu": n, "reads_total": n_total, "key_len": L,
                   "max_distance": wl["d"], "metric": "edit" if wl["edit"] else "hamming",
                   "dissection": wl["method"], "seed": wl["seed"],
                   "parallelism": (f"1 process/GPU over RCCL, plan {res.plan} (fastqdedup_amd/sharded.py)"
                                   if sharded else "single GPU")},
        "result": {"n_unique": res.n_unique, "n_edges": res.n_edges, "n_clusters": res.n_clusters,
                   "n_kept": res.n_kept},
        "stage_ms_per_step": {k: round(v / args.steps, 3) for k, v in stage_sum.items()},
        "record_bytes": sh.stride_words * 4, "planes": sh.planes,
        "dedupe_record_bytes": 12 if compact_records else sh.stride_words * 4,
        "roofline": roofline,
        "job_roofline": job_roofline,
        "kernels": kernels,
        "route": getattr(res, "route", None),
        "kernel_timers": {"timed_steps": args.kernel_timers, "dominant": dominant_name,
                          "note": "HIP event pairs around every kernel launch stop the stream at each record (~0.12 ms of "
                                  "a config-3 step): the timed steps carry the pair of the dominant kernel only -- "
                                  "roofline.avg_launch_ms is measured there, live -- and the other kernels' numbers come "
                                  f"from {SURVEY} untimed survey steps with all pairs on"},
        "host_input": pcie,
    }
    if phases is not None:
        out["sharded_phases_ms_rank0"] = phases
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(ctx, wl, args.cpu_sample or (1_000_000 if L > 100 or wl.get("skew") else 4_000_000))
        except Exception as exc:  # the GPU numbers stand on their own
            out["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if sharded:
        dist.destroy_process_group()
    if "cpu_baseline" in out and out["cpu_baseline"].get("parity") is not True:
        # false: the two disagree; missing: the baseline or the parity run raised -- neither is a pass
        raise SystemExit("PARITY FAILURE: the HIP path and the CPU reference disagree on the cpu_baseline sample"
                         if out["cpu_baseline"].get("parity") is False else
                         f"PARITY UNCHECKED: {out['cpu_baseline'].get('error', 'no parity verdict')}")


if __name__ == "__main__":
    main()
