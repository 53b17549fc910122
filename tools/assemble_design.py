#!/usr/bin/env python3
"""Fills the measured numbers of DESIGN.md's current-state sections from profiles/rNN_*.json|csv (the text around them is
written by hand in DESIGN.md itself, between the markers this script replaces):

    python tools/assemble_design.py r04        # rewrites the tables between <!-- rNN:name --> ... <!-- /rNN:name -->

Tables: `workloads` (ms per step, reads/s, whole-job fraction, PMC traffic, t_e2e, first call, CPU reference),
`kernels_config3` (rocprofv3 average per kernel of the default command, PMC bytes per launch), `sharded` (the one-rank
plan's phases)."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
P = os.path.join(ROOT, "profiles")


def load(name):
    path = os.path.join(P, f"{R}_{name}")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        txt = f.read()
    lines = [ln for ln in txt.splitlines() if ln.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def workloads():
    rows = [("config 3: 50 M × 32 nt, d=1, directional", "bench_config3.json"),
            ("config 2: 10 M × 100 nt, d=1, directional", "bench_config2.json"),
            ("config 4's shape on one GPU: 25 M × 300 nt, d=2, directional", "bench_config4.json"),
            ("config 5: 50 M × 300 nt, `--edit` d=1, adjacency", "bench_config5.json"),
            ("config 5v: the same with a 1 % indel tail", "bench_config5v.json"),
            ("config 3 under the skewed model (`config3_skew`)", "bench_config3_skew.json"),
            ("… dissected by adjacency (`config3_skew_adj`)", "bench_config3_skew_adj.json"),
            ("config 4's shape under the skewed model (`config4_skew`)", "bench_config4_skew.json"),
            ("config 3 through the multi-GPU plan, one rank", "bench_config3_sharded_world1.json"),
            ("config 4's shape through the multi-GPU plan, one rank", "bench_config4_sharded_world1.json")]
    out = ["| workload | ms / step | reads/s (`t_dev`) | whole-job fraction of 8 TB/s | PMC traffic per step | `t_e2e` ms | first call ms | CPU reference reads/s (sample) |",
           "|---|---|---|---|---|---|---|---|"]
    for label, name in rows:
        d = load(name)
        if d is None:
            continue
        jr = d.get("job_roofline", {})
        tr = jr.get("traffic")
        cpu = d.get("cpu_baseline") or {}
        cpu_s = "–"
        if cpu.get("value"):
            cpu_s = f"{cpu['value'] / 1e6:.2f} M ({cpu.get('sample', '').split(' reads')[0].replace('first ', '')} reads; parity {cpu.get('parity')})"
        out.append(f"| {label} | **{d['ms_per_step']:.2f}** | {d['value'] / 1e9:.2f} G | {jr.get('frac', 0):.3f} | "
                   f"{'%.2f GB for %.2f algorithmic' % (tr / 1e9, jr['alg_bytes'] / 1e9) if tr else '–'} | "
                   f"{d.get('t_e2e_ms') or '–'} | {d.get('t_first_call_ms', '–')} | {cpu_s} |")
    return "\n".join(out)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def kernels_config3():
    path = os.path.join(P, f"{R}_kernel_stats_bench_config3.csv")
    if not os.path.exists(path):
        return ""
    pmc = {}
    pj = os.path.join(P, f"{R}_pmc_per_kernel_config3.json")
    if os.path.exists(pj):
        pmc = json.load(open(pj)).get("kernels", {})
    rows = list(csv.DictReader(open(path)))
    steps = None
    for r in rows:
        if "pack_kernel" in r["Name"]:
            steps = int(r["Calls"])
    out = ["| kernel | launches per step | µs per launch | PMC GB per launch |", "|---|---|---|---|"]
    for r in rows:
        n = short(r["Name"])
        if "synth" in n or float(r["AverageNs"]) < 3000:
            continue
        gb = pmc.get(n, {}).get("traffic_bytes_per_launch")
        out.append(f"| `{n}` | {int(r['Calls']) / steps if steps else r['Calls']:.0f} | {float(r['AverageNs']) / 1000:.0f} | "
                   f"{gb / 1e9:.2f} |" if gb else f"| `{n}` | {int(r['Calls']) / steps if steps else r['Calls']:.0f} | {float(r['AverageNs']) / 1000:.0f} | – |")
    return "\n".join(out)


def sharded():
    out = ["| phase | config 3, ms | config 4's shape, ms |", "|---|---|---|"]
    a, b = load("bench_config3_sharded_world1.json"), load("bench_config4_sharded_world1.json")
    if not a:
        return ""
    pa, pb = a.get("sharded_phases_ms_rank0") or {}, (b or {}).get("sharded_phases_ms_rank0") or {}
    for k in list(pa) + [k for k in pb if k not in pa]:
        out.append(f"| {k} | {pa.get(k, '–')} | {pb.get(k, '–')} |")
    out.append(f"| **step (timed, no phase syncs)** | **{a['ms_per_step']:.2f}** | **{b['ms_per_step']:.2f}** |" if b else
               f"| **step** | **{a['ms_per_step']:.2f}** | – |")
    return "\n".join(out)


def main():
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    for name, fn in (("workloads", workloads), ("kernels_config3", kernels_config3), ("sharded", sharded)):
        table = fn()
        if not table:
            continue
        pat = re.compile(rf"(<!-- {R}:{name} -->\n).*?(\n<!-- /{R}:{name} -->)", re.S)
        if not pat.search(s):
            print(f"marker {R}:{name} not found", file=sys.stderr)
            continue
        s = pat.sub(lambda m: m.group(1) + table + m.group(2), s)
    open(path, "w").write(s)


if __name__ == "__main__":
    main()
