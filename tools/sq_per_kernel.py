#!/usr/bin/env python3
"""Where the waves of every kernel of one bench step spend their cycles, from the SQ counters.

One `rocprofv3 --kernel-trace --pmc <8 SQ counters>` pass over `bench.py --steps 1 --warmup 1`
(MI355X_MICROARCH.md "rocprofv3 PMC slots": 8 SQ slots per pass): per kernel, the mean over its
launches of SQ_WAVE_CYCLES and the share of them the waves were parked on s_waitcnt / a barrier
(SQ_WAIT_ANY), stalled at issue (SQ_WAIT_INST_ANY, of which on the LDS queue: SQ_WAIT_INST_LDS) or
issuing (SQ_ACTIVE_INST_ANY); LDS bank-conflict cycles over LDS-active cycles; SQ_BUSY_CYCLES.

    python3 tools/sq_per_kernel.py --workload config3 --out profiles/rNN_sq_per_kernel_config3.json
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_per_kernel import ROOT, short  # noqa: E402

COUNTERS = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS",
            "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_BUSY_CYCLES"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="config3")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    out = tempfile.mkdtemp(prefix="fqd_sq_", dir="/tmp")
    cmd = [exe, "--kernel-trace", "--pmc", *COUNTERS, "--output-format", "csv", "-d", out, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--workload",
           args.workload, "--no-cpu-baseline", "--no-pmc", "--no-host-input", "--no-copy-peak"]
    subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL, timeout=600)
    per = {}
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in COUNTERS:
                per.setdefault(short(r["Kernel_Name"]), {}).setdefault(r["Counter_Name"], []).append(
                    float(r["Counter_Value"]))
    shutil.rmtree(out, ignore_errors=True)
    rows = {}
    for name, v in per.items():
        m = {c: sum(x) / len(x) for c, x in v.items()}
        wc = m.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0:
            continue
        rows[name] = {
            "launches": len(v["SQ_WAVE_CYCLES"]), "SQ_WAVE_CYCLES": int(wc),
            "parked_on_waitcnt_or_barrier": round(m.get("SQ_WAIT_ANY", 0) / wc, 3),
            "stalled_at_issue": round(m.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
            "stalled_at_issue_on_lds": round(m.get("SQ_WAIT_INST_LDS", 0) / wc, 3),
            "issuing": round(m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
            "lds_bank_conflict_share_of_lds_cycles": (round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], 3)
                                                      if m.get("SQ_LDS_IDX_ACTIVE") else None),
            "SQ_BUSY_CYCLES": int(m.get("SQ_BUSY_CYCLES", 0)),
        }
    rows = dict(sorted(rows.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"] * kv[1]["launches"]))
    doc = {"command": "tools/sq_per_kernel.py --workload " + args.workload,
           "note": "shares of SQ_WAVE_CYCLES (quad-cycles summed over the waves of a launch), mean over the launches of "
                   "1 warm-up + 1 timed step; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES",
           "kernels": rows}
    text = json.dumps(doc, indent=1)
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
