#!/usr/bin/env python3
"""HBM traffic of every kernel of one bench step from the PMC counters.

Two SEPARATE `rocprofv3 --kernel-trace --pmc <counter>` passes (FETCH_SIZE, WRITE_SIZE) over
`bench.py --steps 1 --warmup 1`, as MI355X_MICROARCH.md "HBM" prescribes; per kernel the mean
over its launches of (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes (gfx950: FETCH_SIZE reports half
the bytes of wide coalesced streams). Writes one JSON object to stdout / --out.

    python3 tools/pmc_per_kernel.py --workload config3 --out profiles/rNN_pmc_per_kernel_config3.json
"""
import argparse
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(?:<[^(]*?>)?)\(", name)
    return (m.group(1) if m else name)[:80]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="config3")
    ap.add_argument("--out", default="")
    ap.add_argument("extra", nargs="*", help="further bench.py arguments")
    args = ap.parse_args()
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix=f"fqd_pmc_{counter}_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--workload",
               args.workload, "--no-cpu-baseline", "--no-pmc", "--no-host-input", "--no-copy-peak", *args.extra]
        subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != counter:
                    continue
                k = per.setdefault(short(r["Kernel_Name"]), {})
                k.setdefault(counter, []).append(float(r["Counter_Value"]))
        shutil.rmtree(out, ignore_errors=True)
    rows = {}
    for name, v in per.items():
        f, w = v.get("FETCH_SIZE", []), v.get("WRITE_SIZE", [])
        if not f or not w:
            continue
        fk, wk = sum(f) / len(f), sum(w) / len(w)
        rows[name] = {"launches": len(f), "FETCH_SIZE_KB": round(fk, 1), "WRITE_SIZE_KB": round(wk, 1),
                      "traffic_bytes_per_launch": int((2 * fk + wk) * 1024)}
    rows = dict(sorted(rows.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"]))
    doc = {"command": "tools/pmc_per_kernel.py --workload " + args.workload + " " + " ".join(args.extra),
           "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes, mean per launch; separate rocprofv3 --pmc passes",
           "note": "launches cover 1 warm-up + 1 timed step (+ bench.py's synth_kernel); FETCH_SIZE is doubled per "
                   "the gfx950 correction, which is exact for wide coalesced reads and an over-estimate for gathers",
           "kernels": rows}
    text = json.dumps(doc, indent=1)
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
