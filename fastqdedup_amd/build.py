"""Builds ``libfqdedup_hip.so`` in-tree with hipcc for gfx950 (cross-compiles
without a GPU). ``python -m fastqdedup_amd.build`` or ``__graft_entry__.build()``.
``FQD_EXTRA_FLAGS`` adds compiler flags (the tile-shape experiments: ``-DFQD_PACK_NSUB=3``,
``-DFQD_SCATTER12_EPT=16 -DFQD_SCATTER12_ROUNDS=2``; touch the source to rebuild it)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libfqdedup_hip.so")
SOURCES = ["api.hip", "api_search.hip", "api_graph.hip", "api_exchange.hip", "api_trie.hip", "trieorder.hip", "prims.hip", "pack.hip", "collapse.hip", "collapse_lds.hip", "collapse_pairs.hip", "edges.hip", "group.hip", "edit.hip", "exchange.hip", "graph.hip", "quality.hip", "synth.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + os.environ.get("FQD_EXTRA_FLAGS", "").split()


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _save_resources(path: str, remarks) -> None:
    """Per-kernel register/scratch/LDS use as the compiler reports it (-Rpass-analysis=
    kernel-resource-usage), kept next to the object file: tests/test_hip_abi.py checks that no hot
    kernel spills to scratch memory (a select chain on a vector's components once sent a whole
    register array there and cost 40 % of a kernel, unnoticed until the next profile)."""
    import json
    import re
    out, cur = {}, None
    for ln in remarks:
        m = re.search(r"remark:\s+(.*?) \[-Rpass-analysis", ln)
        if not m:
            continue
        text = m.group(1).strip()
        if text.startswith("Function Name:"):
            cur = text.split(":", 1)[1].strip()
            out[cur] = {}
        elif cur and ":" in text:
            k, v = text.rsplit(":", 1)
            try:
                out[cur][k.strip()] = int(v)
            except ValueError:
                pass
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


def kernel_resources() -> dict:
    """{mangled kernel name: {"VGPRs": .., "ScratchSize [bytes/lane]": .., ...}} of the last build."""
    import json
    res = {}
    objdir = os.path.join(HERE, "build")
    for src in SOURCES:
        path = os.path.join(objdir, src.replace(".hip", ".o") + ".resources.json")
        if os.path.exists(path):
            with open(path) as f:
                res.update(json.load(f))
    return res


def build(force: bool = False, verbose: bool = False) -> str:
    headers = [os.path.join(CSRC, "fqd_internal.h"), os.path.join(CSRC, "partition.cuh"), os.path.join(CSRC, "api_ctx.h"),
               os.path.join(os.path.dirname(HERE), "include", "fqdedup_hip.h")]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers) or not os.path.exists(o + ".resources.json"):
            jobs.append([HIPCC, *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(f"{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        remarks = [ln for ln in r.stderr.splitlines() if "kernel-resource-usage" in ln]
        if "-c" in cmd:  # objects without kernels get an empty table, so they are not rebuilt every time
            _save_resources(cmd[-1] + ".resources.json", remarks)
        rest = "\n".join(ln for ln in r.stderr.splitlines() if "kernel-resource-usage" not in ln)
        if verbose and rest.strip():
            print(rest, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
