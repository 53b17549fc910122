#!/usr/bin/env python3
"""Memory-instruction skeleton of a kernel from hipcc's device assembly: the order of global loads,
stores, atomics, s_waitcnt vmcnt and barriers -- enough to see whether the loads of an unrolled loop
are in flight together or wait one by one (a conditional load + use inside one branch does the latter).
   hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S x.hip -o x.s
   tools/isa_skeleton.py x.s kernel_name_substring [max_items]"""
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    limit = int(sys.argv[3]) if len(sys.argv) > 3 else 120
    lines = open(path).read().splitlines()
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\S*)%s(\S*):" % re.escape(pat), lines[i])
        if not m:
            i += 1
            continue
        name = lines[i].split(":")[0]
        out = []
        j = i + 1
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
            t = lines[j].strip()
            mm = re.match(r"(global_load\w*|global_store\w*|global_atomic\w*|buffer_\w+|flat_\w+|s_waitcnt vmcnt\(\d+\)|s_waitcnt lgkmcnt\(0\)|s_barrier|ds_\w+|s_endpgm|scratch_\w+)", t)
            if mm:
                tok = mm.group(1)
                tok = {"s_waitcnt lgkmcnt(0)": "L0"}.get(tok, tok).replace("s_waitcnt ", "").replace("global_", "g.")
                if out and out[-1][0] == tok:
                    out[-1][1] += 1
                else:
                    out.append([tok, 1])
            j += 1
        print(name[:100])
        print("  " + " ".join(f"{t}x{n}" if n > 1 else t for t, n in out[:limit]))
        i = j


if __name__ == "__main__":
    main()
