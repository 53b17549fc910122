#!/bin/bash
# A/B of a COMPILE-TIME switch on one box: [BENCH_ARGS="--workload W --steps K ..."] tools/ab_build.sh file.hip "-DFLAG=1" ["-DOTHER=2" ...]
# (rebuilds the one object with each flag set in turn, runs the same short bench, restores the default build)
cd "$(dirname "$0")/.."
src=$1; shift
run() {
    python bench.py ${BENCH_ARGS:---steps 30 --warmup 8} --no-pmc --no-cpu-baseline --no-host-input --no-copy-peak 2>/dev/null | tail -1 |
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1'.ljust(40), d['ms_per_step'], {k['kernel'].replace('_kernel','')[:22]: k['ms_per_step'] for k in d['kernels'][:8]})"
}
run default
for f in "$@"; do
    touch fastqdedup_amd/csrc/$src
    FQD_EXTRA_FLAGS="$f" python -m fastqdedup_amd.build > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
    run "$f"
done
touch fastqdedup_amd/csrc/$src
python -m fastqdedup_amd.build > /dev/null 2>&1
run default
