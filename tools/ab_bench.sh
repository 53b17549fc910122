#!/bin/bash
# A/B timing of diagnostic switches on ONE box (box-to-box spread is +-3 % on the step): every variant runs
# bench.py with the same steps; prints ms/step and the stage times.   tools/ab_bench.sh "VAR=1" "OTHER=0" ...
cd "$(dirname "$0")/.."
run() {
    env "$@" python bench.py --steps 30 --warmup 8 --no-pmc --no-cpu-baseline --no-host-input --no-copy-peak 2>/dev/null | tail -1 |
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*'.ljust(44), d['ms_per_step'], d['stage_ms_per_step']); print('    ', {k['kernel'].replace('_kernel','')[:22]: k['ms_per_step'] for k in d['kernels'][:9]})"
}
run FQD_AB=base
for v in "$@"; do run $v; done
run FQD_AB=base
