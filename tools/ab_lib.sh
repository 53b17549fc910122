#!/bin/bash
# A/B of environment switches and library variants (tools/ab_variants.sh) on ONE box: each argument is a set of VAR=value
# pairs joined by commas ("base" = none), e.g.   tools/ab_lib.sh base FQD_NO_ONE_KERNEL_COLLAPSE=1 FQD_LIB_VARIANT=v512
cd "$(dirname "$0")/.."
run() {
    local envs=$(echo "$1" | tr ',' ' ')
    [ "$1" = base ] && envs="FQD_AB=base"
    env $envs python bench.py ${BENCH_ARGS:---steps 30 --warmup 8} --no-pmc --no-cpu-baseline --no-host-input --no-copy-peak 2>/dev/null | tail -1 |
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1'.ljust(52), d['ms_per_step'], d['stage_ms_per_step']); print('    ', {k['kernel'].replace('_kernel','')[:24]: k['ms_per_step'] for k in d['kernels'][:10]})"
}
for v in "$@"; do run "$v"; done
run "$1"
