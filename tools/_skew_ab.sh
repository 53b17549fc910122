cd /root/repo
run() {
    env "$@" python bench.py --workload config3_skew --steps 10 --warmup 3 --kernel-timers all --no-pmc --no-cpu-baseline --no-host-input --no-copy-peak 2>/dev/null | tail -1 |
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*'.ljust(44), d['ms_per_step'], d['stage_ms_per_step'], d.get('route')); print('    ', {k['kernel'].replace('_kernel','')[:22]: k['ms_per_step'] for k in d['kernels']})"
}
run FQD_AB=base
run FQD_NO_SPILL_LIST=1
