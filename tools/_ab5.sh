cd /root/repo
run() {
    w=$1; shift
    env "$@" python bench.py --workload $w --steps 8 --warmup 3 --kernel-timers all --no-pmc --no-cpu-baseline --no-host-input --no-copy-peak 2>/dev/null | tail -1 |
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w $*'.ljust(44), d['ms_per_step'], d['stage_ms_per_step'], d.get('job_roofline',{}).get('frac')); print('    ', {k['kernel'].replace('_kernel','')[:26]: k['ms_per_step'] for k in d['kernels']})"
}
run config5v FQD_AB=base
run config5 FQD_AB=base
