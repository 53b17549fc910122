cd /root/repo
for m in all dominant none dominant all; do python bench.py --steps 30 --warmup 8 --no-pmc --no-cpu-baseline --no-host-input --no-copy-peak --kernel-timers $m 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$m', d['ms_per_step'])"; done
