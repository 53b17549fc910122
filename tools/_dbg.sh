cd /root/repo
for w in config4 config3_skew config3 config2; do
  echo "== $w"
  FQD_DEBUG=1 FQD_UF_NO_SAMPLING=1 python bench.py --workload $w --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-host-input --no-copy-peak 2>&1 | grep "union-find" | tail -2
done
