#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box (run through gpurun from the repo root):
#   tools/refresh_profiles.sh rNN      -> gpurun_out/profiles_rNN/*, to be copied into profiles/
# Steps are joined so that a failing GPU step ends the script (no further GPU work after a failure).
set -o pipefail
R=${1:-r02}
OUT=gpurun_out/profiles_$R
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/${R}_bench_config3.json 2> $OUT/bench_config3.err || exit 1
echo "config3 done"
for w in config2 config4 config5 config5v; do
    python3 bench.py --workload $w --no-pmc > $OUT/${R}_bench_$w.json 2> $OUT/bench_$w.err || exit 1
    echo "$w done"
done
python3 bench.py --force-sharded --no-pmc --no-cpu-baseline --no-host-input --warmup 8 > $OUT/${R}_bench_config3_sharded_world1.json 2> $OUT/bench_sharded.err || exit 1
echo "sharded done"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o r -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-input --no-pmc --no-copy-peak > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.log 2>&1) || exit 1
cp $OUT/prof/r_kernel_stats.csv $OUT/${R}_kernel_stats_bench_config3.csv
echo "rocprof config3 done"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof5v -o r -- python3 $GRAFT_REPO_ROOT/bench.py --workload config5v --reads-per-gpu 10000000 --no-cpu-baseline --no-host-input --no-pmc --no-copy-peak --steps 5 --warmup 1 > $GRAFT_REPO_ROOT/$OUT/rocprof_bench_5v.log 2>&1) || exit 1
cp $OUT/prof5v/r_kernel_stats.csv $OUT/${R}_kernel_stats_bench_config5v_10M.csv
echo "rocprof config5v (10 M ragged reads) done"
python3 tools/pmc_per_kernel.py --workload config3 --out $OUT/${R}_pmc_per_kernel_config3.json > /dev/null 2> $OUT/pmc.err || exit 1
echo "pmc done"
rm -rf $OUT/prof $OUT/prof5v
