#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box (run through gpurun from the repo root):
#   tools/refresh_profiles.sh rNN      -> gpurun_out/profiles_rNN/*, to be copied into profiles/
# Steps are joined so that a failing GPU step ends the script (no further GPU work after a failure); every step
# prints a line, so a long run never looks hung.
set -o pipefail
R=${1:-r04}
PART=${2:-all}        # bench1 | bench2 | prof | all (three gpurun calls of <= 20 min each)
OUT=gpurun_out/profiles_$R
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$PART" = "bench1" ] || [ "$PART" = "all" ]; then
python3 bench.py > $OUT/${R}_bench_config3.json 2> $OUT/bench_config3.err || exit 1
echo "config3 done"
# the long-key shapes WITH the PMC passes (job_roofline.traffic), config 5 / 5v without (their child passes take minutes)
for w in config2 config4; do
    python3 bench.py --workload $w > $OUT/${R}_bench_$w.json 2> $OUT/bench_$w.err || exit 1
    echo "$w done"
done
fi
if [ "$PART" = "bench2" ] || [ "$PART" = "all" ]; then
for w in config5 config5v; do
    python3 bench.py --workload $w --no-pmc > $OUT/${R}_bench_$w.json 2> $OUT/bench_$w.err || exit 1
    echo "$w done"
done
python3 bench.py --workload config3_skew --no-pmc --steps 5 --warmup 2 > $OUT/${R}_bench_config3_skew.json 2> $OUT/bench_skew.err || exit 1
echo "config3_skew done"
python3 bench.py --force-sharded --no-pmc --no-cpu-baseline --no-host-input --warmup 8 > $OUT/${R}_bench_config3_sharded_world1.json 2> $OUT/bench_sharded.err || exit 1
python3 bench.py --force-sharded --workload config4 --steps 10 --warmup 3 --no-pmc --no-cpu-baseline --no-host-input > $OUT/${R}_bench_config4_sharded_world1.json 2> $OUT/bench_sharded4.err || exit 1
echo "sharded done"
# the skewed model: adjacency, and at distance 2 beside the uniform job of that shape (no CPU baseline: the oracle's
# quadratic dissection of the 65 536-key component takes minutes at d = 2)
python3 bench.py --workload config3_skew_adj --no-pmc --steps 5 --warmup 2 --no-cpu-baseline --no-host-input > $OUT/${R}_bench_config3_skew_adj.json 2> $OUT/bench_skew_adj.err || exit 1
python3 bench.py --workload config4_skew --no-pmc --steps 5 --warmup 2 --no-cpu-baseline --no-host-input > $OUT/${R}_bench_config4_skew.json 2> $OUT/bench_skew4.err || exit 1
echo "skew variants done"
hipcc --offload-arch=gfx950 -O3 -o /tmp/mb tools/microbench_patterns.hip 2> /dev/null && /tmp/mb > $OUT/${R}_microbench_patterns.json || exit 1
echo "microbench done"
fi
[ "$PART" = "bench1" ] && exit 0
[ "$PART" = "bench2" ] && exit 0
for w in config3 config2 config4 config3_skew config4_skew; do
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$w -o r -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 10 --no-cpu-baseline --no-host-input --no-pmc --no-copy-peak > $GRAFT_REPO_ROOT/$OUT/rocprof_bench_$w.log 2>&1) || exit 1
    cp $OUT/prof_$w/r_kernel_stats.csv $OUT/${R}_kernel_stats_bench_$w.csv
    rm -rf $OUT/prof_$w
    echo "rocprof $w done"
done
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof5v -o r -- python3 $GRAFT_REPO_ROOT/bench.py --workload config5v --reads-per-gpu 10000000 --no-cpu-baseline --no-host-input --no-pmc --no-copy-peak --steps 5 --warmup 1 > $GRAFT_REPO_ROOT/$OUT/rocprof_bench_5v.log 2>&1) || exit 1
cp $OUT/prof5v/r_kernel_stats.csv $OUT/${R}_kernel_stats_bench_config5v_10M.csv
rm -rf $OUT/prof5v
echo "rocprof config5v (10 M ragged reads) done"
for w in config3 config2 config4; do
    python3 tools/pmc_per_kernel.py --workload $w --out $OUT/${R}_pmc_per_kernel_$w.json > /dev/null 2> $OUT/pmc_$w.err || exit 1
    echo "pmc $w done"
done
python3 tools/sq_per_kernel.py --workload config3 --out $OUT/${R}_sq_per_kernel_config3.json > /dev/null 2> $OUT/sq.err || exit 1
echo "sq done"
