set -e
mkdir -p gpurun_out/t
for w in config4_skew config3_skew config3_skew_adj; do
  timeout -k 10 300 python3 bench.py --workload $w --no-pmc --steps 10 --warmup 3 --no-cpu-baseline --no-host-input > gpurun_out/t/b_$w.json 2>> gpurun_out/t/err.log
  python3 -c "
import json,sys;d=json.loads([l for l in open('gpurun_out/t/b_$w.json') if l.startswith('{')][-1]);print('$w',d['ms_per_step'],d['t_first_call_ms'],d['stage_ms_per_step'],{k:v for k,v in d['route'].items() if v and k.startswith('search')})"
done
FQD_SKEW="hot=0.02,ladder=0,lowc_every=100" timeout -k 10 300 python3 bench.py --workload config4_skew --no-pmc --steps 10 --warmup 3 --no-cpu-baseline --no-host-input > gpurun_out/t/b_c4s_noladder.json 2>> gpurun_out/t/err.log
python3 -c "
import json,sys;d=json.loads([l for l in open('gpurun_out/t/b_c4s_noladder.json') if l.startswith('{')][-1]);print('c4skew no ladder',d['ms_per_step'],d['stage_ms_per_step'],{k:v for k,v in d['route'].items() if v and k.startswith('search')})"
FQD_GROUP_TILE_BUDGET=0 FQD_SKEW="hot=0.02,ladder=0,lowc_every=100" timeout -k 10 300 python3 bench.py --workload config4_skew --no-pmc --steps 10 --warmup 3 --no-cpu-baseline --no-host-input > gpurun_out/t/b_c4s_noladder_ref.json 2>> gpurun_out/t/err.log
python3 -c "
import json,sys;d=json.loads([l for l in open('gpurun_out/t/b_c4s_noladder_ref.json') if l.startswith('{')][-1]);print('c4skew no ladder, refinement',d['ms_per_step'],d['stage_ms_per_step'],{k:v for k,v in d['route'].items() if v and k.startswith('search')})"
timeout -k 10 1100 python -m pytest tests/test_hip_parity.py -q -x 2>&1 | tail -5
