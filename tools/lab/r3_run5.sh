#!/bin/bash
# round 3, run 5: lanes / k_verify-workgroup sweep with the early exit, rank simulation at worlds 1-8 (baseline), and the
# rocprofv3 --kernel-trace --stats summary of the bench command
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$PWD
B="--steps 200 --repeats 3 --cpu-seconds 0 --no-pmc --no-hbm-point --no-time-to-register --no-parity --no-exclusive --no-full-count-mode"
for cfg in "S4P_LANES=6 S4P_VERIFY_BLOCKS=512" "S4P_LANES=8 S4P_VERIFY_BLOCKS=512" "S4P_LANES=6 S4P_VERIFY_BLOCKS=384" "S4P_LANES=8 S4P_VERIFY_BLOCKS=384" "S4P_LANES=6 S4P_VERIFY_BLOCKS=256" "S4P_LANES=5 S4P_VERIFY_BLOCKS=512" "S4P_LANES=6 S4P_VERIFY_BLOCKS=768"; do
  v=$(env $cfg timeout 200 python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],4), round(d['roofline']['per_launch']['avg_launch_ms'],4), d['stage_ms_per_step'])")
  echo "$cfg -> $v" >> gpurun_out/r3_run5_sweep.log
done
timeout 600 python tools/sim_world.py > gpurun_out/r3_sim_world_before.jsonl 2> gpurun_out/r3_sim_world.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3stats_bench -o r --output-format csv -- python $R/bench.py --repeats 1 --cpu-seconds 0 --no-parity --no-pmc --no-hbm-point --no-time-to-register --no-exclusive --no-full-count-mode > $R/gpurun_out/r3stats_bench.json 2> $R/gpurun_out/r3stats_bench.err
cd $R
cat gpurun_out/r3_run5_sweep.log; cat gpurun_out/r3_sim_world_before.jsonl
python - <<'PY'
import csv, glob, json
for f in sorted(glob.glob('gpurun_out/r3stats_bench/**/r_kernel_stats.csv', recursive=True)):
    print(f)
    for r in list(csv.DictReader(open(f)))[:8]:
        print('  ', r['Name'][:50], r['Calls'], r['AverageNs'], r['Percentage'])
try:
    d=json.load(open('gpurun_out/r3stats_bench.json')); print('bench under rocprof:', round(d['value']/1e6,2), d['roofline']['per_launch']['avg_launch_ms'])
except Exception as e: print('ERR', e)
PY
