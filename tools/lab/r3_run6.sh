#!/bin/bash
# round 3, run 6: lazy set-2 preparation in k_quads, selection pipeline (drawer / evaluators / assembler), batched device
# selection, 256 k_verify workgroups: parity tests, bench, rank simulation, clean kernel stats of the bench command
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 -k "not 20000" > gpurun_out/r3_run6_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_run6_tests.log
B="--steps 200 --repeats 3 --cpu-seconds 0 --no-pmc --no-hbm-point --no-time-to-register --no-parity --no-exclusive --no-instrumented"
for cfg in "S4P_X=0" "S4P_VERIFY_BLOCKS=512" "S4P_LIB=$R/scratch/libr3_before_lazy.so"; do
  v=$(env $cfg timeout 200 python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],4), d['config']['full_count_mode'] and round(d['config']['full_count_mode']['value']/1e6,2), d['stage_ms_per_step'])")
  echo "$cfg -> $v" >> gpurun_out/r3_run6_ab.log
done
timeout 600 python tools/sim_world.py > gpurun_out/r3_sim_world_after.jsonl 2> gpurun_out/r3_sim_world.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3stats6_bench -o r --output-format csv -- python $R/bench.py --repeats 1 --cpu-seconds 0 --no-parity --no-pmc --no-hbm-point --no-time-to-register --no-exclusive --no-full-count-mode --no-instrumented > $R/gpurun_out/r3stats6_bench.json 2> $R/gpurun_out/r3stats6_bench.err
cd $R
tail -8 gpurun_out/r3_run6_tests.log; cat gpurun_out/r3_run6_ab.log; cat gpurun_out/r3_sim_world_after.jsonl
python - <<'PY'
import csv, glob, json
for f in sorted(glob.glob('gpurun_out/r3stats6_bench/**/r_kernel_stats.csv', recursive=True)):
    for r in list(csv.DictReader(open(f)))[:7]:
        print('  ', r['Name'][:50], r['Calls'], r['AverageNs'], r['Percentage'])
try:
    d=json.load(open('gpurun_out/r3stats6_bench.json')); print('bench under rocprof:', round(d['value']/1e6,2), d['roofline']['per_launch']['avg_launch_ms'])
except Exception as e: print('ERR', e)
PY
