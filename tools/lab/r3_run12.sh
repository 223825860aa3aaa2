#!/bin/bash
# round 3, final pass: the bench line with default flags (+ the rocprofv3 CSVs it collects itself), the driver's command, a
# kernel trace of the bench command restricted to the default-configuration launches, init / time-to-register of configs[2-4]
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r03_bench_final
timeout 900 python bench.py --profile-dir gpurun_out/r03_bench_final > gpurun_out/r03_bench_final.json 2> gpurun_out/r03_bench_final.err
echo "bench rc=$?" > gpurun_out/r3_run12.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_driver_command.json 2> gpurun_out/r03_bench_driver_command.err
echo "bench20 rc=$?" >> gpurun_out/r3_run12.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r03_stats_final" -o r --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --repeats 1 --cpu-seconds 0 --no-parity --no-pmc --no-hbm-point --no-time-to-register --no-exclusive --no-instrumented --no-full-count-mode --no-stage-pass > "$GRAFT_REPO_ROOT/gpurun_out/r03_bench_under_rocprof_final.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/r03_stats_final.err" )
echo "rocprof rc=$?" >> gpurun_out/r3_run12.log
timeout 600 python tools/r3_init_timing.py > gpurun_out/r03_init_and_time_to_register_final.jsonl 2> gpurun_out/r3_init_timing.err
echo "init timing rc=$?" >> gpurun_out/r3_run12.log
python - <<'PY' >> gpurun_out/r3_run12.log
import json, glob
for f in ('r03_bench_final','r03_bench_driver_command','r03_bench_under_rocprof_final'):
    try:
        line=[l for l in open('gpurun_out/%s.json'%f).read().splitlines() if l.startswith('{"metric')][-1]
        d=json.loads(line)
        r=d['roofline']
        print(f, round(d['value']/1e6,2),'M cand/s', round(d['ms_per_step'],4),'ms/step', [round(d['spread'][k]/1e6,1) for k in ('min','max')], 'full', d['config']['full_count_mode'] and round(d['config']['full_count_mode']['value']/1e6,2), 'parity', d['parity'] and (d['parity'].get('bases'), d['parity'].get('mismatches'), d['parity'].get('failed')))
        print('   frac', round(r['frac'],3), r['binding'], 'traffic', r['traffic'], 'per_launch', r['per_launch']['avg_launch_ms'], r['per_launch'].get('exclusive') and r['per_launch']['exclusive']['avg_launch_ms'], 'hbm point', r['hbm_bound_point'] and {k:r['hbm_bound_point'].get(k) for k in ('kernel_ms','measured_GBps','frac','count_mismatches','error')})
        print('   ttr', d['config']['time_to_register'] and d['config']['time_to_register']['seconds'], 'cpu', d.get('cpu_baseline') and (d['cpu_baseline']['value'], d['cpu_baseline']['openmp_all_cores']['value']))
    except Exception as e: print(f,'ERR',e)
import csv
for f in glob.glob('gpurun_out/r03_stats_final/**/r_kernel_stats.csv', recursive=True):
    print(f)
    for r in list(csv.DictReader(open(f)))[:7]:
        print('  ', r['Name'][:50], r['Calls'], r['AverageNs'], r['Percentage'])
print(open('gpurun_out/r03_init_and_time_to_register_final.jsonl').read()[:3000])
PY
cat gpurun_out/r3_run12.log
