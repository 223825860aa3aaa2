#!/bin/bash
# round 4, final pass on the shipped commit: the whole GPU suite; the bench line with default flags (+ the rocprofv3 CSVs it
# collects itself); the driver's command; a kernel trace of the bench command; per-kernel counters with one base in flight;
# init / time-to-register of configs[2-4]; the drop-in (facade) against the C ABI; cold HBM points of k_apply and the sampler
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_final; mkdir -p $O $O/bench_final
cp super4pcs_amd/lib/BUILD_INFO.json $O/BUILD_INFO.json
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=10 > $O/gpu_tests.log 2>&1
echo "pytest rc=$?" >> $O/gpu_tests.log
tail -16 $O/gpu_tests.log
timeout 900 python bench.py --profile-dir $O/bench_final > $O/bench_final.json 2> $O/bench_final.err
echo "bench rc=$?" > $O/log.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
echo "bench20 rc=$?" >> $O/log.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/stats" -o r --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --repeats 1 --cpu-seconds 0 --no-parity --no-pmc --no-hbm-point --no-time-to-register --no-exclusive --no-instrumented --no-full-count-mode --no-stage-pass --no-extra > "$GRAFT_REPO_ROOT/$O/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$O/stats.err" )
echo "rocprof rc=$?" >> $O/log.txt
timeout 400 python tools/r4/prof_kernels.py $O --lanes 1 --steps 100 --passes trace,sq,sq2,tcc,fetch,write > $O/prof_kernels.log 2>&1
echo "prof kernels rc=$?" >> $O/log.txt
timeout 500 python tools/r4/init_timing.py > $O/init_and_time_to_register.jsonl 2> $O/init_timing.err
echo "init timing rc=$?" >> $O/log.txt
g++ -O2 -std=c++17 -Iinclude tests/facade_app/timing.cpp -Lsuper4pcs_amd/lib -lsuper4pcs_amd -Wl,-rpath,$GRAFT_REPO_ROOT/super4pcs_amd/lib -o /tmp/facade_timing && timeout 200 /tmp/facade_timing 1000000 0.004 2000 0.5 > $O/facade_timing.json 2> $O/facade_timing.err
echo "facade timing rc=$?" >> $O/log.txt
python - <<'PY' >> gpurun_out/r04_final/log.txt
import json, glob, csv
O='gpurun_out/r04_final'
for f in ('bench_final','bench_driver_command','bench_under_rocprof'):
    try:
        line=[l for l in open('%s/%s.json'%(O,f)).read().splitlines() if l.startswith('{"metric')][-1]
        d=json.loads(line); r=d['roofline']
        print(f, round(d['value']/1e6,2),'M cand/s', round(d['ms_per_step'],4),'ms/step', [round(d['spread'][k]/1e6,1) for k in ('min','max')], 'full', d['config']['full_count_mode'] and round(d['config']['full_count_mode']['value']/1e6,2), 'parity', d['parity'] and (d['parity'].get('bases'), d['parity'].get('mismatches'), d['parity'].get('failed')))
        print('   frac', r['frac'], r['binding'], 'traffic', r['traffic'], 'per_launch', r['per_launch']['avg_launch_ms'], r['per_launch'].get('exclusive') and r['per_launch']['exclusive']['avg_launch_ms'])
        print('   hbm point', r['hbm_bound_point'] and {k:r['hbm_bound_point'].get(k) for k in ('kernel_ms','measured_GBps','frac','count_mismatches','error')})
        print('   ttr', d['config']['time_to_register'] and d['config']['time_to_register']['seconds'], 'cpu', d.get('cpu_baseline') and (d['cpu_baseline']['value'], d['cpu_baseline']['openmp_all_cores']['value']), 'extra', d.get('extra'))
        print('   provenance', d.get('provenance'))
    except Exception as e: print(f,'ERR',repr(e))
for f in glob.glob(O+'/stats/**/r_kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print('  ', r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
print(open(O+'/init_and_time_to_register.jsonl').read()[:4000])
print(open(O+'/facade_timing.json').read())
print(open(O+'/prof_kernels.log').read()[-1500:])
PY
cat $O/log.txt
