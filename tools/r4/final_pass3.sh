#!/bin/bash
# round 4, last full pass, on the commit that ships (after the batched fourth-point scan, the 128 k-point apply chunks and
# the host-side traces; the four kernels of a base are instruction-identical to the build of final_pass.sh, so its
# per-kernel counters stay): the GPU suite in two halves side by side, the bench line with default flags (+ the rocprofv3
# CSVs it collects itself), the driver's command, a kernel trace of the bench command, init / time-to-register with the
# init breakdown, the drop-in against the C ABI, the simulated world, the selection probe
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_final3; mkdir -p $O $O/bench_final
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1   # a no-op when the snapshot is up to date; never two builds side by side
cp super4pcs_amd/lib/BUILD_INFO.json $O/BUILD_INFO.json
( timeout 800 python -m pytest tests/test_gpu_configs.py -m gpu -q --timeout 600 --durations=8 -p no:cacheprovider > $O/gpu_tests_configs.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests_configs.log ) &
( timeout 800 python -m pytest tests --ignore=tests/test_gpu_configs.py -m gpu -q --timeout 600 --durations=8 -p no:cacheprovider > $O/gpu_tests_rest.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests_rest.log ) &
wait
tail -14 $O/gpu_tests_configs.log > $O/log.txt; tail -14 $O/gpu_tests_rest.log >> $O/log.txt
timeout 900 python bench.py --profile-dir $O/bench_final > $O/bench_final.json 2> $O/bench_final.err
echo "bench rc=$?" >> $O/log.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
echo "bench20 rc=$?" >> $O/log.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/stats" -o r --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --repeats 1 --cpu-seconds 0 --no-parity --no-pmc --no-hbm-point --no-time-to-register --no-exclusive --no-instrumented --no-full-count-mode --no-stage-pass --no-extra > "$GRAFT_REPO_ROOT/$O/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$O/stats.err" )
echo "rocprof rc=$?" >> $O/log.txt
S4P_TRACE_INIT=1 timeout 300 python tools/r4/init_timing.py 2 3 4s > $O/init_and_time_to_register.jsonl 2> $O/init_timing.err
echo "init timing rc=$?" >> $O/log.txt
g++ -O2 -std=c++17 -Iinclude tests/facade_app/timing.cpp -Lsuper4pcs_amd/lib -lsuper4pcs_amd -Wl,-rpath,$GRAFT_REPO_ROOT/super4pcs_amd/lib -o /tmp/facade_timing
for i in 1 2 3 4 5; do timeout 100 /tmp/facade_timing 1000000 0.004 2000 0.5 >> $O/facade_timing.json 2>> $O/facade_timing.err; done
echo "facade timing rc=$?" >> $O/log.txt
S4P_TRACE_CHAIN=1 timeout 200 python tools/sim_world.py > $O/sim_world.jsonl 2> $O/sim_world.err
echo "sim_world rc=$?" >> $O/log.txt
timeout 100 python tools/r4/select_probe.py > $O/select_probe.json 2> $O/select_probe.err
echo "select probe rc=$?" >> $O/log.txt
python - <<'PY' >> gpurun_out/r04_final3/log.txt
import json, glob, csv
O='gpurun_out/r04_final3'
for f in ('bench_final','bench_driver_command','bench_under_rocprof'):
    try:
        line=[l for l in open('%s/%s.json'%(O,f)).read().splitlines() if l.startswith('{"metric')][-1]
        d=json.loads(line); r=d['roofline']
        print(f, round(d['value']/1e6,2),'M cand/s', round(d['ms_per_step'],4),'ms/step', [round(d['spread'][k]/1e6,1) for k in ('min','max')], 'full', d['config']['full_count_mode'] and round(d['config']['full_count_mode']['value']/1e6,2), 'parity', d['parity'] and (d['parity'].get('bases'), d['parity'].get('mismatches'), d['parity'].get('failed')))
        print('   frac', r['frac'], r['binding'], 'traffic', r['traffic'], 'per_launch', r['per_launch']['avg_launch_ms'], r['per_launch'].get('exclusive') and r['per_launch']['exclusive']['avg_launch_ms'])
        print('   ttr', d['config']['time_to_register'] and d['config']['time_to_register']['seconds'], 'cpu', d.get('cpu_baseline') and (d['cpu_baseline']['value'], d['cpu_baseline']['openmp_all_cores']['value']), 'extra', d.get('extra') and d['extra'].get('value'))
        print('   provenance', d.get('provenance'))
    except Exception as e: print(f,'ERR',repr(e))
for f in glob.glob(O+'/stats/**/r_kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print('  ', r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
for f in ('init_and_time_to_register.jsonl','facade_timing.json','sim_world.jsonl','select_probe.json'):
    try: print(open(O+'/'+f).read()[:6000])
    except Exception as e: print(f, 'ERR', repr(e))
PY
grep s4p_trace $O/init_timing.err $O/sim_world.err >> $O/log.txt
cat $O/log.txt
