#!/bin/bash
# round 6, final pass, part A (on the sources that ship): the GPU suite in two halves side by side, the driver's bench command,
# a kernel trace of the bench command with the line printed under it
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
export S4P_WAIT_TIMEOUT_S=600
O=gpurun_out/r06_final; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1   # a no-op when the snapshot is up to date; never two builds side by side
cp super4pcs_amd/lib/BUILD_INFO.json $O/BUILD_INFO.json
( timeout -s KILL 900 python -m pytest tests/test_gpu_configs.py -m gpu -q --timeout 700 --durations=8 -p no:cacheprovider > $O/gpu_tests_configs.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests_configs.log ) &
( timeout -s KILL 900 python -m pytest tests --ignore=tests/test_gpu_configs.py -m gpu -q --timeout 600 --durations=8 -p no:cacheprovider > $O/gpu_tests_rest.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests_rest.log ) &
wait
tail -6 $O/gpu_tests_configs.log > $O/log_a.txt; tail -6 $O/gpu_tests_rest.log >> $O/log_a.txt
timeout -s KILL 420 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
echo "bench20 rc=$?" >> $O/log_a.txt
( cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/stats" -o r --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --repeats 1 --cpu-seconds 0 --no-parity --no-pmc --no-hbm-point --no-time-to-register --no-exclusive --no-instrumented --no-full-count-mode --no-stage-pass --no-extra > "$GRAFT_REPO_ROOT/$O/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$O/stats.err" )
echo "rocprof rc=$?" >> $O/log_a.txt
python - <<'PY' >> gpurun_out/r06_final/log_a.txt
import json, glob, csv
O='gpurun_out/r06_final'
for f in ('bench_driver_command','bench_under_rocprof'):
    try:
        line=[l for l in open('%s/%s.json'%(O,f)).read().splitlines() if l.startswith('{"metric')][-1]
        d=json.loads(line); r=d['roofline']
        print(f, round(d['value']/1e6,2),'M cand/s', round(d['ms_per_step'],4),'ms/step', [round(d['spread'][k]/1e6,1) for k in ('min','max')], 'full', d.get('value_full_count') and round(d['value_full_count']/1e6,2), 'parity', d['parity'] and (d['parity'].get('bases'), d['parity'].get('mismatches'), d['parity'].get('failed')))
        print('   frac', r['frac'], r['binding'], 'per_launch', r['per_launch']['avg_launch_ms'], r['per_launch']['launches'], r['per_launch'].get('exclusive') and r['per_launch']['exclusive']['avg_launch_ms'])
        print('   extra', d.get('extra') and (d['extra'].get('value'), d['extra'].get('parity') and d['extra']['parity'].get('mismatches'), d['extra'].get('error')))
    except Exception as e: print(f,'ERR',repr(e))
for f in glob.glob(O+'/stats/**/r_kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print('  ', r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
PY
cat $O/log_a.txt
