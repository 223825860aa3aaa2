#!/bin/bash
# final pass of round 2: whole -m gpu suite, smoke, bench (+ rocprof kernel stats of the same command)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== full pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/r2_pytest12.log
echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== bench"
timeout 1200 python bench.py > gpurun_out/r2_bench12.json 2> gpurun_out/r2_bench12.err
tail -c 600 gpurun_out/r2_bench12.err
echo "== rocprof stats of the bench command"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2stats_bench -o r --output-format csv -- python $R/bench.py --repeats 1 --cpu-seconds 0 --no-parity --no-pmc --no-hbm-point --no-time-to-register > $R/gpurun_out/r2stats_bench.log 2>&1
S4P_LANES=1 timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2stats_bench_l1 -o r --output-format csv -- python $R/bench.py --repeats 1 --cpu-seconds 0 --no-parity --no-pmc --no-hbm-point --no-time-to-register > $R/gpurun_out/r2stats_bench_l1.log 2>&1
cd $R
python - <<'PY'
import json, csv, glob
try:
    d=json.load(open('gpurun_out/r2_bench12.json'))
    print({k:d[k] for k in ('value','ms_per_step','spread')}, d['parity']['mismatches'])
    r=d['roofline']; print({k:r[k] for k in ('achieved','frac','traffic','avg_launch_ms')}, r['hbm_bound_point'].get('frac'))
    print(d['k_apply'])
    print(d['config']['time_to_register'])
except Exception as e:
    print('bench json unreadable', e)
for f in sorted(glob.glob('gpurun_out/r2stats_bench*/**/r_kernel_stats.csv', recursive=True)):
    print(f)
    for r in list(csv.DictReader(open(f)))[:6]:
        print('  ', r['Name'][:44], r['Calls'], r['AverageNs'], r['Percentage'])
for f in ('gpurun_out/r2stats_bench.log','gpurun_out/r2stats_bench_l1.log'):
    print(open(f).read()[-400:])
PY
