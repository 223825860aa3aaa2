#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== A/B"
run() { S4P_LANES=$1 timeout 300 python tools/ab_one.py 100 3 2>&1 | tail -1 | tee -a gpurun_out/r2_ab10.log; }
run 1
run 3
S4P_NO_QLDS=1 run 1
echo "== full pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r2_pytest10.log
echo "== bench"
timeout 900 python bench.py > gpurun_out/r2_bench10.json 2> gpurun_out/r2_bench10.err
tail -c 800 gpurun_out/r2_bench10.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r2_bench10.json'))
    print({k:d[k] for k in ('value','ms_per_step','spread','parity')})
    r=d['roofline']; print({k:r[k] for k in ('achieved','frac','traffic','traffic_note','avg_launch_ms','pass_fractions','kbar','l2_sweep','hbm_bound_point')})
    print(d['stage_ms_per_step'], d['config']['time_to_register'])
    print(d['cpu_baseline'])
except Exception as e:
    print('bench json unreadable', e)
PY
