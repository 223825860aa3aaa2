#!/bin/bash
# first GPU pass of round 2: parity of the restructured kernels, A/B against the round-1 library, a short bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest (kernels, registration)" 
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_registration.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r2_pytest1.log
echo "== A/B"
for arm in "scratch/libbase_r01.so 1 1" "scratch/libbase_r01.so 3 1" "super4pcs_amd/lib/libsuper4pcs_amd.so 1 1" "super4pcs_amd/lib/libsuper4pcs_amd.so 3 1" "super4pcs_amd/lib/libsuper4pcs_amd.so 1 0" "super4pcs_amd/lib/libsuper4pcs_amd.so 3 0" "super4pcs_amd/lib/libsuper4pcs_amd.so 2 1"; do
  set -- $arm
  S4P_LIB=$PWD/$1 S4P_LANES=$2 S4P_FUSED=$3 timeout 300 python tools/ab_one.py 100 3 2>&1 | tail -2 | tee -a gpurun_out/r2_ab1.log
done
echo "== bench (short)"
timeout 900 python bench.py --steps 100 --repeats 3 --cpu-seconds 4 --no-hbm-point > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err
tail -c 1500 gpurun_out/r2_bench1.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r2_bench1.json'))
    print({k:d[k] for k in ('value','ms_per_step','spread','parity')})
    print(d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['avg_launch_ms'], d['roofline']['pass_fractions'], d['roofline']['kbar'])
    print(d['stage_ms_per_step'], d['config']['time_to_register'])
    print(d['cpu_baseline'])
except Exception as e:
    print('bench json unreadable', e)
PY
