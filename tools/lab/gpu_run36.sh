#!/bin/bash
# round 2, pass 36: current tree (hash-slot guard, bench changes): kernel + registration tests, bench with 448 / 512 verify workgroups
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_registration.py tests/test_gpu_select.py -m gpu -x -q 2>&1 | tail -3
q() { env $1 timeout 600 python bench.py --no-pmc --no-hbm-point --cpu-seconds 0 --no-time-to-register --repeats 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'] / 1e6, 2), [round(d['spread'][k] / 1e6, 1) for k in ('min', 'max')], d['parity']['mismatches'])"; }
q S4P_VERIFY_BLOCKS=512
q S4P_VERIFY_BLOCKS=448
q S4P_VERIFY_BLOCKS=384
q S4P_VERIFY_BLOCKS=512
