#!/bin/bash
# round 2, pass 37: fewer k_verify workgroups (the other lanes' kernels get CUs of their own)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
q() { env $1 timeout 600 python bench.py --no-pmc --no-hbm-point --cpu-seconds 0 --no-time-to-register --repeats 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'] / 1e6, 2), [round(d['spread'][k] / 1e6, 1) for k in ('min', 'max')], d['parity']['mismatches'], round(d['roofline']['exclusive']['avg_launch_ms'], 4))"; }
q S4P_VERIFY_BLOCKS=256
q S4P_VERIFY_BLOCKS=320
q S4P_VERIFY_BLOCKS=384
