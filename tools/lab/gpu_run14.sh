#!/bin/bash
# round 2, pass 14: packed (4-byte) point lists in the exact stage of k_verify: parity, then A/B against the float lists
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== pytest kernels + registration (packed lists)"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_registration.py -m gpu -x -q 2>&1 | tail -5
echo "== A/B"
run() { S4P_LIB=$R/$1 S4P_LANES=$2 timeout 300 python tools/ab_one.py 100 3 2>&1 | tail -1 | tee -a gpurun_out/r2_ab14.log; }
run scratch/libfloatlists.so 1
run super4pcs_amd/lib/libsuper4pcs_amd.so 1
run super4pcs_amd/lib/libsuper4pcs_amd.so 3
run scratch/libfloatlists.so 3
echo "== settled fraction"
python - <<'PY'
import numpy as np
from super4pcs_amd import capi, datasets
P, Q, _ = datasets.bumpy_pair(1_000_000, overlap=0.5, delta=0.004, seed=20140814)
m = capi.Matcher(capi.make_options(0.004, 0.5, 2000), device=0, max_pairs=8 << 20, max_quads=64 << 20)
m.init_full(P, Q)
m.profile_enable(True, True)
m.profile_get(reset=True)
m.perform_n_steps(20)
p = m.profile_get()
print({"queries": p.verify_queries, "l2_pass": p.verify_l2_pass, "point_tests": p.verify_point_tests, "settled": p.verify_settled,
       "settled_per_l2": p.verify_settled / max(p.verify_l2_pass, 1)})
PY
