#!/bin/bash
# round 5, run 21: set-1 preparation inside k_pairs2 (S4P_FUSE_PREP): parity tests, then A/B
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
export S4P_WAIT_TIMEOUT_S=60
O=gpurun_out/r5_run21; mkdir -p $O
( timeout -s KILL 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_registration.py -m gpu -x -q --timeout 300 -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log ); tail -4 $O/tests.log
run() { tag=$1; shift
  env "$@" timeout -s KILL 60 python tools/r5/tp_probe.py ${STEPS:-300} "$tag" > $O/tp_$tag.json 2> $O/tp_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$O/tp_$tag.json")); print(d["tag"], d["runs"][0]["mcand_per_s"], d["runs"][0]["us_per_base"], d["runs"][0]["wait_us"], d["best_count"], d["cand"])
except Exception as e: print("$tag failed", e)
PY
}
run fuse1_a
run fuse0_a S4P_FUSE_PREP=0
run fuse1_b
run fuse0_b S4P_FUSE_PREP=0
STEPS=20 run s20_fuse1
STEPS=20 run s20_fuse0 S4P_FUSE_PREP=0
