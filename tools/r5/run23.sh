#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
export S4P_WAIT_TIMEOUT_S=60
O=gpurun_out/r5_run23; mkdir -p $O
run() { tag=$1; shift
  env "$@" timeout -s KILL 90 python tools/r5/tp_probe.py ${STEPS:-300} "$tag" > $O/tp_$tag.json 2> $O/tp_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$O/tp_$tag.json")); print(d["tag"], d["runs"][0]["mcand_per_s"], d["runs"][0]["us_per_base"], d["runs"][0]["wait_us"], d["best_count"], d["cand"])
except Exception as e: print("$tag failed", e)
PY
}
STEPS=20 run s20_l20g2_q16 S4P_LANES=20 GPU_MAX_HW_QUEUES=16
STEPS=20 run s20_l20g2_q12 S4P_LANES=20 GPU_MAX_HW_QUEUES=12
STEPS=20 run s20_l24g3_q8 S4P_LANES=24 S4P_GROUP=3
STEPS=20 run s20_l21g3_q8 S4P_LANES=21 S4P_GROUP=3
STEPS=20 run s20_l14g2_q8
run s300_l20g2_q16 S4P_LANES=20 GPU_MAX_HW_QUEUES=16
run s300_l20g2_q12 S4P_LANES=20 GPU_MAX_HW_QUEUES=12
run s300_l21g3_q8 S4P_LANES=21 S4P_GROUP=3
