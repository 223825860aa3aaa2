#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
export S4P_WAIT_TIMEOUT_S=20
O=gpurun_out/r5_run18; mkdir -p $O
run() { tag=$1; shift
  env "$@" timeout -s KILL 60 python tools/r5/tp_probe.py ${STEPS:-100} "$tag" > $O/tp_$tag.json 2> $O/tp_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$O/tp_$tag.json")); print(d["tag"], d["runs"][0]["mcand_per_s"], d["runs"][0]["us_per_base"], d["runs"][0]["wait_us"], d["best_count"], d["cand"])
except Exception as e: print("$tag failed", e)
PY
}
run full_l14g2 TP_FULL=1
run full_l6g1 TP_FULL=1 S4P_LANES=6 S4P_GROUP=1
run full_l6g2 TP_FULL=1 S4P_LANES=6 S4P_GROUP=2
run full_l8g2 TP_FULL=1 S4P_LANES=8 S4P_GROUP=2
run full_l14g1 TP_FULL=1 S4P_LANES=14 S4P_GROUP=1 GPU_MAX_HW_QUEUES=16
STEPS=20 run s20_l14g2
STEPS=20 run s20_l8g2 S4P_LANES=8 S4P_GROUP=2
STEPS=20 run s20_l6g1 S4P_LANES=6 S4P_GROUP=1
STEPS=20 run s20_l12g3 S4P_LANES=12 S4P_GROUP=3
STEPS=20 run s20_l14g1q16 S4P_LANES=14 S4P_GROUP=1 GPU_MAX_HW_QUEUES=16
