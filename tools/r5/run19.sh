#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
export S4P_WAIT_TIMEOUT_S=20
O=gpurun_out/r5_run19; mkdir -p $O
run() { tag=$1; shift
  env "$@" timeout -s KILL 60 python tools/r5/tp_probe.py ${STEPS:-100} "$tag" > $O/tp_$tag.json 2> $O/tp_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$O/tp_$tag.json")); print(d["tag"], d["runs"][0]["mcand_per_s"], d["runs"][0]["us_per_base"], d["runs"][0]["wait_us"], d["best_count"], d["cand"])
except Exception as e: print("$tag failed", e)
PY
}
STEPS=20 run s20_l14g2_w5
STEPS=20 run s20_l14g2_w14 TP_WARMUP=14
STEPS=20 run s20_l14g2_w28 TP_WARMUP=28
STEPS=20 run s20_l8g2_w5 S4P_LANES=8
STEPS=20 run s20_l8g2_w16 S4P_LANES=8 TP_WARMUP=16
STEPS=20 run s20_l20g2_w5 S4P_LANES=16
STEPS=300 run s300_l8g2 S4P_LANES=8
STEPS=300 run s300_l10g2 S4P_LANES=10
