#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
export S4P_WAIT_TIMEOUT_S=60
export S4P_TRACE_LAUNCH=1
O=gpurun_out/r5_run22; mkdir -p $O
run() { tag=$1; shift
  env "$@" timeout -s KILL 60 python tools/r5/tp_probe.py ${STEPS:-300} "$tag" > $O/tp_$tag.json 2> $O/tp_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$O/tp_$tag.json")); print(d["tag"], d["runs"][0]["mcand_per_s"], d["runs"][0]["us_per_base"], d["runs"][0]["wait_us"], d["best_count"], d["cand"])
except Exception as e: print("$tag failed", e)
PY
  grep -o '"group_launches.*' $O/tp_$tag.err | tail -1
}
STEPS=20 run s20_fuse1_a
STEPS=20 run s20_fuse0_a S4P_FUSE_PREP=0
STEPS=20 run s20_fuse1_b
STEPS=20 run s20_fuse0_b S4P_FUSE_PREP=0
STEPS=20 run s20_fuse1_c
STEPS=20 run s20_fuse0_c S4P_FUSE_PREP=0
run s300_fuse1
run s300_fuse0 S4P_FUSE_PREP=0
