#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
export S4P_WAIT_TIMEOUT_S=20
O=gpurun_out/r5_run16; mkdir -p $O
run() { tag=$1; shift
  env "$@" timeout -s KILL 60 python tools/r5/tp_probe.py 300 "$tag" > $O/tp_$tag.json 2> $O/tp_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$O/tp_$tag.json")); print(d["tag"], d["runs"][0]["mcand_per_s"], d["runs"][0]["us_per_base"], d["runs"][0]["wait_us"], d["best_count"], d["cand"])
except Exception as e: print("$tag failed", e)
PY
}
run default
run prio_q16 S4P_VERIFY_PRIO=1 GPU_MAX_HW_QUEUES=16
run prio_q16_l12 S4P_VERIFY_PRIO=1 GPU_MAX_HW_QUEUES=16 S4P_LANES=12
run prio_q16_l16 S4P_VERIFY_PRIO=1 GPU_MAX_HW_QUEUES=16 S4P_LANES=16
run prio_q8 S4P_VERIFY_PRIO=1
run vb192 S4P_VERIFY_BLOCKS=192
run vb320 S4P_VERIFY_BLOCKS=320
run default2
