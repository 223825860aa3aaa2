#!/bin/bash
# round 5, run 13: tuning of the group pipeline (adaptive k_prep / k_quads grids are in): lanes x group, k_verify workgroup shapes
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
export S4P_WAIT_TIMEOUT_S=20
O=gpurun_out/r5_run13; mkdir -p $O
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout -s KILL 60 python tools/r5/tp_probe.py 300 "$tag" > $O/tp_$tag.json 2> $O/tp_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$O/tp_$tag.json")); print(d["tag"], d["runs"][0]["mcand_per_s"], d["runs"][0]["us_per_base"], d["runs"][0]["wait_us"], d["best_count"], d["cand"])
except Exception as e: print("$tag failed", e)
PY
}
run l12g3 S4P_LANES=12 S4P_GROUP=3
run l12g3b S4P_LANES=12 S4P_GROUP=3
run l9g3 S4P_LANES=9 S4P_GROUP=3
run l12g2 S4P_LANES=12 S4P_GROUP=2
run l12g3_t384b512 S4P_LANES=12 S4P_GROUP=3 S4P_VERIFY_THREADS=384 S4P_VERIFY_BLOCKS=512
run l12g3_t512b256 S4P_LANES=12 S4P_GROUP=3 S4P_VERIFY_THREADS=512 S4P_VERIFY_BLOCKS=256
run l12g3_t512b512 S4P_LANES=12 S4P_GROUP=3 S4P_VERIFY_THREADS=512 S4P_VERIFY_BLOCKS=512
run l12g3_t1024b256 S4P_LANES=12 S4P_GROUP=3 S4P_VERIFY_THREADS=1024 S4P_VERIFY_BLOCKS=256
run l12g3_t768b512 S4P_LANES=12 S4P_GROUP=3 S4P_VERIFY_THREADS=768 S4P_VERIFY_BLOCKS=512
