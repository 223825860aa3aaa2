#!/bin/bash
# round 5, run 2: is the pipeline bound by the launch thread or by the device?  1, 2 and 3 matchers side by side on one GPU
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5_run2; mkdir -p $O
timeout 100 python tools/r5/tp_probe.py 300 alone > $O/alone.json 2> $O/alone.err
S4P_LANES=8 timeout 100 python tools/r5/tp_probe.py 300 lanes8 > $O/lanes8.json 2> $O/lanes8.err
T=$(python -c "import time; print(time.time()+25)")
for k in 1 2; do TP_START_AT=$T timeout 100 python tools/r5/tp_probe.py 600 two_$k > $O/two_$k.json 2> $O/two_$k.err & done; wait
T=$(python -c "import time; print(time.time()+25)")
for k in 1 2 3; do TP_START_AT=$T timeout 100 python tools/r5/tp_probe.py 600 three_$k > $O/three_$k.json 2> $O/three_$k.err & done; wait
cat $O/*.json
