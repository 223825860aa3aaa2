#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5_run7; mkdir -p $O
for ab in 2 1; do
S4P_ABLATE=$ab S4P_DEBUG=1 S4P_WAIT_TIMEOUT_S=3 timeout 25 python tools/r5/dbg1.py > $O/dbg_ablate$ab.log 2>&1; echo "rc=$?" >> $O/dbg_ablate$ab.log
echo "== ablate $ab"; grep -v "^\[s4p\] wait\|^\[s4p\] k_verify" $O/dbg_ablate$ab.log | tail -6
done
