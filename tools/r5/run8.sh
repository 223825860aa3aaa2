#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5_run8; mkdir -p $O
S4P_ABLATE=8 S4P_DEBUG=1 S4P_WAIT_TIMEOUT_S=3 timeout 20 python tools/r5/dbg1.py > $O/dbg.log 2>&1; echo "rc=$?" >> $O/dbg.log
grep -v "^\[s4p\] wait\|^\[s4p\] k_verify\|File\|self\._chk\|raise" $O/dbg.log | tail -6
