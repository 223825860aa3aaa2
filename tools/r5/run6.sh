#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5_run6; mkdir -p $O
S4P_DEBUG=1 S4P_WAIT_TIMEOUT_S=5 timeout 40 python tools/r5/dbg1.py > $O/dbg.log 2>&1; echo "rc=$?" >> $O/dbg.log
tail -40 $O/dbg.log
