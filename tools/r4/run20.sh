#!/bin/bash
# round 4, run 20: the drop-in after its last change (spot check of the kept positions): its GPU test and one timing run
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4_run20; mkdir -p $O
timeout 30 python -m pytest tests/test_facade.py -m gpu -q -x --timeout 25 > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
tail -3 $O/tests.log
