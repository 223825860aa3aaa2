#!/bin/bash
# round 4, run 5: where a k_verify wave's lifetime goes (cycle-profiling build), the counters rocprofv3 offers, and the
# failing multi-pass records test again
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4_run5; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_registration.py -m gpu -q -x --timeout 280 -k "multi_pass or per_candidate" > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
tail -12 $O/tests.log
for L in 1 6; do
S4P_LANES=$L S4P_LIB=$GRAFT_REPO_ROOT/scratch/libr4_cycprof.so timeout 120 python tools/ab_one.py 100 1 2>&1 | grep -a "cycle prof\|cand_per_s" | tee -a $O/cycprof_lanes$L.log
done
(rocprofv3 --list-avail 2>/dev/null || rocprofv3 -L 2>/dev/null) | grep -a -o "SQ_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|GRBM_[A-Z0-9_]*" | sort -u > $O/counters.txt
wc -l $O/counters.txt
