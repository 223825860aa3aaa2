#!/bin/bash
# round 4, run 18: s4p_transform_points in 128 k-point chunks (parity over chunk boundaries; time-to-register both ways)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4_run18; mkdir -p $O
timeout 240 python -m pytest tests/test_gpu_registration.py tests/test_facade.py -m gpu -q -x --timeout 200 -k "transform_points or facade" > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
tail -5 $O/tests.log
g++ -O2 -std=c++17 -Iinclude tests/facade_app/timing.cpp -Lsuper4pcs_amd/lib -lsuper4pcs_amd -Wl,-rpath,$GRAFT_REPO_ROOT/super4pcs_amd/lib -o /tmp/facade_timing
for i in 1 2 3; do timeout 100 /tmp/facade_timing 1000000 0.004 2000 0.5 >> $O/facade_timing.json 2>> $O/facade_timing.err; done
cat $O/facade_timing.json; tail -3 $O/facade_timing.err
g++ -O2 -std=c++17 -DS4P_FACADE_TRACE -Iinclude tests/facade_app/timing.cpp -Lsuper4pcs_amd/lib -lsuper4pcs_amd -Wl,-rpath,$GRAFT_REPO_ROOT/super4pcs_amd/lib -o /tmp/facade_trace && timeout 100 /tmp/facade_trace 1000000 0.004 2000 0.5 > $O/facade_trace.json 2> $O/facade_trace.err
tail -11 $O/facade_trace.err
