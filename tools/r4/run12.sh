#!/bin/bash
# round 4, run 12: where the drop-in's extra time goes (-DS4P_FACADE_TRACE, S4P_TRACE_INIT) and which stage of the host chain
# bounds a rank of a world of 8 (S4P_TRACE_CHAIN)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4_run12; mkdir -p $O
g++ -O2 -std=c++17 -DS4P_FACADE_TRACE -Iinclude tests/facade_app/timing.cpp -Lsuper4pcs_amd/lib -lsuper4pcs_amd -Wl,-rpath,$GRAFT_REPO_ROOT/super4pcs_amd/lib -o /tmp/facade_trace && S4P_TRACE_INIT=1 timeout 100 /tmp/facade_trace 1000000 0.004 2000 0.5 > $O/facade_trace.json 2> $O/facade_trace.err
cat $O/facade_trace.json; tail -40 $O/facade_trace.err
g++ -O2 -std=c++17 -Iinclude tests/facade_app/timing.cpp -Lsuper4pcs_amd/lib -lsuper4pcs_amd -Wl,-rpath,$GRAFT_REPO_ROOT/super4pcs_amd/lib -o /tmp/facade_timing && timeout 100 /tmp/facade_timing 1000000 0.004 2000 0.5 > $O/facade_timing.json 2> $O/facade_timing.err
cat $O/facade_timing.json
S4P_TRACE_CHAIN=1 timeout 300 python tools/sim_world.py > $O/sim_world.jsonl 2> $O/sim_world.err
echo "sim_world rc=$?"; cat $O/sim_world.jsonl; grep s4p_trace $O/sim_world.err
