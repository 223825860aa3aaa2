#!/bin/bash
# round 4, run 13 / 15: k_select_fourth (13: bound published through the records; 15: workgroup bound in LDS, one atomic per workgroup) -- parity tests, cost of a batch call on an idle GPU, the host chain
# of a rank of a world of 8 at n_P = 4.2 M; the drop-in with the lean SoA / sampled-cloud passes
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4_run15; mkdir -p $O
timeout 240 python -m pytest tests/test_gpu_select.py tests/test_facade.py -m gpu -q -x --timeout 200 > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
tail -5 $O/tests.log
timeout 120 python tools/r4/select_probe.py > $O/select_probe.json 2> $O/select_probe.err
echo "probe rc=$?"; cat $O/select_probe.json; tail -3 $O/select_probe.err
S4P_TRACE_CHAIN=1 timeout 200 python tools/sim_world.py --only-big > $O/sim_world_big.jsonl 2> $O/sim_world.err
echo "sim_world rc=$?"; cat $O/sim_world_big.jsonl; grep s4p_trace $O/sim_world.err
g++ -O2 -std=c++17 -Iinclude tests/facade_app/timing.cpp -Lsuper4pcs_amd/lib -lsuper4pcs_amd -Wl,-rpath,$GRAFT_REPO_ROOT/super4pcs_amd/lib -o /tmp/facade_timing && timeout 100 /tmp/facade_timing 1000000 0.004 2000 0.5 > $O/facade_timing.json 2> $O/facade_timing.err
cat $O/facade_timing.json; tail -3 $O/facade_timing.err
g++ -O2 -std=c++17 -DS4P_FACADE_TRACE -Iinclude tests/facade_app/timing.cpp -Lsuper4pcs_amd/lib -lsuper4pcs_amd -Wl,-rpath,$GRAFT_REPO_ROOT/super4pcs_amd/lib -o /tmp/facade_trace && timeout 100 /tmp/facade_trace 1000000 0.004 2000 0.5 > $O/facade_trace.json 2> $O/facade_trace.err
tail -14 $O/facade_trace.err
