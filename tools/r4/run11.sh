#!/bin/bash
# round 4, run 11: the batched fourth-point scan (k_select_fourth: one pass over P per batch of attempts) before the last full
# pass -- its parity tests, one GPU playing rank 0 of a world of 1 / 8 at n_P = 4.2 M, and the drop-in's time-to-register with
# the threaded AoS <-> SoA passes
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4_run11; mkdir -p $O
timeout 240 python -m pytest tests/test_gpu_select.py -m gpu -q -x --timeout 200 > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
tail -5 $O/tests.log
timeout 200 python tools/sim_world.py --only-big > $O/sim_world_big.jsonl 2> $O/sim_world.err
echo "sim_world rc=$?"; cat $O/sim_world_big.jsonl
g++ -O2 -std=c++17 -Iinclude tests/facade_app/timing.cpp -Lsuper4pcs_amd/lib -lsuper4pcs_amd -Wl,-rpath,$GRAFT_REPO_ROOT/super4pcs_amd/lib -o /tmp/facade_timing && timeout 100 /tmp/facade_timing 1000000 0.004 2000 0.5 > $O/facade_timing.json 2> $O/facade_timing.err
cat $O/facade_timing.json; tail -3 $O/facade_timing.err
