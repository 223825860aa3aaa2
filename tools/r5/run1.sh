#!/bin/bash
# round 5, run 1: baseline on today's box: bench workload through Perform_N_steps with the launch-thread breakdown
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5_run1; mkdir -p $O
S4P_TRACE_LAUNCH=1 timeout 120 python tools/ab_one.py 200 3 > $O/ab.json 2> $O/ab.err
timeout 100 python tools/r4/host_probe.py 300 > $O/host_probe.json 2> $O/host_probe.err
cat $O/ab.json $O/host_probe.json; grep s4p_trace $O/ab.err | tail -3
