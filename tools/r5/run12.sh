#!/bin/bash
# round 5, run 12: the whole GPU suite on the base-group build, two halves side by side
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
export S4P_WAIT_TIMEOUT_S=300
O=gpurun_out/r5_run12; mkdir -p $O
python -c "from super4pcs_amd import build as B; B.build()" > $O/build.log 2>&1
( timeout -s KILL 800 python -m pytest tests/test_gpu_configs.py -m gpu -q --timeout 700 --durations=10 -p no:cacheprovider > $O/gpu_tests_configs.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests_configs.log ) &
( timeout -s KILL 800 python -m pytest tests --ignore=tests/test_gpu_configs.py -m gpu -q --timeout 600 --durations=8 -p no:cacheprovider > $O/gpu_tests_rest.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests_rest.log ) &
wait
tail -25 $O/gpu_tests_configs.log; tail -25 $O/gpu_tests_rest.log
