#!/bin/bash
# round 2, pass 23: capacity growth (roll back, grow, replay the base); k_verify block size per structure (HBM-bound point)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== registration tests"
timeout 900 python -m pytest tests/test_gpu_registration.py -m gpu -x -q 2>&1 | tail -6
echo "== bench: HBM-bound point, timed region"
timeout 900 python bench.py --no-pmc --cpu-seconds 0 --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'] / 1e6, 2), d['parity']['mismatches'], d['roofline']['hbm_bound_point'])"
echo "== config4 test (1024-thread path on a structure beyond the cache)"
timeout 900 python -m pytest tests/test_gpu_configs.py::test_config4_part_in_whole_10m_scene -m gpu -x -q 2>&1 | tail -3
