#!/bin/bash
# round 3, run 10: helper threads on / off per world size (one GPU playing rank 0), chunked share test
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 300 python tools/sim_world.py --threads-ab > gpurun_out/r3_run10_threads_ab.jsonl 2> gpurun_out/r3_run10_threads_ab.err
timeout 300 python -m pytest "tests/test_gpu_registration.py::test_quad_slices_are_a_partition_of_the_base" -m gpu -q --timeout 200 > gpurun_out/r3_run10_tests.log 2>&1
tail -3 gpurun_out/r3_run10_tests.log
python - <<'PY'
import json, collections
rows=collections.defaultdict(list)
for l in open('gpurun_out/r3_run10_threads_ab.jsonl'):
    d=json.loads(l); rows[(d['world'], d['helper_threads'])].append(d['ms_per_window'])
for k in sorted(rows): print(k, sorted(rows[k]))
PY
