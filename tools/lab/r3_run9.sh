#!/bin/bash
# round 3, run 9: parity cases at SURVEY 8d's sample sizes (configs[3] n = 20 000, configs[4] n = 5000), the chunked share test,
# fixed cost of short Perform_N_steps calls, 20 000-point sample with 512 workgroups per chunk pass
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1500 python -m pytest "tests/test_gpu_registration.py::test_quad_slices_are_a_partition_of_the_base" tests/test_gpu_configs.py::test_config3_lidar_pair_5m_points tests/test_gpu_configs.py::test_config4_part_in_whole_10m_scene -m gpu -q --timeout 900 --durations=8 > gpurun_out/r3_run9_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_run9_tests.log
S4P_TRACE_CALL=1 timeout 300 python tools/r3_short_calls.py > gpurun_out/r3_run9_short_calls.log 2>&1
timeout 900 python bench.py --sample 20000 --steps 2 --warmup 0 --repeats 1 --no-parity > gpurun_out/r3_run9_bench_sample20000.json 2> gpurun_out/r3_run9_bench20000.err
echo "bench20000 rc=$?" >> gpurun_out/r3_run9_tests.log
python - <<'PY' >> gpurun_out/r3_run9_tests.log
import json
for f in ('r3_run9_bench_sample20000',):
    try:
        line=[l for l in open('gpurun_out/%s.json'%f).read().splitlines() if l.startswith('{"metric')][-1]
        d=json.loads(line)
        print(f, round(d['value']/1e6,2),'M cand/s', round(d['ms_per_step'],4),'ms/step', d['stage_ms_per_step'])
    except Exception as e: print(f,'ERR',e)
PY
tail -30 gpurun_out/r3_run9_tests.log; cat gpurun_out/r3_run9_short_calls.log | tail -40
