#!/bin/bash
# round 3, run 4: early exit (best-count hint) + reworked bench.py (parity over all timed bases, byte model from the timed
# bases, pmc valu / l2 / traffic, HBM-bound point under rocprofv3), second line at the 20 000-point sample
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_registration.py tests/test_gpu_kernels.py tests/test_gpu_sharding.py tests/test_facade.py -m gpu -x -q --timeout 900 > gpurun_out/r3_run4_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_run4_tests.log
timeout 900 python bench.py --profile-dir gpurun_out/r3_bench_profile > gpurun_out/r3_run4_bench.json 2> gpurun_out/r3_run4_bench.err
echo "bench rc=$?" >> gpurun_out/r3_run4_tests.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-hbm-point --cpu-seconds 0 > gpurun_out/r3_run4_bench_steps20.json 2>> gpurun_out/r3_run4_bench.err
echo "bench20 rc=$?" >> gpurun_out/r3_run4_tests.log
timeout 900 python bench.py --sample 20000 --steps 2 --warmup 0 --repeats 1 > gpurun_out/r3_run4_bench_sample20000.json 2> gpurun_out/r3_run4_bench20000.err
echo "bench20000 rc=$?" >> gpurun_out/r3_run4_tests.log
python - <<'PY' >> gpurun_out/r3_run4_tests.log
import json
for f in ('r3_run4_bench','r3_run4_bench_steps20','r3_run4_bench_sample20000'):
    try:
        d=json.load(open('gpurun_out/%s.json'%f))
        print(f, round(d['value']/1e6,2),'M cand/s', round(d['ms_per_step'],4),'ms/step; parity', d['parity'] and (d['parity']['bases'], d['parity']['mismatches'], d['parity'].get('failed')),
              'full', d['config'].get('full_count_mode') and round(d['config']['full_count_mode']['value']/1e6,2), 'pruned frac', round(d['config']['early_exit']['fraction'],3))
        r=d['roofline']; print('   frac', round(r['frac'],3), 'binding', r['binding'], 'traffic', r['traffic'], 'hbm point', r['hbm_bound_point'] and {k:r['hbm_bound_point'].get(k) for k in ('kernel_ms','measured_GBps','frac','count_mismatches','error')})
    except Exception as e: print(f,'ERR',e)
PY
tail -5 gpurun_out/r3_run4_bench.err gpurun_out/r3_run4_bench20000.err >> gpurun_out/r3_run4_tests.log
tail -40 gpurun_out/r3_run4_tests.log
