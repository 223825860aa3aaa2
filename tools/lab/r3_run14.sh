#!/bin/bash
# round 3, run 14: sweep of k_verify with packed two-query locating (v_pk_fma_f32 + operand selectors), direct ballots:
# kernel-level parity tests, then A/B against the previous library on the same box
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py::test_config2_1m_pair_fused_path tests/test_gpu_registration.py::test_compute_transformation_matches_oracle -m gpu -x -q --timeout 600 > gpurun_out/r3_run14_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_run14_tests.log
tail -4 gpurun_out/r3_run14_tests.log
B="--no-pmc --no-hbm-point --cpu-seconds 0 --no-parity --no-time-to-register --no-stage-pass --no-instrumented --repeats 5"
for cfg in "S4P_X=0" "S4P_LIB=$GRAFT_REPO_ROOT/scratch/libr3_before_pk.so" "S4P_X=1" "S4P_LIB=$GRAFT_REPO_ROOT/scratch/libr3_before_pk.so"; do
  v=$(env $cfg timeout 200 python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), [round(d['spread'][k]/1e6,1) for k in ('min','max')], d['config']['full_count_mode'] and round(d['config']['full_count_mode']['value']/1e6,2), round(d['roofline']['per_launch']['avg_launch_ms'],4), d['roofline']['per_launch']['exclusive'] and round(d['roofline']['per_launch']['exclusive']['avg_launch_ms'],4))")
  echo "$cfg -> $v" | tee -a gpurun_out/r3_run14_ab.log
done
