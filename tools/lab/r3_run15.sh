#!/bin/bash
# round 3, run 15: sweep steps of 2 (default build) vs 4 chunks (scratch/libr3_sweep4.so): kernel-level parity for both, A/B
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for lib in "" "$GRAFT_REPO_ROOT/scratch/libr3_sweep4.so"; do
  S4P_LIB=$lib timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py::test_config2_1m_pair_fused_path -m gpu -x -q --timeout 600 > gpurun_out/r3_run15_tests_$(basename "${lib:-default}").log 2>&1
  echo "pytest [$lib] rc=$?"; tail -2 gpurun_out/r3_run15_tests_$(basename "${lib:-default}").log
done
B="--no-pmc --no-hbm-point --cpu-seconds 0 --no-parity --no-time-to-register --no-stage-pass --no-instrumented --repeats 5"
for cfg in "S4P_X=0" "S4P_LIB=$GRAFT_REPO_ROOT/scratch/libr3_sweep4.so" "S4P_X=1" "S4P_LIB=$GRAFT_REPO_ROOT/scratch/libr3_sweep4.so" "S4P_LIB=$GRAFT_REPO_ROOT/scratch/libr3_sweep4.so S4P_VERIFY_BLOCKS=512" "S4P_VERIFY_BLOCKS=512"; do
  v=$(env $cfg timeout 200 python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), [round(d['spread'][k]/1e6,1) for k in ('min','max')], d['config']['full_count_mode'] and round(d['config']['full_count_mode']['value']/1e6,2), round(d['roofline']['per_launch']['avg_launch_ms'],4), d['roofline']['per_launch']['exclusive'] and round(d['roofline']['per_launch']['exclusive']['avg_launch_ms'],4))")
  echo "$cfg -> $v" | tee -a gpurun_out/r3_run15_ab.log
done
