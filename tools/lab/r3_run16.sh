#!/bin/bash
# round 3, run 16: where run 15's gain comes from.  With 512-entry queues the quantised query copy no longer fits the 80 KB LDS
# budget at 768 threads, so the default build sweeps float queries from global memory.  Arms: default (4 chunks, float queries),
# 4 chunks + LDS copy (budget 112 KB), 2 chunks + LDS copy (run 14's build), 2 chunks + float queries (S4P_NO_QLDS=1)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
S4P_LIB=$GRAFT_REPO_ROOT/scratch/libr3_sweep4_qlds.so timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py::test_config2_1m_pair_fused_path -m gpu -x -q --timeout 600 2>&1 | tail -2
B="--no-pmc --no-hbm-point --cpu-seconds 0 --no-parity --no-time-to-register --no-stage-pass --no-instrumented --repeats 5"
for cfg in "S4P_X=0" "S4P_LIB=$GRAFT_REPO_ROOT/scratch/libr3_sweep4_qlds.so" "S4P_LIB=$GRAFT_REPO_ROOT/scratch/libr3_sweep2_now.so" "S4P_LIB=$GRAFT_REPO_ROOT/scratch/libr3_sweep2_now.so S4P_NO_QLDS=1" "S4P_X=1" "S4P_LIB=$GRAFT_REPO_ROOT/scratch/libr3_sweep4_qlds.so"; do
  v=$(env $cfg timeout 200 python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), [round(d['spread'][k]/1e6,1) for k in ('min','max')], d['config']['full_count_mode'] and round(d['config']['full_count_mode']['value']/1e6,2), round(d['roofline']['per_launch']['avg_launch_ms'],4), d['roofline']['per_launch']['exclusive'] and round(d['roofline']['per_launch']['exclusive']['avg_launch_ms'],4))")
  echo "$cfg -> $v" | tee -a gpurun_out/r3_run16_ab.log
done
