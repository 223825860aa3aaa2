#!/bin/bash
# round 4, run 6: per-wave phases of the lean k_verify with three time stamps per wave (light cycle profile), extra SQ counters
# of the sweep (scoring ablated to sweep + drain), and the bench arm for reference
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4_run6; mkdir -p $O
B="--no-pmc --no-hbm-point --cpu-seconds 0 --no-parity --no-time-to-register --no-stage-pass --no-instrumented --no-exclusive --repeats 1 --no-full-count-mode"
for L in 1 6; do
S4P_LANES=$L S4P_LIB=$GRAFT_REPO_ROOT/scratch/libr4_cycprof2.so timeout 120 python bench.py $B 2>&1 | grep -a "cycle prof" | tail -3 | tee -a $O/cycprof2_lanes$L.log
done
cat > /tmp/passes.py <<'PY'
import sys
sys.path.insert(0, "tools/r4")
import prof_kernels as P
P.PASSES["lds"] = ["SQ_WAIT_INST_LDS", "SQ_INST_LEVEL_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS"]
P.PASSES["misc"] = ["SQ_INSTS_BRANCH", "SQ_IFETCH", "SQ_IFETCH_LEVEL", "SQ_INST_CYCLES_SALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"]
P.PASSES["lvl"] = ["SQ_INSTS_SMEM", "SQ_INST_LEVEL_SMEM", "SQ_INST_LEVEL_VMEM", "SQ_BUSY_CYCLES", "SQ_LEVEL_WAVES", "SQ_ACTIVE_INST_ANY"]
sys.argv = ["prof_kernels.py"] + sys.argv[1:]
P.main()
PY
S4P_ABLATE=1 timeout 300 python /tmp/passes.py $O --lanes 1 --steps 60 --passes trace,sq,lds,misc,lvl --tag ablate1 2>&1 | tail -2
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_run6/kernels_ablate1.json"))
print(json.dumps(d["kernels"]["k_verify"]))
PY
