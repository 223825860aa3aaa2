#!/bin/bash
# round 4, run 8: the shipped k_verify (lean, 76 VGPRs, no deferral) -> kernel / registration tests, the new reduced-scale
# whole registrations of configs[3] / [4] and the chunked-winner test, bench arm, cold HBM points of k_apply and the sampler
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r4_run8; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_registration.py -m gpu -q -x --timeout 400 > $O/tests.log 2>&1
echo "pytest rc=$?" >> $O/tests.log
tail -4 $O/tests.log
timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q -x --timeout 500 -k "reduced_scale or chunked_winner" > $O/tests2.log 2>&1
echo "pytest rc=$?" >> $O/tests2.log
tail -8 $O/tests2.log
B="--no-pmc --no-hbm-point --cpu-seconds 0 --no-parity --no-time-to-register --no-stage-pass --no-instrumented --no-exclusive --repeats 3"
for cfg in "S4P_X=default"; do
  v=$(env $cfg timeout 60 python bench.py $B 2>$O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), [round(d['spread'][k]/1e6,1) for k in ('min','max')], d['config']['full_count_mode'] and round(d['config']['full_count_mode']['value']/1e6,2), round(d['roofline']['per_launch']['avg_launch_ms'],4))" 2>>$O/err.log)
  echo "$cfg -> $v" | tee -a $O/ab.log
done
cd /tmp
for what in "apply 134217728" "sampler 10000000"; do
  tag=$(echo $what | cut -d' ' -f1)
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=$(mktemp -d /tmp/hbm_XXXX)
    timeout 200 rocprofv3 --pmc $ctr --kernel-trace -d $d -o h --output-format csv -- python $GRAFT_REPO_ROOT/tools/r4/hbm_points.py $what > $GRAFT_REPO_ROOT/$O/hbm_${tag}_${ctr}.log 2>&1
    for f in $(find $d -name "*counter_collection.csv"); do grep -a "k_apply\|k_vox\|Kernel_Name" $f | cut -d, -f1-40 > $GRAFT_REPO_ROOT/$O/hbm_${tag}_${ctr}_counters.csv; done
    for f in $(find $d -name "*kernel_trace.csv"); do grep -a "k_apply\|k_vox\|Kernel_Name" $f > $GRAFT_REPO_ROOT/$O/hbm_${tag}_${ctr}_trace.csv; done
    rm -rf $d
  done
done
cd $GRAFT_REPO_ROOT
tail -2 $O/hbm_apply_FETCH_SIZE.log; head -3 $O/hbm_apply_FETCH_SIZE_counters.csv | cut -c1-400; head -3 $O/hbm_apply_FETCH_SIZE_trace.csv | cut -c1-400
