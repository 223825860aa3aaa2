#!/bin/bash
# round 6, final pass, part C (same sources as parts A and B): the kernels of a device pass alone (one base per launch) with their
# counters, the per-wave phase profile of the lab build, the N > 1 path of the bench with both ranks on one GPU
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
export S4P_WAIT_TIMEOUT_S=300
O=gpurun_out/r06_final; mkdir -p $O/lanes1
cp super4pcs_amd/lib/BUILD_INFO.json $O/BUILD_INFO_c.json
timeout -s KILL 500 python tools/prof_kernels.py $O/lanes1 --lanes 1 --steps 60 --passes trace,sq,sq2,tcc,lds > $O/lanes1/log.txt 2>&1
echo "kern rc=$?" > $O/log_c.txt; tail -2 $O/lanes1/log.txt >> $O/log_c.txt
BASES=12 timeout -s KILL 300 python tools/r6/wave_prof.py > $O/wave_profile.txt 2>&1
echo "wave_prof rc=$?" >> $O/log_c.txt; grep "lean sweep" $O/wave_profile.txt | tail -3 >> $O/log_c.txt
S4P_BENCH_ONE_GPU=1 timeout -s KILL 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --cpu-seconds 0 --no-pmc --no-hbm-point --no-time-to-register --no-exclusive --no-instrumented --no-full-count-mode --no-stage-pass --no-extra > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks_one_gpu.err
echo "2 ranks rc=$?" >> $O/log_c.txt
python - <<'PY' >> gpurun_out/r06_final/log_c.txt
import json
try:
    line=[l for l in open('gpurun_out/r06_final/bench_2ranks_one_gpu.json').read().splitlines() if l.startswith('{"metric')][-1]
    d=json.loads(line); print('2 ranks on one GPU', round(d['value']/1e6,2), 'M cand/s', d['n_gpus'], d['config'].get('ranks'), 'parity', d['parity'] and (d['parity'].get('bases'), d['parity'].get('mismatches')))
except Exception as e: print('2 ranks ERR', repr(e))
PY
cat $O/log_c.txt
