#!/bin/bash
# round 3, run 7: split-base sharding on the GPU (2 and 3 ranks on one GPU), 2-rank bench dry runs in both modes, CPU
# time-to-register on the box's host cores (in the background), second line at the 20 000-point sample
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
(timeout 900 python tools/r3_cpu_ttr.py > gpurun_out/r3_cpu_time_to_register.json 2> gpurun_out/r3_cpu_ttr.err) &
CPU_PID=$!
timeout 900 python -m pytest tests/test_gpu_sharding.py -m gpu -x -q --timeout 600 > gpurun_out/r3_run7_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_run7_tests.log
for mode in base split; do
  S4P_BENCH_ONE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 4 --repeats 2 --shard-mode $mode > gpurun_out/r3_run7_bench_2ranks_$mode.json 2> gpurun_out/r3_run7_bench_2ranks_$mode.err
  echo "bench 2 ranks $mode rc=$?" >> gpurun_out/r3_run7_tests.log
done
timeout 900 python bench.py --sample 20000 --steps 2 --warmup 0 --repeats 1 > gpurun_out/r3_run7_bench_sample20000.json 2> gpurun_out/r3_run7_bench20000.err
echo "bench20000 rc=$?" >> gpurun_out/r3_run7_tests.log
wait $CPU_PID
python - <<'PY' >> gpurun_out/r3_run7_tests.log
import json
for f in ('r3_run7_bench_2ranks_base','r3_run7_bench_2ranks_split','r3_run7_bench_sample20000'):
    try:
        d=json.load(open('gpurun_out/%s.json'%f))
        print(f, round(d['value']/1e6,2),'M cand/s', round(d['ms_per_step'],4),'ms/step', d['scaling'], d['config'].get('shard_mode'), d['config'].get('collective'), 'parity', d['parity'] and (d['parity'].get('mismatches'), d['parity'].get('failed')))
    except Exception as e: print(f,'ERR',e)
print(open('gpurun_out/r3_cpu_time_to_register.json').read())
PY
tail -30 gpurun_out/r3_run7_tests.log; tail -3 gpurun_out/r3_run7_bench_2ranks_split.err
