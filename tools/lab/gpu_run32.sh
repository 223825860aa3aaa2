#!/bin/bash
# round 2, pass 32: the sharded bench path with the final build, 2 and 4 ranks sharing the one GPU of the box (gloo callbacks
# for the collective; on an 8-GPU node the same loop runs over RCCL)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for n in 2 4; do
  S4P_BENCH_ONE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29510 + n)) bench.py --gpus $n --steps 60 --warmup 5 --repeats 2 --no-pmc --no-hbm-point --cpu-seconds 0 --no-time-to-register 2>gpurun_out/r2_shard32_$n.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ranks $n', round(d['value'] / 1e6, 2), d['n_gpus'], d['steps'], round(d['ms_per_step'], 4), d.get('parity'), d['config']['parallelism'][:60])"
  tail -3 gpurun_out/r2_shard32_$n.err | grep -v amdgpu.ids | cut -c1-300
done
