#!/bin/bash
# round 5, run 26: bench.py --gpus 2 on ONE GPU (gloo through the callback provider): the N > 1 code path of the bench on the base-group build
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
export S4P_WAIT_TIMEOUT_S=120
O=gpurun_out/r5_run26; mkdir -p $O
S4P_BENCH_ONE_GPU=1 timeout -s KILL 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2ranks_dryrun.json 2> $O/bench_2ranks_dryrun.err; echo "rc=$?"
tail -3 $O/bench_2ranks_dryrun.err
python - <<'PY'
import json
try:
    line=[l for l in open('gpurun_out/r5_run26/bench_2ranks_dryrun.json').read().splitlines() if l.startswith('{"metric')][-1]
    d=json.loads(line)
    print('value', round(d['value']/1e6,1), 'n_gpus', d['n_gpus'], 'scaling', d['scaling'], 'ms/step', round(d['ms_per_step'],4), 'parity', d['parity'] and (d['parity'].get('mismatches'), d['parity'].get('failed')))
    print('collective', d['config']['collective'], d['config']['shard_mode'], d['config']['ranks'])
except Exception as e: print('ERR', repr(e))
PY
