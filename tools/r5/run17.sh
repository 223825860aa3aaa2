#!/bin/bash
# round 5, run 17: the driver's bench command on the base-group build (checks the reworked bench.py end to end)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
export S4P_WAIT_TIMEOUT_S=120
O=gpurun_out/r5_run17; mkdir -p $O
timeout -s KILL 420 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; echo "rc=$?" >> $O/bench20.err
tail -5 $O/bench20.err
python - <<'PY'
import json
try:
    line=[l for l in open('gpurun_out/r5_run17/bench20.json').read().splitlines() if l.startswith('{"metric')][-1]
    d=json.loads(line)
    print('value', round(d['value']/1e6,1), 'full', d.get('value_full_count') and round(d['value_full_count']/1e6,1), 'ms/step', round(d['ms_per_step'],4), 'spread', [round(d['spread'][k]/1e6,1) for k in ('min','max')])
    print('parity', d['parity'] and {k:d['parity'].get(k) for k in ('bases','mismatches','failed')})
    r=d['roofline']; print('frac', r['frac'], 'binding', r['binding'], 'traffic', r['traffic'])
    print('kernels', json.dumps(r.get('kernels'), indent=0)[:1500])
    print('per_launch', r['per_launch']['avg_launch_ms'], r['per_launch']['candidates_per_launch'], r['per_launch'].get('exclusive'))
    print('extra', json.dumps(d.get('extra'))[:1200])
    print('cpu', d.get('cpu_baseline') and d['cpu_baseline']['value'], 'ttr', d['config']['time_to_register'])
except Exception as e:
    print('ERR', repr(e))
PY
