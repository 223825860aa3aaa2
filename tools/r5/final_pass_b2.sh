#!/bin/bash
# round 5, final pass, part B once more for the bench line alone (bench.py changed after part B: the L2 request rate is now taken
# over the counter pass's own kernel time; device / host sources unchanged: same digest as parts A and B)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
export S4P_WAIT_TIMEOUT_S=600
O=gpurun_out/r05_final; mkdir -p $O $O/bench_final
timeout -s KILL 1100 python bench.py --profile-dir $O/bench_final --ttr-configs 1,3,4s > $O/bench_final.json 2> $O/bench_final.err
echo "bench rc=$?"
python - <<'PY'
import json
O='gpurun_out/r05_final'
line=[l for l in open(O+'/bench_final.json').read().splitlines() if l.startswith('{"metric')][-1]
d=json.loads(line); r=d['roofline']
print('bench_final', round(d['value']/1e6,2),'M cand/s', round(d['ms_per_step'],4),'ms/step', [round(d['spread'][k]/1e6,1) for k in ('min','max')], 'full', d.get('value_full_count') and round(d['value_full_count']/1e6,2), 'parity', d['parity'] and (d['parity'].get('bases'), d['parity'].get('mismatches')))
print('   l2', r['l2'] and (r['l2']['frac'], r['l2'].get('launch_ms')), 'valu', r['valu'] and r['valu']['frac'], 'binding', r['binding'])
print('   provenance', d.get('provenance'))
PY
