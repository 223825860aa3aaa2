#!/bin/bash
# round 6, final pass, part B (same sources as part A): the bench line with default flags (+ the rocprofv3 CSVs it collects itself
# and the time-to-register of the other BASELINE configs), the drop-in against the C ABI, the simulated world
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
export S4P_WAIT_TIMEOUT_S=600
O=gpurun_out/r06_final; mkdir -p $O $O/bench_final
cp super4pcs_amd/lib/BUILD_INFO.json $O/BUILD_INFO_b.json
timeout -s KILL 1100 python bench.py --profile-dir $O/bench_final --ttr-configs 1,3,4s > $O/bench_final.json 2> $O/bench_final.err
echo "bench rc=$?" > $O/log_b.txt
g++ -O2 -std=c++17 -Iinclude tests/facade_app/timing.cpp -Lsuper4pcs_amd/lib -lsuper4pcs_amd -Wl,-rpath,$GRAFT_REPO_ROOT/super4pcs_amd/lib -o /tmp/facade_timing
for i in 1 2 3; do GPU_MAX_HW_QUEUES=8 timeout -s KILL 100 /tmp/facade_timing 1000000 0.004 2000 0.5 >> $O/facade_timing.json 2>> $O/facade_timing.err; done
# the same program in the environment a C++ application has by default (4 hardware queues: the group streams take two priority levels)
for i in 1 2 3; do env -u GPU_MAX_HW_QUEUES timeout -s KILL 100 /tmp/facade_timing 1000000 0.004 2000 0.5 >> $O/facade_timing_default_env.json 2>> $O/facade_timing.err; done
echo "facade timing rc=$?" >> $O/log_b.txt
S4P_TRACE_CHAIN=1 timeout -s KILL 200 python tools/sim_world.py > $O/sim_world.jsonl 2> $O/sim_world.err
echo "sim_world rc=$?" >> $O/log_b.txt
python - <<'PY' >> gpurun_out/r06_final/log_b.txt
import json
O='gpurun_out/r06_final'
try:
    line=[l for l in open(O+'/bench_final.json').read().splitlines() if l.startswith('{"metric')][-1]
    d=json.loads(line); r=d['roofline']
    print('bench_final', round(d['value']/1e6,2),'M cand/s', round(d['ms_per_step'],4),'ms/step', [round(d['spread'][k]/1e6,1) for k in ('min','max')], 'full', d.get('value_full_count') and round(d['value_full_count']/1e6,2), 'parity', d['parity'] and (d['parity'].get('bases'), d['parity'].get('mismatches'), d['parity'].get('failed')))
    print('   frac', r['frac'], r['binding'], 'traffic', r['traffic'], 'per_launch', r['per_launch']['avg_launch_ms'], r['per_launch']['launches'], r['per_launch'].get('exclusive'))
    for k,v in (r.get('kernels') or {}).items(): print('   ', k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
    print('   hbm point', r.get('hbm_bound_point') and {k:r['hbm_bound_point'].get(k) for k in ('kernel_ms','frac','measured_frac','count_mismatches','error')})
    print('   ttr', d['config']['time_to_register'], 'cpu', d.get('cpu_baseline') and (d['cpu_baseline']['value'], d['cpu_baseline'].get('openmp_all_cores') and d['cpu_baseline']['openmp_all_cores']['value']))
    print('   extra', json.dumps(d.get('extra'))[:3000])
    print('   provenance', d.get('provenance'))
except Exception as e: print('bench_final ERR',repr(e))
for f in ('facade_timing.json','facade_timing_default_env.json','sim_world.jsonl'):
    try: print(open(O+'/'+f).read()[:4000])
    except Exception as e: print(f, 'ERR', repr(e))
PY
cat $O/log_b.txt
