#!/bin/bash
# round 4, final pass, second part (same library build as part 1; bench.py with the per-rank fields): the launch contract for
# N > 1 as a dry run with both ranks on the one GPU of the box, then the three bench legs again so that the committed lines carry
# the digest of the bench.py that ships
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04_final; mkdir -p $O $O/bench_final
cp super4pcs_amd/lib/BUILD_INFO.json $O/BUILD_INFO.json
S4P_BENCH_ONE_GPU=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2ranks_dryrun.json 2> $O/bench_2ranks_dryrun.err
echo "bench 2 ranks rc=$?" > $O/log2.txt
rm -f $O/bench_final/*.csv
timeout 900 python bench.py --profile-dir $O/bench_final > $O/bench_final.json 2> $O/bench_final.err
echo "bench rc=$?" >> $O/log2.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
echo "bench20 rc=$?" >> $O/log2.txt
timeout 420 python -m pytest tests/test_facade.py tests/test_gpu_configs.py -m gpu -q -x --timeout 400 --durations=5 -k "facade or reduced_scale" > $O/gpu_tests_part2.log 2>&1
echo "pytest part 2 rc=$?" >> $O/log2.txt
tail -12 $O/gpu_tests_part2.log >> $O/log2.txt
python - <<'PY' >> gpurun_out/r04_final/log2.txt
import json, glob, csv
O='gpurun_out/r04_final'
for f in ('bench_2ranks_dryrun','bench_final','bench_driver_command'):
    try:
        line=[l for l in open('%s/%s.json'%(O,f)).read().splitlines() if l.startswith('{"metric')][-1]
        d=json.loads(line); r=d['roofline']
        print(f, round(d['value']/1e6,2),'M cand/s', round(d['ms_per_step'],4),'ms/step', [round(d['spread'][k]/1e6,1) for k in ('min','max')], 'parity', d['parity'] and (d['parity'].get('bases'), d['parity'].get('mismatches'), d['parity'].get('failed')))
        print('   ranks', d['config'].get('ranks'), d['config'].get('collective'))
        print('   frac', r['frac'], r['binding'], 'traffic', r['traffic'], 'per_launch', r['per_launch']['avg_launch_ms'])
        print('   extra', d.get('extra'), 'bench_py', d.get('provenance',{}).get('bench_py_sha16'), d.get('provenance',{}).get('git_sha'))
    except Exception as e: print(f,'ERR',repr(e))
for f in glob.glob(O+'/stats/**/r_kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print('  ', r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
PY
cat $O/log2.txt
