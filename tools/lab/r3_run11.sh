#!/bin/bash
# round 3, run 11: the whole GPU suite with the helper-thread policy (threads where they pay) and k_verify-only events in the
# timed region; short bench lines (default flags minus the profiling passes; the driver's 20-step command)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 --durations=10 > gpurun_out/r3_run11_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_run11_tests.log
timeout 300 python bench.py --no-pmc --no-hbm-point --cpu-seconds 0 --parity-bases 10 --no-time-to-register --no-exclusive > gpurun_out/r3_run11_bench.json 2> gpurun_out/r3_run11_bench.err
echo "bench rc=$?" >> gpurun_out/r3_run11_tests.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-hbm-point --cpu-seconds 0 --no-time-to-register --no-exclusive > gpurun_out/r3_run11_bench_steps20.json 2> gpurun_out/r3_run11_bench20.err
echo "bench20 rc=$?" >> gpurun_out/r3_run11_tests.log
python - <<'PY' >> gpurun_out/r3_run11_tests.log
import json
for f in ('r3_run11_bench','r3_run11_bench_steps20'):
    try:
        line=[l for l in open('gpurun_out/%s.json'%f).read().splitlines() if l.startswith('{"metric')][-1]
        d=json.loads(line)
        print(f, round(d['value']/1e6,2),'M cand/s', round(d['ms_per_step'],4),'ms/step', [round(d['spread'][k]/1e6,1) for k in ('min','max')], 'full', d['config']['full_count_mode'] and round(d['config']['full_count_mode']['value']/1e6,2), 'parity', d['parity'] and (d['parity'].get('bases'), d['parity'].get('mismatches'), d['parity'].get('failed')), d['stage_ms_per_step'], d['roofline']['per_launch']['avg_launch_ms'], d['roofline']['frac'])
    except Exception as e: print(f,'ERR',e)
PY
tail -32 gpurun_out/r3_run11_tests.log
