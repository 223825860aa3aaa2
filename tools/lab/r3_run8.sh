#!/bin/bash
# round 3, run 8: what run 7 found (shares of a base defined on order keys, byte model and collective warm-up for N > 1),
# the parity cases at SURVEY 8d's sample sizes for configs[3] / [4], and the 20 000-point sample on an otherwise idle host
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_sharding.py tests/test_gpu_registration.py::test_quad_slices_are_a_partition_of_the_base tests/test_gpu_configs.py::test_config3_lidar_pair_5m_points tests/test_gpu_configs.py::test_config4_part_in_whole_10m_scene -m gpu -x -q --timeout 900 --durations=12 > gpurun_out/r3_run8_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_run8_tests.log
for mode in base split; do
  S4P_BENCH_ONE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --shard-mode $mode > gpurun_out/r3_run8_bench_2ranks_$mode.json 2> gpurun_out/r3_run8_bench_2ranks_$mode.err
  echo "bench 2 ranks $mode rc=$?" >> gpurun_out/r3_run8_tests.log
done
timeout 900 python bench.py --sample 20000 --steps 2 --warmup 0 --repeats 1 --no-parity > gpurun_out/r3_run8_bench_sample20000.json 2> gpurun_out/r3_run8_bench20000.err
echo "bench20000 rc=$?" >> gpurun_out/r3_run8_tests.log
python - <<'PY' >> gpurun_out/r3_run8_tests.log
import json
for f in ('r3_run8_bench_2ranks_base','r3_run8_bench_2ranks_split','r3_run8_bench_sample20000'):
    try:
        line=[l for l in open('gpurun_out/%s.json'%f).read().splitlines() if l.startswith('{"metric')][-1]
        d=json.loads(line)
        print(f, round(d['value']/1e6,2),'M cand/s', round(d['ms_per_step'],4),'ms/step', d['scaling'], d['spread']['min'], d['spread']['max'], 'frac', d['roofline']['frac'], 'parity', d['parity'] and (d['parity'].get('mismatches'), d['parity'].get('failed')))
    except Exception as e: print(f,'ERR',e)
PY
tail -40 gpurun_out/r3_run8_tests.log
