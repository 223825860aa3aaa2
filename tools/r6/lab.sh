#!/bin/bash
# round 6 lab pass: TAG [what...]   what = quick | kern | bench20 | bench200 | tests | testsfast | trace
# Runs on the GPU box from the repo root; everything lands in gpurun_out/r6_<TAG>/ and a short summary is printed last.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
export S4P_WAIT_TIMEOUT_S=${S4P_WAIT_TIMEOUT_S:-120}
TAG=$1; shift
O=gpurun_out/r6_$TAG; mkdir -p $O
BQ="--cpu-seconds 0 --no-pmc --no-hbm-point --no-time-to-register --no-exclusive --no-instrumented --no-full-count-mode --no-stage-pass --no-extra"
for what in "$@"; do
  case $what in
    testsfast)
      timeout -s KILL 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_registration.py -m gpu -x -q --timeout 300 -p no:cacheprovider > $O/testsfast.log 2>&1; echo "testsfast rc=$?" >> $O/summary.txt; tail -4 $O/testsfast.log >> $O/summary.txt ;;
    tests)
      timeout -s KILL 1100 python -m pytest tests -m gpu -x -q --timeout 700 --durations=6 -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/summary.txt; tail -12 $O/tests.log >> $O/summary.txt ;;
    kern)
      timeout -s KILL 400 python tools/prof_kernels.py $O --lanes 1 --steps 60 --passes trace,sq,tcc > $O/kern.log 2>&1; echo "kern rc=$?" >> $O/summary.txt; tail -3 $O/kern.log >> $O/summary.txt ;;
    kernlib:*)
      lib=${what#kernlib:}
      S4P_LIB=$PWD/scratch/lib$lib.so timeout -s KILL 400 python tools/prof_kernels.py $O --lanes 1 --steps 60 --passes trace --tag $lib > $O/kern_$lib.log 2>&1; echo "kern $lib rc=$?" >> $O/summary.txt; tail -2 $O/kern_$lib.log | head -1 >> $O/summary.txt ;;
    benchlib:*)
      lib=${what#benchlib:}
      S4P_LIB=$PWD/scratch/lib$lib.so timeout -s KILL 400 python bench.py --gpus 1 --no-parity --repeats 3 $BQ > $O/bench200np_$lib.json 2> $O/bench200np_$lib.err; echo "bench200np $lib rc=$?" >> $O/summary.txt ;;
    kernenv:*)
      spec=${what#kernenv:}
      env $(echo $spec | tr ',' ' ') timeout -s KILL 400 python tools/prof_kernels.py $O --lanes 1 --steps 60 --passes trace --tag ${spec//[,=]/_} > $O/kern_${spec//[,=]/_}.log 2>&1; echo "kern $spec rc=$?" >> $O/summary.txt; tail -2 $O/kern_${spec//[,=]/_}.log | head -1 >> $O/summary.txt ;;
    benchdef:*)
      spec=${what#benchdef:}
      env $(echo $spec | tr ',' ' ') timeout -s KILL 400 python bench.py --gpus 1 --no-parity --repeats 3 $BQ > $O/bench200np_${spec//[,=]/_}.json 2> $O/bench200np_${spec//[,=]/_}.err; echo "bench200np $spec rc=$?" >> $O/summary.txt ;;
    b20env:*)
      spec=${what#b20env:}; lib=${spec%%,*}; envs=${spec#*,}
      env $(echo $envs | tr ',' ' ') S4P_LIB=$PWD/scratch/lib$lib.so timeout -s KILL 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-parity $BQ > $O/bench20_${spec//[,=]/_}.json 2> $O/bench20_${spec//[,=]/_}.err; echo "bench20 $spec rc=$?" >> $O/summary.txt ;;
    benchenv:*)
      spec=${what#benchenv:}; lib=${spec%%,*}; envs=${spec#*,}
      env $(echo $envs | tr ',' ' ') S4P_LIB=$PWD/scratch/lib$lib.so timeout -s KILL 400 python bench.py --gpus 1 --no-parity --repeats 3 $BQ > $O/bench200np_${spec//[,=]/_}.json 2> $O/bench200np_${spec//[,=]/_}.err; echo "bench200np $spec rc=$?" >> $O/summary.txt ;;
    trace)
      timeout -s KILL 300 python tools/prof_kernels.py $O --lanes 1 --steps 60 --passes trace > $O/trace.log 2>&1; echo "trace rc=$?" >> $O/summary.txt; tail -3 $O/trace.log >> $O/summary.txt ;;
    bench20)
      timeout -s KILL 400 python bench.py --gpus 1 --steps 20 --warmup 5 $BQ > $O/bench20.json 2> $O/bench20.err; echo "bench20 rc=$?" >> $O/summary.txt ;;
    bench200)
      timeout -s KILL 400 python bench.py --gpus 1 $BQ > $O/bench200.json 2> $O/bench200.err; echo "bench200 rc=$?" >> $O/summary.txt ;;
    bench200np)
      timeout -s KILL 400 python bench.py --gpus 1 --no-parity --repeats 3 $BQ > $O/bench200np.json 2> $O/bench200np.err; echo "bench200np rc=$?" >> $O/summary.txt ;;
  esac
done
python - "$O" <<'PY' >> $O/summary.txt
import json, sys, glob
O = sys.argv[1]
for f in sorted(glob.glob(O + '/bench*.json')):
    try:
        line = [l for l in open(f).read().splitlines() if l.startswith('{"metric')][-1]
        d = json.loads(line)
        p = d.get('parity') or {}
        print(f.split('/')[-1], round(d['value'] / 1e6, 2), 'M cand/s', round(d['ms_per_step'], 4), 'ms/step', [round(d['spread'][k] / 1e6, 1) for k in ('min', 'max')],
              'parity', (p.get('bases'), p.get('mismatches'), p.get('failed')))
    except Exception as e:
        print(f, 'ERR', repr(e))
PY
cat $O/summary.txt
