#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
echo "== ablations (fine0 = pipelined sweep + lane loop)"
for ab in 0 1 2; do
  S4P_LIB=$R/scratch/libfine0.so S4P_LANES=1 S4P_ABLATE=$ab timeout 300 python tools/ab_one.py 100 2 2>&1 | tail -1 | tee -a gpurun_out/r2_ab4.log
done
echo "== TCP/TA counters"
cd /tmp
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TD_BUSY_avr TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum"; do
  i=$((i+1))
  S4P_LIB=$R/scratch/libfine0.so S4P_LANES=1 timeout 200 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/r4pmc_$i -o p --output-format csv -- python $R/tools/ab_one.py 30 1 > $R/gpurun_out/r4pmc_$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, collections, glob, os
for f in sorted(glob.glob('gpurun_out/r4pmc_*/**/p_counter_collection.csv', recursive=True)):
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_verify<false>' in r['Kernel_Name']:
            d[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[1], {k: '%.4g'%(sum(v)/len(v)) for k,v in sorted(d.items())})
for f in sorted(glob.glob('gpurun_out/r4pmc_*.log')):
    t=open(f).read()
    if 'rror' in t: print(f, t[-300:])
PY
