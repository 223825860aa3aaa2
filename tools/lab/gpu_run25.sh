#!/bin/bash
# round 2, pass 25: SQ counters of the final kernels (one base in flight), two passes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
S4P_LANES=1 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/r2pmc25a -o p -- python tools/ab_one.py 30 1 > gpurun_out/r2pmc25a.log 2>&1
S4P_LANES=1 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/r2pmc25b -o p -- python tools/ab_one.py 30 1 > gpurun_out/r2pmc25b.log 2>&1
python - <<'PY'
import csv, glob, collections, json
out = {}
for tag in ('a', 'b'):
    for f in glob.glob('gpurun_out/r2pmc25%s/**/p_counter_collection.csv' % tag, recursive=True):
        d = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            for k in ('k_verify<false', 'k_pairs', 'k_prep', 'k_quads'):
                if k in n:
                    d[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in d.items():
            out.setdefault(k, {}).update({a: {"mean_per_launch": sum(b) / len(b), "launches": len(b)} for a, b in v.items()})
json.dump({"note": "rocprofv3 --pmc (two passes) --kernel-trace -- python tools/ab_one.py 30 1 with S4P_LANES=1, final build of round 2; SQ_* summed over all waves, quad-cycles", "kernels": out}, open('gpurun_out/r2_pmc25.json', 'w'), indent=1)
for k, v in out.items():
    print(k, {a: round(b['mean_per_launch'] / 1e6, 3) for a, b in sorted(v.items())})
PY
