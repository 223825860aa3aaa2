#!/bin/bash
# round 2, pass 16: where the small kernels of a base spend their time -- kernel trace of the new build (one k_prep launch per
# set) against the previous build, and SQ counters of the small kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
stats() {  # name lib extra-env
  env S4P_LIB=$R/$2 S4P_LANES=1 $3 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2stats16_$1 -o r -- python tools/ab_one.py 60 1 > gpurun_out/r2stats16_$1.log 2>&1
  python - "$1" <<'PY'
import csv, glob, sys
for f in glob.glob('gpurun_out/r2stats16_%s/**/r_kernel_stats.csv' % sys.argv[1], recursive=True):
    print("--", sys.argv[1])
    for r in list(csv.DictReader(open(f)))[:9]:
        print("  %-46s %6s %10.1f us %6s %%" % (r['Name'][:46], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
}
stats new super4pcs_amd/lib/libsuper4pcs_amd.so S4P_PREP_SPLIT=1
stats old scratch/libfloatlists.so S4P_X=1
echo "== PMC small kernels (new build)"
S4P_LANES=1 S4P_PREP_SPLIT=1 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/r2pmc16 -o p -- python tools/ab_one.py 20 1 > gpurun_out/r2pmc16.log 2>&1
python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/r2pmc16/**/p_counter_collection.csv', recursive=True):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'k_pairs' in n or 'k_prep' in n or 'k_quads' in n:
            key = n[:14] + ('/grid' + r['Grid_Size'] if 'Grid_Size' in r else '')
            d[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in d.items():
        print(k, {a: round(sum(b) / len(b)) for a, b in sorted(v.items())}, 'launches', len(next(iter(v.values()))))
PY
