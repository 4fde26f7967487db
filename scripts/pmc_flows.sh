#!/bin/bash
# PMC passes on the env step over the flow source (65 536 envs x 60 candidates, scripts/time_env_step.py): instruction and wait
# counters per launch, each set in its own rocprofv3 run (--pmc with --kernel-trace only).  Usage: bash scripts/pmc_flows.sh <tag>
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$name -o p -- python scripts/time_env_step.py --traffic flows --sizes 65536 --steps 30 > $OUT/pmc_$name.log 2>&1
done
python - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if 'env_step_kernel' not in k: continue
        acc[(k.split('(')[0], r.get('Grid_Size', '?'))][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(acc.items()):
    print(k)
    for c in sorted(d):
        v = d[c]
        print('  %-24s n=%-4d mean=%.5g' % (c, len(v), sum(v) / len(v)))
PY
