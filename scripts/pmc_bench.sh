#!/bin/bash
# PMC counters of the dominant kernel inside the real bench (eager, so each launch is a dispatch record)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc_bench
mkdir -p $OUT
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$tag -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $OUT/$tag.log 2>&1 || echo "pass $tag failed"
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'ELi1EEv' in n or 'Li6ELi1' in n or ', 6, 1>' in n:
            agg[n[:80]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s mean %.6g  min %.6g max %.6g (n=%d)" % (c, sum(v)/len(v), min(v), max(v), len(v)))
PY
