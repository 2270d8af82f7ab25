#!/bin/bash
# PMC counters for one conv shape (separate passes; no tracing domains mixed with --pmc besides kernel-trace)
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$1
mkdir -p $OUT
FILTER="$2"; CFG="$3"; MODE="${4:-fwd}"
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$tag -- python scripts/conv_bench.py --filter "$FILTER" --cfgs $CFG --cfgs64 $CFG --iters 3 --mode $MODE > $OUT/$tag.log 2>&1 || echo "pass $tag failed"
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'conv_gemm' in r['Kernel_Name'] or 'wgrad' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s mean %.4g  (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
