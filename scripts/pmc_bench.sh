#!/bin/bash
# PMC counters of the dominant kernel inside the real bench (eager, so each launch is a dispatch record)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
# usage: pmc_bench.sh [OUT_DIR [extra bench.py flags...]]   e.g.  pmc_bench.sh gpurun_out/pmc_c5 --dtype f16 --depth 101 --batch-per-gpu 4
OUT=${1:-gpurun_out/pmc_bench}
shift || true
mkdir -p $OUT
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$tag -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-graph "$@" > $OUT/$tag.log 2>&1 || echo "pass $tag failed"
done
python scripts/pmc_summarize.py $OUT
