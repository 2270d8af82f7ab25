#!/bin/bash
# Round artifacts on one MI355X: the default bench line, the rocprofv3 --kernel-trace --stats summary of the same
# command, the step timeline, PMC passes of the dominant kernel (launch-plan replay, so that every launch is a dispatch
# record), the box ops under rocprofv3, the --no-graph line (libtdn executor), config C5.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/final
rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py > $O/prof_bench.json 2> $O/prof.err || exit 1
python scripts/trace_timeline.py $O/prof/*/*_kernel_trace.csv > $O/timeline.txt 2>&1
bash scripts/pmc_bench.sh $O/pmc > $O/pmc_passes.log 2>&1 || exit 1
python scripts/pmc_summarize.py $O/pmc > $O/pmc_summary.txt || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/box -- python scripts/box_bench.py > $O/box_bench.jsonl 2> $O/box.err || exit 1
python bench.py --no-graph --no-cpu-baseline --no-secondary > $O/bench_nograph.json 2> $O/bench_nograph.err || exit 1
python bench.py --dtype f16 --depth 101 --batch-per-gpu 4 --no-cpu-baseline --no-secondary > $O/bench_c5.json 2> $O/bench_c5.err || exit 1
for b in 1 4; do python bench.py --batch-per-gpu $b --no-cpu-baseline --no-secondary --no-kernel-timer > $O/bench_b$b.json 2>/dev/null; done
# keep what is judged small: drop the raw traces (tens of MB), keep the stats / counter CSVs
find $O -name "*_kernel_trace.csv" -size +3M -delete
find $O -name "*counter_collection.csv" -size +3M -delete
echo done
