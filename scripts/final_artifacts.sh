#!/bin/bash
# Round artifacts on one MI355X: the default bench line, the rocprofv3 --kernel-trace --stats summary of the same
# command, PMC passes of the dominant kernel (eager launches so that every launch is a dispatch record), config C5.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/final
rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py > $O/prof_bench.json 2> $O/prof.err || exit 1
bash scripts/pmc_bench.sh > $O/pmc_passes.log 2>&1 || exit 1
python scripts/pmc_summarize.py > $O/pmc_summary.txt || exit 1
python bench.py --dtype f16 --depth 101 --batch-per-gpu 4 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err || exit 1
echo done
