#!/bin/bash
# Round artifacts on one MI355X (two gpurun calls: PART=1, PART=2): the default bench line, the rocprofv3 --kernel-trace
# --stats summary of the same command, the step timeline, PMC passes (launch-plan replay, so that every launch is a
# dispatch record), the box ops under rocprofv3, the --no-graph line (libtdn executor), config C5, batch dependence,
# per-layer conv benches (graph-replay timing), the fixed-vs-per-K-step sweep and the halo kernel's ablation runs.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/final
PART=${PART:-1}
if [ "$PART" = "1" ]; then
  rm -rf $O; mkdir -p $O
  python bench.py > $O/bench.json 2> $O/bench.err || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py > $O/prof_bench.json 2> $O/prof.err || exit 1
  python scripts/trace_timeline.py $O/prof/*/*_kernel_trace.csv > $O/timeline.txt 2>&1
  bash scripts/pmc_bench.sh $O/pmc > $O/pmc_passes.log 2>&1 || exit 1
  python scripts/pmc_summarize.py $O/pmc > $O/pmc_summary.txt || exit 1
  # the halo kernel with 8 x 16 patches only (every fragment inside one patch row): LDS bank conflicts against the default patches
  mkdir -p $O/pmc816
  TDN_HALO_TH=8 TDN_HALO_TW=16 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc816/lds -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-graph > $O/pmc816/lds.log 2>&1 || echo "pmc816 failed"
  python scripts/pmc_summarize.py $O/pmc816 > $O/pmc816_summary.txt 2>&1
  find $O -name "*_kernel_trace.csv" -size +3M -delete
  find $O -name "*counter_collection.csv" -size +3M -delete
else
  mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/box -- python scripts/box_bench.py > $O/box_bench.jsonl 2> $O/box.err || exit 1
  python bench.py --no-graph --no-cpu-baseline --no-secondary > $O/bench_nograph.json 2> $O/bench_nograph.err || exit 1
  python bench.py --dtype f16 --depth 101 --batch-per-gpu 4 --no-cpu-baseline --no-secondary > $O/bench_c5.json 2> $O/bench_c5.err || exit 1
  for b in 1 2 4; do python bench.py --batch-per-gpu $b --no-cpu-baseline --no-secondary --no-kernel-timer > $O/bench_b$b.json 2>/dev/null; done
  TDN_HALO=0 python bench.py --no-cpu-baseline --no-secondary --no-kernel-timer > $O/bench_nohalo.json 2>/dev/null
  python scripts/halo_bench.py --mode fwd --graph --no-epi > $O/convbench_fwd.log 2>&1
  python scripts/halo_bench.py --mode dgrad --graph --no-epi > $O/convbench_dgrad.log 2>&1
  python scripts/halo_bench.py --mode fwd --graph --no-epi --batch 1 > $O/convbench_fwd_b1.log 2>&1
  python scripts/halo_ksweep.py > $O/ksweep_l3.log 2>&1
  python scripts/halo_ksweep.py 2 200 336 256 > $O/ksweep_p2.log 2>&1
  if [ -f torch_detection_amd/libtdn_trace.so ]; then python scripts/halo_ablate.py > $O/halo_ablate.log 2>&1; fi
  python scripts/wgrad_group_bench.py > $O/wgrad_group_bench.log 2>&1
  { python scripts/block_bench.py --C 64; python scripts/block_bench.py --C 128 --H 100 --W 168; python scripts/block_bench.py --head; } > $O/block_bench.log 2>&1
  for v in 0 64 1; do TDN_BLOCK_FUSE=$v python bench.py --no-cpu-baseline --no-secondary --no-kernel-timer > $O/bench_fuse$v.json 2>/dev/null; done
  TDN_BLOCK_HEAD=0 python bench.py --no-cpu-baseline --no-secondary --no-kernel-timer > $O/bench_head0.json 2>/dev/null
  TDN_BLOCK_BITS=0 python bench.py --no-cpu-baseline --no-secondary --no-kernel-timer > $O/bench_bits0.json 2>/dev/null
  find $O -name "*_kernel_trace.csv" -size +3M -delete
fi
echo done
