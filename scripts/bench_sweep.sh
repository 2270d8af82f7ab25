#!/bin/bash
# usage: bench_sweep.sh OUT "ENV1=a ENV2=b" "ENV1=c" ...   — one bench.py run (no CPU baseline, no kernel timer) per
# quoted environment setting, value + ms/step appended to OUT
cd "$(dirname "$0")/.."
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"
for cfg in "$@"; do
  line=$(env $cfg timeout -k 10 120 python bench.py --no-cpu-baseline --no-kernel-timer --no-secondary ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$cfg -> $line" | tee -a "$OUT"
done
