#!/bin/bash
# usage: sweep_env.sh VAR v1 v2 ...   -> bench.py images/sec per value (STEPS=40 by default: on one box two runs of
# the same configuration agree to ~0.1 %)
cd "$(dirname "$0")/.."
VAR=$1; shift
for v in "$@"; do
  r=$(env $VAR=$v timeout -k 10 200 python bench.py --steps ${STEPS:-40} --warmup 5 --no-cpu-baseline --no-kernel-timer --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$VAR=$v -> $r"
done
