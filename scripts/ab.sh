#!/bin/bash
# usage: ab.sh VAR a b [pairs]   -> interleaved bench.py runs of VAR=a / VAR=b on one box (img/s, ms)
cd "$(dirname "$0")/.."
VAR=$1; A=$2; B=$3; PAIRS=${4:-3}
for i in $(seq 1 $PAIRS); do
  for v in $A $B; do
    r=$(env $VAR=$v timeout -k 10 200 python bench.py --steps ${STEPS:-40} --warmup 5 --no-cpu-baseline --no-kernel-timer --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "$VAR=$v -> $r"
  done
done
