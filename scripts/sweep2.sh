#!/bin/bash
# usage: sweep2.sh "VAR=a" "VAR=b" ... : interleaved rounds of bench.py under each env setting (img/s, ms)
cd "$(dirname "$0")/.."
ROUNDS=${ROUNDS:-2}
for r in $(seq 1 $ROUNDS); do
  for kv in "$@"; do
    res=$(env $kv timeout -k 10 200 python bench.py --steps ${STEPS:-40} --warmup 5 --no-cpu-baseline --no-kernel-timer --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "$kv -> $res"
  done
done
