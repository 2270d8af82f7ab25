#!/bin/bash
# In-step A/B of generic-GEMM tile configurations per GEMM shape (libtdn_trace.so, TDN_CFG_RULE): whole-step img/s with
# ONE shape forced to ONE configuration at a time; "none" rows in between track the box's drift.
# usage: cfg_autotune.sh "M:N:K" cfg1 cfg2 ...   (several shapes: call it several times)
cd "$(dirname "$0")/.."
export TDN_LIB=libtdn_trace.so
shape=$1; shift
run() {
  env TDN_CFG_RULE=$1 timeout -k 10 200 python bench.py --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --no-kernel-timer --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"
}
echo "$shape none -> $(run none)"
for c in "$@"; do echo "$shape cfg $c -> $(run $shape:$c)"; done
