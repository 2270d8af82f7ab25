export TMPDIR=/tmp; O=gpurun_out/exp1; mkdir -p $O
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-secondary"
for i in 1 2; do
for v in default 0 1; do
  if [ $v = default ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  r=$($B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "HIP_FORCE_DEV_KERNARG=$v -> $r" | tee -a $O/kernarg.log
done; done
unset HIP_FORCE_DEV_KERNARG
