#!/bin/bash
# usage: tools/ab.sh <tag> [VAR=val ...]   parity tests, then bench.py under each
# environment setting given (one run per argument; "-" = default environment)
tag=$1; shift
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread > gpurun_out/t_$tag.log 2>&1
rc=$?; tail -3 gpurun_out/t_$tag.log
[ $rc -ne 0 ] && exit $rc
for setting in "$@"; do
  if [ "$setting" = "-" ]; then
    timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --probe 0 > gpurun_out/ab_${tag}.tmp 2>&1 || exit 1
  else
    timeout -k 10 300 env $setting python bench.py --steps 10 --warmup 3 --no-cpu-baseline --probe 0 > gpurun_out/ab_${tag}.tmp 2>&1 || exit 1
  fi
  echo "$setting: $(grep -o '"value": [0-9.]*' gpurun_out/ab_${tag}.tmp) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_${tag}.tmp)" | tee -a gpurun_out/ab_$tag.log
done
