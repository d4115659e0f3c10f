#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_loss_curve.py -x -q -m gpu --timeout 400 --timeout-method thread > gpurun_out/t_res.log 2>&1
rc=$?; tail -5 gpurun_out/t_res.log; [ $rc -ne 0 ] && exit $rc
for r in 0 256 512 768; do
  PCM_DW_RESIDENT=$r timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --probe 0 > gpurun_out/b_res$r.log 2>&1 || exit 1
  echo resident=$r $(grep -o '"value": [0-9.]*' gpurun_out/b_res$r.log)
done
