#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread > gpurun_out/t_x.log 2>&1
rc=$?; tail -2 gpurun_out/t_x.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --probe 0 > gpurun_out/b_x.log 2>&1 || exit 1
echo value $(grep -o '"value": [0-9.]*' gpurun_out/b_x.log)
bash tools/prof_default.sh x --no-cpu-baseline --probe 0 --steps 3 --warmup 1 | tail -1 | cut -c1-100
