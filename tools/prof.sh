#!/bin/bash
# usage: tools/prof.sh <tag>   (runs on the GPU box: tests, then a profiled bench)
tag=$1
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread > gpurun_out/t_$tag.log 2>&1
rc=$?; echo exit=$rc >> gpurun_out/t_$tag.log; tail -4 gpurun_out/t_$tag.log
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_$tag
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o r --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --probe 0 > gpurun_out/prof_$tag.log 2>&1
grep -o "\"value\": [0-9.]*" gpurun_out/prof_$tag.log
