#!/bin/bash
# usage: tools/pmc.sh <tag> <counters...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_$tag
timeout -s KILL 300 rocprofv3 --pmc "$@" -d gpurun_out/pmc_$tag -o p --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --probe 0 > gpurun_out/pmc_$tag.log 2>&1
echo exit=$?
