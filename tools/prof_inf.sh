#!/bin/bash
# usage: tools/prof_inf.sh <tag> [bench_inference flags]: bench_inference.py under
# rocprofv3 --kernel-trace --stats
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_$tag
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o r --output-format csv -- python3 bench_inference.py "$@" > gpurun_out/prof_$tag.log 2>&1
echo exit=$?; grep -o '"metric": "[^"]*", "value": [0-9.]*' gpurun_out/prof_$tag.log
