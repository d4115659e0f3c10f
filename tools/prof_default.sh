#!/bin/bash
# usage: tools/prof_default.sh <tag> [bench flags]: the default bench.py run under
# rocprofv3 --kernel-trace --stats (the roofline probe and rocprof see the same run)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_$tag
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o r --output-format csv -- python3 bench.py "$@" > gpurun_out/prof_$tag.log 2>&1
echo exit=$?; tail -1 gpurun_out/prof_$tag.log | cut -c1-400
# the per-dispatch trace is large (gpurun merges at most 64 MiB back): keep the stats
rm -f gpurun_out/prof_$tag/r_kernel_trace.csv
