#!/bin/bash
# usage: tools/trace_chain.sh <tag> [VAR=val ...]: one bench step under rocprofv3
# --kernel-trace (per-dispatch begin / end timestamps) for timeline analysis
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/trace_$tag
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/trace_$tag -o r --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --probe 0 --inference 0 > gpurun_out/trace_$tag.log 2>&1
echo exit=$?
python3 tools/chain_timeline.py gpurun_out/trace_$tag/r_kernel_trace.csv > gpurun_out/trace_$tag.txt 2>&1
tail -70 gpurun_out/trace_$tag.txt
rm -f gpurun_out/trace_$tag/r_kernel_trace.csv
