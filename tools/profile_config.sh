#!/bin/bash
# Kernel trace + HBM counters of any bench.py configuration: tools/profile_config.sh <tag> <bench.py args...>
# -> gpurun_out/prof_<tag>_{trace,fetch,write}; condense with tools/summarize_profile.py <tag>. GPU box, repo root.
R=$PWD
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_trace -- python3 $R/bench.py --cpu-seconds 0 "$@" > $R/gpurun_out/prof_${tag}_trace.json || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${tag}_fetch -- python3 $R/bench.py --cpu-seconds 0 "$@" > /dev/null || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${tag}_write -- python3 $R/bench.py --cpu-seconds 0 "$@" > /dev/null || exit 1
echo done $tag
