#!/bin/bash
# HBM fetch/write bytes of the stacked-capture adapters (sum_transpose_kernel) at the headline size. GPU box, repo root.
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_capture_fetch -- python3 $R/bench.py --cpu-seconds 0 --layout capture --steps 3 --warmup 1 > /dev/null || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_capture_write -- python3 $R/bench.py --cpu-seconds 0 --layout capture --steps 3 --warmup 1 > /dev/null || exit 1
echo done
