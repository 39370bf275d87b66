#!/bin/bash
# Kernel trace of the tall-tile schedule (N=22, 256 channels, one averaged period). GPU box, repo root.
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01_n22_trace -- python3 $R/bench.py --cpu-seconds 0 --order 22 --channels 256 --periods 1 --steps 10 > $R/gpurun_out/prof_r01_n22_trace.json || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_r01_n22_fetch -- python3 $R/bench.py --cpu-seconds 0 --order 22 --channels 256 --periods 1 --steps 5 --warmup 2 > /dev/null || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_r01_n22_write -- python3 $R/bench.py --cpu-seconds 0 --order 22 --channels 256 --periods 1 --steps 5 --warmup 2 > /dev/null || exit 1
echo done
