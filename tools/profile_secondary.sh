#!/bin/bash
# Kernel trace of the kernels either side of the headline path (emphasis filter, spectrum, layout adapters) and of
# the single-period configurations; outputs under gpurun_out/prof_r01_secondary*. Run on the GPU box from the repo root.
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01_secondary -- python3 $R/tools/secondary_bench.py > $R/gpurun_out/prof_r01_secondary.log || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01_r1 -- python3 $R/bench.py --cpu-seconds 0 --order 20 --channels 1024 --periods 1 --steps 10 > $R/gpurun_out/prof_r01_r1.json || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_r01_r1_fetch -- python3 $R/bench.py --cpu-seconds 0 --order 20 --channels 1024 --periods 1 --steps 5 --warmup 2 > /dev/null || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_r01_r1_write -- python3 $R/bench.py --cpu-seconds 0 --order 20 --channels 1024 --periods 1 --steps 5 --warmup 2 > /dev/null || exit 1
echo done
