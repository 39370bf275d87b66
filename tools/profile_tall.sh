#!/bin/bash
# Kernel trace and counters of the tall-tile schedule (N=22, 256 channels, one averaged period). GPU box, repo root.
R=$PWD
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --cpu-seconds 0 --order 22 --channels 256 --periods 1 --steps 5 --warmup 2"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01_n22_trace -- python3 $R/bench.py --cpu-seconds 0 --order 22 --channels 256 --periods 1 --steps 10 > $R/gpurun_out/prof_r01_n22_trace.json || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_r01_n22_fetch -- $B > /dev/null || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_r01_n22_write -- $B > /dev/null || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVES --output-format csv -d $R/gpurun_out/prof_r01_n22_sq -- $B > /dev/null || exit 1
echo done
