#!/bin/bash
# Round-1 profiling recipe (run on the GPU box from the repo root; outputs under gpurun_out/prof_r01*).
# Kernel trace + stats in one run, each PMC set in a run of its own (rocprofv3 does not split counters over passes).
R=$PWD
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01_trace -- python3 $R/bench.py --cpu-seconds 0 > $R/gpurun_out/prof_r01_trace.json || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_r01_fetch -- $B > /dev/null || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_r01_write -- $B > /dev/null || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVES --output-format csv -d $R/gpurun_out/prof_r01_sq -- $B > /dev/null || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/prof_r01_tcc -- $B > /dev/null || exit 1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_r01_grbm -- $B > /dev/null || exit 1
echo done
