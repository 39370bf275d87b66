#!/bin/bash
# Round-2 profiling recipe (run on the GPU box from the repo root; outputs under gpurun_out/prof_<tag>_*).
# Kernel trace + stats in one run, each PMC set in a run of its own (rocprofv3 does not split counters over passes).
# The headline workload only (--configs none: the extra BASELINE configs reuse the same kernel names at N=20 and
# would mix into the per-kernel averages; --cpu-seconds 0: the CPU leg launches no kernels).
# usage: tools/profile_r02.sh [tag] [extra bench.py arguments]
TAG=${1:-r02}; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --configs none $*"
O=$R/gpurun_out/prof_${TAG}
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_trace -- python3 $R/bench.py --cpu-seconds 0 --configs none "$@" > ${O}_trace.json 2> ${O}_trace.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d ${O}_fetch -- $B > /dev/null 2>> ${O}_pmc.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d ${O}_write -- $B > /dev/null 2>> ${O}_pmc.err || exit 1
# LIGHT=1: kernel trace and HBM traffic only (the other BASELINE configurations)
[ -n "$LIGHT" ] && { echo done; exit 0; }
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVES --output-format csv -d ${O}_sq -- $B > /dev/null 2>> ${O}_pmc.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d ${O}_tcc -- $B > /dev/null 2>> ${O}_pmc.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d ${O}_grbm -- $B > /dev/null 2>> ${O}_pmc.err || exit 1
# where the scatter pass loses its time: write requests by size, read requests by size, issue stalls
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d ${O}_wrreq -- $B > /dev/null 2>> ${O}_pmc.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d ${O}_rdreq -- $B > /dev/null 2>> ${O}_pmc.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d ${O}_sq2 -- $B > /dev/null 2>> ${O}_pmc.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_WR_UNCACHED_32B_sum --output-format csv -d ${O}_tcc2 -- $B > /dev/null 2>> ${O}_pmc.err || exit 1
echo done
