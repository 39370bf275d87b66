#!/bin/bash
# kernel trace of the kernels either side of the headline path (tools/secondary_bench.py), round 2
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02_secondary -- python3 $R/tools/secondary_bench.py > $R/gpurun_out/secondary_r02.txt 2> $R/gpurun_out/secondary_r02.err || exit 1
echo done
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02_ingest -- python3 $R/tools/ingest_timing.py > $R/gpurun_out/ingest_r02.txt 2> $R/gpurun_out/ingest_r02.err || exit 1
echo done
