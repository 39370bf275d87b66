#!/bin/bash
# Round 3: single-period batches in channel chunks small enough for the intermediate to stay in the 256 MiB Infinity
# Cache (MI355X_MICROARCH.md residency rule: the lines plus everything moved between two uses fit ~256 MiB), as a same-box
# A/B of four builds: workspace stores / loads temporal (0) or non-temporal (1); capture loads and response stores stay
# non-temporal in all of them. Builds: make -C zignal_amd/csrc OUT=../lib_wsSL EXTRA="-DZMLS_NT_WORK_ST=S -DZMLS_NT_WORK_LD=L".
# usage: tools/ic_sweep.sh > gpurun_out/ic_sweep.txt
line() { python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$1', round(d['ms_per_step'],4), 'ms  frac', round(r['frac'],4), ' passes', [round(v,4) for v in r['pass_ms']])"; }
run() {   # label order channels steps reserve [tile]
  local t=""; [ -n "$6" ] && t="--tile $6"
  for v in ws11 ws00 ws01 ws10; do
    unset ZMLS_LIBRARY; [ $v != ws11 ] && export ZMLS_LIBRARY=$PWD/zignal_amd/lib_$v/libzmls.so
    timeout -k 10 120 python bench.py --cpu-seconds 0 --configs none --plain-alloc --order $2 --channels $3 --periods 1 --steps $4 --warmup 3 --reserve-channels $5 $t \
      | line "$1 chunk=$5 ${6:+tile=$6 }$v" || exit 1
  done
}
# intermediate per chunk at N=20: 8 MiB per channel (mid tile: 16-channel slabs = 128 MiB; small tile: 8-channel slabs = 64 MiB)
run "N20x1024" 20 1024 10 0
run "N20x1024" 20 1024 10 64
run "N20x1024" 20 1024 10 32
run "N20x1024" 20 1024 10 16
run "N20x1024" 20 1024 10 16 0
run "N20x1024" 20 1024 10 8 0
# N=18: 2 MiB per channel
run "N18x4096" 18 4096 10 0
run "N18x4096" 18 4096 10 128
run "N18x4096" 18 4096 10 64
run "N18x4096" 18 4096 10 32
run "N18x4096" 18 4096 10 16
# N=16: 0.5 MiB per channel
run "N16x16384" 16 16384 10 0
run "N16x16384" 16 16384 10 512
run "N16x16384" 16 16384 10 256
run "N16x16384" 16 16384 10 128
run "N16x16384" 16 16384 10 64
# the BASELINE shapes themselves
run "N18x64" 18 64 50 0
run "N18x64" 18 64 50 32
run "N18x64" 18 64 50 16
