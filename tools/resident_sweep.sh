#!/bin/bash
# Round 3: the cache-resident schedule (zmls_plan_set_resident) against the streaming one, same box, same build.
# usage: tools/resident_sweep.sh > gpurun_out/resident_sweep.txt
line() { python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$1', round(d['ms_per_step'],4), 'ms  frac', round(r['frac'],4), ' passes', [round(v,4) for v in r['pass_ms']])"; }
run() {   # label order channels periods steps resident...
  local lab=$1 n=$2 c=$3 p=$4 s=$5; shift 5
  for r in "$@"; do
    timeout -k 10 150 python bench.py --cpu-seconds 0 --configs none --plain-alloc --order $n --channels $c --periods $p --steps $s --warmup 2 --resident $r \
      | line "$lab resident=$r" || exit 1
  done
}
run "N20x1024x1" 20 1024 1 10 off on:128 on:192 on:256 on:384 on:256:2 off on:256
run "N18x64x1" 18 64 1 50 off on:256 on:256:2 off on:256
run "N18x4096x1" 18 4096 1 10 off on:128 on:192 on:256 on:384 on:256:2
run "N16x16384x1" 16 16384 1 10 off on:128 on:256 on:384
run "N20x512x2" 20 512 2 10 off on:128 on:256 on:256:2
run "N20x512x4" 20 512 4 10 off on:128 on:256
run "N20x512x16" 20 512 16 5 off on:256
run "N21x256x1" 21 256 1 10 off on:256 on:384
run "N24x256x1" 24 256 1 4 off on:128 on:256 on:256:1 on:64 on:384 off on:256
run "N23x256x1" 23 256 1 4 off on:128 on:256
run "N24x32x1" 24 32 1 4 off on:128 on:256
