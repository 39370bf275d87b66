#!/bin/bash
# the one-launch schedule of two-pass orders (zmls_plan_set_resident flag 8) against the cache-resident and the streaming schedules, alternated on one box
line() { python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$1', round(d['ms_per_step'],4), 'ms  frac', round(r['frac'],4), ' passes', [round(v,4) for v in r['pass_ms']])"; }
run() {   # label order channels periods steps extra resident...
  local lab=$1 n=$2 c=$3 p=$4 s=$5 x=$6; shift 6
  for r in "$@"; do
    timeout -k 10 200 python bench.py --cpu-seconds 0 --configs none $x --order $n --channels $c --periods $p --steps $s --warmup 2 --resident $r \
      | line "$lab resident=$r" || exit 1
  done
}
run "N20x1024x1" 20 1024 1 10 "--plain-alloc" off on:256 on:256:8 on:256 on:256:8
run "N18x4096x1" 18 4096 1 10 "--plain-alloc" off on:256 on:256:8 on:256:8
run "N16x16384x1" 16 16384 1 10 "--plain-alloc" on:256 on:256:8
run "N18x64x1" 18 64 1 50 "--plain-alloc" on:256 on:256:8 on:256 on:256:8
run "N20x512x2" 20 512 2 10 "--plain-alloc" on:256 on:256:8
run "N20x512x4" 20 512 4 10 "--plain-alloc" on:256 on:256:8
run "headline" 20 512 16 20 "" on:256 on:256:8 on:256 on:256:8
