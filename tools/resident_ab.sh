#!/bin/bash
# headline and single-period shapes: streaming schedule against the cache-resident one, one and two lanes; alternated
line() { python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$1', round(d['ms_per_step'],4), 'ms  frac', round(r['frac'],4), ' passes', [round(v,4) for v in r['pass_ms']], 'placement', d['capture_placement'] and round(d['capture_placement']['chosen_ms'],2))"; }
run() {   # label order channels periods steps extra resident...
  local lab=$1 n=$2 c=$3 p=$4 s=$5 x=$6; shift 6
  for r in "$@"; do
    timeout -k 10 200 python bench.py --cpu-seconds 0 --configs none $x --order $n --channels $c --periods $p --steps $s --warmup 2 --resident $r \
      | line "$lab resident=$r" || exit 1
  done
}
run "headline" 20 512 16 20 "" off on:256 on:128 on:512 on:256:4 on:512:4 off on:256 on:256:4
run "N20x1024x1" 20 1024 1 10 "--plain-alloc" off on:256 on:256:4 on:384:4 on:512:4 off on:256:4
run "N18x4096x1" 18 4096 1 10 "--plain-alloc" off on:256 on:256:4 on:384:4 on:512:4
run "N20x512x2" 20 512 2 10 "--plain-alloc" off on:256 on:256:4 on:512:4
run "N20x512x4" 20 512 4 10 "--plain-alloc" off on:256 on:256:4 on:512:4
run "N20x512x8" 20 512 8 10 "--plain-alloc" off on:256 on:256:4 on:512:4
run "N24x256x1" 24 256 1 4 "--plain-alloc" off on:256 on:256:4 on:512:4
run "N22x256x1" 22 256 1 6 "--plain-alloc" off on:512 on:1024:4
