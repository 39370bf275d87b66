#!/bin/bash
# A/B of two builds of libzmls.so on one box: the headline batch, alternating, N rounds. usage: tools/ab_libs.sh <other .so> [rounds] [bench args]
OTHER=$1; ROUNDS=${2:-3}; shift; shift
line() { python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$1', round(d['ms_per_step'],3), 'ms', [round(v,3) for v in r['pass_ms']], 'placement', d['capture_placement'] and round(d['capture_placement']['chosen_ms'],2))"; }
for i in $(seq $ROUNDS); do
  timeout -k 10 200 python bench.py --cpu-seconds 0 --configs none "$@" | line "base " || exit 1
  ZMLS_LIBRARY=$OTHER timeout -k 10 200 python bench.py --cpu-seconds 0 --configs none "$@" | line "other" || exit 1
done
