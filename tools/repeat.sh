#!/bin/bash
# tuning helper: repeat bench.py (no CPU leg) with the given extra args, print one short line per run
n=$1; shift
for i in $(seq 1 $n); do
  timeout -k 10 300 python bench.py --cpu-seconds 0 "$@" | python -c "import json,sys; d=json.load(sys.stdin); print('run probe', round(d['capture_placement_probe_ms'] or 0,2), 'ms;', round(d['value'],1), 'Gs/s', round(d['ms_per_step'],3), 'ms frac', round(d['roofline']['frac'],4), [round(v,3) for v in d['roofline']['pass_ms']])" || exit 1
done
