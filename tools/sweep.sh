#!/bin/bash
# tuning helper: bench.py over tile configurations, no CPU leg (run on the GPU box)
for t in "$@"; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --configs none --tile $t | python -c "import json,sys; d=json.load(sys.stdin); print('tile $t', round(d['value'],1), 'Gs/s', round(d['ms_per_step'],3), 'ms frac', round(d['roofline']['frac'],4), [round(v,3) for v in d['roofline']['pass_ms']])" || exit 1
done
