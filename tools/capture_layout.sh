#!/bin/bash
# bench.py on the stacked single-channel layout (C, R*L) through the layout adapters; one short line per case
for a in "--periods 16 --channels 512" "--periods 1 --channels 1024" "--periods 4 --channels 512 --order 16"; do
  timeout -k 10 300 python bench.py --cpu-seconds 0 --steps 5 --warmup 2 --layout capture $a | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('capture-major $a', round(d['ms_per_step'],3), 'ms frac', round(r['frac'],4), [round(v,3) for v in r['pass_ms']])" || exit 1
done
