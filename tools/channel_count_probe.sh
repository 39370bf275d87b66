#!/bin/bash
# Does the period stride (L x C x 8 bytes, within 4 KiB of 4 GiB at N=20 x 512 channels) matter? The headline at nearby channel counts.
line() { python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$1', round(d['ms_per_step'],3), 'ms', [round(v,3) for v in r['pass_ms']], 'frac', round(r['frac'],4), 'gather GB/s', round(r['per_kernel'][0]['GBps']))"; }
for C in 512 496 528 480 544 512; do
  timeout -k 10 200 python bench.py --cpu-seconds 0 --configs none --channels $C --steps 10 | line "C=$C" || exit 1
done
