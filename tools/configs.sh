#!/bin/bash
# BASELINE.json configs 2-5 at full single-GPU size through bench.py (no CPU leg); one short line each.
# Run on the GPU box from the repo root.
line() { python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$1', round(d['value'],1), 'Gs/s', round(d['ms_per_step'],3), 'ms frac', round(r['frac'],4), [round(v,3) for v in r['pass_ms']])"; }
timeout -k 10 200 python bench.py --cpu-seconds 0 --order 18 --channels 64 --periods 1 --steps 50 | line "cfg2 N18 64ch R1 f64" || exit 1
timeout -k 10 200 python bench.py --cpu-seconds 0 --order 20 --channels 1024 --periods 1 --steps 20 | line "cfg5 N20 1024ch R1 f64" || exit 1
timeout -k 10 200 python bench.py --cpu-seconds 0 --order 20 --channels 1024 --periods 1 --dtype f32 --steps 20 | line "cfg5 N20 1024ch R1 f32" || exit 1
timeout -k 10 300 python bench.py --cpu-seconds 0 --order 24 --channels 256 --periods 1 --steps 5 --warmup 1 | line "cfg4 N24 256ch R1 f64" || exit 1
timeout -k 10 200 python bench.py --cpu-seconds 0 --order 16 --channels 512 --periods 4 --steps 50 | line "N16 512ch R4 f64" || exit 1
