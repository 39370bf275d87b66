#!/bin/bash
# Efficiency map: order x channels x periods (frame-major f64), one short line per point. GPU box, repo root.
line() { python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; k=r['per_kernel']; print('$1', '%8.4f ms  frac %.3f ' % (d['ms_per_step'], r['frac']), ' '.join('%.0f' % (x['GBps'] or 0) for x in k), 'GB/s per pass')"; }
for N in 11 12 13 14 15 16 17 18 19 20; do
  for C in 64 1024 16384; do
    for R in 1 4; do
      bytes=$(( (R + 2) * (1 << N) * C * 8 ))
      if [ $bytes -gt 12000000000 ]; then continue; fi
      steps=20; if [ $bytes -lt 500000000 ]; then steps=100; fi
      timeout -k 10 120 python bench.py --cpu-seconds 0 --order $N --channels $C --periods $R --steps $steps --warmup 3 | line "N=$N C=$C R=$R" || exit 1
    done
  done
done
