#!/bin/bash
line() { python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$1', round(d['ms_per_step'],3), 'ms', [round(v,3) for v in r['pass_ms']])"; }
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --cpu-seconds 0 --configs none | line "base          " || exit 1
  ZMLS_LIBRARY=$PWD/zignal_amd/lib_a/libzmls.so timeout -k 10 200 python bench.py --cpu-seconds 0 --configs none | line "nt work stores" || exit 1
  ZMLS_LIBRARY=$PWD/zignal_amd/lib_b/libzmls.so timeout -k 10 200 python bench.py --cpu-seconds 0 --configs none | line "nt everything " || exit 1
done
for L in "" a b; do
  [ -n "$L" ] && export ZMLS_LIBRARY=$PWD/zignal_amd/lib_$L/libzmls.so
  timeout -k 10 200 python bench.py --cpu-seconds 0 --configs none --order 20 --channels 1024 --periods 1 --steps 10 | line "cfg5 lib=$L" || exit 1
done
