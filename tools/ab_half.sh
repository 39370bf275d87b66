#!/bin/bash
line() { python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$1', round(d['ms_per_step'],3), 'ms', [round(v,3) for v in r['pass_ms']])"; }
for i in 1 2; do
for v in base h8 h32; do
  unset ZMLS_LIBRARY; [ $v != base ] && export ZMLS_LIBRARY=$PWD/zignal_amd/lib_$v/libzmls.so
  timeout -k 10 200 python bench.py --cpu-seconds 0 --configs none | line "headline $v" || exit 1
done
done
