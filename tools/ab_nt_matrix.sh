#!/bin/bash
# which streams of the passes should be non-temporal? builds in zignal_amd/lib_<variant> (g = gather loads, s = workspace
# stores, l = workspace loads, r = response stores); headline, then config 5 (N=20 x 1024 x 1) and config 4 (N=24 x 256 x 1)
line() { python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$1', round(d['ms_per_step'],3), 'ms', [round(v,3) for v in r['pass_ms']])"; }
VARS="base g gs gsl gslr gsr"
for i in 1 2; do
  for v in $VARS; do
    unset ZMLS_LIBRARY; [ $v != base ] && export ZMLS_LIBRARY=$PWD/zignal_amd/lib_$v/libzmls.so
    timeout -k 10 200 python bench.py --cpu-seconds 0 --configs none | line "headline $v" || exit 1
  done
done
for v in $VARS; do
  unset ZMLS_LIBRARY; [ $v != base ] && export ZMLS_LIBRARY=$PWD/zignal_amd/lib_$v/libzmls.so
  timeout -k 10 200 python bench.py --cpu-seconds 0 --configs none --order 20 --channels 1024 --periods 1 --steps 10 | line "cfg5 $v" || exit 1
  timeout -k 10 200 python bench.py --cpu-seconds 0 --configs none --order 24 --channels 256 --periods 1 --steps 4 --warmup 1 | line "cfg4 $v" || exit 1
  timeout -k 10 200 python bench.py --cpu-seconds 0 --configs none --order 18 --channels 64 --periods 1 --steps 50 | line "cfg2 $v" || exit 1
done
