#!/bin/bash
# Orders 21 and 22: the tall tile's two 11-bit passes against the mid tile's three passes (tools/README.md).
set -e
mkdir -p gpurun_out
for cfg in "21 256 1" "22 256 1" "22 64 4" "21 512 4"; do
  set -- $cfg
  for tile in 2 -1; do
    echo "== N=$1 C=$2 R=$3 tile=$tile"
    timeout -k 10 200 python bench.py --order $1 --channels $2 --periods $3 --tile $tile --steps 10 --warmup 2 --cpu-seconds 0 \
      | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['ms_per_step'],3),'ms/step frac',round(r['frac'],3),'pass_ms',[round(v,3) for v in r['pass_ms']], 'err', d['system_check_max_abs_err'])"
  done
done
