set -e
for tile in 2 0 1024 1026; do
  echo "== N=24 C=256 R=1 tile=$tile"
  timeout -k 10 200 python bench.py --order 24 --channels 256 --periods 1 --tile $tile --steps 5 --warmup 2 --cpu-seconds 0 \
    | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['ms_per_step'],3),'ms/step frac',round(r['frac'],3),'pass_ms',[round(v,3) for v in r['pass_ms']], 'err', d['system_check_max_abs_err'])"
done
