line() { python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$1', round(d['ms_per_step'],4), 'ms')"; }
for d in 1 128 512 2048 8192; do
  ZMLS_PIPE_DELAY=$d timeout -k 10 200 python bench.py --cpu-seconds 0 --configs none --plain-alloc --order 20 --channels 1024 --periods 1 --steps 10 --warmup 2 --resident on:256:8 | line "N20x1024x1 delay=$d" || exit 1
  ZMLS_PIPE_DELAY=$d timeout -k 10 200 python bench.py --cpu-seconds 0 --configs none --plain-alloc --order 18 --channels 4096 --periods 1 --steps 10 --warmup 2 --resident on:256:8 | line "N18x4096x1 delay=$d" || exit 1
  ZMLS_PIPE_DELAY=$d timeout -k 10 200 python bench.py --cpu-seconds 0 --configs none --plain-alloc --order 16 --channels 16384 --periods 1 --steps 10 --warmup 2 --resident on:256:8 | line "N16x16384x1 delay=$d" || exit 1
done
