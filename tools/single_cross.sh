line() { python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$1', '%8.4f ms  frac %.3f' % (d['ms_per_step'], r['frac']))"; }
for N in 10 8 6; do
for C in 256 1024 4096; do
for t in 0 2; do
  timeout -k 10 120 python bench.py --cpu-seconds 0 --order $N --channels $C --periods 4 --steps 200 --tile $t | line "N=$N C=$C R=4 tile $t" || exit 1
done; done; done
