line() { python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$1', round(d['value'],1), 'Gs/s', round(d['ms_per_step'],3), 'ms frac', round(r['frac'],4), [round(v,3) for v in r['pass_ms']])"; }
for rc in 0 16 32 64 128; do
timeout -k 10 200 python bench.py --cpu-seconds 0 --order 20 --channels 1024 --periods 1 --steps 20 --plain-alloc --reserve-channels $rc | line "N20 1024ch R1 f64 reserve $rc" || exit 1
done
for rc in 0 64 128; do
timeout -k 10 200 python bench.py --cpu-seconds 0 --order 18 --channels 1024 --periods 1 --steps 20 --plain-alloc --reserve-channels $rc | line "N18 1024ch R1 f64 reserve $rc" || exit 1
done
