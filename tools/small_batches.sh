line() { python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$1', round(d['value'],1), 'Gs/s', round(d['ms_per_step'],4), 'ms frac', round(r['frac'],4), [round(v,4) for v in r['pass_ms']])"; }
for t in 2 0; do
timeout -k 10 200 python bench.py --cpu-seconds 0 --order 18 --channels 64 --periods 1 --steps 200 --tile $t | line "N18 64ch R1 tile $t" || exit 1
timeout -k 10 200 python bench.py --cpu-seconds 0 --order 16 --channels 64 --periods 1 --steps 200 --tile $t | line "N16 64ch R1 tile $t" || exit 1
timeout -k 10 200 python bench.py --cpu-seconds 0 --order 16 --channels 512 --periods 4 --steps 100 --tile $t | line "N16 512ch R4 tile $t" || exit 1
timeout -k 10 200 python bench.py --cpu-seconds 0 --order 18 --channels 256 --periods 1 --steps 100 --tile $t | line "N18 256ch R1 tile $t" || exit 1
timeout -k 10 200 python bench.py --cpu-seconds 0 --order 14 --channels 256 --periods 2 --steps 200 --tile $t | line "N14 256ch R2 tile $t" || exit 1
done
