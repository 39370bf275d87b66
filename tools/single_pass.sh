line() { python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$1', round(d['value'],1), 'Gs/s', round(d['ms_per_step'],4), 'ms frac', round(r['frac'],4), [round(v,4) for v in r['pass_ms']])"; }
timeout -k 10 200 python bench.py --cpu-seconds 0 --order 10 --channels 16384 --periods 4 --steps 50 | line "N10 16384ch R4" || exit 1
timeout -k 10 200 python bench.py --cpu-seconds 0 --order 10 --channels 16384 --periods 1 --steps 50 | line "N10 16384ch R1" || exit 1
timeout -k 10 200 python bench.py --cpu-seconds 0 --order 8 --channels 65536 --periods 4 --steps 50 | line "N8 65536ch R4" || exit 1
timeout -k 10 200 python bench.py --cpu-seconds 0 --order 12 --channels 8192 --periods 4 --steps 50 | line "N12 8192ch R4 (two passes)" || exit 1
