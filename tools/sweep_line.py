import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d['ms_per_step'],4), [(x['zipf'], round(x['ms_per_step'],4), round(x['lookup_ms'],4), round(x['lookup_rows_ready_ms'],4)) for x in d['skew_sweep']])
