import json,sys
for line in sys.stdin:
    if not line.startswith('{'): continue
    d=json.loads(line)
    print(d['ms_per_step'], {k:round(v['ms_avg'],3) for k,v in d.get('kernels',{}).items()})
