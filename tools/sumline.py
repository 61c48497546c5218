import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']
        print(j['config']['workload'][:40], '|', j['value'], j['ms_per_step'], 'fwd', r['kernel_ms_avg'], 'other', r.get('other_kernel'), j.get('checks'))
