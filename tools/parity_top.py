import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1]); p=d['parity']['oracle_fp64_full']
    top=sorted(((v,k) for k,v in p.items() if isinstance(v,float)), reverse=True)[:4]
    print(f.split('/')[-1], 'ms %.3f'%d['ms_per_step'], top)
