import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=j['roofline']; c=j['config']
print('ms',j['ms_per_step'],'value',j['value'],'frac',r['frac'],'traffic',r['traffic'],'trace',r['rocprof_kernel_trace'])
print('open',c.get('gf_solve',{}).get('sector_open_ms'),'apply_host',c.get('apply_host',{}).get('ratio_to_floor'),{k:v.get('ms_per_product') for k,v in c.get('other_workloads',{}).items()})
