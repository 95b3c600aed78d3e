import json, sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')][0]
d=json.loads(l)
print({k:d[k] for k in ('value','ms_per_step','segments_per_s','path_tflops')})
r=d.get('roofline'); 
if r: print('roofline', r['kernel'], round(r['achieved'],1), 'TF frac', round(r['frac'],3))
if 'conv_stack' in d:
    print('conv s/step',round(d['conv_stack']['seconds_per_step'],4),'tflops',round(d['conv_stack']['tflops'],1))
    for k,v in d['conv_stack']['kernels'].items(): print("%-60s %6.1f TF %7.2f ms %d"%(k,v['tflops'],v['ms_per_step'],v['launches_per_step']))
if 'cpu_baseline' in d: print(d['cpu_baseline'], d.get('speedup_vs_cpu'))
