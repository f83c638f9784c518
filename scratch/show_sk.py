import json,glob,collections
best=collections.defaultdict(list)
for f in sorted(glob.glob('/root/repo/gpurun_out/cb_sk/*.log')):
    for l in open(f):
        if l.startswith('{'):
            r=json.loads(l); best[(r['stride'],r['C'],r['V'])].append((r['tflops'],r['variant'],r['us']))
        elif 'Error' in l or 'error' in l: print(f,l.strip())
for k,v in best.items():
    v0=[x for x in v if x[1]==0][0]
    print(k,'default',v0[0],'TF',v0[2],'us | ', ' '.join(f"{var}:{tf}" for tf,var,us in sorted(v,reverse=True)[:9]))
