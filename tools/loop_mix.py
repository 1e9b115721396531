import re,collections,sys
f=sys.argv[1]
lines=open(f).read().split('\n')
addr={}
for i,l in enumerate(lines):
    m=re.search(r'//\s*([0-9A-F]{12}):',l)
    if m: addr[i]=int(m.group(1),16)
inv={v:k for k,v in addr.items()}
best=None
for i,l in enumerate(lines):
    m=re.match(r'\s+s_cbranch_\w+\s+(\d+)',l)
    if m and i in addr:
        off=int(m.group(1))
        if off>=32768:
            tgt=addr[i]+4+(off-65536)*4
            if tgt in inv:
                n=i-inv[tgt]
                if any('mfma' in x for x in lines[inv[tgt]:i]) and (best is None or n<best[2]): best=(inv[tgt],i,n)
a,b,_=best
ops=collections.Counter(re.match(r'\s+(\S+)',l).group(1) for l in lines[a:b+1] if re.match(r'\s+(\S+)',l))
valu=sum(v for k,v in ops.items() if k.startswith('v_') and 'mfma' not in k)
mf=sum(v for k,v in ops.items() if 'mfma' in k)
print(f,'loop lines',a,b,'VALU',valu,'MFMA',mf,'ratio %.2f'%(valu/max(mf,1)),'SALU',sum(v for k,v in ops.items() if k.startswith('s_')),'LDS',sum(v for k,v in ops.items() if k.startswith('ds_')),'VMEM',sum(v for k,v in ops.items() if k.startswith(('global','buffer'))))
print(ops.most_common(30))
