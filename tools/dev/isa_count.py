import sys, collections
s=open(sys.argv[1]).read(); name=sys.argv[2]
i=s.index(name+':'); j=s.index('.Lfunc_end', i)
body=s[i:j].split('\n')
idx=[k for k,l in enumerate(body) if 's_barrier' in l]
loop=body[idx[0]:]
c=collections.Counter()
for l in loop:
    l=l.strip()
    if not l or l.startswith(('.',';','//')) or l.endswith(':'): continue
    op=l.split()[0]
    if op.startswith('v_mfma'): c['mfma']+=1
    elif op.startswith('v_'): c['valu']+=1
    elif op.startswith('s_waitcnt'): c['waitcnt']+=1
    elif op.startswith('s_'): c['salu']+=1
    elif op.startswith('ds_'): c['lds']+=1
    elif op.startswith(('global_','buffer_','flat_','scratch_')): c['vmem']+=1
print(dict(c))
vc=collections.Counter(l.strip().split()[0] for l in loop if l.strip().startswith('v_') and not l.strip().startswith('v_mfma'))
print(vc.most_common(16))
