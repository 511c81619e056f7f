import json, sys
rows=json.load(open(sys.argv[1]))
cfgs={0:'128x128',1:'128x64',2:'64x128',3:'64x64',4:'256x32',5:'128x32',6:'32x128'}
def nm(c):
    if c is None: return ''
    return ('P' if c>=7 else ' ')+cfgs.get(c%7,str(c))
agg={}
for r in rows:
    k=(r['key'],nm(r['cfg']),r['splitk'])
    a=agg.setdefault(r['key'],[0,0.0,0.0,nm(r['cfg']),r['splitk']])
    a[0]+=1;a[1]+=r['ms'];a[2]+=r['gflop']
tot=sum(r['ms'] for r in rows)
print('total %.3f ms' % tot)
for k,a in sorted(agg.items(), key=lambda kv:-kv[1][1]):
    print('%-42s x%-2d %-9s sk%-3s %7.4f ms/launch %7.3f ms total  %6.1f TF' % (k,a[0],a[3],a[4],a[1]/a[0],a[1],(a[2]/a[1] if a[1] else 0)))
