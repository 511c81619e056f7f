import json, sys
rows=json.load(open(sys.argv[1]))
X3={40:'f16 256x128',41:'f16 128x128a',42:'f16 128x128b',43:'f16 256x64',44:'f16 128x64',45:'f16 256x128w8',46:'f16 128x256',47:'f16 64x128',48:'f16 64x64',39:'x3 64x64',31:'x3 256x128',32:'x3 128x128a',33:'x3 128x128b',34:'x3 256x64',35:'x3 128x64',36:'x3 256x128w8',37:'x3 128x256',38:'x3 64x128'}
def nm(c):
    if c is None: return ''
    if c is not None and c >= 49:            # f16x2 tiles with 3 (49..57) / 4 (58..66) LDS stages
        return X3.get(40 + (c - 49) % 9, '?') + '/%d' % (3 + (c - 49) // 9)
    return X3.get(c, 'f32 cfg%d' % c)
agg={}
for r in rows:
    k=(r['key'],nm(r['cfg']),r['splitk'])
    a=agg.setdefault(r['key'],[0,0.0,0.0,nm(r['cfg']),r['splitk']])
    a[0]+=1;a[1]+=r['ms'];a[2]+=r['gflop']
tot=sum(r['ms'] for r in rows)
print('total %.3f ms' % tot)
for k,a in sorted(agg.items(), key=lambda kv:-kv[1][1]):
    print('%-42s x%-2d %-13s sk%-3s %7.4f ms/launch %7.3f ms total  %6.1f TF' % (k,a[0],a[3],a[4],a[1]/a[0],a[1],(a[2]/a[1] if a[1] else 0)))
