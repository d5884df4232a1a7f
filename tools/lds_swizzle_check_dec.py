OFF=1
import itertools
g0=list(range(0,4))+list(range(12,16))+list(range(20,28)); g1=list(range(4,12))+list(range(16,20))+list(range(28,32))
groups=[g0,g1,[l+32 for l in g0],[l+32 for l in g1]]
def rd(swz):
    worst=0; tot=0; n=0
    for W in range(8):
        for ks in range(12):
            for grp in groups:
                slots={}
                for l in grp:
                    q=l&15; g=l>>4
                    G=OFF+128*W+8*q+4*ks+g
                    P=swz(G)
                    slots.setdefault(P%16,set()).add(P)
                m=max(len(v) for v in slots.values())
                worst=max(worst,m); tot+=m; n+=1
    return worst, tot/n
def wr(swz):
    worst=0; tot=0;n=0
    for k in range(9):
        for half in range(2):
          for tb in range(0,256,64):
            for pair in range(2):
                banks={}
                for l in range(32):
                    tid=tb+half*32+l
                    idx=tid+256*k
                    j=4*idx-2+2*pair+8   # +8 front offset
                    G=j>>3
                    a=swz(G)*16+(j&7)*2
                    banks.setdefault((a//4)%32,set()).add(a)
                m=max(len(v) for v in banks.values()); worst=max(worst,m); tot+=m;n+=1
    return worst, tot/n
print('linear', rd(lambda G:G), wr(lambda G:G))
print('x^(G>>4)&7', rd(lambda G:G^((G>>4)&7)), wr(lambda G:G^((G>>4)&7)))
# linear maps: low4 ^= M * bits(4..7)
best=[]
for cols in itertools.product(range(16),repeat=4):
    def swz(G,cols=cols):
        x=0
        for b in range(4):
            if (G>>(4+b))&1: x^=cols[b]
        return G^x
    r=rd(swz)
    if r[0]==1:
        best.append((wr(swz),cols))
best.sort()
print(len(best),best[:10])
print("---- alternate pair order on odd lanes")
def wr2(swz, sel):
    worst=0; tot=0;n=0
    for k in range(9):
        for half in range(2):
          for tb in range(0,256,64):
            for pair in range(2):
                banks={}
                for l in range(32):
                    tid=tb+half*32+l
                    idx=tid+256*k
                    p=pair^sel(tid)
                    j=4*idx-2+2*p+8
                    G=j>>3
                    a=swz(G)*16+(j&7)*2
                    banks.setdefault((a//4)%32,set()).add(a)
                m=max(len(v) for v in banks.values()); worst=max(worst,m); tot+=m;n+=1
    return worst, tot/n
sels={'l&1':lambda t:t&1,'(l>>1)&1':lambda t:(t>>1)&1,'(l>>4)&1':lambda t:(t>>4)&1,'(l>>3)&1':lambda t:(t>>3)&1,'(l>>2)&1':lambda t:(t>>2)&1}
res=[]
for w,cols in best:
    def swz(G,cols=cols):
        x=0
        for b in range(4):
            if (G>>(4+b))&1: x^=cols[b]
        return G^x
    for n,s in sels.items():
        res.append((wr2(swz,s),cols,n))
res.sort()
print(res[:8])
