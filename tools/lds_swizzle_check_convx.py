import itertools
P=29
groups=[list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32)),
        [l+32 for l in list(range(0,4))+list(range(12,16))+list(range(20,28))], [l+32 for l in list(range(4,12))+list(range(16,20))+list(range(28,32))]]
def read_conf(s, P):
    worst=0
    for kh in range(3):
        for col in range(4):
            for grp in groups:
                slots={}
                for l in grp:
                    row=(l&15)+kh; g=l>>4
                    a=((row*P+col)*4+(g^s(row)))*16
                    slots.setdefault((a//16)%16,set()).add(a)
                worst=max(worst,max(len(v) for v in slots.values()))
    return worst
def write_conf(s,P,width=8):
    # ds_write_b64: 4 groups of 16 contiguous lanes, bank=(a/4)%32 ; conflict = max distinct addresses per bank
    worst=0
    for ch in range(2):
      for col in range(4):
        for g in range(4):
            banks={}
            for r in range(16):
                a=((r*P+col)*4+(g^s(r)))*16+ch*8
                for w in range(0,width,4):
                    banks.setdefault(((a+w)//4)%32,set()).add(a+w)
            worst=max(worst,max(len(v) for v in banks.values()))
    return worst
cands={'old ((i>>2)&1)<<1': lambda i:((i>>2)&1)<<1, '(i>>1)&3': lambda i:(i>>1)&3, '(i>>2)&3': lambda i:(i>>2)&3, 'i&3':lambda i:i&3,
       '((i>>1)&1)|(((i>>3)&1)<<1)': lambda i:((i>>1)&1)|(((i>>3)&1)<<1), '((i>>1)&3)^((i>>3)&1)': lambda i:((i>>1)&3)^((i>>3)&1)}
for P in (29,33):
  for n,f in cands.items():
    print(P, n, 'read', read_conf(f,P), 'write64', write_conf(f,P))
# brute force over tables with period 8 in row
best=[]
for tab in itertools.product(range(4),repeat=8):
    f=lambda i,tab=tab: tab[i&7]
    r=read_conf(f,29)
    if r==1:
        w=write_conf(f,29)
        best.append((w,tab))
best.sort()
print(len(best), best[:5])
