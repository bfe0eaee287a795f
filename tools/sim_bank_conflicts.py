"""LDS bank-conflict model of k_sweep_hc's spin accesses at 16^3 (verdict r3, item 2): every pass reads the own site and
the six neighbours of up to 192 sites of one hyperplane, 64 lanes per instruction, 32 banks, two 32-lane halves per b32 read.
Prints the modelled LDS cycles of one sweep for the site-order layout, for the best paddings of the x-plane / row strides,
and the conflict-free count.  (python tools/sim_bank_conflicts.py; no GPU.)"""
import numpy as np, itertools
L=16; N=L**3
# schedule: levels by x+y+z, within level by site index; passes = chunks of <=256 sites of one level
sites=np.arange(N); x=sites//256; y=(sites//16)%16; z=sites%16
lev=x+y+z
order=np.lexsort((sites,lev))
passes=[]
for c in range(3*L-2):
    s=order[lev[order]==c]
    for k in range(0,len(s),256): passes.append(s[k:k+256])
print("passes",len(passes))
def cycles32(addr):  # addr: dword addresses of 64 lanes (array), -1 inactive; b32: two halves of 32 lanes
    tot=0
    for h in range(0,len(addr),32):
        a=addr[h:h+32]; a=a[a>=0]
        if len(a)==0: continue
        a=np.unique(a)
        tot+=np.bincount(a%32).max()
    return tot
def run(pad):   # pad(site)->lds index
    tot=0; ideal=0
    for s in passes:
        sx,sy,sz=s//256,(s//16)%16,s%16
        nbrs=[s]
        for dx,dy,dz in ((1,0,0),(-1,0,0),(0,1,0),(0,-1,0),(0,0,1),(0,0,-1)):
            nbrs.append(((sx+dx)%L)*256+((sy+dy)%L)*16+(sz+dz)%L)
        for n in nbrs:
            a=pad(n)
            for w in range(0,len(a),64):
                tot+=cycles32(a[w:w+64]); ideal+=(min(64,len(a)-w)+31)//32
    return tot,ideal
print("identity",run(lambda s:s))
best=[]
for px in range(0,33):
    for py in (0,1,2,3):
        t,i=run(lambda s:s+px*(s//256)+py*(s//16))
        best.append((t,px,py))
best.sort(); print(best[:8]); 
