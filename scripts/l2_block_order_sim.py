"""LRU model of ONE XCD's L2 (4 MiB = 32 768 lines) for pass B's tiles of a row group under different dispatch orders of the prefix blocks
(option block_order): every tile reads its own lines when it starts and the lines of its out-of-block source columns
half a window later; `inflight` tiles are resident at a time.  Prints lines fetched per ideal line (1.0 = every line of the panel once).
Pure Python / numpy, no GPU.   usage: l2_block_order_sim.py C5 11 1   |   l2_block_order_sim.py C3 12 2
(model name, low orbitals per block, row groups per 128-byte line: 1 for 8-row complex tiles, 2 for 4-row ones)
Round 4, C5: largest-first 3.18 (measured 117.8 GB / 37.8 GB = 3.1), by particle number of the high orbitals 2.00 (measured 1.97)."""
import numpy as np, sys
from collections import OrderedDict
def model_edges(name):
    e=[]
    if name=='C5':
        Ns=18
        for i in range(6): e.append((i,(i+1)%6))
        for ib in range(2):
            o=6+6*ib
            for i in range(6): e.append((o+i,o+(i+1)%6)); e.append((i,o+i))
    else:
        Ns=16
        sq=[(0,1),(2,3),(0,2),(1,3)]
        e+=sq
        for ib in range(3):
            o=4+4*ib
            e+=[(o+a,o+b) for a,b in sq]; e+=[(i,o+i) for i in range(4)]
    return Ns,list(set(tuple(sorted(x)) for x in e))
def build(name,L):
    Ns,edges=model_edges(name); npart=Ns//2
    states=np.array([s for s in range(1<<Ns) if bin(s).count('1')==npart])
    idx={int(s):i for i,s in enumerate(states)}
    hi=states>>L
    bstart=[0]+[i for i in range(1,len(states)) if hi[i]!=hi[i-1]]+[len(states)]
    nb=len(bstart)-1
    block_of=np.zeros(len(states),int)
    for k in range(nb): block_of[bstart[k]:bstart[k+1]]=k
    src=[[] for _ in range(nb)]
    for i,s in enumerate(states):
        s=int(s)
        for a,b in edges:
            for x,y in ((a,b),(b,a)):
                if (s>>y)&1 and not (s>>x)&1:
                    j=idx[s^(1<<y)^(1<<x)]
                    if block_of[j]!=block_of[i]: src[block_of[i]].append(j)
    pc=np.array([bin(int(states[bstart[k]])>>L).count('1') for k in range(nb)])
    return states,bstart,[np.array(x) for x in src],pc
def sim(bstart,src,order,share=1,pair=False,cap_lines=32768,inflight=64,ngroups=10,gdelay=None):
    """share = row groups per line (2 for R=4 complex).  Returns lines fetched per (own line-equivalents) in steady state."""
    nb=len(bstart)-1
    if pair: tiles=[(2*p+h,k) for p in range(ngroups//2) for k in order for h in (0,1)]
    else: tiles=[(g,k) for g in range(ngroups) for k in order]
    ev=[]
    gd=inflight if gdelay is None else gdelay
    for t,(g,k) in enumerate(tiles):
        ev.append((2*t,0,g,k)); ev.append((2*t+gd,1,g,k))
    ev.sort()
    lru=OrderedDict(); miss=np.zeros(ngroups)
    def touch(key,g):
        if key in lru: lru.move_to_end(key)
        else:
            miss[g]+=1; lru[key]=1
            if len(lru)>cap_lines: lru.popitem(last=False)
    for _,kind,g,k in ev:
        lg=g//share
        if kind==0:
            for c in range(bstart[k],bstart[k+1]): touch((lg,c),g)
        else:
            for c in src[k]: touch((lg,int(c)),g)
    mid=slice(ngroups//2-share, ngroups//2+share) if share>1 else slice(ngroups-3,ngroups-1)
    n=bstart[-1]
    ng=mid.stop-mid.start
    return miss[mid].sum()/(n*ng/share)   # lines fetched / ideal lines
if __name__=='__main__':
    name=sys.argv[1]; L=int(sys.argv[2]); share=int(sys.argv[3])
    states,bstart,src,pc=build(name,L)
    nb=len(bstart)-1; sizes=np.diff(bstart)
    orders={'size-desc':sorted(range(nb),key=lambda k:-sizes[k]),'popcount':sorted(range(nb),key=lambda k:pc[k]),'natural':list(range(nb))}
    for nm,o in orders.items():
        for pair in ((False,True) if share>1 else (False,)):
            for infl in (32,64,128):
                print(f"{name} L={L} share={share} {nm:10s} pair={pair} inflight={infl}: fetched/ideal = {sim(bstart,src,o,share,pair,inflight=infl):.3f}")
