"""Host-only statistics of the prefix-block decomposition of one spin sector (no GPU needed):
block sizes, in-block entries per row, row slots / block hops per block.  Used to dimension the
register-resident tables of the pipelined job kernels (csrc/hxv_jobs.hip).
usage: plan_stats.py [C2|C3|C4|C5] [lowbits]"""
import sys
from itertools import combinations
from math import comb

import numpy as np

sys.path.insert(0, "cdmft-lanc-ed_amd")
from hxv import models


def hops_of(m, spin):
    L, O, B = m.Nlat, m.Norb, m.Nbath
    imp = lambda il, io: io + il * O
    bath = lambda il, io, ib: L * O + imp(il, io) + ib * L * O
    hops = []
    for il in range(L):
        for jl in range(L):
            for io in range(O):
                for jo in range(O):
                    t = m.impHloc[il, jl, spin, spin, io, jo]
                    a, b = imp(il, io), imp(jl, jo)
                    if a != b and t != 0:
                        hops.append((a, b, t))
    for ib in range(B):
        for il in range(L):
            for jl in range(L):
                for io in range(O):
                    for jo in range(O):
                        t = m.Hbath[il, jl, spin, spin, io, jo, ib]
                        a, b = bath(il, io, ib), bath(jl, jo, ib)
                        if a != b and t != 0:
                            hops.append((a, b, t))
    for il in range(L):
        for io in range(O):
            for ib in range(B):
                V = m.Vbath[il, spin, io, ib]
                if V != 0:
                    hops.append((bath(il, io, ib), imp(il, io), V))
                    hops.append((imp(il, io), bath(il, io, ib), V))
    return hops


def stats(m, n, spin, L):
    ns = m.Ns
    states = sorted(sum(1 << b for b in c) for c in combinations(range(ns), n))
    idx = {s: i for i, s in enumerate(states)}
    hops = hops_of(m, spin)
    blk = [s >> L for s in states]
    starts = [0] + [i for i in range(1, len(states)) if blk[i] != blk[i - 1]] + [len(states)]
    nb = len(starts) - 1
    bidx = np.zeros(len(states), dtype=int)
    for k in range(nb):
        bidx[starts[k]:starts[k + 1]] = k
    kin = np.zeros(len(states), dtype=int)
    per_pair = {}
    for j, s in enumerate(states):
        for a, b, t in hops:
            if (s >> b) & 1 and not (s >> a) & 1:
                i = idx[(s & ~(1 << b)) | (1 << a)]
                if bidx[i] == bidx[j]:
                    kin[i] += 1
                else:
                    per_pair.setdefault((bidx[i], bidx[j]), []).append((i - starts[bidx[i]], j - starts[bidx[j]], a >= L and b >= L))
    sizes = [starts[k + 1] - starts[k] for k in range(nb)]
    nrs = np.zeros(nb, dtype=int)
    nbh = np.zeros(nb, dtype=int)
    for (bi, bj), ents in per_pair.items():
        uniform = len(ents) == sizes[bi] == sizes[bj] and all(e[0] == e[1] for e in ents)
        if uniform:
            nbh[bi] += 1
        else:
            mult = {}
            for e in ents:
                mult[e[0]] = mult.get(e[0], 0) + 1
            nrs[bi] += max(mult.values())
    print(f"  spin {spin}: dim {len(states)} L={L} nblocks {nb} sizes min/max {min(sizes)}/{max(sizes)} kin max {kin.max()} mean {kin.mean():.2f} "
          f"row-slots/block max {nrs.max()} block-hops/block max {nbh.max()} hops {len(hops)}")
    return sizes


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "C3"
    m, (nup, ndw) = {"C2": (models.hm_1dchain(), (6, 6)), "C3": (models.hm_2dsquare(Nbath=3), (8, 8)),
                     "C4": (models.bhz_2d(Nbath=1), (8, 8)), "C5": (models.hm_ring(6, 2), (9, 9))}[which]
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    print(which, m.name, "Ns", m.Ns)
    stats(m, nup, 0, L)
    if m.Nspin > 1:
        stats(m, ndw, m.Nspin - 1, L)
