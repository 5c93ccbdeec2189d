#!/usr/bin/env python3
"""VERDICT r5 item 1a, host only: what does the orbital -> bit assignment of a spin's basis cost the out-of-block phases?

The engine sorts a spin's configurations by integer value, orbital p <-> bit p-1 as the reference numbers them (impurity first, then the
replicas, ED_SETUP.f90:367-375,563-568).  A prefix block = the states that share their high Ns-L bits.  Hops that touch a high orbital leave the
block: "block hops" when both orbitals are high (whole block, one coefficient, a contiguous run) and "row slots" otherwise (one table word per
row, lanes whose hop is ruled out idle).  For a given assignment this script rebuilds that classification from the model's hop graph and
replays pass A's out-of-block gathers (one thread per block row, waves of 64 consecutive rows, C columns of `pitch` elements):
  * out-of-block entries per row, block hops per row, row slots per block row (the plan's max_outer / slots statistics),
  * wave-gathers actually issued (a wave whose 64 lanes are all idle skips the slot), live lanes per wave-gather,
  * distinct 128-byte lines per wave-gather and per row -- the currency the gathers are paid in (LABNOTES: "the phase pays per distinct segment").
It also reports how many row-slot gathers are RUNS (consecutive live rows reading consecutive source rows with one coefficient): those need no
table word and no idle lanes.

    python scripts/bitorder_sim.py [C3|C4|C5] [--search N]
"""
import itertools
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "cdmft-lanc-ed_amd"))
from hxv import models  # noqa: E402


def hop_edges(m, spin):
    """Ordered pairs (a <- b, amplitude) of one spin, orbitals 0-based in the reference's numbering (sparse/H_up.f90:8-87)."""
    L, O, B = m.Nlat, m.Norb, m.Nbath
    nimp = L * O
    s = spin if m.Nspin > 1 else 0
    e = {}

    def add(a, b, t):
        if a != b and t != 0:
            e[(a, b)] = e.get((a, b), 0) + t

    for il in range(L):
        for jl in range(L):
            for io in range(O):
                for jo in range(O):
                    add(io + il * O, jo + jl * O, m.impHloc[il, jl, s, s, io, jo])
                    for ib in range(B):
                        add(nimp + io + il * O + ib * nimp, nimp + jo + jl * O + ib * nimp, m.Hbath[il, jl, s, s, io, jo, ib])
    for ib in range(B):
        for il in range(L):
            for io in range(O):
                v = m.Vbath[il, s, io, ib]
                add(io + il * O, nimp + io + il * O + ib * nimp, v)
                add(nimp + io + il * O + ib * nimp, io + il * O, v)
    return e


def popcount(x):
    x = x - ((x >> 1) & 0x55555555)
    x = (x & 0x33333333) + ((x >> 2) & 0x33333333)
    x = (x + (x >> 4)) & 0x0F0F0F0F
    return (x * 0x01010101 >> 24) & 0xFF


def analyse(ns, npart, edges, pos, L, C=4, pitch_lines=True, verbose=False):
    """pos[o] = bit of orbital o.  Returns a dict of statistics for pass A on this spin (and the counts pass B shares)."""
    allst = np.arange(1 << ns, dtype=np.int64)
    st = allst[popcount(allst) == npart]  # sorted ascending
    dim = len(st)
    index_of = np.full(1 << ns, -1, dtype=np.int64)
    index_of[st] = np.arange(dim)
    hi = st >> L
    starts = np.flatnonzero(np.r_[True, hi[1:] != hi[:-1]])
    bounds = np.r_[starts, dim]
    block_of = np.searchsorted(starts, np.arange(dim), side="right") - 1
    nb = len(starts)
    # out-of-block entries: for every ordered hop a <- b the (target row, source row, sign) triples
    tgt_all, src_all, sgn_all, hop_all = [], [], [], []
    n_in = 0
    for hid, ((a, b), t) in enumerate(edges.items()):
        pa, pb = pos[a], pos[b]
        ok = ((st >> pb) & 1 == 1) & ((st >> pa) & 1 == 0)
        s = st[ok]
        lo, hi_ = min(pa, pb), max(pa, pb)
        between = (s >> (lo + 1)) & ((1 << (hi_ - lo - 1)) - 1)
        sg = 1 - 2 * (popcount(between) & 1)
        tg = index_of[s - (1 << pb) + (1 << pa)]
        sr = np.flatnonzero(ok)
        oob = block_of[tg] != block_of[sr]
        n_in += int((~oob).sum())
        tgt_all.append(tg[oob])
        src_all.append(sr[oob])
        sgn_all.append(sg[oob])
        hop_all.append(np.full(int(oob.sum()), hid))
    tgt = np.concatenate(tgt_all)
    src = np.concatenate(src_all)
    sgn = np.concatenate(sgn_all)
    hop = np.concatenate(hop_all)
    amp = np.array([abs(t) for t in edges.values()])
    res = dict(dim=dim, nblocks=nb, max_block=int(np.diff(bounds).max()), n_in=n_in / dim, n_out=len(tgt) / dim)
    # group by (target block, source block, multiplicity) exactly like the plan: the k-th entry of a row from one source block
    order = np.lexsort((src, tgt))
    tgt, src, sgn, hop = tgt[order], src[order], sgn[order], hop[order]
    tb, sb = block_of[tgt], block_of[src]
    # multiplicity index of an entry among the entries of its row with the same source block
    key = tgt * nb + sb
    o2 = np.argsort(key, kind="stable")
    kk = np.zeros(len(key), dtype=np.int64)
    ks = key[o2]
    first = np.r_[True, ks[1:] != ks[:-1]]
    run_id = np.cumsum(first) - 1
    run_start = np.flatnonzero(first)
    kk[o2] = np.arange(len(key)) - run_start[run_id]
    slot_key = (tb * nb + sb) * 64 + kk
    uniq = np.unique(slot_key)
    n_bh = n_rs = 0
    bh_rows = rs_rows = 0
    gathers = live = lines = 0
    run_gathers = run_lines = 0
    runs_total = 0
    rs_entries = 0
    run_entries = 0
    for sk in uniq:
        sel = slot_key == sk
        t_, s_, g_, h_ = tgt[sel], src[sel], sgn[sel], hop[sel]
        b = int(sk // 64 // nb)
        sbk = int(sk // 64 % nb)
        r0, r1 = bounds[b], bounds[b + 1]
        s0 = bounds[sbk]
        n = r1 - r0
        if len(t_) == n and bounds[sbk + 1] - s0 == n and np.all(s_ - s0 == t_ - r0) and np.all(g_ * amp[h_] == g_[0] * amp[h_[0]]):
            n_bh += 1
            bh_rows += n
            continue
        n_rs += 1
        rs_rows += n
        rs_entries += len(t_)
        # runs: consecutive target rows, consecutive sources, one signed coefficient
        brk = np.r_[True, (np.diff(t_) != 1) | (np.diff(s_) != 1) | (np.diff(g_ * amp[h_]) != 0)]
        runs_total += int(brk.sum())
        rl = np.diff(np.r_[np.flatnonzero(brk), len(t_)])
        run_entries += int(rl[rl >= 16].sum())
        ent = dict(zip(t_.tolist(), s_.tolist()))
        for w0 in range(r0, r1, 64):
            rows = range(w0, min(w0 + 64, r1))
            srcs = [ent.get(i, -1) for i in rows]
            lv = sum(1 for x in srcs if x >= 0)
            if lv == 0:
                continue
            ln = {x // 8 for x in srcs if x >= 0}
            if lv < len(srcs):
                ln.add(s0 // 8)  # idle lanes read the block's first row
            gathers += 1
            live += lv
            lines += len(ln) * C
    nwaves = sum((bounds[k + 1] - bounds[k] + 63) // 64 for k in range(nb))
    res.update(block_hops_per_row=bh_rows / dim, row_slots_per_row=rs_rows / dim, rs_live_per_row=rs_entries / dim, n_rs=n_rs, n_bh=n_bh,
               wave_gathers_per_wave=gathers / nwaves, live_per_gather=live / max(gathers, 1), lines_per_gather=lines / max(gathers, 1),
               rs_lines_per_row=lines / dim, bh_lines_per_row=bh_rows / dim * C / 8.0, runs=runs_total,
               long_run_share=run_entries / max(rs_entries, 1))
    res["oob_lines_per_row"] = res["rs_lines_per_row"] + res["bh_lines_per_row"]
    return res


def fmt(r):
    return (f"in {r['n_in']:.2f} out {r['n_out']:.2f} | bh/row {r['block_hops_per_row']:.2f} rs/row {r['row_slots_per_row']:.2f} (live {r['rs_live_per_row']:.2f}) "
            f"| wave-gathers/wave {r['wave_gathers_per_wave']:.2f} live {r['live_per_gather']:.1f} lines/gather {r['lines_per_gather']:.1f} "
            f"| OOB lines/row: rs {r['rs_lines_per_row']:.3f} + bh {r['bh_lines_per_row']:.3f} = {r['oob_lines_per_row']:.3f} | runs {r['runs']} long-run share {r['long_run_share']:.2f}")


def workload(name):
    if name == "C3":
        return models.hm_2dsquare(Nbath=3), 8, 12
    if name == "C4":
        return models.bhz_2d(), 8, 12
    if name == "C5":
        return models.hm_ring(), 9, 12
    if name == "C2":
        return models.hm_1dchain(), 6, 12
    raise SystemExit(name)


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
    m, npart, L = workload(wl)
    ns = m.Ns
    nimp = m.Nimp
    for spin in range(2 if m.Nspin > 1 else 1):
        edges = hop_edges(m, spin)
        und = {tuple(sorted(k)) for k in edges}
        print(f"== {wl} spin {spin}: Ns={ns} n={npart} L={L}, {len(und)} undirected edges")
        ident = list(range(ns))
        cands = {"reference order": ident}
        if wl == "C3":
            # orbitals: cluster 0-3, replica r: 4+4r .. 7+4r.  bit lists are written LOW -> HIGH as orbital ids
            def order(lowhigh):
                pos = [0] * ns
                for bit, o in enumerate(lowhigh):
                    pos[o] = bit
                return pos
            cl, r1, r2, r3 = [0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]
            cands["r1 r2 cluster | r3"] = order(r1 + r2 + cl + r3)
            cands["r1 r2 cluster(rev) | r3"] = order(r1 + r2 + cl[::-1] + r3)
            cands["cluster r1 r2 | r3 (=ref)"] = order(cl + r1 + r2 + r3)
            cands["r1 cluster r2 | r3"] = order(r1 + cl + r2 + r3)
        for nm, pos in cands.items():
            r = analyse(ns, npart, edges, pos, L)
            print(f"{nm:28s} {fmt(r)}")
        if "--search" in sys.argv:
            n_iter = int(sys.argv[sys.argv.index("--search") + 1])
            rng = np.random.default_rng(1)
            best = min(cands.values(), key=lambda p: analyse(ns, npart, edges, p, L)["oob_lines_per_row"])
            best = list(best)
            bv = analyse(ns, npart, edges, best, L)["oob_lines_per_row"]
            for it in range(n_iter):
                p = list(best)
                i, j = rng.choice(ns, 2, replace=False)
                p[i], p[j] = p[j], p[i]
                v = analyse(ns, npart, edges, p, L)["oob_lines_per_row"]
                if v < bv:
                    best, bv = p, v
                    print(f"  it {it}: {bv:.3f}  pos={best}")
            r = analyse(ns, npart, edges, best, L)
            print(f"{'best found':28s} {fmt(r)}\n   pos (orbital -> bit) = {best}")


if __name__ == "__main__":
    main()


def constructive(ns, npart, edges, L, top=6):
    """Candidates by construction: every high set of Ns-L orbitals among those touched by the fewest hops, low orbitals ordered by the number
    of hops that tie them to the high set (most tied = highest low bit: its row slots are the longest runs)."""
    und = {}
    for (a, b) in edges:
        und.setdefault(tuple(sorted((a, b))), 1)
    und = list(und)
    nh = ns - L
    scored = []
    for H in itertools.combinations(range(ns), nh):
        hs = set(H)
        touch = sum(1 for (a, b) in und if a in hs or b in hs)
        scored.append((touch, H))
    scored.sort()
    best_touch = scored[0][0]
    out = []
    for touch, H in scored:
        if touch > best_touch:
            break
        hs = set(H)
        deg = {o: 0 for o in range(ns) if o not in hs}
        for (a, b) in und:
            if a in hs and b not in hs:
                deg[b] += 1
            if b in hs and a not in hs:
                deg[a] += 1
        low = sorted(deg, key=lambda o: (deg[o], -o))  # least tied first = lowest bit
        pos = [0] * ns
        for bit, o in enumerate(low + list(H)):
            pos[o] = bit
        out.append((H, pos))
    return best_touch, out


if __name__ == "__main__" and "--construct" in sys.argv:
    wl = sys.argv[1]
    m, npart, L = workload(wl)
    for spin in range(2 if m.Nspin > 1 else 1):
        edges = hop_edges(m, spin)
        touch, cands = constructive(m.Ns, npart, edges, L)
        print(f"== {wl} spin {spin}: {len(cands)} high sets touched by {touch} hops (minimum)")
        res = []
        for H, pos in cands[:40]:
            r = analyse(m.Ns, npart, edges, pos, L)
            res.append((r["oob_lines_per_row"], H, pos, r))
        res.sort(key=lambda x: x[0])
        for v, H, pos, r in res[:3]:
            print(f"high {H}: {fmt(r)}\n   pos = {pos}")
