"""Expected LDS bank-conflict ratio of the in-block gathers, computed on the host from the tables the engine builds (no GPU).

VERDICT r2 asked why SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE stayed at 0.48 (pass B) / 0.41 (pass A) after the XOR swizzle.
MI355X_MICROARCH.md, section LDS: a wave's ds_read_b128 is served in 4 fixed groups of 16 lanes, one LDS cycle per group when the
16 lanes hit 16 different 16-byte bank quads (64 banks x 4 B); every further DISTINCT address on a busy quad adds a cycle
(identical addresses broadcast).  SQ_LDS_BANK_CONFLICT counts the extra cycles, SQ_LDS_IDX_ACTIVE all cycles.
This script replays the in-block hop loops of both passes on the C3 tables (block = prefix block of 12 low orbitals, tile layouts of
csrc/hxv_tiled.hip) and counts those cycles, next to what 16 uniformly random quads would give.
usage: lds_conflicts.py [C2|C3|C4]"""
import sys
from math import comb

import numpy as np

sys.path.insert(0, "cdmft-lanc-ed_amd")
sys.path.insert(0, ".")
from hxv import models
from oracle.oracle import OracleSector

GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
GROUPS += [[l + 32 for l in g] for g in GROUPS]


def cycles(quads):
    """LDS cycles of one ds_read_b128 wave-instruction: per lane group, the largest number of DISTINCT addresses on one bank quad.
    quads: 64 (address // 16) values, -1 for an inactive lane."""
    tot = 0
    for g in GROUPS:
        per = {}
        for l in g:
            a = quads[l]
            if a >= 0:
                per.setdefault(a % 16, set()).add(a)
        tot += max((len(v) for v in per.values()), default=0)
    return tot


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
    m, (nup, ndw) = {"C2": (models.hm_1dchain(), (6, 6)), "C3": (models.hm_2dsquare(Nbath=3), (8, 8)), "C4": (models.bhz_2d(Nbath=1), (8, 8))}[wl]
    orc = OracleSector(m, nup, ndw)
    L = 12 if m.Ns >= 12 else m.Ns
    rng = np.random.default_rng(0)
    rnd = np.mean([cycles(list(rng.integers(0, 1 << 20, 64))) for _ in range(2000)])
    print(f"{wl}: 64 uniformly random 16-byte addresses: {rnd:.2f} cycles per instruction (4 without conflicts) -> conflict ratio {(rnd - 4) / rnd:.2f}")
    for which, amap in (("up", orc.map_up()), ("dw", orc.map_dw())):
        rp, cols, _ = orc.csr(which)
        cols = cols - 1
        dim = len(amap)
        hi = amap >> L
        starts = [0] + [i for i in range(1, dim) if hi[i] != hi[i - 1]] + [dim]
        blk = np.zeros(dim, dtype=np.int64)
        for k in range(len(starts) - 1):
            blk[starts[k]:starts[k + 1]] = k
        # in-block ELL lists in the CSR (row-list) order, offsets relative to the block
        lists = [[int(c - starts[blk[i]]) for c in cols[rp[i]:rp[i + 1]] if blk[c] == blk[i]] for i in range(dim)]
        act = conf = 0
        for k in range(len(starts) - 1):
            b0, n = starts[k], starts[k + 1] - starts[k]
            if which == "up":
                order = list(range(n))                       # pass A: natural row order, tile lds[cc*n + row], C = 4 columns
            else:
                order = sorted(range(n), key=lambda q: -len(lists[b0 + q]))  # pass B: columns sorted by in-block count (stable)
            for w0 in range(0, n, 64):
                lanes = order[w0:w0 + 64]
                kmax = max(len(lists[b0 + q]) for q in lanes)
                for slot in range(kmax):
                    offs = [lists[b0 + q][slot] if slot < len(lists[b0 + q]) else 0 for q in lanes] + [-1] * (64 - len(lanes))
                    if which == "up":
                        for cc in range(4):
                            c = cycles([(cc * n + o) if o >= 0 else -1 for o in offs])
                            act += c
                            conf += c - 4 if len(lanes) == 64 else max(0, c - (len(lanes) + 15) // 16)
                    else:
                        for rr in range(4):                  # tile [column][4 rows], row position XOR-swizzled by (column >> 2) & 3
                            c = cycles([(o * 4 + (rr ^ ((o >> 2) & 3))) if o >= 0 else -1 for o in offs])
                            act += c
                            conf += c - 4 if len(lanes) == 64 else max(0, c - (len(lanes) + 15) // 16)
        print(f"  pass {'A' if which == 'up' else 'B'} ({which} hops, {len(starts) - 1} blocks): in-block gathers take {act} LDS cycles per tile set, "
              f"{conf} of them conflicts -> ratio {conf / act:.2f}")


if __name__ == "__main__":
    main()
