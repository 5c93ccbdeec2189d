#!/usr/bin/env python3
"""VERDICT r4 item 7, settled on the host first: would pass A's row-slot gathers touch fewer 128-byte lines on a 2-column x 4-row
line-tiled copy of v than on the column-major layout?  Replays the out-of-block row-slot gathers of pass A for the C3 sector (8,8)
(H_up from the CPU oracle, blocks of 12 low orbitals, one thread per block row, C = 4 columns, waves of 64 consecutive rows) and counts the
distinct 128-byte lines one wave-gather touches in both layouts.  No GPU.   python scripts/rowslot_lines_sim.py"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "cdmft-lanc-ed_amd"))
sys.path.insert(0, str(ROOT))
from hxv import models  # noqa: E402
from oracle.oracle import OracleSector  # noqa: E402

L, C, PITCH = 12, 4, 12872
m = models.hm_2dsquare(Nbath=3)
orc = OracleSector(m, 8, 8)
mp = orc.map_up().astype(np.int64)
rp, cols, vals = orc.csr("up")
cols = cols - 1
dim = len(mp)
hi = mp >> L
starts = np.flatnonzero(np.r_[True, hi[1:] != hi[:-1]])
block_of = np.searchsorted(starts, np.arange(dim), side="right") - 1
bounds = np.r_[starts, dim]
tot = {"colmajor": 0, "tiled": 0, "gathers": 0, "live": 0}
for b in range(len(starts)):
    r0, r1 = bounds[b], bounds[b + 1]
    # out-of-block entries of this block grouped by source block; a slot = (source block, multiplicity index)
    by_src = {}
    for i in range(r0, r1):
        seen = {}
        for p in range(rp[i], rp[i + 1]):
            sb = block_of[cols[p]]
            if sb == b:
                continue
            k = seen.get(sb, 0)
            seen[sb] = k + 1
            by_src.setdefault((sb, k), {})[i] = cols[p]
    for (sb, k), ent in by_src.items():
        s0, ns = bounds[sb], bounds[sb + 1] - bounds[sb]
        nb = r1 - r0
        # a block hop (whole block, identity on the low orbitals) is a contiguous run, not a row slot
        if len(ent) == nb and ns == nb and all(ent[i] - s0 == i - r0 for i in ent):
            continue
        for w0 in range(r0, r1, 64):
            rows = np.arange(w0, min(w0 + 64, r1))
            src = np.array([ent.get(i, s0) for i in rows])         # an empty slot gathers the block's first row (coefficient 0)
            live = sum(1 for i in rows if i in ent)
            if live == 0:
                continue
            # column-major [col][pitch]: line = (col, row*16 // 128); tiled: line = (row // 4, col // 2)
            cm = {(c, int(s) // 8) for s in src for c in range(C)}
            tl = {(int(s) // 4, c // 2) for s in src for c in range(C)}
            tot["colmajor"] += len(cm)
            tot["tiled"] += len(tl)
            tot["gathers"] += 1
            tot["live"] += live
print(f"C3 (8,8), H_up, L={L}, C={C}: {tot['gathers']} wave-gathers of row slots, {tot['live'] / tot['gathers']:.1f} live lanes per wave-gather")
print(f"  distinct 128-byte lines per wave-gather: column-major {tot['colmajor'] / tot['gathers']:.2f}, 2-col x 4-row tiled {tot['tiled'] / tot['gathers']:.2f} "
      f"(ratio {tot['tiled'] / tot['colmajor']:.3f})")
