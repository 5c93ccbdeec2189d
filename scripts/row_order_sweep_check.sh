#!/bin/bash
# Every Lanczos sector of C3 through ED_DIAG's loop (scripts/solve_sweep.py part c) with the device row order and with HXV_ROW_ORDER=0: the lowest
# eigenvalues of all sectors must agree (round 6: the order differs from sector to sector -- it follows nup).  -> gpurun_out/r06_row_order_sweep.txt
O=gpurun_out; mkdir -p $O
# (the first process on a fresh box pays more for every fresh hipMalloc -- the sectors that set a new maximum size: a throw-away run goes first)
HXV_ROW_ORDER=0 PARTS=c python scripts/solve_sweep.py > $O/sweep_warm.txt 2>&1
PARTS=c python scripts/solve_sweep.py > $O/sweep_on.txt 2>&1
HXV_ROW_ORDER=0 PARTS=c python scripts/solve_sweep.py > $O/sweep_off.txt 2>&1
python3 - <<'PY' > $O/r06_row_order_sweep.txt
import re
def load(f):
    out = {}
    for ln in open(f):
        m = re.match(r"\(\s*(\d+),\s*(\d+)\)\s+(\d+)\s+(\w+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s+(\S+)", ln)
        if m:
            out[(int(m.group(1)), int(m.group(2)))] = (m.group(4), float(m.group(9)), int(m.group(11)), float(m.group(12)))
    tail = [ln.strip() for ln in open(f) if ln.startswith("# ") or ln.startswith("## ")]
    return out, tail
on, t_on = load("gpurun_out/sweep_on.txt")
off, t_off = load("gpurun_out/sweep_off.txt")
lz = [k for k in on if on[k][0] == "lanczos"]
worst = max(abs(on[k][3] - off[k][3]) for k in lz)
print("device row order ON :", *t_on[-2:], sep="\n  ")
print("HXV_ROW_ORDER=0     :", *t_off[-2:], sep="\n  ")
print(f"{len(lz)} Lanczos sectors; worst |E0(on) - E0(off)| = {worst:.3e}; products on/off = {sum(on[k][2] for k in lz)} / {sum(off[k][2] for k in lz)}; "
      f"solve time on/off = {sum(on[k][1] for k in lz) / 1e3:.2f} / {sum(off[k][1] for k in lz) / 1e3:.2f} s")
PY
cat $O/r06_row_order_sweep.txt
