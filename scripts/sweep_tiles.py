"""Timing sweep of the tiled kernels on the C3 sector (run on the GPU box).
usage: sweep_tiles.py [cfg ...]   cfg = kbA,C,TA,kbB,R,TB,sort"""
import sys, time, itertools
sys.path.insert(0, "cdmft-lanc-ed_amd")
import torch, hxv
from hxv import models

import os
m = models.hm_2dsquare(Nbath=3)
NUP, NDW = map(int, os.environ.get("SECTOR", "8,8").split(","))
sec = hxv.HxvSector.from_model(m, NUP, NDW)
print("sector", NUP, NDW, "DimUp", sec.DimUp, "DimDw", sec.DimDw)
v = torch.randn(sec.fullElems, dtype=torch.float64, device="cuda") + 1j * torch.randn(sec.fullElems, dtype=torch.float64, device="cuda")
hv = torch.empty_like(v)
torch.cuda.synchronize()
alg = 32 * sec.Dim


def t(nrep=3):
    sec.time_apply(v, hv, 1)
    return sec.time_apply(v, hv, nrep)


sec.set_option("kernel", 1)
cfgs = sys.argv[1:] or ["64,4,1024,64,4,1024,2", "64,4,512,64,4,512,2", "64,4,1024,64,4,1024,0", "64,4,1024,64,4,1024,1",
                        "32,4,512,32,4,512,2", "32,2,1024,32,2,1024,2", "128,8,1024,128,8,1024,2", "64,4,1024,32,4,512,2",
                        "64,4,1024,128,8,1024,2", "16,4,256,16,4,256,2"]
for cfg in cfgs:
    parts = list(map(int, cfg.split(",")))
    kbA, C, TA, kbB, R, TB, srt = parts[:7]
    padA, padB = (parts[7:9] + [0, 0])[:2] if len(parts) > 7 else (0, 0)
    sec.set_option("wt_cols", int(os.environ.get("WT_COLS", "4")))
    try:
        sec.set_option("lds_min_kb_up", padA); sec.set_option("lds_min_kb_dw", padB)
        for k, val in (("lds_budget_kb_up", kbA), ("cols_per_tile", C), ("threads_up", TA), ("lds_budget_kb_dw", kbB),
                       ("rows_per_tile", R), ("threads_dw", TB), ("sort_mode", srt)):
            sec.set_option(k, val)
    except Exception as e:
        print(cfg, "skip", e); continue
    g = sec.get_option
    info = (f"bits {g('tile_bits_up')}/{g('tile_bits_dw')} slots up in/out {g('slots_in_up_x100')/100:.1f}/{g('slots_out_up_x100')/100:.1f} "
            f"dw in {g('slots_in_dw_x100')/100:.1f} bh/rs up {g('bh_up_x100')/100:.1f}/{g('rs_up_x100')/100:.1f} dw {g('bh_dw_x100')/100:.1f}/{g('rs_dw_x100')/100:.1f} out-frac up {g('n_out_up')/(g('n_in_up')+g('n_out_up')):.2f} dw {g('n_out_dw')/(g('n_in_dw')+g('n_out_dw')):.2f}")
    sec.set_option("passes", 1); ta = t()
    sec.set_option("passes", 2); tb = t()
    sec.set_option("passes", 3); tt = t()
    print(f"A(kb={kbA},C={C},T={TA},pad={padA}) B(kb={kbB},R={R},T={TB},pad={padB}) sort={srt}: A {ta:.2f} ms  B {tb:.2f} ms  total {tt:.2f} ms  ({alg/tt/1e6:.0f} GB/s alg)  {info}", flush=True)
