"""Timing sweep of the tiled kernels on the C3 sector (run on the GPU box)."""
import sys, time, itertools
sys.path.insert(0, "cdmft-lanc-ed_amd")
import torch, hxv
from hxv import models

m = models.hm_2dsquare(Nbath=3)
sec = hxv.HxvSector.from_model(m, 8, 8)
v = torch.randn(sec.Dim, dtype=torch.float64, device="cuda") + 1j * torch.randn(sec.Dim, dtype=torch.float64, device="cuda")
hv = torch.empty_like(v)
torch.cuda.synchronize()
alg = 32 * sec.Dim


def t(nrep=3):
    sec.time_apply(v, hv, 1)
    return sec.time_apply(v, hv, nrep)


sec.set_option("kernel", 0)
print("naive ms", t(), flush=True)
sec.set_option("kernel", 1)
cfgs = sys.argv[1:] or ["64,4,8,256", "32,4,8,256", "16,4,8,256", "64,4,8,512", "128,8,8,512", "32,2,4,256", "64,8,16,512", "32,8,16,256", "16,2,4,256", "8,2,4,256"]
for cfg in cfgs:
    kb, C, R, T = map(int, cfg.split(","))
    try:
        sec.set_option("lds_budget_kb", kb); sec.set_option("cols_per_tile", C); sec.set_option("rows_per_tile", R)
        sec.set_option("threads_up", T); sec.set_option("threads_dw", T)
    except Exception as e:
        print(cfg, "skip", e); continue
    g = sec.get_option
    info = f"bits {g('tile_bits_up')}/{g('tile_bits_dw')} kin/kout up {g('k_in_up')}/{g('k_out_up')} dw {g('k_in_dw')}/{g('k_out_dw')} out-frac up {g('n_out_up')/(g('n_in_up')+g('n_out_up')):.2f} dw {g('n_out_dw')/(g('n_in_dw')+g('n_out_dw')):.2f}"
    sec.set_option("passes", 1); ta = t()
    sec.set_option("passes", 2); tb = t()
    sec.set_option("passes", 3); tt = t()
    print(f"kb={kb} C={C} R={R} T={T}: A {ta:.2f} ms  B {tb:.2f} ms  total {tt:.2f} ms  ({alg/tt/1e6:.0f} GB/s alg)  {info}", flush=True)
