"""Run nrep products of one tiling config on the C3 sector (target of rocprofv3 runs)."""
import sys
sys.path.insert(0, "/root/repo/cdmft-lanc-ed_amd")
import torch, hxv
from hxv import models

cfg = sys.argv[1] if len(sys.argv) > 1 else "64,4,8,512"
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kernel = int(sys.argv[3]) if len(sys.argv) > 3 else 1
kb, C, R, T = map(int, cfg.split(","))
m = models.hm_2dsquare(Nbath=3)
sec = hxv.HxvSector.from_model(m, 8, 8)
sec.set_option("kernel", kernel)
if kernel == 1:
    sec.set_option("lds_budget_kb", kb); sec.set_option("cols_per_tile", C); sec.set_option("rows_per_tile", R)
    sec.set_option("threads_up", T); sec.set_option("threads_dw", T)
v = torch.randn(sec.Dim, dtype=torch.float64, device="cuda") + 1j * torch.randn(sec.Dim, dtype=torch.float64, device="cuda")
hv = torch.empty_like(v)
torch.cuda.synchronize()
print("ms", sec.time_apply(v, hv, nrep))
