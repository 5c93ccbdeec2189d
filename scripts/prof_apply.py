"""Run nrep products of one tiling config on the C3 sector (target of rocprofv3 runs)."""
import sys
sys.path.insert(0, "/root/repo/cdmft-lanc-ed_amd")
import torch, hxv
from hxv import models

cfg = sys.argv[1] if len(sys.argv) > 1 else "64,4,1024,64,4,1024,0"
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kernel = int(sys.argv[3]) if len(sys.argv) > 3 else 1
kbA, C, TA, kbB, R, TB, srt = map(int, cfg.split(","))
m = models.hm_2dsquare(Nbath=3)
sec = hxv.HxvSector.from_model(m, 8, 8)
sec.set_option("kernel", kernel)
if kernel == 1:
    for k, val in (("lds_budget_kb_up", kbA), ("cols_per_tile", C), ("threads_up", TA), ("lds_budget_kb_dw", kbB),
                   ("rows_per_tile", R), ("threads_dw", TB), ("sort_mode", srt)):
        sec.set_option(k, val)
v = torch.randn(sec.fullElems, dtype=torch.float64, device="cuda") + 1j * torch.randn(sec.fullElems, dtype=torch.float64, device="cuda")
hv = torch.empty_like(v)
torch.cuda.synchronize()
print("ms", sec.time_apply(v, hv, nrep))
