"""One product + a plain 2.65 GB device copy (known bytes) for calibrating the TCC counters."""
import sys
sys.path.insert(0, "/root/repo/cdmft-lanc-ed_amd")
import torch, hxv
from hxv import models
cfg = sys.argv[1] if len(sys.argv) > 1 else "64,4,1024,64,4,1024,0"
kbA, C, TA, kbB, R, TB, srt = map(int, cfg.split(","))
m = models.hm_2dsquare(Nbath=3)
sec = hxv.HxvSector.from_model(m, 8, 8)
for k, val in (("lds_budget_kb_up", kbA), ("cols_per_tile", C), ("threads_up", TA), ("lds_budget_kb_dw", kbB), ("rows_per_tile", R), ("threads_dw", TB), ("sort_mode", srt)):
    sec.set_option(k, val)
v = torch.randn(sec.fullElems, dtype=torch.float64, device="cuda") + 1j * torch.randn(sec.fullElems, dtype=torch.float64, device="cuda")
hv = torch.empty_like(v)
torch.cuda.synchronize()
for _ in range(2):
    hv.copy_(v)          # calibration: reads 2.650 GB, writes 2.650 GB
    torch.cuda.synchronize()
    sec.apply_device(v, hv)
    torch.cuda.synchronize()
