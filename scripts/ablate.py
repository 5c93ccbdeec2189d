import sys, os
os.environ.setdefault("HXV_EXPERIMENTS", "1")
sys.path.insert(0, "cdmft-lanc-ed_amd")
import torch, hxv
from hxv import models
m = models.hm_2dsquare(Nbath=3)
NUP, NDW = map(int, os.environ.get("SECTOR", "8,8").split(","))
sec = hxv.HxvSector.from_model(m, NUP, NDW)
v = torch.randn(sec.fullElems, dtype=torch.float64, device="cuda") + 1j * torch.randn(sec.fullElems, dtype=torch.float64, device="cuda")
hv = torch.empty_like(v)
torch.cuda.synchronize()
def t(nrep=3):
    sec.time_apply(v, hv, 1); return sec.time_apply(v, hv, nrep)
cfgs = sys.argv[1:] or ["64,4,1024,64,4,1024,0"]
for cfg in cfgs:
    kbA, C, TA, kbB, R, TB, srt = map(int, cfg.split(","))
    for k, val in (("lds_budget_kb_up", kbA), ("cols_per_tile", C), ("threads_up", TA), ("lds_budget_kb_dw", kbB), ("rows_per_tile", R), ("threads_dw", TB), ("sort_mode", srt)):
        sec.set_option(k, val)
    print("sector", NUP, NDW, "cfg", cfg, "bits", sec.get_option("tile_bits_up"), sec.get_option("tile_bits_dw"))
    for name, dbg in (("full", 0), ("no-outer", 1), ("no-inner", 2), ("no-hops", 3), ("B no-hvread", 4), ("B no-outer no-hvread", 5), ("nothing", 7), ("full, nt hv", 8)):
        sec.set_option("debug", dbg)
        sec.set_option("passes", 1); ta = t()
        sec.set_option("passes", 2); tb = t()
        print(f"  {name:24s} A {ta:.2f} ms   B {tb:.2f} ms", flush=True)
    sec.set_option("debug", 0); sec.set_option("passes", 3)
