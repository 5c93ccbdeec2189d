"""In-process A/B of debug bits on the full product (same box, interleaved)."""
import sys
sys.path.insert(0, "cdmft-lanc-ed_amd")
import torch, hxv
from hxv import models
m = models.hm_2dsquare(Nbath=3)
sec = hxv.HxvSector.from_model(m, 8, 8)
v = torch.randn(sec.fullElems, dtype=torch.float64, device="cuda") + 1j * torch.randn(sec.fullElems, dtype=torch.float64, device="cuda")
hv = torch.empty_like(v)
torch.cuda.synchronize()
bits = [int(x) for x in sys.argv[1:]] or [0, 16]
res = {b: [] for b in bits}
for rep in range(4):
    for b in bits:
        sec.set_option("debug", b)
        sec.time_apply(v, hv, 1)
        res[b].append(sec.time_apply(v, hv, 5))
for b in bits:
    print("debug", b, "ms", " ".join(f"{x:.3f}" for x in res[b]), " min", f"{min(res[b]):.3f}")
