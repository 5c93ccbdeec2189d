"""In-process A/B of option sets on the full product. usage: ab_opts.py "k=v,k=v" "k=v" ..."""
import sys
sys.path.insert(0, "cdmft-lanc-ed_amd")
import torch, hxv
from hxv import models
m = models.hm_2dsquare(Nbath=3)
sec = hxv.HxvSector.from_model(m, 8, 8)
v = torch.randn(sec.fullElems, dtype=torch.float64, device="cuda") + 1j * torch.randn(sec.fullElems, dtype=torch.float64, device="cuda")
hv = torch.empty_like(v)
torch.cuda.synchronize()
sets = sys.argv[1:]
res = {s: [] for s in sets}
for rep in range(3):
    for s in sets:
        for kv in s.split(","):
            if kv:
                k, val = kv.split("="); sec.set_option(k, int(val))
        sec.time_apply(v, hv, 1)
        res[s].append(sec.time_apply(v, hv, 5))
for s in sets:
    print(f"{s:60s} ms " + " ".join(f"{x:.3f}" for x in res[s]) + f"  min {min(res[s]):.3f}")
