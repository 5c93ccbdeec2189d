"""Correctness of an option set at C3: tiled kernels vs the one-thread-per-element kernel. usage: check_opts.py "k=v,k=v" """
import sys
sys.path.insert(0, "cdmft-lanc-ed_amd")
import torch, hxv
from hxv import models
sec = hxv.HxvSector.from_model(models.hm_2dsquare(Nbath=3), 8, 8)
v = torch.randn(sec.fullElems, dtype=torch.float64, device="cuda") + 1j * torch.randn(sec.fullElems, dtype=torch.float64, device="cuda")
sec.set_option("kernel", 0); ref = sec.apply_device(v).clone()
sec.set_option("kernel", 1)
for kv in sys.argv[1].split(","):
    k, val = kv.split("="); sec.set_option(k, int(val))
out = sec.apply_device(v)
print(sys.argv[1], "rel err vs naive", ((out - ref).abs().max() / ref.abs().max()).item())
