"""Target of rocprofv3 runs: apply an option set ("k=v,k=v") to the C3 sector (WORKLOAD env: C2..C5) and run nrep products."""
import os, sys
sys.path.insert(0, "/root/repo/cdmft-lanc-ed_amd")
import torch, hxv
from hxv import models
wl = os.environ.get("WORKLOAD", "C3")
m, (nup, ndw) = {"C2": (models.hm_1dchain(), (6, 6)), "C3": (models.hm_2dsquare(Nbath=3), (8, 8)),
                 "C4": (models.bhz_2d(Nbath=1), (8, 8)), "C4K": (models.bhz_2d(Nbath=1, Ust=0.5, Jh=0.1, Jx=0.1, Jp=0.1), (8, 8)), "C5": (models.hm_ring(6, 2), (9, 9))}[wl]
sec = hxv.HxvSector.from_model(m, nup, ndw)
for kv in (sys.argv[1] if len(sys.argv) > 1 else "").split(","):
    if kv:
        k, val = kv.split("="); sec.set_option(k, int(val))
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if os.environ.get("REAL", "0") == "1":
    # REAL-vector product (real H): the kernels that carry hxv_lanczos_eigh / hxv_eigh_lowest / the real Green's-function channels
    v = sec.pad_real(torch.randn(sec.Dim, dtype=torch.float64, device="cuda"))
    hv = torch.empty_like(v)
    sec.apply_device_real(v, hv)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(nrep):
        sec.apply_device_real(v, hv)
    e1.record()
    torch.cuda.synchronize()
    print("ms (real vectors)", e0.elapsed_time(e1) / nrep)
else:
    v = torch.randn(sec.fullElems, dtype=torch.float64, device="cuda") + 1j * torch.randn(sec.fullElems, dtype=torch.float64, device="cuda")
    hv = torch.empty_like(v)
    torch.cuda.synchronize()
    print("ms", sec.time_apply(v, hv, nrep))
