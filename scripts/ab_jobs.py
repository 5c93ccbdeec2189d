"""In-process A/B of the pipelined job kernels against the one-tile-per-workgroup kernels (C3 unless WORKLOAD is set).
usage: ab_jobs.py "k=v,k=v" ...   each set is applied on top of the defaults; prints full product, pass B alone, difference."""
import os, sys
os.environ.setdefault("HXV_EXPERIMENTS", "1")  # the passes / job_debug timing options
sys.path.insert(0, "cdmft-lanc-ed_amd")
import torch, hxv
from hxv import models
wl = os.environ.get("WORKLOAD", "C3")
m, (nup, ndw) = {"C2": (models.hm_1dchain(), (6, 6)), "C3": (models.hm_2dsquare(Nbath=3), (8, 8)),
                 "C4": (models.bhz_2d(Nbath=1), (8, 8)), "C4K": (models.bhz_2d(Nbath=1, Ust=0.5, Jh=0.1, Jx=0.1, Jp=0.1), (8, 8)), "C5": (models.hm_ring(6, 2), (9, 9))}[wl]
sec = hxv.HxvSector.from_model(m, nup, ndw)
v = torch.randn(sec.fullElems, dtype=torch.float64, device="cuda") + 1j * torch.randn(sec.fullElems, dtype=torch.float64, device="cuda")
hv = torch.empty_like(v)
ref = None
torch.cuda.synchronize()
sets = sys.argv[1:] or ["job_up=0", "job_up=1"]
defaults = {"job_up": 1, "job_cols": 1, "job_groups": 100, "job_stages": 4, "job_debug": 0}
for s in sets:
    opts = dict(defaults)
    for kv in s.split(","):
        if kv:
            k, val = kv.split("="); opts[k] = int(val)
    for k, val in opts.items():
        sec.set_option(k, val)
    sec.set_option("passes", 3)
    sec.time_apply(v, hv, 1)
    torch.cuda.synchronize()
    if ref is None:
        sec.set_option("job_up", 0); sec.time_apply(v, hv, 1); torch.cuda.synchronize(); ref = hv.clone(); sec.set_option("job_up", opts["job_up"]); sec.time_apply(v, hv, 1); torch.cuda.synchronize()
    err = float((hv - ref).abs().max() / ref.abs().max())
    full = min(sec.time_apply(v, hv, 5) for _ in range(3))
    sec.set_option("passes", 2)
    sec.time_apply(v, hv, 1)
    tb = min(sec.time_apply(v, hv, 5) for _ in range(3))
    sec.set_option("passes", 3)
    print(f"{s:50s} active {sec.get_option('job_up_active')}  full {full:.3f} ms  passB {tb:.3f}  passA {full - tb:.3f}  relerr-vs-tiled {err:.2e}", flush=True)
