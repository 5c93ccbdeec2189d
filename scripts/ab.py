"""In-process A/B of option sets on one workload (WORKLOAD env: C2|C3|C4|C4K|C5, default C3).
usage: ab.py "k=v,k=v" "k=v" ...   every set is applied on top of the library defaults of a FRESH handle.
Prints, per set: full product, pass B alone, pass A alone (ms, best of 3 x 5), relative error against the first set
(the first set should be the shipped defaults: "") and the plan it produced."""
import os, sys
os.environ.setdefault("HXV_EXPERIMENTS", "1")  # the `passes` timing option
sys.path.insert(0, "cdmft-lanc-ed_amd")
import torch, hxv
from hxv import models

wl = os.environ.get("WORKLOAD", "C3")
m, (nup, ndw) = {"C2": (models.hm_1dchain(), (6, 6)), "C3": (models.hm_2dsquare(Nbath=3), (8, 8)),
                 "C4": (models.bhz_2d(Nbath=1), (8, 8)), "C4K": (models.bhz_2d(Nbath=1, Ust=0.5, Jh=0.1, Jx=0.1, Jp=0.1), (8, 8)),
                 "C5": (models.hm_ring(6, 2), (9, 9))}[wl]
sets = sys.argv[1:] or [""]
ref = None
v = hv = None
for s in sets:
    sec = hxv.HxvSector.from_model(m, nup, ndw)
    if v is None:
        n = sec.fullElems
        v = torch.empty(n, dtype=torch.complex128, device="cuda")
        vr = torch.view_as_real(v).view(-1)
        g = torch.Generator(device="cuda").manual_seed(5)
        for a in range(0, 2 * n, 1 << 28):
            b = min(a + (1 << 28), 2 * n)
            vr[a:b] = torch.randn(b - a, dtype=torch.float64, device="cuda", generator=g)
        hv = torch.empty(sec.localElems, dtype=torch.complex128, device="cuda")
    try:
        for kv in s.split(","):
            if kv:
                k, val = kv.split("=")
                sec.set_option(k, int(val))
        sec.time_apply(v, hv, 1)
        torch.cuda.synchronize()
        if ref is None:
            ref = hv.clone()
            err = 0.0
        else:
            err = 0.0
            mx = 0.0
            for a in range(0, hv.numel(), 1 << 27):
                b = min(a + (1 << 27), hv.numel())
                err = max(err, (hv[a:b] - ref[a:b]).abs().max().item())
                mx = max(mx, ref[a:b].abs().max().item())
            err /= mx
        nrep = 3 if wl == "C5" else 5
        full = min(sec.time_apply(v, hv, nrep) for _ in range(3))
        sec.set_option("passes", 2)
        sec.time_apply(v, hv, 1)
        tb = min(sec.time_apply(v, hv, nrep) for _ in range(3))
        sec.set_option("passes", 1)
        sec.time_apply(v, hv, 1)
        ta = min(sec.time_apply(v, hv, nrep) for _ in range(3))
        plan = "bits %d/%d blocks %d/%d maxblk %d/%d C %d R %d" % (sec.get_option("tile_bits_up"), sec.get_option("tile_bits_dw"),
                                                                 sec.get_option("nblocks_up"), sec.get_option("nblocks_dw"),
                                                                 sec.get_option("max_block_up"), sec.get_option("max_block_dw"),
                                                                 sec.get_option("cols_per_tile"), sec.get_option("rows_per_tile"))
        print(f"{wl} [{s:58s}] full {full:8.3f} ms  passB {tb:8.3f}  passA {ta:8.3f}  relerr {err:.1e}  {plan}", flush=True)
    except Exception as e:  # an option set the plan refuses: report and go on
        print(f"{wl} [{s:58s}] FAILED: {e}", flush=True)
    sec.close()
    del sec
