"""Same-box A/B of option "real_dw_pairs" (round 5): pass B of the REAL-vector product as the complex kernel on pairs of rows.
WORKLOAD=C3|C2|C5; prints ms per real product and per real Lanczos iteration for 0 / 1 and whether the two products are bit-identical."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cdmft-lanc-ed_amd"))
import torch, hxv
from hxv import models
wl = os.environ.get("WORKLOAD", "C3")
m, (nup, ndw) = {"C2": (models.hm_1dchain(), (6, 6)), "C3": (models.hm_2dsquare(Nbath=3), (8, 8)), "C3o": (models.hm_2dsquare(Nbath=3), (9, 8)), "C5": (models.hm_ring(6, 2), (9, 9))}[wl]
sec = hxv.HxvSector.from_model(m, nup, ndw)
v = sec.pad_real(torch.randn(sec.Dim, dtype=torch.float64, device="cuda"))
out = {}
for opt in (0, 1, 0, 1):
    sec.set_option("real_dw_pairs", opt)
    hv = torch.empty_like(v)
    sec.apply_device_real(v, hv)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30 if wl != "C5" else 5
    e0.record()
    for _ in range(n):
        sec.apply_device_real(v, hv)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    lz = sec.time_lanczos(20 if wl != "C5" else 4)
    print(f"{wl} DimUp={sec.DimUp} real_dw_pairs={opt}: real product {ms:.4f} ms, real Lanczos iteration {lz:.4f} ms", flush=True)
    out[opt] = sec.unpad_real(hv).clone()
print("bit-identical products:", torch.equal(out[0], out[1]), " max|diff| =", (out[0] - out[1]).abs().max().item())
