"""Same-box A/B of option "wt_colmajor" (round 5): the blocked dw-hop scratch with column-major patches (pass A reads Rp*16 contiguous bytes
per column and patch instead of every lane its own 64-byte stretch C times).  WORKLOAD=C3|C4|C5|C2: complex product, real product (real H),
Lanczos iterations; results must be bit-identical."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cdmft-lanc-ed_amd"))
import torch, hxv
from hxv import models
wl = os.environ.get("WORKLOAD", "C3")
m, (nup, ndw) = {"C2": (models.hm_1dchain(), (6, 6)), "C3": (models.hm_2dsquare(Nbath=3), (8, 8)), "C4": (models.bhz_2d(Nbath=1), (8, 8)), "C5": (models.hm_ring(6, 2), (9, 9))}[wl]
sec = hxv.HxvSector.from_model(m, nup, ndw)
n = sec.fullElems
v = torch.empty(n, dtype=torch.complex128, device="cuda")
vr = torch.view_as_real(v).view(-1)
g = torch.Generator(device="cuda").manual_seed(7)
for a in range(0, 2 * n, 1 << 28):
    b = min(a + (1 << 28), 2 * n)
    vr[a:b] = torch.randn(b - a, dtype=torch.float64, device="cuda", generator=g)
torch.view_as_real(v).view(-1, sec.pitch, 2)[:, sec.DimUp:, :] = 0.0
hv = torch.empty_like(v)
real = sec.real_vectors_available
if real:
    xr = sec.pad_real(torch.randn(sec.Dim, dtype=torch.float64, device="cuda", generator=g))
    hr = torch.empty_like(xr)
out, outr = {}, {}
reps = 30 if wl != "C5" else 4
for opt in (0, 1, 0, 1):
    sec.set_option("wt_colmajor", opt)
    sec.time_apply(v, hv, 2)
    ms = sec.time_apply(v, hv, reps)
    out[opt] = hv.clone() if wl != "C5" else hv[: 1 << 24].clone()
    line = f"{wl} wt_colmajor={opt}: complex product {ms:.4f} ms"
    if real:
        sec.apply_device_real(xr, hr)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            sec.apply_device_real(xr, hr)
        e1.record()
        torch.cuda.synchronize()
        outr[opt] = hr.clone() if wl != "C5" else hr[: 1 << 24].clone()
        line += f", real product {e0.elapsed_time(e1) / reps:.4f} ms"
    if wl != "C5":
        sec.set_option("real_vectors", 0)
        lz = sec.time_lanczos(10)
        sec.set_option("real_vectors", 1)
        line += f", complex Lanczos iteration {lz:.4f} ms"
        if real:
            line += f", real Lanczos iteration {sec.time_lanczos(10):.4f} ms"
    print(line, flush=True)
print("bit-identical complex products:", torch.equal(out[0], out[1]), (" real products: " + str(torch.equal(outr[0], outr[1]))) if real else "")
