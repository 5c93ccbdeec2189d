"""BASELINE config 5 (Ns=18 half-filled sector, Dim = 2 363 904 400, 37.8 GB per vector) on ONE GPU:
tiled kernels vs the one-thread-per-element kernel at full size, and timing."""
import sys, time
sys.path.insert(0, "cdmft-lanc-ed_amd")
import torch, hxv
from hxv import models
m = models.hm_ring(6, 2)
t0 = time.time(); sec = hxv.HxvSector.from_model(m, 9, 9); print("build s", round(time.time() - t0, 2), "Dim", sec.Dim, sec.stats(), flush=True)
print("bits", sec.get_option("tile_bits_up"), sec.get_option("tile_bits_dw"), "blocks", sec.get_option("nblocks_up"), sec.get_option("nblocks_dw"), flush=True)
n = sec.fullElems
v = torch.empty(n, dtype=torch.complex128, device="cuda")
vr = torch.view_as_real(v)
chunk = 1 << 28
g = torch.Generator(device="cuda").manual_seed(5)
for a in range(0, 2 * n, chunk):
    b = min(a + chunk, 2 * n)
    vr.view(-1)[a:b] = torch.randn(b - a, dtype=torch.float64, device="cuda", generator=g)
hv = torch.empty_like(v); hv0 = torch.empty_like(v)
sec.set_option("kernel", 0); sec.apply_device(v, hv0)
sec.set_option("kernel", 1); sec.apply_device(v, hv)
torch.cuda.synchronize()
err = 0.0; mx = 0.0
for a in range(0, n, 1 << 27):
    b = min(a + (1 << 27), n)
    err = max(err, (hv[a:b] - hv0[a:b]).abs().max().item()); mx = max(mx, hv0[a:b].abs().max().item())
print("tiled vs naive rel err", err / mx, flush=True)
assert err / mx < 1e-13
for k in (1, 0):
    sec.set_option("kernel", k)
    ms = sec.time_apply(v, hv, 3)
    print("kernel", k, "ms", round(ms, 2), "GB/s alg", round(32 * sec.Dim / ms / 1e6, 1), flush=True)
