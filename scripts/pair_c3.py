"""Per-iteration time of the paired real tridiagonalisation at C3 against the single runs (complex vectors / real vectors)."""
import sys, time
sys.path.insert(0, "cdmft-lanc-ed_amd")
import numpy as np, torch, hxv
from hxv import models
m = models.hm_2dsquare(Nbath=3)
sec = hxv.HxvSector.from_model(m, 8, 8)
g = torch.Generator(device="cuda").manual_seed(3)
va = torch.zeros(sec.localElems, dtype=torch.complex128, device="cuda")
vb = torch.zeros_like(va)
for v in (va, vb):
    x = torch.randn(sec.DimDw, sec.DimUp, dtype=torch.float64, device="cuda", generator=g)
    x /= x.norm()
    torch.view_as_real(v).view(sec.DimDw, sec.pitch, 2)[:, : sec.DimUp, 0] = x
def timed(f, n1, n2):
    f(n1); torch.cuda.synchronize()
    t0 = time.perf_counter(); f(n1); torch.cuda.synchronize(); t1 = time.perf_counter(); f(n2); torch.cuda.synchronize(); t2 = time.perf_counter()
    return ((t2 - t1) - (t1 - t0)) / (n2 - n1) * 1e3
ms_pair = timed(lambda n: sec.lanczos_tridiag_pair(va, vb, n), 5, 25)
sec.set_option("real_vectors", 1)
ms_real = timed(lambda n: sec.lanczos_tridiag(va, n), 5, 25)
sec.set_option("real_vectors", 0)
ms_cplx = timed(lambda n: sec.lanczos_tridiag(va, n), 5, 25)
print(f"C3 tridiag per iteration: paired {ms_pair:.3f} ms per pair = {ms_pair/2:.3f} ms per channel; single real-vector run {ms_real:.3f} ms; single complex-vector run {ms_cplx:.3f} ms")
