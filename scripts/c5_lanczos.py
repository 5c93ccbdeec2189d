"""BASELINE config 5 (Ns=18 half-filled sector, Dim = 2 363 904 400) on ONE GPU: ground state by the device Lanczos
(sp_lanc_eigh call shape).  H is real, so the driver runs on real vectors (18.9 GB each)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cdmft-lanc-ed_amd"))
import torch, hxv
from hxv import models
m = models.hm_ring(6, 2)
sec = hxv.HxvSector.from_model(m, 9, 9)
print("Dim", sec.Dim, "real vectors available", sec.real_vectors_available, flush=True)
for mode in (1, 0):
    sec.set_option("real_vectors", mode)
    ms = sec.time_lanczos(5)
    print(f"Ns=18 Lanczos iteration real_vectors={mode}: {ms:.1f} ms", flush=True)
    torch.cuda.empty_cache()
sec.set_option("real_vectors", 1)
t0 = time.time()
e0, vec, nit = sec.lanczos_eigh(300, 1e-12, native=True)
torch.cuda.synchronize()
print(f"Ns=18 lanczos_eigh (real vectors, two-pass with eigenvector): E0={e0:.10f} iterations={nit} {time.time()-t0:.1f}s", flush=True)
hv = sec.apply_device(vec)
r = 0.0
for a in range(0, vec.numel(), 1 << 27):
    b = min(a + (1 << 27), vec.numel())
    r += (hv[a:b] - e0 * vec[a:b]).abs().pow(2).sum().item()
print(f"  residual |H x - E0 x| = {r ** 0.5:.2e}", flush=True)
