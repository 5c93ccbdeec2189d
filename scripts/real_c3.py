"""C3: product on REAL vectors vs complex vectors (same kernels, half the bytes)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cdmft-lanc-ed_amd"))
import torch, hxv
from hxv import models
sec = hxv.HxvSector.from_model(models.hm_2dsquare(Nbath=3), 8, 8)
L = hxv.engine.load_library()
n = L.hxv_realvec_elems(sec._h)
x = sec.pad_real(torch.randn(sec.Dim, dtype=torch.float64, device="cuda"))
hr = torch.zeros(n, dtype=torch.float64, device="cuda")
xc = sec.pad(sec.unpad_real(x).to(torch.complex128)); hc = torch.zeros(sec.localElems, dtype=torch.complex128, device="cuda")
def t(f, nrep=20):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(nrep): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / nrep * 1e3
tc = t(lambda: sec.apply_device(xc, hc)); tr = t(lambda: sec.apply_device_real(x, hr))
err = (sec.unpad_real(hr) - sec.unpad(hc).real).abs().max().item() / sec.unpad(hc).abs().max().item()
print(f"C3 complex product {tc:.3f} ms   real product {tr:.3f} ms   ratio {tc/tr:.2f}   rel diff {err:.1e}")
for mode in (0, 1):
    sec.set_option("real_vectors", mode)
    ms = sec.time_lanczos(20)
    print(f"C3 Lanczos iteration real_vectors={mode}: {ms:.3f} ms  ({1e3/ms:.1f} matvecs/s)  real_last={sec.get_option('lanczos_real_last')}", flush=True)
del x, hr, xc, hc
torch.cuda.empty_cache()
for mode in (1,):
    sec.set_option("real_vectors", mode)
    t0 = time.time(); ev, X, nconv, nmv = sec.eigh_lowest(2, 20, native=True); torch.cuda.synchronize(); dt = time.time() - t0
    print(f"C3 eigh_lowest real_vectors={mode}: E={ev} nconv={nconv} matvecs={nmv} {dt:.2f}s ({dt/nmv*1e3:.1f} ms/step)", flush=True)
    r = (sec.apply_device(X[0].contiguous()) - ev[0] * X[0]).norm().item()
    print(f"  residual {r:.2e}")
    del X; torch.cuda.empty_cache()
    t0 = time.time(); e0, vec, nit = sec.lanczos_eigh(512, 1e-13, native=True); torch.cuda.synchronize(); dt = time.time() - t0
    print(f"C3 lanczos_eigh real_vectors={mode}: E0={e0:.12f} iterations={nit} {dt:.2f}s")
