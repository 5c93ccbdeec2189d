"""Drop-in host-array call (spHtimesV_p => gpuMatVec_main => hxv_apply_host) at C3: PCIe-inclusive time per product."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cdmft-lanc-ed_amd"))
import numpy as np, torch, hxv
from hxv import models
wl = os.environ.get("SECTOR", "C3")
m, (nup, ndw) = (models.hm_2dsquare(Nbath=3), (8, 8)) if wl == "C3" else (models.hm_1dchain(), (6, 6))
sec = hxv.HxvSector.from_model(m, nup, ndw)
v = np.empty(sec.Dim, dtype=np.complex128); v.real = 1.0; v.imag = 0.5
hv = np.empty_like(v)
for it in range(4):
    t0 = time.perf_counter(); sec.apply_host(v, hv); dt = time.perf_counter() - t0
    print(f"{wl} apply_host call {it}: {dt*1e3:.1f} ms  ({2 * 16 * sec.Dim / dt / 1e9:.1f} GB/s over PCIe incl. kernels)", flush=True)
