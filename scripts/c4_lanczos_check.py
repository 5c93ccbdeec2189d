"""C4 (complex BHZ): single-vector Lanczos before and after the thick-restart solver on the same handle."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cdmft-lanc-ed_amd"))
import torch, hxv
from hxv import models
m = models.bhz_2d(Nbath=1)
sec = hxv.HxvSector.from_model(m, 8, 8)
e0, vec, nit = sec.lanczos_eigh(512, 1e-13, native=True)
print("before: E0", e0, "iterations", nit, flush=True)
del vec
ev, X, nconv, nmv = sec.eigh_lowest(2, 20, native=True)
print("eigh_lowest", ev, nmv, "full/local", sec.get_option("eigh_last_full_passes"), sec.get_option("eigh_last_local_passes"), flush=True)
del X
torch.cuda.empty_cache()
e0, vec, nit = sec.lanczos_eigh(512, 1e-13, native=True)
print("after: E0", e0, "iterations", nit, flush=True)
ev, X, nconv, nmv = sec.eigh_lowest(2, 20, native=True)
hv = sec.apply_device(X[0].contiguous())
print("residual", (hv - ev[0] * X[0]).norm().item(), flush=True)
del X, hv
torch.cuda.empty_cache()
e0, vec, nit = sec.lanczos_eigh(512, 1e-13, native=True)
print("after apply_device: E0", e0, "iterations", nit, flush=True)
sec2 = hxv.HxvSector.from_model(m, 8, 8)
sec2.set_option("eigh_measure_all", 0)
sec2.set_option("real_vectors", 1)
ev, X, nconv, nmv = sec2.eigh_lowest(2, 20, native=True)
del X
torch.cuda.empty_cache()
e0, vec, nit = sec2.lanczos_eigh(512, 1e-13, native=True)
print("fresh handle with options set: E0", e0, "iterations", nit, flush=True)
