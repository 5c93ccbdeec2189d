"""Edge inputs of the device drivers: zero / NaN start vectors, nlanc beyond Dim, one-dimensional sectors, tiny tolerances."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cdmft-lanc-ed_amd"))
import numpy as np, torch, hxv
from hxv import models
m = models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6])
sec = hxv.HxvSector.from_model(m, 3, 3)
def show(what, f):
    try:
        r = f()
        print(f"{what}: returned {r}", flush=True)
    except hxv.HxvError as e:
        print(f"{what}: HxvError {str(e)[:140]}", flush=True)
z = torch.zeros(sec.Dim, dtype=torch.complex128, device="cuda")
show("tridiag from a zero vector", lambda: [x[:3] if hasattr(x, '__len__') else x for x in sec.lanczos_tridiag(z, 10)])
nanv = z.clone(); nanv[5] = float("nan")
show("tridiag from a vector with a NaN", lambda: [x[:3] if hasattr(x, '__len__') else x for x in sec.lanczos_tridiag(nanv, 10)])
v = torch.randn(sec.Dim, dtype=torch.complex128, device="cuda"); v /= v.norm()
show("tridiag nlanc = Dim + 50", lambda: sec.lanczos_tridiag(v, sec.Dim + 50)[2])
show("pair from zero vectors", lambda: [r[2] for r in sec.lanczos_tridiag_pair(z, z, 10)])
show("eigh threshold 0", lambda: sec.lanczos_eigh(512, 0.0, want_vector=False)[0])
show("eigh_lowest tol 1e-30", lambda: sec.eigh_lowest(2, 12, tol=1e-30, want_vectors=False)[0])
sec.close()
one = hxv.HxvSector.from_model(m, 0, 0)     # Dim = 1
x1 = torch.ones(1, dtype=torch.complex128, device="cuda")
show("Dim=1 product", lambda: one.apply_device(one.pad(x1)).cpu().numpy()[:1])
show("Dim=1 tridiag", lambda: one.lanczos_tridiag(x1, 5))
show("Dim=1 eigh", lambda: one.lanczos_eigh(10, 1e-12, want_vector=False)[0])
show("Dim=1 eigh_lowest", lambda: one.eigh_lowest(1, 4, want_vectors=False))
one.close()
