"""Launch-bound sizes: a 200-step tridiagonalisation at C2 (Ns=12) with and without the hipGraph path."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cdmft-lanc-ed_amd"))
import numpy as np, torch, hxv
from hxv import models
for name, m, (nu, nd) in (("C2", models.hm_1dchain(), (6, 6)), ("Ns=8", models.hm_1dchain(Nlat=2, Nbath=3), (4, 4)), ("C3 (4,8)", models.hm_2dsquare(Nbath=3), (4, 8))):
    sec = hxv.HxvSector.from_model(m, nu, nd)
    v = np.random.default_rng(0).standard_normal(sec.Dim); v = (v / np.linalg.norm(v)).astype(np.complex128)
    dv = sec.pad(torch.from_numpy(v).cuda())
    for g in (0, 1, 0, 1):
        sec.set_option("lanczos_graph", g)
        torch.cuda.synchronize(); t0 = time.perf_counter(); a, b, n = sec.lanczos_tridiag(dv, 200); dt = time.perf_counter() - t0
        print(f"{name} Dim={sec.Dim}: lanczos_graph={g}: 200 steps {dt*1e3:.2f} ms ({dt/200*1e6:.1f} us/step)", flush=True)
