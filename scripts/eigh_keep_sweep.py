"""hxv_eigh_lowest: products and time against the share of the basis a thick restart keeps (option eigh_keep_pct)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cdmft-lanc-ed_amd"))
import numpy as np, torch, hxv
from hxv import models
cases = {"C3": (models.hm_2dsquare(Nbath=3), (8, 8), 2, 20), "C4": (models.bhz_2d(Nbath=1), (8, 8), 2, 20), "C2e4": (models.hm_1dchain(eps_bath=[0.3, 0.6]), (6, 6), 4, 40),
         "C2e2": (models.hm_1dchain(eps_bath=[0.3, 0.6]), (6, 6), 2, 20), "sq1": (models.hm_2dsquare(Nbath=1), (4, 4), 3, 24), "bhz0": (models.bhz_2d(Nbath=0, Ust=0.4, Jh=0.1), (4, 4), 4, 30),
         "C3_4": (models.hm_2dsquare(Nbath=3), (8, 8), 4, 40)}
for name in sys.argv[1].split(","):
    m, (nup, ndw), neig, ncv = cases[name]
    sec = hxv.HxvSector.from_model(m, nup, ndw)
    for keep in (int(x) for x in sys.argv[2].split(",")):
        sec.set_option("eigh_keep_pct", keep)
        best = None
        for rep in range(2):
            torch.cuda.synchronize(); t = time.time()
            ev, _, nconv, nmv = sec.eigh_lowest(neig, ncv, want_vectors=False)
            torch.cuda.synchronize(); dt = time.time() - t
            best = dt if best is None else min(best, dt)
        print(f"{name:5s} neigen={neig} ncv={ncv} keep={keep:2d}%: E0={ev[0]:.10f} E_last={ev[-1]:.10f} nconv={nconv} products={nmv} {best:.3f}s full/local {sec.get_option('eigh_last_full_passes')}/{sec.get_option('eigh_last_local_passes')}", flush=True)
    sec.close()
