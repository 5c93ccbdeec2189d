"""hxv_eigh_lowest against LAPACK on small sectors over (model, neigen, ncv, tol): complex H, Kanamori terms, large bases, neigen close to ncv, loose tol."""
import os, sys
ROOT = os.path.join(os.path.dirname(__file__), "..")
sys.path.insert(0, os.path.join(ROOT, "cdmft-lanc-ed_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, hxv
from hxv import models
from oracle.oracle import OracleSector
bad = 0
cases = [("bhz", models.bhz_2d(Nbath=0), (4, 4)), ("bhzK", models.bhz_2d(Nbath=0, Ust=0.7, Jh=0.2, Jx=0.2, Jp=0.15), (4, 4)), ("bhz35", models.bhz_2d(Nbath=0, Ust=0.4, Jh=0.1), (3, 5)),
         ("sq1", models.hm_2dsquare(Nbath=1), (4, 4)), ("chain8", models.hm_1dchain(Nlat=2, Nbath=3), (4, 4))]
for name, m, (nup, ndw) in cases:
    ref = np.linalg.eigvalsh(OracleSector(m, nup, ndw).dense())
    sec = hxv.HxvSector.from_model(m, nup, ndw)
    for neig, ncv, tol in ((2, 40, 0.0), (4, 40, 0.0), (8, 64, 0.0), (10, 20, 0.0), (15, 20, 0.0), (3, 33, 1e-6), (6, 48, 1e-8), (1, 64, 0.0), (20, 64, 0.0)):
        try:
            ev, X, nconv, nmv = sec.eigh_lowest(neig, ncv, 512, tol)
        except hxv.HxvError as e:
            print(f"{name} neig={neig} ncv={ncv} tol={tol}: refused: {str(e)[:90]}"); continue
        # degenerate copies may come in any order; compare as multisets against LAPACK's lowest neig
        err = np.abs(np.sort(ev) - ref[:neig]).max()
        lim = 1e-9 if tol == 0.0 else 50 * tol * max(1.0, np.abs(ref[:neig]).max())
        Xh = X.cpu().numpy().T
        orth = np.abs(Xh.conj().T @ Xh - np.eye(neig)).max()
        ok = nconv == neig and err < lim and orth < 1e-7
        bad += 0 if ok else 1
        print(f"{name:6s} Dim={sec.Dim:6d} neig={neig:2d} ncv={ncv:2d} tol={tol:g}: nconv={nconv} products={nmv} max|E-LAPACK|={err:.1e} orth={orth:.1e} {'ok' if ok else 'FAIL'}", flush=True)
    sec.close()
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
