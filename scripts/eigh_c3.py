"""C3 (Ns=16, Dim=165 636 900): the default spectrum call of ED_DIAG (sp_eigh, Neigen=2, Nblock=20) on one MI355X,
next to the single-vector Lanczos (sp_lanc_eigh)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cdmft-lanc-ed_amd"))
import torch
import hxv
from hxv import models

wl = os.environ.get("SECTOR", "C3")
m, (nup, ndw) = (models.hm_2dsquare(Nbath=3), (8, 8)) if wl == "C3" else (models.bhz_2d(Nbath=1), (8, 8)) if wl == "C4" else (models.hm_1dchain(), (6, 6))
sec = hxv.HxvSector.from_model(m, nup, ndw)
sec.set_option("eigh_measure_all", int(os.environ.get("MEASURE_ALL", 0)))
sec.set_option("lanczos_fused", int(os.environ.get("FUSED", 1)))
sec.set_option("real_vectors", int(os.environ.get("REAL_VECTORS", 1)))
sec.set_option("eigh_degenerate", int(os.environ.get("DEGENERATE", 0)))   # 1: plus the check rounds for hidden copies of degenerate levels
sec.set_option("eigh_fuse_restart", int(os.environ.get("FUSE_RESTART", 1)))
if "KEEP" in os.environ:
    sec.set_option("eigh_keep_pct", int(os.environ["KEEP"]))
neigen, ncv = int(os.environ.get("NEIGEN", 2)), int(os.environ.get("NCV", 20))
for rep in range(int(os.environ.get("REPS", 1))):  # (a second run finds the Krylov basis in the engine's buffer cache)
    t = time.time()
    ev, X, nconv, nmv = sec.eigh_lowest(neigen, ncv, native=True)
    torch.cuda.synchronize()
    dt = time.time() - t
    print(f"{wl} eigh_lowest neigen={neigen} ncv={ncv} degenerate={sec.get_option('eigh_degenerate')}: E={ev} nconv={nconv} matvecs={nmv} "
          f"(search {sec.get_option('eigh_last_search_products')} + check {sec.get_option('eigh_last_check_products')}) {dt:.2f}s ({dt / nmv * 1e3:.1f} ms per Lanczos step; "
          f"Gram-Schmidt passes: {sec.get_option('eigh_last_full_passes')} whole-basis, {sec.get_option('eigh_last_local_passes')} local)", flush=True)
    if rep + 1 < int(os.environ.get("REPS", 1)):
        del X
hv = sec.apply_device(X[0].contiguous())
r = (hv - ev[0] * X[0]).norm().item()
print(f"  residual |H x0 - E0 x0| = {r:.2e}", flush=True)
del X, hv
torch.cuda.empty_cache()
t = time.time()
e0, vec, nit = sec.lanczos_eigh(512, 1e-13, native=True)
torch.cuda.synchronize()
dt = time.time() - t
print(f"{wl} lanczos_eigh: E0={e0:.12f} iterations={nit} {dt:.2f}s   |E0 - eigh_lowest| = {abs(e0 - ev[0]):.2e}", flush=True)
