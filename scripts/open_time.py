"""How long does opening a sector take (host build of maps / tables / plan + uploads)?  The reference re-opens sectors for every Lanczos run
(build_Hv_sector, ED_HAMILTONIAN.f90:39-143): 56 Green's-function channels per solve."""
import sys, time
sys.path.insert(0, "cdmft-lanc-ed_amd")
import torch, hxv
from hxv import models
torch.cuda.init()
for name, m, secs in (("C3", models.hm_2dsquare(Nbath=3), [(8, 8), (9, 8), (8, 8), (7, 8)]), ("C4", models.bhz_2d(Nbath=1), [(8, 8), (9, 8)]), ("C5", models.hm_ring(6, 2), [(9, 9), (9, 9)])):
    for nup, ndw in secs:
        t0 = time.perf_counter()
        s = hxv.HxvSector.from_model(m, nup, ndw)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        s.close()
        t2 = time.perf_counter()
        print(f"{name} sector ({nup},{ndw}) Dim={s.Dim}: open {1e3 * (t1 - t0):.1f} ms, close {1e3 * (t2 - t1):.1f} ms", flush=True)
