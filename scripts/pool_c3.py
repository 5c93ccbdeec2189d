"""Effect of the device-buffer cache at C3: eigh_lowest on a fresh handle with a cold / warm cache."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cdmft-lanc-ed_amd"))
import torch, hxv
from hxv import models
m = models.hm_2dsquare(Nbath=3)
for rep in range(3):
    t0 = time.time(); sec = hxv.HxvSector.from_model(m, 8, 8); tb = time.time() - t0
    t0 = time.time(); ev, _, nconv, nmv = sec.eigh_lowest(2, 20, want_vectors=False); te = time.time() - t0
    t0 = time.time(); e0, _, nit = sec.lanczos_eigh(512, 1e-12, want_vector=False); tl = time.time() - t0
    sec.close()
    print(f"run {rep}: build {tb:.2f}s  eigh_lowest {te:.2f}s ({nmv} products)  lanczos_eigh {tl:.2f}s ({nit} it)  pool {hxv.pool_stats()}", flush=True)
