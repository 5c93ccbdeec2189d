"""Lanczos iteration time (complex and real vectors, fused recurrence) of one library build: HXV_LIB=<lib> python scripts/lz_ab.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cdmft-lanc-ed_amd"))
import torch, hxv
from hxv import models
sec = hxv.HxvSector.from_model(models.hm_2dsquare(Nbath=3), 8, 8)
sec.set_option("real_vectors", 0)
sec.time_lanczos(5)
c = min(sec.time_lanczos(20) for _ in range(3))
sec.set_option("real_vectors", 1)
sec.time_lanczos(5)
r = min(sec.time_lanczos(20) for _ in range(3))
print(f"{os.environ.get('HXV_LIB', 'shipped')}: complex iteration {c:.4f} ms, real iteration {r:.4f} ms", flush=True)
