"""Lanczos iteration time (complex and real vectors) with pass A as jobs / one tile per workgroup."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cdmft-lanc-ed_amd"))
import torch, hxv
from hxv import models
sec = hxv.HxvSector.from_model(models.hm_2dsquare(Nbath=3), 8, 8)
for rep in range(2):
    for job in (1, 0):
        sec.set_option("job_up", job)
        for mode in (0, 1):
            sec.set_option("real_vectors", mode)
            ms = sec.time_lanczos(20)
            print(f"job_up={job} real_vectors={mode}: {ms:.3f} ms per iteration", flush=True)
