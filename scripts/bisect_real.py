"""Bisect helper: real- and complex-vector Lanczos iteration time at C3 with the package + library of another tree.
usage: bisect_real.py <tree root> [k=v,...]   (one process per tree: one libhxv.so per process)"""
import os, sys
root = os.path.abspath(sys.argv[1])
sys.path.insert(0, os.path.join(root, "cdmft-lanc-ed_amd"))
import torch, hxv
from hxv import models
sec = hxv.HxvSector.from_model(models.hm_2dsquare(Nbath=3), 8, 8)
for kv in (sys.argv[2].split(",") if len(sys.argv) > 2 and sys.argv[2] else []):
    k, v = kv.split("="); sec.set_option(k, int(v))
out = []
for mode in (1, 0):
    sec.set_option("real_vectors", mode)
    sec.time_lanczos(5)
    ms = min(sec.time_lanczos(20) for _ in range(3))
    out.append(f"real_vectors={mode}: {ms:.3f} ms/it")
print(f"{os.path.basename(root):12s} {' '.join(sys.argv[2:3]):24s} " + "   ".join(out), flush=True)
