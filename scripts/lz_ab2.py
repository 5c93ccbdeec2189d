"""A/B of option sets on the Lanczos iteration (complex and real vectors) at C3.  usage: lz_ab2.py "k=v,..." ..."""
import sys
sys.path.insert(0, "cdmft-lanc-ed_amd")
import torch, hxv
from hxv import models
m = models.hm_2dsquare(Nbath=3)
for s in sys.argv[1:] or [""]:
    sec = hxv.HxvSector.from_model(m, 8, 8)
    for kv in s.split(","):
        if kv:
            k, val = kv.split("="); sec.set_option(k, int(val))
    out = []
    for rv in (0, 1):
        sec.set_option("real_vectors", rv)
        sec.time_lanczos(10)
        out.append(min(sec.time_lanczos(20) for _ in range(2)))
    print(f"C3 [{s:30s}] lanczos ms/iter complex {out[0]:.3f} real {out[1]:.3f}", flush=True)
    sec.close()
