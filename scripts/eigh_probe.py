"""Probe of hxv_eigh_lowest over (neigen, ncv) and options on one sector; prints E, nconv, products, residuals."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cdmft-lanc-ed_amd"))
import numpy as np, torch, hxv
from hxv import models
wl = os.environ.get("SECTOR", "C2")
m, (nup, ndw) = {"C2": (models.hm_1dchain(), (6, 6)), "C2e": (models.hm_1dchain(eps_bath=[0.3, 0.6]), (6, 6)), "sq1": (models.hm_2dsquare(Nbath=1), (4, 4))}[wl]
for spec in sys.argv[1:]:
    neigen, ncv, real, fused, mall = (int(x) for x in spec.split(","))
    sec = hxv.HxvSector.from_model(m, nup, ndw)
    sec.set_option("real_vectors", real); sec.set_option("lanczos_fused", fused); sec.set_option("eigh_measure_all", mall)
    ev, X, nconv, nmv = sec.eigh_lowest(neigen, ncv, native=True)
    res = [float((sec.apply_device(X[i].contiguous()) - ev[i] * X[i]).norm().item()) for i in range(neigen)]
    print(f"{wl} neigen={neigen} ncv={ncv} real={real} fused={fused} measure_all={mall}: E={np.array2string(ev, precision=8)} nconv={nconv} matvecs={nmv} residuals={['%.1e' % r for r in res]} full/local passes {sec.get_option('eigh_last_full_passes')}/{sec.get_option('eigh_last_local_passes')}", flush=True)
    sec.close()
