import sys
sys.path.insert(0, "cdmft-lanc-ed_amd")
import numpy as np, torch, hxv
from hxv import models
m = models.hm_1dchain()
sec = hxv.HxvSector.from_model(m, 6, 6)
rng = np.random.default_rng(7)
xa = rng.standard_normal(sec.Dim); xb = rng.standard_normal(sec.Dim)
xa /= np.linalg.norm(xa); xb /= np.linalg.norm(xb)
da = torch.from_numpy(xa.astype(np.complex128)).cuda(); db = torch.from_numpy(xb.astype(np.complex128)).cuda()
sec.set_option("lanczos_graph", 0); sec.set_option("real_vectors", 0); sec.set_option("job_up", 0)
for nl in (8, 9, 10, 12):
    s1 = sec.lanczos_tridiag(da, nl); p1 = sec.lanczos_tridiag_pair(da, db, nl)
    print(nl, "alpha ulps", [(float(x - y) / np.spacing(abs(y))) for x, y in zip(p1[0][0], s1[0])][-4:], "beta ulps", [(float(x - y) / np.spacing(abs(y)) if y else 0.0) for x, y in zip(p1[0][1], s1[1])][-4:], "nsteps", s1[2], p1[0][2])
