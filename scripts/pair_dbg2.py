import sys
sys.path.insert(0, "cdmft-lanc-ed_amd")
import numpy as np, torch, hxv
from hxv import models
m = models.hm_1dchain()
sec = hxv.HxvSector.from_model(m, 6, 6)
rng = np.random.default_rng(7)
xa = rng.standard_normal(sec.Dim); xb = rng.standard_normal(sec.Dim)
v1 = torch.from_numpy(xa.astype(np.complex128)).cuda()
v2 = torch.from_numpy(xa + 1j * xb).cuda()
h1 = sec.apply_device(v1); h2 = sec.apply_device(v2); torch.cuda.synchronize()
print("product: real parts bit-identical:", bool((h1.real == h2.real).all()), "max diff", float((h1.real - h2.real).abs().max()), "imag of H(x,0) max", float(h1.imag.abs().max()))
