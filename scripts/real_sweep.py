"""C3 real-vector product under a few tile options (real mode uses 2x the complex cols/rows per tile)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cdmft-lanc-ed_amd"))
import torch, hxv
from hxv import models
sec = hxv.HxvSector.from_model(models.hm_2dsquare(Nbath=3), 8, 8)
L = hxv.engine.load_library()
n = L.hxv_realvec_elems(sec._h)
x = sec.pad_real(torch.randn(sec.Dim, dtype=torch.float64, device="cuda"))
hr = torch.zeros(n, dtype=torch.float64, device="cuda")
def t(f, nrep=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(nrep): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / nrep * 1e3
for C, R, wc in [(4, 4, 4), (2, 4, 4), (4, 2, 4), (4, 4, 8), (4, 4, 2), (8, 8, 8)]:
    sec.set_option("cols_per_tile", C); sec.set_option("rows_per_tile", R); sec.set_option("wt_cols", wc)
    tr = t(lambda: sec.apply_device_real(x, hr))
    print(f"complex options C={C} R={R} wt_cols={wc} (real: C={min(8,2*C)} R={min(8,2*R)}): real product {tr:.3f} ms", flush=True)
