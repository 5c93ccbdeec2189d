set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest1.log; cat gpurun_out/pytest1.log
python __graft_entry__.py smoke 2>&1 | tail -3
python - <<'PY' 2>&1 | tee gpurun_out/first_bench.log
import sys, time
sys.path.insert(0, "cdmft-lanc-ed_amd")
import torch, hxv
from hxv import models
m = models.hm_2dsquare(Nbath=3)
t=time.time(); sec = hxv.HxvSector.from_model(m, 8, 8); print("build s", time.time()-t, sec.stats(), "bits", sec.get_option("tile_bits_up"), sec.get_option("tile_bits_dw"))
v = torch.randn(sec.Dim, dtype=torch.float64, device="cuda") + 1j*torch.randn(sec.Dim, dtype=torch.float64, device="cuda")
hv = torch.empty_like(v)
for k in (1, 0):
    sec.set_option("kernel", k)
    sec.time_apply(v, hv, 2)
    ms = sec.time_apply(v, hv, 5)
    print("kernel", k, "ms", ms, "GB/s alg", 32*sec.Dim/ms/1e6)
PY
