"""Do pass A (without the wt add) and pass B speed up when they run concurrently on two streams?
(tells how much slack a better-overlapped design could recover)"""
import sys, time
sys.path.insert(0, "cdmft-lanc-ed_amd")
import torch, hxv
from hxv import models
m = models.hm_2dsquare(Nbath=3)
sa = hxv.HxvSector.from_model(m, 8, 8); sa.set_option("passes", 1)
sb = hxv.HxvSector.from_model(m, 8, 8); sb.set_option("passes", 2)
v = torch.randn(sa.fullElems, dtype=torch.float64, device="cuda") + 1j * torch.randn(sa.fullElems, dtype=torch.float64, device="cuda")
h1 = torch.empty_like(v); h2 = torch.empty_like(v)
s1 = torch.cuda.Stream(); s2 = torch.cuda.Stream()
def run(conc, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        if conc:
            with torch.cuda.stream(s1): sa.apply_device(v, h1)
            with torch.cuda.stream(s2): sb.apply_device(v, h2)
        else:
            sa.apply_device(v, h1); sb.apply_device(v, h2)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
run(False, 2); run(True, 2)
print("sequential A;B  ms", run(False))
print("concurrent A||B ms", run(True))
