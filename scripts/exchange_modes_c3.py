"""The three exchanges of a split sector executed by THREAD ranks on one GPU (C3, P ranks): wall time per product of all ranks together.
The 'links' are device-to-device copies here, so this shows the engine-side work of each mode (slab copies, pack / unpack, panel kernels),
not link time.  usage: exchange_modes_c3.py [P]"""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cdmft-lanc-ed_amd"))
import numpy as np, torch, hxv
from hxv import models

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
m, (nup, ndw) = models.hm_2dsquare(Nbath=3), (8, 8)
ser = hxv.HxvSector.from_model(m, nup, ndw)
v = torch.randn(ser.Dim, dtype=torch.complex128, device="cuda")
hv = torch.empty_like(v)
ser.time_apply(v, hv, 2)
t_ser = ser.time_apply(v, hv, 10)
ser.close()
del v, hv
torch.cuda.empty_cache()
print(f"C3 unsplit product: {t_ser:.3f} ms", flush=True)
for exchange in ("allgather", "halo", "alltoall"):
    hxv.set_exchange_default(exchange)
    bar = threading.Barrier(P)
    out = {}

    def rank(r, group):
        sec = hxv.HxvSector.from_model(m, nup, ndw, rank=r, nranks=P)
        sec.comm_init_local(group)
        x = torch.randn(sec.localElems, dtype=torch.complex128, device="cuda")
        if exchange != "alltoall":
            home = sec.slab_home()
            home.copy_(x)
            x = home
        y = torch.empty(sec.localElems, dtype=torch.complex128, device="cuda")
        for _ in range(3):
            sec.apply_device_slab(x, y)
        torch.cuda.synchronize()
        bar.wait()
        t0 = time.perf_counter()
        for _ in range(10):
            sec.apply_device_slab(x, y)
        torch.cuda.synchronize()
        bar.wait()
        out[r] = (time.perf_counter() - t0) / 10 * 1e3
        mode = sec.exchange_mode
        sec.close()
        return mode

    try:
        modes = hxv.run_ranks(P, rank)
    finally:
        hxv.set_exchange_default("allgather")
    print(f"C3 x {P} thread ranks, {modes[0]:9s}: {max(out.values()):7.3f} ms per product (all ranks together, one GPU)", flush=True)
