"""Exchange mode 2 at C3 with P thread ranks on one GPU: fused form against the overlapped form (option exchange_overlap).  On one GPU the
'links' are device copies and the two streams share the same CUs, so this shows the COST of the overlapped form (its extra add pass), not
its gain.  usage: exchange_overlap_c3.py [P]"""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cdmft-lanc-ed_amd"))
import torch, hxv
from hxv import models

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
m, (nup, ndw) = models.hm_2dsquare(Nbath=3), (8, 8)
hxv.set_exchange_default("alltoall")
for ov in (0, 1):
    bar = threading.Barrier(P)
    out = {}

    def rank(r, group):
        sec = hxv.HxvSector.from_model(m, nup, ndw, rank=r, nranks=P)
        group.join(sec)
        sec.set_option("exchange_overlap", ov)
        x = torch.randn(sec.localElems, dtype=torch.complex128, device="cuda")
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
        sec.close()

    hxv.run_ranks(P, rank)
    print(f"C3 x {P} thread ranks, two transposes, exchange_overlap={ov}: {max(out.values()):7.3f} ms per product (all ranks together, one GPU)", flush=True)
hxv.set_exchange_default("allgather")
