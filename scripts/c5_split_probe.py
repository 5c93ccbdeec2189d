"""Ns=18 (Dim = 2 363 904 400 > 2^31) on TWO thread ranks of one GPU: every rank's slab of the split product against the same slab of the
unsplit product, for the all-gather and for the two-transposes exchange (64-bit offsets, WtRange pieces, block order at this size)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cdmft-lanc-ed_amd"))
import torch, hxv
from hxv import models
m, (nup, ndw) = models.hm_ring(6, 2), (9, 9)
ser = hxv.HxvSector.from_model(m, nup, ndw)
n = ser.fullElems
v = torch.empty(n, dtype=torch.complex128, device="cuda")
vr = torch.view_as_real(v).view(-1)
g = torch.Generator(device="cuda").manual_seed(5)
for a in range(0, 2 * n, 1 << 28):
    b = min(a + (1 << 28), 2 * n)
    vr[a:b] = torch.randn(b - a, dtype=torch.float64, device="cuda", generator=g)
v.view(-1, ser.pitch)[:, ser.DimUp:] = 0
ref = ser.apply_device(v).clone()
pitch, dimdw = ser.pitch, ser.DimDw
ser.close(); hxv.pool_trim()
bad = 0
for exchange in ("allgather", "alltoall"):
    hxv.set_exchange_default(exchange)
    def rank(r, group):
        sec = hxv.HxvSector.from_model(m, nup, ndw, rank=r, nranks=2)
        group.join(sec)
        c0 = sec.mpiIshift // sec.DimUp
        x = v[c0 * pitch:(c0 + sec.mpiQdw) * pitch].clone()
        t0 = time.time()
        y = sec.apply_device_slab(x)
        torch.cuda.synchronize()
        r_ = ref[c0 * pitch:(c0 + sec.mpiQdw) * pitch]
        err = 0.0; mx = 0.0
        for a in range(0, y.numel(), 1 << 27):
            b = min(a + (1 << 27), y.numel())
            d = (y[a:b] - r_[a:b]).view(-1, 1)
            err = max(err, (y[a:b] - r_[a:b]).abs().max().item()); mx = max(mx, r_[a:b].abs().max().item())
        mode = sec.exchange_mode
        sec.close()
        return err / mx, mode, time.time() - t0
    try:
        out = hxv.run_ranks(2, rank)
    finally:
        hxv.set_exchange_default("allgather")
    hxv.pool_trim(); torch.cuda.empty_cache()
    for r, (e, mode, dt) in enumerate(out):
        ok = e <= 1e-13
        bad += 0 if ok else 1
        print(f"C5 rank {r}/2 {mode}: rel err of the slab against the unsplit product {e:.1e} {'ok' if ok else 'FAIL'}", flush=True)
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
