"""Soak of the split-sector code on ONE GPU (thread ranks): random models / sectors / rank counts / exchanges / TRANSPORT (thread-rank
transport or the RCCL branches through tests/rccl_double) -- every rank's product through hxv_apply_device_slab against the oracle (with the
two-transposes exchange also in its overlapped form), and a short tridiagonalisation against the serial handle.
usage: ranks_soak.py [first] [last]"""
import os, sys, time, traceback
ROOT = os.path.join(os.path.dirname(__file__), "..")
sys.path.insert(0, os.path.join(ROOT, "cdmft-lanc-ed_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, hxv
import test_gpu_fuzz as tf
from oracle.oracle import OracleSector

def one(seed):
    rng = np.random.default_rng(5000 + seed)
    m = tf._random_model(rng)
    Ns = m.Ns
    nup, ndw = int(np.clip(Ns // 2 + rng.integers(-1, 2), 1, Ns - 1)), int(np.clip(Ns // 2 + rng.integers(-1, 2), 1, Ns - 1))
    full = OracleSector(m, nup, ndw)
    if full.DimDw < 2 or full.Dim > 400000:
        return "skipped"
    nranks = int(rng.integers(2, min(4, full.DimDw) + 1))
    exchange = ["allgather", "halo", "alltoall"][int(rng.integers(3))]
    transport = ["local", "rccl"][seed % 2]
    overlap = int(rng.integers(2))
    v = rng.standard_normal(full.Dim) + 1j * rng.standard_normal(full.Dim)
    v /= np.linalg.norm(v)
    ref = full.spMatVec_main(v)
    ser = hxv.HxvSector.from_model(m, nup, ndw)
    fused = int(rng.integers(2))
    ser.set_option("lanczos_fused", fused)
    nl = min(8, full.Dim)
    a0, b0, n0 = ser.lanczos_tridiag(torch.from_numpy(v).cuda(), nl)
    ser.close()
    hxv.set_exchange_default(exchange)

    def rank(r, group):
        sec = hxv.HxvSector.from_model(m, nup, ndw, rank=r, nranks=nranks)
        sec.set_option("lanczos_fused", fused)
        group.join(sec)
        sec.set_option("exchange_overlap", overlap)
        lo, hi = sec.mpiIshift, sec.mpiIshift + sec.vecDim
        got = sec.unpad(sec.apply_device_slab(sec.pad(torch.from_numpy(v[lo:hi].copy()).cuda(), sec.mpiQdw))).cpu().numpy() if sec.vecDim else np.zeros(0, complex)
        a, b, n = sec.lanczos_tridiag(torch.from_numpy(v[lo:hi].copy()).cuda(), nl)
        mode = sec.exchange_mode
        sec.close()
        return lo, hi, got, a, b, n, mode

    try:
        res = hxv.run_ranks(nranks, rank, transport=transport)
    finally:
        hxv.set_exchange_default("allgather")
    scale = max(np.abs(ref).max(), 1e-300)
    for lo, hi, got, a, b, n, mode in res:
        assert np.abs(got - ref[lo:hi]).max() <= 2e-13 * scale, ("product", seed, nranks, exchange, mode)
        k = min(n, n0, 5)
        closes = n != n0 and min(np.abs(b[min(n, n0):max(n, n0)]).max() if max(n, n0) <= len(b) else 1.0, np.abs(b0[min(n, n0):max(n, n0)]).max() if max(n, n0) <= len(b0) else 1.0) < 1e-9
        # (a Krylov space that closes: the residual norm sits at the breakdown threshold and the two runs may stop one step apart)
        assert (n == n0 or closes) and np.abs(a[:k] - a0[:k]).max() <= 1e-9 * max(1.0, np.abs(a0).max()), ("tridiag", seed, nranks, exchange, mode, n, n0, full.Dim, b, b0)
    return f"{nranks} ranks {res[0][6]} {transport}"

import __graft_entry__ as ge
os.environ["HXV_RCCL_LIB"] = str(ge.build_rccl_double())   # (only handles joined through hxv_comm_init -- transport "rccl" -- load it)
a, b = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 200)
bad, t0, seen = [], time.time(), {}
for seed in range(a, b):
    try:
        r = one(seed)
        seen[r] = seen.get(r, 0) + 1
    except Exception:
        bad.append(seed)
        print("seed", seed, "FAILED\n" + traceback.format_exc()[-1800:], flush=True)
    if seed % 25 == 0:
        print(f"... seed {seed}, {time.time() - t0:.0f} s, failures so far {bad}", flush=True)
print(f"seeds {a}..{b - 1}: {len(bad)} failures {bad} in {time.time() - t0:.0f} s; cases {seen}")
sys.exit(1 if bad else 0)
