"""Exploratory parity probe beyond the suite's sectors: off-diagonal (nup != ndw), nearly empty / nearly full and one-column sectors of the
Ns=16 models at FULL size -- a slab of the product against the oracle's matrices -- and the ladder operators between neighbouring full-size
sectors against a vectorised numpy restatement of the master's loop (ED_GF_NORMAL.f90:180-199)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(__file__), "..")
sys.path.insert(0, os.path.join(ROOT, "cdmft-lanc-ed_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, hxv
from hxv import models
from oracle.oracle import OracleSector
from test_gpu_parity import _slab_reference

def rel(a, b): return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)

bad = 0
for name, m in (("C3", models.hm_2dsquare(Nbath=3)), ("C4K", models.bhz_2d(Nbath=1, Ust=0.5, Jh=0.1, Jx=0.1, Jp=0.1))):
    for nup, ndw in ((9, 8), (8, 7), (7, 9), (10, 6), (5, 11), (12, 4), (2, 14), (15, 1), (1, 15), (8, 0), (0, 8), (16, 8), (8, 16), (3, 3), (13, 13)):
        t0 = time.time()
        full = hxv.HxvSector.from_model(m, nup, ndw)
        dim, dimdw = full.Dim, full.DimDw
        full.close()
        nranks = max(1, min(64, dimdw, dim // 2_000_000 + 1))
        rank = nranks // 3
        sec = hxv.HxvSector.from_model(m, nup, ndw, rank=rank, nranks=nranks)
        orc = OracleSector(m, nup, ndw, rank, nranks)
        rng = np.random.default_rng(nup * 17 + ndw)
        v = rng.standard_normal(dim) + 1j * rng.standard_normal(dim)
        ref = _slab_reference(orc, v)
        dv = torch.from_numpy(sec.to_gather_layout(v, nranks)).cuda()
        errs = []
        for kernel in (1, 0):
            sec.set_option("kernel", kernel)
            hv = sec.unpad(sec.apply_device(dv)).cpu().numpy()
            errs.append(rel(hv, ref))
        ok = max(errs) <= 1e-13
        bad += 0 if ok else 1
        print(f"{name} sector ({nup:2d},{ndw:2d}) Dim={dim:>11d} slab {rank}/{nranks}: rel err tiled {errs[0]:.1e} naive {errs[1]:.1e} {'ok' if ok else 'FAIL'}  ({time.time() - t0:.1f}s)", flush=True)
        sec.close(); orc.close()
        del dv, v, ref

# ladder operators at full size: (8,8) -> (9,8) [c^dagger up], (8,8) -> (8,7) [c dw]
m = models.hm_2dsquare(Nbath=3)
sa = hxv.HxvSector.from_model(m, 8, 8)
mu_a, md_a = sa.maps()
g = torch.Generator(device="cuda").manual_seed(3)
psi = torch.randn(sa.Dim, dtype=torch.float64, device="cuda", generator=g) + 1j * torch.randn(sa.Dim, dtype=torch.float64, device="cuda", generator=g)
P = psi.cpu().numpy().reshape(sa.DimDw, sa.DimUp)      # [idw][iup]
for (tnup, tndw, spin, create, pos) in ((9, 8, 0, True, 5), (8, 7, 1, False, 11), (7, 8, 0, False, 0), (8, 9, 1, True, 15)):
    sb = hxv.HxvSector.from_model(m, tnup, tndw)
    mu_b, md_b = sb.maps()
    out, n2 = sa.apply_ladder(sb, pos, spin, create, psi)
    got = out.cpu().numpy().reshape(sb.DimDw, sb.DimUp)
    src, dst = (mu_a, mu_b) if spin == 0 else (md_a, md_b)
    bit = 1 << pos
    occ = (src & bit) != 0
    act = ~occ if create else occ
    tgt_state = np.where(create, src | bit, src & ~bit)
    lut = -np.ones(1 << 16, dtype=np.int64); lut[dst] = np.arange(len(dst))
    tgt = lut[tgt_state[act]]
    par = np.array([bin(int(s) & (bit - 1)).count("1") & 1 for s in src[act]])
    sgn = np.where(par == 1, -1.0, 1.0)
    ref = np.zeros((sb.DimDw, sb.DimUp), dtype=complex)
    if spin == 0:
        ref[:, tgt] = P[:, act] * sgn[None, :]
    else:
        ref[tgt, :] = P[act, :] * sgn[:, None]
    e = np.abs(got - ref).max()
    ok = e < 1e-14 and abs(n2 - np.vdot(ref, ref).real) < 1e-9 * n2
    bad += 0 if ok else 1
    print(f"ladder C3 (8,8)->({tnup},{tndw}) {'c^dagger' if create else 'c'} pos {pos} spin {spin}: max err {e:.1e} norm2 {n2:.6e} {'ok' if ok else 'FAIL'}", flush=True)
    sb.close()
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
