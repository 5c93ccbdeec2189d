"""The callers' usage pattern as a whole: every sector of a model opened, solved and closed the way ED_DIAG.f90:78-260
does, every Green's-function channel of build_gf_normal (ED_GF_NORMAL.f90:36-110) with its sector opened and closed around it, and the
sector-image cache that serves the re-opens.  The harness (scripts/harness.py) only drives the C-ABI; the numbers are compared with the CPU
oracle's matrices (tests/golden/c2_sector_sweep.json, scripts/make_golden_c2_sweep.py) and with the oracle's own Lanczos."""
import json
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"


def test_reopened_sector_shares_its_image_and_multiplies_bit_for_bit(built):
    import torch
    import hxv
    from hxv import models

    hxv.sector_cache_clear()
    m = models.hm_1dchain()
    st0 = hxv.sector_cache_stats()
    a = hxv.HxvSector.from_model(m, 6, 6)
    assert a.get_option("open_cache_hit") == 0 and a.get_option("open_us_plan") > 0
    v = torch.randn(a.Dim, dtype=torch.complex128, device="cuda")
    ha = a.apply_device(v).clone()
    ea, _, _, _ = a.eigh_lowest(2, 20, want_vectors=False)
    a.close()
    b = hxv.HxvSector.from_model(m, 6, 6)                      # same model bytes, sector, split, device: the image of the first open
    assert b.get_option("open_cache_hit") == 1 and b.get_option("open_us_host") == 0 and b.get_option("open_us_plan") == 0
    c = hxv.HxvSector.from_model(m, 6, 6)                      # two OPEN handles share one image too
    assert c.get_option("open_cache_hit") == 1
    hb, hc = b.apply_device(v), c.apply_device(v)
    torch.cuda.synchronize()
    assert torch.equal(ha, hb) and torch.equal(ha, hc)
    eb, _, _, _ = b.eigh_lowest(2, 20, want_vectors=False)
    assert np.array_equal(ea, eb)
    # a handle-private plan (a tiling option) leaves the shared image alone
    b.set_option("cols_per_tile", 2)
    hb2 = b.apply_device(v)
    hc2 = c.apply_device(v)
    torch.cuda.synchronize()
    assert torch.equal(hc2, ha) and (hb2 - ha).abs().max().item() < 1e-12
    b.close()
    c.close()
    # another bath is another key (the next DMFT iteration must never see a stale image)
    m2 = models.hm_1dchain(eps_bath=[0.31, -0.2])
    d = hxv.HxvSector.from_model(m2, 6, 6)
    assert d.get_option("open_cache_hit") == 0
    d.close()
    # another sector, another split: misses; the same split again: a hit
    e = hxv.HxvSector.from_model(m, 6, 6, rank=1, nranks=3)
    assert e.get_option("open_cache_hit") == 0
    e.close()
    e = hxv.HxvSector.from_model(m, 6, 6, rank=1, nranks=3)
    assert e.get_option("open_cache_hit") == 1
    e.close()
    st1 = hxv.sector_cache_stats()
    assert st1["hits"] - st0["hits"] == 3 and st1["entries"] == 3
    hxv.sector_cache_clear()
    assert hxv.sector_cache_stats()["entries"] == 0
    f = hxv.HxvSector.from_model(m, 6, 6)
    assert f.get_option("open_cache_hit") == 0
    hf = f.apply_device(v)
    torch.cuda.synchronize()
    assert torch.equal(hf, ha)
    f.close()


def test_reopening_the_headline_sector_costs_milliseconds(built):
    """a re-open of a cached Ns=16 sector is served by the image (no host build, no plan build; ~0.3 ms against 20 ms cold,
    printed), identical products."""
    import time
    import torch
    import hxv
    from hxv import models

    hxv.sector_cache_clear()
    m = models.hm_2dsquare(Nbath=3)
    t0 = time.perf_counter()
    a = hxv.HxvSector.from_model(m, 9, 8)
    cold_ms = (time.perf_counter() - t0) * 1e3
    v = torch.randn(a.fullElems, dtype=torch.complex128, device="cuda")
    torch.view_as_real(v).view(-1, a.pitch, 2)[:, a.DimUp:, :] = 0.0
    ha = a.apply_device(v).clone()
    a.close()
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        b = hxv.HxvSector.from_model(m, 9, 8)
        times.append((time.perf_counter() - t0) * 1e3)
        # a re-open builds nothing: no host description, no tile plan (the image of the first open serves it)
        assert b.get_option("open_cache_hit") == 1 and b.get_option("open_us_host") == 0 and b.get_option("open_us_plan") == 0
        hb = b.apply_device(v)
        torch.cuda.synchronize()
        assert torch.equal(ha, hb)
        b.close()
    # (wall-clock is printed, not asserted: a shared box must not turn a correct cache red; what a re-open may not do is asserted below)
    print(f"Ns=16 sector (9,8): cold open {cold_ms:.1f} ms, cached re-opens {['%.2f' % t for t in times]} ms")
    hxv.sector_cache_clear()


def test_every_sector_of_c2_against_the_oracle(built):
    """ED_DIAG's sweep at Ns=12: all 169 sectors opened, solved (sp_eigh call shape: Neigen=2, Nblock=20) and closed; E0 and E1 within
    1e-10 of the oracle's matrices (LAPACK below Dim=3000, ARPACK above).  Edge sectors (nup or ndw in {0, Ns}: DimUp or DimDw = 1) and
    the sectors the reference would hand to LAPACK (Dim <= 1024) included."""
    import hxv
    from hxv import models
    from harness import diag_sweep

    gold = json.loads((GOLD / "c2_sector_sweep.json").read_text())
    ref = {(r["nup"], r["ndw"]): r for r in gold["sectors"]}
    m = models.hm_1dchain()
    assert m.name == gold["model"]
    recs = diag_sweep(m)
    assert len(recs) == 169 == len(ref)
    worst = 0.0
    for r in recs:
        g = ref[(r["nup"], r["ndw"])]
        assert r["dim"] == g["dim"]
        ne = min(2, r["dim"])
        assert r["nconv"] == ne, r
        err = np.abs(np.array(r["evals"][:ne]) - np.array(g["lowest"][:ne])).max()
        worst = max(worst, err)
        assert err <= 1e-10, (r, g)
    nl = [r for r in recs if r["lanczos"]]
    assert len(nl) == sum(1 for g in ref.values() if g["dim"] > 1024)
    # the edge sectors (nup or ndw in {0, Ns}: DimUp = 1 or DimDw = 1, one of the two passes has nothing to couple) through the PRODUCT as well:
    # H x v against the oracle's spMatVec_main, tiled kernels and the one-thread-per-element cross-check
    import torch
    from hxv.models import deterministic_vector
    from oracle.oracle import OracleSector

    Ns, nedge = 12, 0
    for nup in range(Ns + 1):
        for ndw in range(Ns + 1):
            if nup not in (0, Ns) and ndw not in (0, Ns):
                continue
            sec = hxv.HxvSector.from_model(m, nup, ndw)
            v = deterministic_vector(sec.Dim)
            want = OracleSector(m, nup, ndw).spMatVec_main(v)
            for kern in (1, 0):
                sec.set_option("kernel", kern)
                got = sec.apply_device(torch.from_numpy(v).cuda()).cpu().numpy()
                assert np.abs(got - want).max() <= 1e-13 * max(np.abs(want).max(), 1e-300), (nup, ndw, kern)
            sec.close()
            nedge += 1
    assert nedge == 48
    print(f"C2 sweep: {len(recs)} sectors ({len(nl)} above the Lanczos threshold), worst |dE| {worst:.2e}, open {sum(r['open_ms'] for r in recs):.0f} ms, "
          f"solve {sum(r['solve_ms'] for r in recs):.0f} ms, close {sum(r['close_ms'] for r in recs):.0f} ms")


def _apply_op_host(psi, maps_from, maps_to, pos, spin, create):
    """c / c^dagger on orbital `pos` (0-based) of one spin, numpy restatement of ED_GF_NORMAL.f90:180-199 (sign: occupied orbitals below
    pos on the same spin only)."""
    src = maps_from[spin]
    dst = maps_to[spin]
    du_f, dd_f, du_t, dd_t = len(maps_from[0]), len(maps_from[1]), len(maps_to[0]), len(maps_to[1])
    P = psi.reshape((du_f, dd_f), order="F")
    out = np.zeros((du_t, dd_t), dtype=complex)
    pos_of = {int(s): k for k, s in enumerate(dst)}
    bit = 1 << pos
    for k, s in enumerate(src):
        s = int(s)
        if bool(s & bit) == create:
            continue
        sgn = -1.0 if bin(s & (bit - 1)).count("1") % 2 else 1.0
        t = pos_of[(s | bit) if create else (s & ~bit)]
        if spin == 0:
            out[t, :] += sgn * P[k, :]
        else:
            out[:, t] += sgn * P[:, k]
    return out.reshape(-1, order="F")


@pytest.mark.parametrize("symmetric", [False, True])
def test_green_function_channels_in_the_callers_order(built, symmetric):
    """build_gf_normal's channel list for a 4-site cluster (56 tridiagonalisations, 32 with ed_gf_symmetric), each with its sector opened
    and closed around it; alanc / blanc of one channel of every kind against the ORACLE's Lanczos on the same start vector, paired
    against unpaired runs, and the re-opens served by the cache."""
    import hxv
    from hxv import models
    from harness import gf_solve, gf_channels
    from oracle.oracle import OracleSector

    hxv.sector_cache_clear()
    m = models.hm_1dchain(Nlat=4, Nbath=1)          # Ns = 8, sector (4,4): Dim 4900; targets (5,4), (3,4): 3920
    chans = gf_channels(m, symmetric)
    assert len(chans) == (32 if symmetric else 56)
    assert sum(1 for c in chans if c["kind"] == "mix_xi") == (0 if symmetric else 24)
    nl = 60
    recs, summ = gf_solve(m, 4, 4, nlanc=nl, symmetric=symmetric, pair=True, keep_tridiag=True, keep_psi=True)
    recs1, summ1 = gf_solve(m, 4, 4, nlanc=nl, symmetric=symmetric, pair=False, keep_tridiag=True)
    assert summ["channels"] == len(chans) == summ1["channels"]
    assert summ["channels_complex"] == (0 if symmetric else 24) and summ["channels_paired"] == 32 and summ1["channels_paired"] == 0
    # every open after the first of each target sector is a cache hit: 2 cold opens in the first solve (paired: one open per two real
    # channels), none in the second (one open per channel)
    assert summ["sector_opens"] == len(chans) - 16 and summ["sector_opens"] - summ["sector_open_cache_hits"] == 2
    assert summ1["sector_opens"] == summ1["sector_open_cache_hits"] == len(chans)
    key = lambda r: (r["kind"], r["create"], tuple(r["terms"]))
    by1 = {key(r): r for r in recs1}
    wm = np.pi / 50.0 * (2 * np.arange(1, 17) - 1)
    for r in recs:
        r1 = by1[key(r)]
        assert r["nsteps"] == r1["nsteps"] == nl and abs(r["norm2"] - r1["norm2"]) < 1e-12
        # (entry by entry on the early steps only: once extremal Ritz values converge the recurrence amplifies rounding differences; the
        #  quantity the consumer builds from the whole run -- the continued fraction on the Matsubara grid, ED_GF_NORMAL.f90:949-973 -- is
        #  compared instead: single Ritz values of a 60-step run are not converged and differ between two roundings of the same run)
        assert np.abs(r["alanc"][:10] - r1["alanc"][:10]).max() < 1e-10 and np.abs(r["blanc"][:10] - r1["blanc"][:10]).max() < 1e-10
        gs_ = []
        for x in (r, r1):
            ev, Z = np.linalg.eigh(np.diag(x["alanc"]) + np.diag(x["blanc"][1:], 1) + np.diag(x["blanc"][1:], -1))
            gs_.append((Z[0, :] ** 2 / (1j * wm[:, None] - (ev[None, :] - summ["e0"]))).sum(axis=1))
        assert np.abs(gs_[0] - gs_[1]).max() < 1e-9, np.abs(gs_[0] - gs_[1]).max()
    # the oracle's Lanczos (SciFortran's recurrence restated, oracle/hxv_oracle.c) on the same start vectors
    psi = summ["psi"]
    gs_o = OracleSector(m, 4, 4)
    maps0 = (gs_o.map_up(), gs_o.map_dw())
    import scipy.sparse.linalg as sla
    from helpers_matrix import oracle_full_matrix

    w0 = sla.eigsh(oracle_full_matrix(gs_o), k=1, which="SA", tol=1e-13)[0][0]
    assert abs(summ["e0"] - w0) < 1e-10
    seen = set()
    for r in recs:
        kd = (r["kind"], r["create"])
        if kd in seen:
            continue
        seen.add(kd)
        tu, td = r["sector"]
        orc = OracleSector(m, tu, td)
        maps1 = (orc.map_up(), orc.map_dw())
        vin = sum(complex(cf) * _apply_op_host(psi, maps0, maps1, orb, 0, r["create"]) for orb, cf in r["terms"])
        n2 = np.vdot(vin, vin).real
        assert abs(n2 - r["norm2"]) < 1e-10
        a, b = orc.lanc_tridiag(vin / np.sqrt(n2), 12)
        assert np.abs(a - r["alanc"][:12]).max() < 1e-9 and np.abs(b[1:] - r["blanc"][1:12]).max() < 1e-9, kd
    assert len(seen) == (4 if symmetric else 6)
    hxv.sector_cache_clear()


def test_sector_cache_cap_and_switch(built):
    """HXV_SECTOR_CACHE_MB caps the cached images (least recently used out first, open handles keep theirs alive); HXV_SECTOR_CACHE=0
    turns the cache off.  Read once per process: checked in child processes."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = r"""
import sys
sys.path.insert(0, sys.argv[1] + "/cdmft-lanc-ed_amd")
import numpy as np, torch, hxv
from hxv import models
m = models.hm_1dchain()
v = None
ref = {}
held = hxv.HxvSector.from_model(m, 6, 6)                       # stays open while its image is evicted from the cache
for rep in range(2):
    for nup in (4, 5, 6, 7):
        s = hxv.HxvSector.from_model(m, nup, 6)
        x = torch.ones(s.Dim, dtype=torch.complex128, device="cuda") * (0.5 + 0.25j)
        y = s.apply_device(x).cpu().numpy()
        if rep == 0:
            ref[nup] = y
        else:
            assert np.array_equal(ref[nup], y)
        s.close()
x = torch.ones(held.Dim, dtype=torch.complex128, device="cuda") * (0.5 + 0.25j)
assert np.array_equal(held.apply_device(x).cpu().numpy(), ref[6])
held.close()
st = hxv.sector_cache_stats()
print("STATS", st["entries"], st["bytes"], st["hits"], st["misses"])
"""
    for env_extra, check in (({"HXV_SECTOR_CACHE_MB": "2"}, lambda e, b, h, m: e <= 3 and b <= 2 << 20 and m >= 5 and h + m == 9),   # (4 images of ~0.8 MB: at most two stay)
                             ({"HXV_SECTOR_CACHE": "0"}, lambda e, b, h, m: e == 0 and h == 0),
                             ({}, lambda e, b, h, m: e == 4 and h == 5 and m == 4)):                                     # (the held (6,6) + three more sectors miss once)
        env = dict(os.environ, **env_extra)
        r = subprocess.run([sys.executable, "-c", prog, root], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, (env_extra, r.stdout[-500:], r.stderr[-2000:])
        e, b, h, m = (int(x) for x in [l for l in r.stdout.splitlines() if l.startswith("STATS")][-1].split()[1:])
        assert check(e, b, h, m), (env_extra, e, b, h, m)


@pytest.mark.parametrize("spin", [0, 1])
def test_green_function_channels_of_a_complex_two_orbital_model(built, spin):
    """The same channel loop for BHZ 2x2 (Norb = 2, Nspin = 2, complex H, Ns = 8): impurity orbitals iorb + (ilat-1)*Norb, both spin indices
    (spin-dw operators change ndw: sectors (4,5) / (4,3)); nothing pairs (complex H: complex vectors throughout).  One channel of every
    kind against the oracle's Lanczos on the start vector built on the host (G against the Lehmann sum for this model:
    tests/test_gpu_lanczos.py::test_impurity_green_function_vs_lehmann)."""
    import hxv
    from hxv import models
    from harness import gf_solve, gf_channels
    from oracle.oracle import OracleSector

    hxv.sector_cache_clear()
    m = models.bhz_2d(Nbath=0)
    chans = gf_channels(m)
    nimp = m.Nlat * m.Norb
    assert len(chans) == 2 * nimp + 4 * nimp * (nimp - 1)            # 8 orbitals: 16 + 224
    sub = [c for c in chans if all(o < 3 for o, _ in c["terms"])]      # (the channels among the first three orbitals: 6 + 24)
    recs, summ = gf_solve(m, 4, 4, nlanc=50, spin=spin, keep_tridiag=True, keep_psi=True, channels=sub)
    assert summ["channels"] == len(sub) == 30 and summ["channels_paired"] == 0
    assert all(r["sector"] == ((4, 5) if r["create"] else (4, 3)) if spin == 1 else r["sector"] == ((5, 4) if r["create"] else (3, 4)) for r in recs)
    psi = summ["psi"]
    gs_o = OracleSector(m, 4, 4)
    maps0 = (gs_o.map_up(), gs_o.map_dw())
    import scipy.sparse.linalg as sla
    from helpers_matrix import oracle_full_matrix

    w0 = np.sort(sla.eigsh(oracle_full_matrix(gs_o), k=2, which="SA", tol=1e-13)[0])
    assert abs(summ["e0"] - w0[0]) < 1e-10 and w0[1] - w0[0] > 1e-6
    seen = set()
    for r in recs:
        kd = (r["kind"], r["create"])
        if kd in seen:
            continue
        seen.add(kd)
        tu, td = r["sector"]
        orc = OracleSector(m, tu, td)
        maps1 = (orc.map_up(), orc.map_dw())
        vin = sum(complex(cf) * _apply_op_host(psi, maps0, maps1, orb, spin, r["create"]) for orb, cf in r["terms"])
        n2 = np.vdot(vin, vin).real
        assert abs(n2 - r["norm2"]) < 1e-10
        a, b = orc.lanc_tridiag(vin / np.sqrt(n2), 12)
        assert np.abs(a - r["alanc"][:12]).max() < 1e-9 and np.abs(b[1:] - r["blanc"][1:12]).max() < 1e-9, kd
    assert len(seen) == 6
    hxv.sector_cache_clear()
