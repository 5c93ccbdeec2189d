"""Device Lanczos (sp_lanc_tridiag / sp_lanc_eigh call shapes) against the CPU oracle:
  - tridiagonal coefficients vs the oracle's restatement (same start vector)         tol 1e-10 relative
  - lowest eigenvalue vs LAPACK on the oracle's dense sector Hamiltonian             tol 1e-10 absolute (BASELINE)
  - impurity Green's function from the continued fraction vs the full-ED Lehmann sum  tol 1e-9 absolute
    (formulas of ED_GF_NORMAL.f90:915-975: weights norm2*Z(1,j)^2, poles +-(E_j-E0), wm = pi/beta*(2n-1))."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_tridiag_matches_oracle(built):
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    m = models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6])
    sec = hxv.HxvSector.from_model(m, 3, 3)
    orc = OracleSector(m, 3, 3)
    v = models.deterministic_vector(sec.Dim)
    v /= np.linalg.norm(v)
    a_ref, b_ref = orc.lanc_tridiag(v, 40)
    a, b, n = sec.lanczos_tridiag(torch.from_numpy(v).cuda(), 40)
    assert n == len(a_ref) == 40
    assert np.abs(a - a_ref).max() <= 1e-10 * np.abs(a_ref).max()
    assert np.abs(b - b_ref).max() <= 1e-10 * np.abs(b_ref).max()
    assert b[0] == 0.0   # blanc(1) unused, ED_GF_NORMAL.f90:949-951


@pytest.mark.parametrize("case", ["C1", "chain", "bhz", "C2"])
def test_lanczos_eigh_ground_state(built, case):
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    if case == "C1":
        m, (nup, ndw) = models.plaquette_2x2_nobath(U=4.0, t=1.0, hfmode=False), (2, 2)
    elif case == "chain":
        m, (nup, ndw) = models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6]), (3, 3)
    elif case == "bhz":
        m, (nup, ndw) = models.bhz_2d(Nbath=0), (4, 4)
    else:
        m, (nup, ndw) = models.hm_1dchain(), (6, 6)
    sec = hxv.HxvSector.from_model(m, nup, ndw)
    e0, vec, nit = sec.lanczos_eigh(nitermax=512, threshold=1e-13)
    if case == "C2":
        # Dim=853776: no dense reference; scipy ARPACK on the oracle's matrices (same family as sp_eigh)
        import scipy.sparse as sp
        import scipy.sparse.linalg as sla

        orc = OracleSector(m, nup, ndw)
        du, dd = orc.DimUp, orc.DimDw
        rp, cols, vals = orc.csr("up")
        Hup = sp.csr_matrix((vals, cols - 1, rp), shape=(du, du))
        rp, cols, vals = orc.csr("dw")
        Hdw = sp.csr_matrix((vals, cols - 1, rp), shape=(dd, dd))
        H = sp.diags(orc.diag()) + sp.kron(Hdw, sp.identity(du)) + sp.kron(sp.identity(dd), Hup)
        ref = sla.eigsh(H.tocsr(), k=1, which="SA", tol=1e-13)[0][0]
    else:
        orc = OracleSector(m, nup, ndw)
        ref = np.linalg.eigvalsh(orc.dense())[0]
    assert abs(e0 - ref) <= 1e-10, (case, e0, ref, nit)
    if case == "C1":
        assert abs(e0 - (-2.10274848)) < 5e-9   # reference-built dense H, SURVEY.md 8c
    # the returned vector is the eigenvector: residual through the engine's own product
    hv = sec.apply_device(vec)
    torch.cuda.synchronize()
    res = (hv - e0 * vec).abs().max().item()
    assert res < 1e-7, res
    assert abs(vec.abs().pow(2).sum().item() - 1.0) < 1e-12


def _apply_op(psi, maps_from, maps_to, pos, spin, create):
    """c / c^dagger on orbital `pos` (0-based) of one spin: vvinit(j) = sgn*psi(i), ED_GF_NORMAL.f90:180-199.
    maps_* = (map_up, map_dw) of the two sectors.  Sign counts occupied orbitals below pos on the same spin only."""
    mu_f, md_f = maps_from
    mu_t, md_t = maps_to
    du_f, dd_f, du_t, dd_t = len(mu_f), len(md_f), len(mu_t), len(md_t)
    P = psi.reshape((du_f, dd_f), order="F")
    out = np.zeros((du_t, dd_t), dtype=complex)
    src = mu_f if spin == 0 else md_f
    dst = mu_t if spin == 0 else md_t
    pos_of = {int(s): k for k, s in enumerate(dst)}
    bit = 1 << pos
    for k, s in enumerate(src):
        s = int(s)
        occ = bool(s & bit)
        if occ == create:
            continue
        sgn = -1.0 if bin(s & (bit - 1)).count("1") % 2 else 1.0
        t = pos_of[(s | bit) if create else (s & ~bit)]
        if spin == 0:
            out[t, :] += sgn * P[k, :]
        else:
            out[:, t] += sgn * P[:, k]
    return out.reshape(-1, order="F")


@pytest.mark.parametrize("model_name", ["chain_B1", "plaquette"])
def test_impurity_green_function_vs_lehmann(built, model_name):
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    if model_name == "chain_B1":
        m, N = models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.25, -0.4], U=2.0), 3
    else:
        m, N = models.plaquette_2x2_nobath(U=4.0, t=1.0, hfmode=True), 2
    beta, Lmats = 50.0, 64
    wm = np.pi / beta * (2 * np.arange(1, Lmats + 1) - 1)   # ED_GF_SHARED.f90:49
    gs = hxv.HxvSector.from_model(m, N, N)
    e0, psi, _ = gs.lanczos_eigh(nitermax=512, threshold=1e-14)
    psi = psi.cpu().numpy()
    maps0 = gs.maps()
    orc0 = OracleSector(m, N, N)
    w0, U0 = np.linalg.eigh(orc0.dense())
    assert abs(w0[0] - e0) < 1e-10 and w0[1] - w0[0] > 1e-6   # non-degenerate ground state
    psi_ref = U0[:, 0]
    G = np.zeros(Lmats, dtype=complex)
    Gref = np.zeros(Lmats, dtype=complex)
    site, spin = 0, 0
    for create, (nu, nd) in ((True, (N + 1, N)), (False, (N - 1, N))):
        sec = hxv.HxvSector.from_model(m, nu, nd)
        maps1 = sec.maps()
        # --- engine: Lanczos continued fraction
        vv = _apply_op(psi, maps0, maps1, site, spin, create)
        norm2 = float(np.vdot(vv, vv).real)
        nl = min(sec.Dim, 200)                                # lanc_nGFiter, ED_GF_NORMAL.f90:204-207
        a, b, n = sec.lanczos_tridiag(torch.from_numpy(vv / np.sqrt(norm2)).cuda(), nl, threshold=1e-12)
        a, b = a[:n], b[:n]
        ev, Z = np.linalg.eigh(np.diag(a) + np.diag(b[1:], 1) + np.diag(b[1:], -1))   # :949-953
        sign = 1.0 if create else -1.0
        for j in range(n):
            G += norm2 * Z[0, j] ** 2 / (1j * wm - sign * (ev[j] - e0))                # :958-973
        # --- reference: Lehmann sum over the full spectrum of the N+-1 sector
        orc = OracleSector(m, nu, nd)
        w1, U1 = np.linalg.eigh(orc.dense())
        vr = _apply_op(psi_ref, maps0, maps1, site, spin, create)
        amp = np.abs(U1.conj().T @ vr) ** 2
        for k in range(len(w1)):
            Gref += amp[k] / (1j * wm - sign * (w1[k] - w0[0]))
        sec.close()
    assert np.abs(G - Gref).max() <= 1e-9, np.abs(G - Gref).max()
    # sum rule: total spectral weight of c^dagger and c channels = 1
    assert abs((G * 1j * wm)[-1].real - 1.0) < 0.2


def test_fused_and_plain_recurrence_agree(built):
    """The fused Lanczos (pass-A epilogue, unnormalised vectors) and the plain one produce the same tridiagonal."""
    import torch
    import hxv
    from hxv import models

    m = models.hm_1dchain()          # C2, Ns=12, Dim=853776: tiled kernels with several blocks
    sec = hxv.HxvSector.from_model(m, 6, 6)
    sec.set_option("lds_budget_kb_up", 16)   # force several prefix blocks (block hops + row slots in the epilogue path)
    sec.set_option("lds_budget_kb_dw", 32)
    v = models.deterministic_vector(sec.Dim)
    v /= np.linalg.norm(v)
    dv = torch.from_numpy(v).cuda()
    sec.set_option("lanczos_fused", 1)
    a1, b1, n1 = sec.lanczos_tridiag(dv, 60)
    sec.set_option("lanczos_fused", 0)
    a0, b0, n0 = sec.lanczos_tridiag(dv, 60)
    assert n0 == n1 == 60
    assert np.abs(a1 - a0).max() <= 1e-10 * np.abs(a0).max()
    assert np.abs(b1 - b0).max() <= 1e-10 * np.abs(b0).max()
    sec.close()
    # with the Jx / Jp block: folded into pass A the fused recurrence covers it too; as its own pass the plain recurrence runs
    m = models.bhz_2d(Nbath=1, Ust=0.5, Jh=0.1, Jx=0.2, Jp=0.1)   # Ns = 12
    sec = hxv.HxvSector.from_model(m, 6, 5)
    v = models.deterministic_vector(sec.Dim)
    v /= np.linalg.norm(v)
    dv = torch.from_numpy(v).cuda()
    runs = []
    for fold, fused in ((1, 1), (1, 0), (0, 1)):
        sec.set_option("fold_nd", fold)
        sec.set_option("lanczos_fused", fused)
        runs.append(sec.lanczos_tridiag(dv, 40))
    for a, b, n in runs[1:]:
        assert n == runs[0][2] == 40
        assert np.abs(a - runs[0][0]).max() <= 1e-10 * np.abs(runs[0][0]).max()
        assert np.abs(b - runs[0][1]).max() <= 1e-10 * np.abs(runs[0][1]).max()
    sec.close()


@pytest.mark.parametrize("spin,create", [(0, True), (0, False), (1, True), (1, False)])
def test_ladder_operator_on_device_matches_host(built, spin, create):
    """hxv_apply_ladder (GF start vectors on device) vs the host restatement of ED_GF_NORMAL.f90:180-199."""
    import torch
    import hxv
    from hxv import models

    m = models.bhz_2d(Nbath=0)   # Ns=8
    N = (4, 3)
    d = 1 if create else -1
    to = (N[0] + d, N[1]) if spin == 0 else (N[0], N[1] + d)
    s0 = hxv.HxvSector.from_model(m, *N)
    s1 = hxv.HxvSector.from_model(m, *to)
    psi = models.deterministic_vector(s0.Dim)
    for orbital in (0, 3, 7):
        ref = _apply_op(psi, s0.maps(), s1.maps(), orbital, spin, create)
        out, n2 = s0.apply_ladder(s1, orbital, spin, create, torch.from_numpy(psi).cuda())
        assert np.array_equal(out.cpu().numpy(), ref)           # a signed permutation: bit-exact
        assert abs(n2 - np.vdot(ref, ref).real) <= 1e-12 * max(1.0, n2)
    with pytest.raises(hxv.HxvError):
        s0.apply_ladder(s0, 0, spin, create, torch.from_numpy(psi).cuda())   # wrong target sector


def test_host_vector_lanczos_entries(built):
    """hxv_lanczos_tridiag_host / _eigh_host: host vectors in the reference layout, one PCIe copy per run."""
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    m = models.hm_2dsquare(Nbath=1)
    sec = hxv.HxvSector.from_model(m, 4, 4)     # DimUp = 70: exercises the padded device pitch (72)
    assert sec.pitch == 72
    orc = OracleSector(m, 4, 4)
    v = models.deterministic_vector(sec.Dim)
    v /= np.linalg.norm(v)
    # 25 steps: beyond ~30 the plain recurrence loses orthogonality and two fp64 implementations drift apart
    a_ref, b_ref = orc.lanc_tridiag(v, 25)
    a, b, n = sec.lanczos_tridiag_host(v, 25)
    assert n == 25 and np.abs(a - a_ref).max() <= 1e-10 * np.abs(a_ref).max() and np.abs(b - b_ref).max() <= 1e-10 * np.abs(b_ref).max()
    e0, vec, nit = sec.lanczos_eigh_host(512, 1e-14)
    H = orc.dense()
    assert abs(e0 - np.linalg.eigvalsh(H)[0]) <= 1e-10
    assert np.abs(H @ vec - e0 * vec).max() < 1e-8 and abs(np.vdot(vec, vec).real - 1) < 1e-12


# ---- several lowest eigenpairs: the sp_eigh (P-ARPACK) call of ED_DIAG.f90:152-160 on the device ----------------
@pytest.mark.parametrize("case,neigen,ncv", [("C1", 1, 10), ("C1", 2, 20), ("C1", 4, 36), ("chain", 2, 20), ("chain", 3, 12),
                                              ("bhz", 2, 20), ("chain8", 4, 20), ("kanamori", 2, 20)])
def test_eigh_lowest_vs_lapack(built, case, neigen, ncv):
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    if case == "C1":
        m, (nup, ndw) = models.plaquette_2x2_nobath(U=4.0, t=1.0, hfmode=False), (2, 2)
    elif case == "chain":
        m, (nup, ndw) = models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6]), (3, 3)
    elif case == "chain8":
        m, (nup, ndw) = models.hm_1dchain(Nlat=2, Nbath=3), (4, 4)
    elif case == "kanamori":
        m, (nup, ndw) = models.bhz_2d(Nbath=0, Jx=0.3, Jp=0.2, Jh=0.3, Ust=1.0), (4, 4)
    else:
        m, (nup, ndw) = models.bhz_2d(Nbath=0), (4, 4)
    sec = hxv.HxvSector.from_model(m, nup, ndw)
    sec.set_option("eigh_degenerate", 1)        # compared with a DENSE spectrum: every copy of a degenerate level is wanted (default: ARPACK's single Krylov space)
    Hd = OracleSector(m, nup, ndw).dense()
    ref = np.linalg.eigvalsh(Hd)
    ev, X, nconv, nmv = sec.eigh_lowest(neigen, ncv)
    assert nconv == neigen and nmv > 0
    assert nmv == sec.get_option("eigh_last_search_products") + sec.get_option("eigh_last_check_products")
    assert np.abs(ev - ref[:neigen]).max() < 1e-10                      # BASELINE: E within 1e-10 of the CPU reference
    X = X.cpu().numpy().T                                               # (Dim, neigen)
    assert np.abs(X.conj().T @ X - np.eye(neigen)).max() < 1e-11
    assert np.linalg.norm(Hd @ X - X * ev, axis=0).max() < 1e-9
    if case == "C1" and neigen == 1:
        assert abs(ev[0] - (-2.10274848)) < 5e-9                        # survey-recorded reference value
    # host variant: eig_basis(Dim, Neigen) in the reference's layout
    ev_h, basis, nconv_h, _ = sec.eigh_lowest_host(neigen, ncv)
    assert np.abs(ev_h - ev).max() < 1e-12 and basis.shape == (sec.Dim, neigen)
    assert np.linalg.norm(Hd @ basis - basis * ev_h, axis=0).max() < 1e-9


def test_eigh_lowest_matches_numpy_restatement_and_arpack_C2(built):
    """C2 (Dim = 853 776): device thick-restart Lanczos == its numpy restatement (same start vector, same restart rule)
    and == scipy ARPACK (the algorithm family of sp_eigh) on the oracle's matrices."""
    import scipy.sparse.linalg as sla
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector
    from helpers_matrix import oracle_full_matrix
    from trlan_numpy import trlan_lowest

    m = models.hm_1dchain()
    sec = hxv.HxvSector.from_model(m, 6, 6)
    assert sec.get_option("eigh_degenerate") == 0                           # default: one Krylov space, like ARPACK
    sec.set_option("eigh_degenerate", 1)                                    # asked for: plus the check round for hidden copies
    ev_d, _, nconv_d, nmv_d = sec.eigh_lowest(2, 20, want_vectors=False)
    n_search, n_check = sec.get_option("eigh_last_search_products"), sec.get_option("eigh_last_check_products")
    sec.set_option("eigh_degenerate", 0)                                    # the bare algorithm of the numpy restatement
    ev, X, nconv, nmv = sec.eigh_lowest(2, 20)
    assert nconv == 2 and nconv_d == 2 and np.abs(ev_d - ev).max() < 1e-11
    assert nmv < nmv_d <= 2 * nmv                                           # the check round stops once the residual bound clears E_2
    assert n_search == nmv and n_check == nmv_d - nmv and sec.get_option("eigh_last_check_products") == 0
    H = oracle_full_matrix(OracleSector(m, 6, 6))
    ref = np.sort(sla.eigsh(H, k=2, which="SA", ncv=20, tol=1e-13)[0])
    assert np.abs(ev - ref).max() < 1e-10
    Xh = X.cpu().numpy().T
    assert np.linalg.norm(H @ Xh - Xh * ev, axis=0).max() < 1e-9
    ev_np, _, _, nmv_np, _ = trlan_lowest(lambda v: H @ v, sec.Dim, 2, 20)
    assert np.abs(ev - ev_np).max() < 1e-11
    assert abs(nmv - nmv_np) <= 0.25 * nmv_np                         # same algorithm: same matvec count up to rounding-driven restarts
    # the single-vector Lanczos agrees on the ground state
    e0, _, _ = sec.lanczos_eigh(512, 1e-13, want_vector=False)
    assert abs(e0 - ev[0]) < 1e-10


def test_eigh_lowest_argument_errors(built):
    import hxv
    from hxv import models

    sec = hxv.HxvSector.from_model(models.plaquette_2x2_nobath(), 2, 2)
    sec.eigh_lowest(2, 100, want_vectors=False)          # Dim=36 clamps the basis to 36 vectors: fine
    big = hxv.HxvSector.from_model(models.hm_1dchain(Nlat=2, Nbath=2), 3, 3)
    with pytest.raises(hxv.HxvError, match="ncv > 64"):
        big.eigh_lowest(2, 100)                          # Dim=400 does not
    with pytest.raises(hxv.HxvError, match="neigen > Dim"):
        sec.eigh_lowest(37, 10)
    shard = hxv.HxvSector.from_model(models.hm_1dchain(Nlat=2, Nbath=2), 3, 3, rank=0, nranks=2)
    with pytest.raises(hxv.HxvError, match="hxv_comm_init"):   # a split sector needs the engine's communicator first
        shard.eigh_lowest(1, 10)


def test_engine_spectra_vs_numbers_recorded_from_the_reference(built):
    """The engine's own lowest eigenvalues against the spectra the survey recorded from the REFERENCE's dense sector
    Hamiltonians (tests/golden/survey_known_answers.json; 8 decimals recorded): the one pin that does not pass through
    the oracle at all.  nnz and dimensions of the engine's matrices against the recorded ones as well."""
    import json
    from pathlib import Path
    import hxv
    from hxv import models

    gold = json.loads((Path(__file__).parent / "golden" / "survey_known_answers.json").read_text())
    g = gold["C1_plaquette_2x2_U4_t1_hfF_sector_2_2"]
    sec = hxv.HxvSector.from_model(models.plaquette_2x2_nobath(U=4.0, t=1.0, hfmode=False), 2, 2)
    assert sec.Dim == g["Dim"] and len(sec.csr("up")[1]) == g["nnz_up"]
    ev, _, nconv, _ = sec.eigh_lowest(4, 36, want_vectors=False)       # Dim=36: the Krylov space closes, all pairs exact
    assert np.allclose(ev, g["lowest"], atol=5e-9)
    assert abs(sec.lanczos_eigh(512, 1e-13, want_vector=False)[0] - g["lowest"][0]) < 5e-9
    m = models.bhz_2d(Nbath=0)
    g = gold["BHZ_2x2_Norb2_Nspin2_Nbath0_sector_4_4"]
    sec = hxv.HxvSector.from_model(m, 4, 4)
    assert sec.Dim == g["Dim"] and len(sec.csr("up")[1]) == g["nnz_up"] and len(sec.csr("dw")[1]) == g["nnz_dw"]
    ev, _, nconv, _ = sec.eigh_lowest(2, 20, want_vectors=False)       # (the third level is degenerate: one Krylov space sees one copy)
    assert nconv == 2 and np.allclose(ev, g["lowest"][:2], atol=5e-9)
    g = gold["BHZ_2x2_Norb2_Nspin2_Nbath0_sector_3_5"]
    sec = hxv.HxvSector.from_model(m, 3, 5)
    assert sec.Dim == g["Dim"]
    ev, _, nconv, _ = sec.eigh_lowest(3, 30, want_vectors=False)
    assert nconv == 3 and np.allclose(ev, g["lowest"][:3], atol=5e-9)


def test_eigh_lowest_recovers_degenerate_levels(built):
    """ED_DIAG.f90:234-244 keeps every state within gs_threshold of the minimum, so a doubly degenerate level must come back
    with both copies.  BHZ 2x2 sector (4,4): E = -5.80307083, -5.69466351 (x2), -5.61135083 (values recorded from the
    reference's dense H).  One Krylov space sees one copy per level; the locking rounds find the other."""
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    m = models.bhz_2d(Nbath=0)
    sec = hxv.HxvSector.from_model(m, 4, 4)
    Hd = OracleSector(m, 4, 4).dense()
    ref = np.linalg.eigvalsh(Hd)
    assert abs(ref[1] - ref[2]) < 1e-10 and np.allclose(ref[:4], [-5.80307083, -5.69466351, -5.69466351, -5.61135083], atol=5e-9)
    sec.set_option("eigh_degenerate", 1)                      # ask for the copies (default: ARPACK's single Krylov space)
    ev, vecs, nconv, nmv = sec.eigh_lowest(3, 20, 512, 0.0)
    assert nconv == 3 and np.abs(ev - ref[:3]).max() < 1e-10, (ev, ref[:4])
    X = vecs.cpu().numpy()                                    # [3, Dim]
    assert np.abs(X.conj() @ X.T - np.eye(3)).max() < 1e-9    # the two copies are orthonormal
    for i in range(3):
        assert np.linalg.norm(Hd @ X[i] - ev[i] * X[i]) < 1e-8
    ev4, _, nconv4, _ = sec.eigh_lowest(4, 24, 512, 0.0, want_vectors=False)
    assert nconv4 == 4 and np.abs(ev4 - ref[:4]).max() < 1e-10
    sec.set_option("eigh_degenerate", 0)                      # single Krylov space: every value is an eigenvalue, copies only by luck
    ev0, _, _, nmv0 = sec.eigh_lowest(3, 20, 512, 0.0, want_vectors=False)
    assert all(np.abs(ref - e).min() < 1e-9 for e in ev0) and nmv0 <= nmv
    sec.close()
    # no degeneracy: the extra round finds nothing below the wanted set and the result is unchanged
    m2 = models.hm_1dchain(eps_bath=[0.3, 0.6])
    s2 = hxv.HxvSector.from_model(m2, 6, 6)
    s2.set_option("eigh_degenerate", 1)
    e1, _, n1, _ = s2.eigh_lowest(2, 20, 512, 0.0, want_vectors=False)
    s2.set_option("eigh_degenerate", 0)
    e0, _, n0, _ = s2.eigh_lowest(2, 20, 512, 0.0, want_vectors=False)
    assert n1 == 2 and n0 == 2 and np.abs(e1 - e0).max() < 1e-11
    s2.close()


def test_eigh_lowest_partial_reorthogonalisation(built):
    """The omega-recurrence (estimated loss of orthogonality, whole-basis Gram-Schmidt only when an estimate passes
    sqrt(eps)) against the round-1 scheme that measured every projection at every step: same eigenpairs, residuals and
    orthonormality to the same tolerances, most steps local."""
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector
    from helpers_matrix import oracle_full_matrix

    for m, sector, neig, ncv in ((models.hm_1dchain(eps_bath=[0.3, 0.6]), (6, 6), 3, 24), (models.bhz_2d(Nbath=0, Ust=0.4, Jh=0.1), (4, 4), 4, 30),
                                 (models.hm_2dsquare(Nbath=1), (4, 4), 2, 12)):
        sec = hxv.HxvSector.from_model(m, *sector)
        H = oracle_full_matrix(OracleSector(m, *sector))
        res = {}
        for mode in (1, 0):
            sec.set_option("eigh_measure_all", mode)
            ev, X, nconv, nmv = sec.eigh_lowest(neig, ncv, 512, 0.0)
            Xh = X.cpu().numpy().T
            assert nconv == neig
            assert np.linalg.norm(H @ Xh - Xh * ev, axis=0).max() < 1e-8, (m.name, mode)
            assert np.abs(Xh.conj().T @ Xh - np.eye(neig)).max() < 1e-9, (m.name, mode)
            res[mode] = (ev, nmv, sec.get_option("eigh_last_full_passes"), sec.get_option("eigh_last_local_passes"))
        assert np.abs(res[0][0] - res[1][0]).max() < 1e-10, m.name
        assert res[1][3] == 0 and res[0][3] > res[0][2], (m.name, res)      # measure-all has no local passes; the default is mostly local
        assert res[0][1] <= 1.3 * res[1][1] + 20, (m.name, res)               # and needs about as many products
        sec.close()


# ---- REAL-vector mode of the device Lanczos drivers (H real + real start vector -> double instead of complex(8)) ---
def test_real_vector_lanczos_matches_complex_mode(built):
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    m = models.hm_1dchain(Nlat=2, Nbath=3)              # Ns=8, sector (4,4), Dim=4900, real H
    sec = hxv.HxvSector.from_model(m, 4, 4)
    assert sec.real_vectors_available and sec.get_option("real_vectors") == 1
    Hd = OracleSector(m, 4, 4).dense()
    ref = np.linalg.eigvalsh(Hd)
    rng = np.random.default_rng(5)
    v = rng.standard_normal(sec.Dim)
    v /= np.linalg.norm(v)
    vin = torch.from_numpy(v.astype(np.complex128)).cuda()
    res = {}
    for mode in (1, 0):
        sec.set_option("real_vectors", mode)
        a, b, n = sec.lanczos_tridiag(vin, 30)
        assert sec.get_option("lanczos_real_last") == mode
        e0, vec, nit = sec.lanczos_eigh(512, 1e-13)
        assert sec.get_option("lanczos_real_last") == mode
        ev, X, nconv, nmv = sec.eigh_lowest(2, 20)
        assert sec.get_option("lanczos_real_last") == mode and nconv == 2
        res[mode] = (a, b, e0, vec.cpu().numpy(), ev, X.cpu().numpy())
        assert abs(e0 - ref[0]) < 1e-10 and np.abs(ev - ref[:2]).max() < 1e-10
        x = res[mode][3]
        assert np.linalg.norm(Hd @ x - e0 * x) < 1e-8 and abs(np.linalg.norm(x) - 1) < 1e-12
        Xh = res[mode][5].T
        assert np.linalg.norm(Hd @ Xh - Xh * ev, axis=0).max() < 1e-9
    # same start vector -> the same tridiagonal matrix in both modes
    assert np.abs(res[1][0] - res[0][0]).max() < 1e-11 and np.abs(res[1][1] - res[0][1]).max() < 1e-11
    assert np.abs(res[1][3].imag).max() == 0.0        # real mode returns a real eigenvector in the complex layout
    # a start vector with an imaginary part keeps the complex path even when real vectors are enabled
    sec.set_option("real_vectors", 1)
    vc = models.deterministic_vector(sec.Dim)
    vc /= np.linalg.norm(vc)
    sec.lanczos_tridiag(torch.from_numpy(vc).cuda(), 10)
    assert sec.get_option("lanczos_real_last") == 0
    # both fused and plain recurrences exist in real mode
    sec.set_option("lanczos_fused", 0)
    a2, b2, _ = sec.lanczos_tridiag(vin, 30)
    assert sec.get_option("lanczos_real_last") == 1
    assert np.abs(a2 - res[1][0]).max() < 1e-11 and np.abs(b2 - res[1][1]).max() < 1e-11


def test_real_vector_mode_not_used_for_complex_h(built):
    import hxv
    from hxv import models

    sec = hxv.HxvSector.from_model(models.bhz_2d(Nbath=0), 4, 4)
    sec.lanczos_eigh(512, 1e-12, want_vector=False)
    assert sec.get_option("lanczos_real_last") == 0
    sec.eigh_lowest(1, 10, want_vectors=False)
    assert sec.get_option("lanczos_real_last") == 0


def test_device_resident_green_function_with_mixed_channels(built):
    """The whole buildgf step on the device: ground state (hxv_lanczos_eigh) -> start vectors c^dagger_i|gs>,
    (c^dagger_i + c^dagger_j)|gs>, (c^dagger_i + xi c^dagger_j)|gs> and the c counterparts (hxv_apply_ladder[_axpy];
    ED_GF_NORMAL.f90:180-199, 370-406, 746-780, 827-861) -> tridiagonal (hxv_lanczos_tridiag) -> continued fraction.
    Each channel O is checked against the Lehmann sum  sum_n |<n|O|gs>|^2 / (i w -+ (E_n - E0))  of the dense sector."""
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    m, N = models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.25, -0.4], U=2.0), 3
    beta, Lmats = 50.0, 32
    wm = np.pi / beta * (2 * np.arange(1, Lmats + 1) - 1)
    gs = hxv.HxvSector.from_model(m, N, N)
    e0, psi, _ = gs.lanczos_eigh(512, 1e-14, native=True)      # padded device vector, stays on the device
    w0, U0 = np.linalg.eigh(OracleSector(m, N, N).dense())
    assert abs(w0[0] - e0) < 1e-10
    # fix the arbitrary sign of the reference eigenvector to the device one
    psi_h = gs.unpad(psi).cpu().numpy()
    psi_ref = U0[:, 0] * np.sign(np.vdot(U0[:, 0], psi_h).real)
    maps0 = gs.maps()
    i_site, j_site, spin = 0, 1, 0
    for create, (nu, nd) in ((True, (N + 1, N)), (False, (N - 1, N))):
        sec = hxv.HxvSector.from_model(m, nu, nd)
        maps1 = sec.maps()
        w1, U1 = np.linalg.eigh(OracleSector(m, nu, nd).dense())
        sign = 1.0 if create else -1.0
        # the reference's four operator combinations per pair (i,j): O = c_i, c_i + c_j, c_i +- xi c_j (dagger for create)
        for cj in (0.0, 1.0, 1j if create else -1j):
            vv, n2 = gs.apply_ladder(sec, i_site, spin, create, psi)
            if cj != 0.0:
                vv, n2 = gs.apply_ladder(sec, j_site, spin, create, psi, coef=cj, out=vv)
            ref = _apply_op(psi_ref, maps0, maps1, i_site, spin, create)
            if cj != 0.0:
                ref = ref + cj * _apply_op(psi_ref, maps0, maps1, j_site, spin, create)
            assert abs(n2 - np.vdot(ref, ref).real) < 1e-9
            nl = min(sec.Dim, 150)
            a, b, n = sec.lanczos_tridiag(vv / np.sqrt(n2), nl, threshold=1e-12)
            assert sec.get_option("lanczos_real_last") == (0 if isinstance(cj, complex) and cj.imag != 0 else 1)
            ev, Z = np.linalg.eigh(np.diag(a[:n]) + np.diag(b[1:n], 1) + np.diag(b[1:n], -1))
            G = (n2 * Z[0, :] ** 2 / (1j * wm[:, None] - sign * (ev[None, :] - e0))).sum(axis=1)
            amp = np.abs(U1.conj().T @ ref) ** 2
            Gref = (amp[None, :] / (1j * wm[:, None] - sign * (w1[None, :] - w0[0]))).sum(axis=1)
            assert np.abs(G - Gref).max() <= 1e-9, (create, cj, np.abs(G - Gref).max())
        sec.close()


def test_sharded_lanczos_driver_on_device_matches_native_driver(built):
    """hxv.ShardedLanczos (the MpiStatus=T flavour of sp_lanc_tridiag / sp_lanc_eigh: torch vector ops + all-reduced dots
    around the per-rank product) at world size 1 against the in-library drivers: same start vector, same numbers."""
    import torch
    import hxv
    from hxv import models

    m = models.hm_2dsquare(Nbath=1)                       # DimUp = 70 -> pitch 72: pad rows in play
    sec = hxv.HxvSector.from_model(m, 4, 4)
    sec.set_option("real_vectors", 0)                     # same complex start vector as the sharded driver
    sh = hxv.ShardedHxv(sec.DimUp, sec.DimDw, 0, 1, sec.apply_device, pitch=sec.pitch)
    lz = hxv.ShardedLanczos(sh)
    e0, vec, nit = lz.eigh(512, 1e-13, device="cuda")
    e0n, vecn, nitn = sec.lanczos_eigh(512, 1e-13, native=True)
    assert abs(e0 - e0n) < 1e-11 and abs(nit - nitn) <= 2
    ov = abs(torch.vdot(vec, vecn).item())
    assert abs(ov - 1.0) < 1e-9
    v = models.deterministic_vector(sec.Dim)
    v /= np.linalg.norm(v)
    dv = sec.pad(torch.from_numpy(v).cuda())
    a, b, n = lz.tridiag(dv, 30)
    an, bn, nn = sec.lanczos_tridiag(dv, 30)
    assert n == nn == 30 and np.abs(a - an).max() < 1e-10 and np.abs(b - bn).max() < 1e-10


def test_device_buffer_cache_reuses_vector_sized_buffers(built):
    """include/hxv.h 'device-buffer cache': what one handle frees (dw-hop scratch, Lanczos vectors, Krylov basis) serves
    the next handle instead of a fresh hipMalloc (~25 ms per GB on this platform)."""
    import hxv
    from hxv import models

    m = models.hm_1dchain(Nlat=2, Nbath=3)
    hxv.pool_trim()
    s0 = hxv.pool_stats()
    assert s0["cached_bytes"] == 0
    a = hxv.HxvSector.from_model(m, 4, 4)
    e_a = a.eigh_lowest(2, 12, want_vectors=False)[0]
    e0_a = a.lanczos_eigh(300, 1e-12, want_vector=False)[0]
    a.close()
    s1 = hxv.pool_stats()
    assert s1["cached_bytes"] > 0 and s1["misses"] > s0["misses"]
    b = hxv.HxvSector.from_model(m, 4, 4)
    e_b = b.eigh_lowest(2, 12, want_vectors=False)[0]
    e0_b = b.lanczos_eigh(300, 1e-12, want_vector=False)[0]
    s2 = hxv.pool_stats()
    assert s2["hits"] >= s1["hits"] + 3            # basis, scratch, Lanczos vectors came from the cache
    assert np.array_equal(e_a, e_b) and e0_a == e0_b   # recycled (dirty) memory changes nothing
    b.close()
    hxv.pool_trim()
    assert hxv.pool_stats()["cached_bytes"] == 0


@pytest.mark.parametrize("case", ["chain8", "C2_blocks", "tiny_breakdown", "bhz_complex"])
def test_graph_captured_tridiagonalisation_equals_the_stepwise_one(built, case):
    """hxv_lanczos_tridiag runs iterations 1.. on the device alone, three per hipGraph (option lanczos_graph): the same
    kernels on the same data as the host-stepped recurrence, so alanc/blanc agree to the last bit, including a breakdown."""
    import time
    import torch
    import hxv
    from hxv import models

    if case == "chain8":
        sec, nl = hxv.HxvSector.from_model(models.hm_1dchain(Nlat=2, Nbath=3), 4, 4), 60
    elif case == "C2_blocks":
        sec, nl = hxv.HxvSector.from_model(models.hm_1dchain(), 6, 6), 100
        sec.set_option("lds_budget_kb_up", 16)
        sec.set_option("lds_budget_kb_dw", 32)
    elif case == "tiny_breakdown":
        sec, nl = hxv.HxvSector.from_model(models.plaquette_2x2_nobath(), 2, 2), 60      # Dim = 36 < nlanc
    else:
        sec, nl = hxv.HxvSector.from_model(models.bhz_2d(Nbath=0), 4, 4), 50
    rng = np.random.default_rng(2)
    for real_start in (True, False):
        v = rng.standard_normal(sec.Dim) + (0 if real_start else 1j) * rng.standard_normal(sec.Dim)
        v = (v / np.linalg.norm(v)).astype(np.complex128)
        dv = torch.from_numpy(v).cuda()
        out = {}
        for g in (1, 0):
            sec.set_option("lanczos_graph", g)
            t0 = time.perf_counter()
            out[g] = sec.lanczos_tridiag(dv, nl, threshold=1e-10)
            out[g] += (time.perf_counter() - t0,)
        (a1, b1, n1, _), (a0, b0, n0, _) = out[1], out[0]
        if case == "tiny_breakdown":
            # past ~Dim steps the recurrence runs on rounding noise: compare the common, meaningful part
            m = min(n0, n1, 20)
            assert m >= 10 and np.allclose(a1[:m], a0[:m], rtol=0, atol=1e-9) and np.allclose(b1[:m], b0[:m], rtol=0, atol=1e-9)
        else:
            assert n1 == n0 == nl
            assert np.array_equal(a1, a0) and np.array_equal(b1, b0)


def test_lanczos_after_eigh_on_a_fresh_handle_c4(built):
    """The single-vector Lanczos right after the thick-restart solver on a handle that has not run Lanczos before: its
    three vectors then come out of the pool block the solver's basis has just returned, and the zero-fill of a recycled
    block has to be ordered with the kernels that follow (the handle's stream does not synchronise with the null stream;
    at C4 size a null-stream memset used to land after the start vector had been written: E0 = 0 after one iteration)."""
    import torch
    import hxv
    from hxv import models

    sec = hxv.HxvSector.from_model(models.bhz_2d(Nbath=1), 8, 8)
    ev, X, nconv, nmv = sec.eigh_lowest(2, 20, native=True)
    del X
    torch.cuda.empty_cache()
    e0, vec, nit = sec.lanczos_eigh(512, 1e-13, native=True)
    assert nit > 20 and abs(e0 - ev[0]) < 1e-9, (e0, ev, nit)
    hv = sec.apply_device(vec)
    assert (hv - e0 * vec).norm().item() < 1e-8
    sec.close()
    del hv, vec
    torch.cuda.empty_cache()
    hxv.pool_trim()  # (the Ns=18 tests that follow need nearly all of the HBM)


@pytest.mark.parametrize("case", ["chain8", "plaquette_ns8", "C2", "unequal_norms", "breakdown_in_one"])
def test_paired_tridiagonalisation_is_two_single_runs_bit_for_bit(built, case):
    """hxv_lanczos_tridiag_pair: two REAL start vectors as Re / Im of one complex Lanczos vector (real H, H(x+iy) = Hx + iHy;
    the independent Green's-function channels of ED_GF_NORMAL.f90:123-306).  Every alanc/blanc equals, bit for bit, what
    hxv_lanczos_tridiag gives for that start vector through the same kernels (real_vectors = 0, same job_up) -- THAT is the bit-identical
    pairing hxv.h states; against the default real-vector path of the single driver the numbers agree to rounding only (checked below on the
    consumer's quantity) -- and agrees with the oracle's recurrence."""
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    if case == "C2":
        m, (nup, ndw), nl = models.hm_1dchain(), (6, 6), 30
    elif case == "plaquette_ns8":
        m, (nup, ndw), nl = models.hm_2dsquare(Nbath=1), (4, 4), 30          # Ns = 8, Dim 4900 (the pairing hxv.h names, at Ns = 8)
    elif case == "breakdown_in_one":
        m, (nup, ndw), nl = models.hm_1dchain(Nlat=2, Nbath=1), (2, 2), 40   # Dim 36 < nl: the Krylov spaces close
    else:
        m, (nup, ndw), nl = models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6]), (3, 3), 25
    sec = hxv.HxvSector.from_model(m, nup, ndw)
    assert sec.real_vectors_available
    rng = np.random.default_rng(7)
    xa = rng.standard_normal(sec.Dim)
    xb = rng.standard_normal(sec.Dim)
    xa /= np.linalg.norm(xa)
    xb /= np.linalg.norm(xb)
    if case == "unequal_norms":
        xa *= 3.0      # the driver normalises each component by itself, like SciFortran's first iteration
        xb *= 0.25
    if case == "breakdown_in_one":
        # vector a lives in a small invariant subspace: an eigenvector of H -> its recurrence stops after one step
        orc = OracleSector(m, nup, ndw)
        w, U = np.linalg.eigh(orc.dense())
        xa = np.real(U[:, 3] * np.exp(-1j * np.angle(U[np.abs(U[:, 3]).argmax(), 3])))
        xa /= np.linalg.norm(xa)
    da = torch.from_numpy(xa.astype(np.complex128)).cuda()
    db = torch.from_numpy(xb.astype(np.complex128)).cuda()
    sec.set_option("lanczos_graph", 0)
    sec.set_option("real_vectors", 0)
    for job_up in (2, 0):   # the fused product as pipelined jobs (default, where the plan allows) and as one tile per workgroup
        sec.set_option("job_up", job_up)
        (aa, ba, na), (ab, bb, nb) = sec.lanczos_tridiag_pair(da, db, nl)
        assert sec.get_option("lanczos_real_last") == 2
        a1, b1, n1 = sec.lanczos_tridiag(da, nl)
        a2, b2, n2 = sec.lanczos_tridiag(db, nl)
        assert (na, nb) == (n1, n2)
        if case == "breakdown_in_one":
            assert na < 5 and nb > na
        # bit-identical in every step both runs made (entries past a breakdown stay zero in both)
        assert np.array_equal(aa[:na], a1[:n1]) and np.array_equal(ba[:na], b1[:n1]), (job_up, np.abs(aa - a1).max(), np.abs(ba - b1).max())
        assert np.array_equal(ab[:nb], a2[:n2]) and np.array_equal(bb[:nb], b2[:n2]), (job_up, np.abs(ab - a2).max(), np.abs(bb - b2).max())
    if case != "breakdown_in_one":
        # the single driver's DEFAULT path (real vectors) is another summation order: equal to rounding, not bit for bit
        sec.set_option("real_vectors", 1)
        a3, b3, n3 = sec.lanczos_tridiag(db, nl)
        assert sec.get_option("lanczos_real_last") == 1 and n3 == nb
        k3 = min(n3, 8)
        assert np.abs(ab[:k3] - a3[:k3]).max() <= 1e-10 * max(1.0, np.abs(a3).max()) and np.abs(bb[:k3] - b3[:k3]).max() <= 1e-10 * max(1.0, np.abs(b3).max())
        sec.set_option("real_vectors", 0)
        orc = OracleSector(m, nup, ndw)
        ar, br = orc.lanc_tridiag(xb.astype(np.complex128) / np.linalg.norm(xb), nl)
        k = min(len(ar), 12)   # (the early steps: later ones amplify rounding differences of the start)
        assert np.abs(ab[:k] - ar[:k]).max() <= 1e-9 * np.abs(ar).max()
    # a complex start vector is refused
    with pytest.raises(hxv.HxvError):
        sec.lanczos_tridiag_pair(da * (1 + 0.5j), db, 4)
    sec.close()


def test_paired_tridiagonalisation_refused_for_complex_h(built):
    import torch
    import hxv
    from hxv import models

    m = models.bhz_2d(Nbath=0)
    sec = hxv.HxvSector.from_model(m, 4, 4)
    v = torch.ones(sec.Dim, dtype=torch.complex128, device="cuda")
    with pytest.raises(hxv.HxvError):
        sec.lanczos_tridiag_pair(v, v, 4)
    sec.close()


def test_eigh_lowest_loose_tolerance_returns_no_duplicate(built):
    """with tol far above machine epsilon the locked pairs are only converged to tol, every product re-injects a
    component along them, and a locking round that did not subtract them could converge back onto a locked state and return it
    twice.  Non-degenerate spectrum, tol = 1e-6: the returned values are the two lowest DISTINCT eigenvalues."""
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    m = models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6])
    sec = hxv.HxvSector.from_model(m, 3, 3)
    w = np.linalg.eigvalsh(OracleSector(m, 3, 3).dense())
    assert w[1] - w[0] > 1e-3 and w[2] - w[1] > 1e-3           # non-degenerate
    sec.set_option("eigh_degenerate", 1)                       # the locking rounds are what this test is about
    for rv in (1, 0):
        sec.set_option("real_vectors", rv)
        for neigen in (2, 3):
            ev, vecs, nconv, nmv = sec.eigh_lowest(neigen, 12, 300, 1e-6)
            assert nconv == neigen
            assert np.abs(ev - w[:neigen]).max() < 1e-5, (ev, w[:neigen])   # accuracy of a 1e-6 residual, and no repeated value
            assert np.min(np.diff(ev)) > 1e-4
            G = (vecs.conj() @ vecs.T).cpu().numpy()
            assert np.abs(G - np.eye(neigen)).max() < 1e-6
    sec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("model_name", ["hubbard_norb1", "bhz_complex", "kanamori_complex", "two_orbital_real_nd"])
def test_fused_epilogue_matches_plain_recurrence_for_every_kernel_family(built, model_name):
    """the fused Lanczos epilogue (pass A tile kernel with C = 2 / 4 columns, the job kernel, real-vector kernels, the
    spH0nd variant) against the unfused recurrence on the same handle -- alanc / blanc of a run long enough that the previous-vector
    term (xm non-null) and several out-of-block slots are exercised -- for every kernel family the launcher can pick."""
    import torch
    import hxv
    from hxv import models

    m, (nup, ndw) = {
        "hubbard_norb1": (models.hm_1dchain(Nlat=2, Nbath=3), (4, 4)),                                   # Ns = 8, real H, one orbital
        "bhz_complex": (models.bhz_2d(Nbath=0, Ust=0.3, Jh=0.1), (4, 3)),                                # complex H, Norb = 2
        "kanamori_complex": (models.bhz_2d(Nbath=0, Ust=0.7, Jh=0.2, Jx=0.2, Jp=0.15), (4, 4)),          # + spH0nd folded into pass A
        "two_orbital_real_nd": (models.bhz_2d(Nbath=0, lam=0.0, Ust=0.5, Jh=0.1, Jx=0.3, Jp=0.1), (3, 5)),
    }[model_name]
    sec = hxv.HxvSector.from_model(m, nup, ndw)
    v = models.deterministic_vector(sec.Dim)
    v /= np.linalg.norm(v)
    dv = torch.from_numpy(v).cuda()
    vr = torch.from_numpy((v.real / np.linalg.norm(v.real)).astype(np.complex128)).cuda()   # a real start vector
    sec.set_option("lds_budget_kb_up", 8)    # several prefix blocks per spin: block hops and row slots in the epilogue path
    sec.set_option("lds_budget_kb_dw", 8)
    sec.set_option("lanczos_fused", 0)
    sec.set_option("real_vectors", 0)
    ref = sec.lanczos_tridiag(dv, 24)
    ref_r = sec.lanczos_tridiag(vr, 24)
    tried = 0
    for cols in (2, 4):
        for job_up in (0, 1, 2):
            for real_vectors in (0, 1):
                sec.set_option("cols_per_tile", cols)
                sec.set_option("job_up", job_up)
                sec.set_option("real_vectors", real_vectors)
                sec.set_option("lanczos_fused", 1)
                for start, want in ((dv, ref), (vr, ref_r)):
                    a, b, n = sec.lanczos_tridiag(start, 24)
                    assert n == want[2], (model_name, cols, job_up, real_vectors)
                    assert np.abs(a - want[0]).max() <= 1e-10 * np.abs(want[0]).max(), (model_name, cols, job_up, real_vectors)
                    assert np.abs(b - want[1]).max() <= 1e-10 * np.abs(want[1]).max(), (model_name, cols, job_up, real_vectors)
                    tried += 1
    assert tried == 24
    sec.close()


def test_library_owned_device_vectors_and_pcie_counters(built):
    """hxv_vector_alloc / _from_host / _to_host / _free (what the Fortran glue keeps between calls) and the h2d / d2h byte counters of
    hxv_get_stats: a round trip is exact, device-resident drivers leave the counters alone, host-array entry points count their slabs."""
    import ctypes as C
    import hxv
    from hxv import models

    m = models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6])
    sec = hxv.HxvSector.from_model(m, 3, 3)
    L = hxv.load_library()
    v = models.deterministic_vector(sec.Dim)
    v /= np.linalg.norm(v)
    d = C.c_void_p()
    assert L.hxv_vector_alloc(sec._h, C.byref(d)) == 0 and d.value
    assert sec.stats()["h2d_bytes"] == 0 and sec.stats()["d2h_bytes"] == 0
    assert L.hxv_vector_from_host(sec._h, v.ctypes.data, d) == 0
    back = np.zeros_like(v)
    assert L.hxv_vector_to_host(sec._h, d, back.ctypes.data) == 0
    assert np.array_equal(back, v)
    st = sec.stats()
    assert st["h2d_bytes"] == 16 * sec.Dim and st["d2h_bytes"] == 16 * sec.Dim
    a = np.zeros(10)
    b = np.zeros(10)
    n = C.c_int32()
    assert L.hxv_lanczos_tridiag(sec._h, d, 10, a.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double)), 1e-12, C.byref(n)) == 0
    assert sec.stats()["h2d_bytes"] == 16 * sec.Dim                          # device-resident: nothing moved
    a2, b2, _ = sec.lanczos_tridiag_host(v, 10)
    assert np.allclose(a, a2, rtol=0, atol=1e-13) and sec.stats()["h2d_bytes"] == 32 * sec.Dim
    sec.apply_host(v)
    st = sec.stats()
    assert st["h2d_bytes"] == 48 * sec.Dim and st["d2h_bytes"] == 32 * sec.Dim
    assert L.hxv_vector_free(sec._h, d) == 0
    sec.close()


_C2E = {}


def _c2e_matrix():
    """oracle matrix of C2 with bath levels (Dim = 853 776) and its 8 lowest eigenvalues from ARPACK, built once per session"""
    if not _C2E:
        import scipy.sparse.linalg as sla
        from hxv import models
        from oracle.oracle import OracleSector
        from helpers_matrix import oracle_full_matrix

        _C2E["model"] = models.hm_1dchain(eps_bath=[0.3, 0.6])
        _C2E["H"] = oracle_full_matrix(OracleSector(_C2E["model"], 6, 6))
        _C2E["ref"] = np.sort(sla.eigsh(_C2E["H"], k=8, which="SA", ncv=48, tol=1e-12)[0])
    return _C2E["model"], _C2E["H"], _C2E["ref"]


@pytest.mark.parametrize("neig,ncv,real_vectors,fused", [(4, 40, 1, 1), (4, 40, 0, 1), (4, 40, 1, 0), (2, 33, 1, 1), (6, 60, 1, 1), (8, 64, 0, 1)])
def test_eigh_lowest_large_krylov_basis(built, neig, ncv, real_vectors, fused):
    """Nblock = lanc_ncv_factor * max(Neigen, lanc_nstates_sector) (ED_DIAG.f90:96) passes 30 as soon as a run asks for more than two states
    per sector (finite temperature).  Round 4 found that cycles of more than ~30 steps LOST the converged Ritz vectors (energies of 1e25
    reported as converged): the forced second cleaning step and the first step of a cycle each made ONE classical Gram-Schmidt pass over
    "H q", which measures a projection before the local / arrow terms are taken out -- the new vector inherited the overlap the step was
    there to remove while the estimates said "clean".  Eigenvalues against ARPACK on the oracle's matrix (C2, Dim = 853 776), residuals,
    orthonormality, and the same answer as the measure-everything mode."""
    import hxv

    m, H, ref = _c2e_matrix()
    sec = hxv.HxvSector.from_model(m, 6, 6)
    sec.set_option("real_vectors", real_vectors)
    sec.set_option("lanczos_fused", fused)
    sec.set_option("eigh_degenerate", 1)                         # (the reference values hold every copy of a degenerate level)
    ev, X, nconv, nmv = sec.eigh_lowest(neig, ncv, 512, 0.0)
    Xh = X.cpu().numpy().T
    assert nconv == neig
    assert np.abs(ev - ref[:neig]).max() < 1e-9, (ev, ref)
    assert np.linalg.norm(H @ Xh - Xh * ev, axis=0).max() < 1e-8
    assert np.abs(Xh.conj().T @ Xh - np.eye(neig)).max() < 1e-9
    assert sec.get_option("eigh_last_local_passes") > 2 * sec.get_option("eigh_last_full_passes")   # still mostly local steps
    sec.set_option("eigh_measure_all", 1)
    ev_all, _, nconv_all, _ = sec.eigh_lowest(neig, ncv, 512, 0.0, want_vectors=False)
    assert nconv_all == neig and np.abs(ev_all - ev).max() < 1e-10
    sec.close()


@pytest.mark.parametrize("site,spin", [(0, 0), (1, 1)])
def test_impurity_green_function_free_fermion_closed_form(built, site, spin):
    """Row N1 against a CLOSED FORM, no oracle in the loop: without interaction the impurity Green's function of the ground state is the one-body
    resolvent, G_ii(i w_n) = [(i w_n - h)^-1]_ii, h assembled straight from the model arrays.  Engine only: ground state (hxv_lanczos_eigh),
    c^dagger / c on the device (hxv_apply_ladder), hxv_lanczos_tridiag, the consumer's continued-fraction assembly (ED_GF_NORMAL.f90:915-975)."""
    import torch
    import hxv
    from hxv import models

    m, N = models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.25, -0.4], U=0.0, hfmode=False, xmu=0.1), 3
    nimp, ns = 2, 6
    h = np.zeros((ns, ns))
    for il in range(2):
        for jl in range(2):
            h[il, jl] += m.impHloc[il, jl, 0, 0, 0, 0].real
            for ib in range(2):
                h[nimp + il + ib * nimp, nimp + jl + ib * nimp] += m.Hbath[il, jl, 0, 0, 0, 0, ib].real
    for ib in range(2):
        for il in range(2):
            h[il, nimp + il + ib * nimp] += m.Vbath[il, 0, 0, ib]
            h[nimp + il + ib * nimp, il] += m.Vbath[il, 0, 0, ib]
    for a in range(nimp):
        h[a, a] -= m.xmu
    eps, phi = np.linalg.eigh(h)
    assert eps[N] - eps[N - 1] > 1e-3                           # closed shell: the ground state of (3,3) is unique
    beta, Lmats = 50.0, 64
    wm = np.pi / beta * (2 * np.arange(1, Lmats + 1) - 1)
    Gexact = (phi[site, :] ** 2 / (1j * wm[:, None] - eps[None, :])).sum(axis=1)
    gs = hxv.HxvSector.from_model(m, N, N)
    e0, psi, _ = gs.lanczos_eigh(nitermax=512, threshold=1e-14, native=True)
    assert abs(e0 - 2 * eps[:N].sum()) < 1e-10
    G = np.zeros(Lmats, dtype=complex)
    for create in (True, False):
        d = 1 if create else -1
        sec = hxv.HxvSector.from_model(m, N + d * (spin == 0), N + d * (spin == 1))
        vv, norm2 = gs.apply_ladder(sec, site, spin, create, psi, out=torch.zeros(sec.localElems, dtype=torch.complex128, device="cuda"))
        a, b, n = sec.lanczos_tridiag(vv, min(sec.Dim, 200), threshold=1e-12)
        a, b = a[:n], b[:n]
        ev, Z = np.linalg.eigh(np.diag(a) + np.diag(b[1:], 1) + np.diag(b[1:], -1))
        sign = 1.0 if create else -1.0
        for j in range(n):
            G += norm2 * Z[0, j] ** 2 / (1j * wm - sign * (ev[j] - e0))
        sec.close()
    gs.close()
    assert np.abs(G - Gexact).max() <= 1e-9, np.abs(G - Gexact).max()
