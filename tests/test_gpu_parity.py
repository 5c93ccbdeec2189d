"""GPU parity tests proper: the HIP path, called through the C-ABI (ctypes), against the CPU
oracle on the same seeded inputs.  Tolerance (stated): max|Hv_gpu - Hv_oracle| / max|Hv_oracle|
<= 1e-13 (fp64, different summation order only)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-13


def _models():
    from hxv import models

    return [
        ("C1", models.plaquette_2x2_nobath(), [(2, 2), (1, 3), (0, 4), (4, 4), (0, 0), (3, 2)]),
        ("chain_B2_eps", models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6]), [(3, 3), (2, 4), (1, 1), (6, 0)]),
        ("square_B1", models.hm_2dsquare(Nbath=1), [(4, 4), (3, 5), (5, 4)]),
        ("bhz_B0", models.bhz_2d(Nbath=0), [(4, 4), (3, 5)]),
        ("bhz_B0_ust", models.bhz_2d(Nbath=0, Ust=0.7, Jh=0.2, xmu=0.1), [(4, 4), (2, 5)]),
        ("bhz_B0_kanamori", models.bhz_2d(Nbath=0, Ust=0.7, Jh=0.2, Jx=0.2, Jp=0.15), [(4, 4), (3, 5), (1, 6)]),
    ]


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("kernel", [0, 1])
def test_hxv_small_sectors_match_oracle(built, kernel):
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    for name, m, sectors in _models():
        for nup, ndw in sectors:
            sec = hxv.HxvSector.from_model(m, nup, ndw)
            orc = OracleSector(m, nup, ndw)
            assert (sec.DimUp, sec.DimDw, sec.Dim) == (orc.DimUp, orc.DimDw, orc.Dim)
            sec.set_option("kernel", kernel)
            v = models.deterministic_vector(sec.Dim)
            ref = orc.spMatVec_main(v)
            hv = sec.apply_device(torch.from_numpy(v).cuda())
            torch.cuda.synchronize()
            assert _rel(hv.cpu().numpy(), ref) <= TOL, (name, nup, ndw, kernel)
            hv2 = sec.apply_host(v)
            assert _rel(hv2, ref) <= TOL, (name, nup, ndw, "host")
            sec.close()
            orc.close()


@pytest.mark.parametrize("kernel", [0, 1])
def test_hxv_c2_ns12_matches_oracle(built, kernel):
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    m = models.hm_1dchain(eps_bath=[0.3, 0.6])
    sec = hxv.HxvSector.from_model(m, 6, 6)
    orc = OracleSector(m, 6, 6)
    sec.set_option("kernel", kernel)
    v = models.deterministic_vector(sec.Dim)
    ref = orc.spMatVec_main(v)
    hv = sec.apply_device(torch.from_numpy(v).cuda())
    torch.cuda.synchronize()
    assert _rel(hv.cpu().numpy(), ref) <= TOL


def test_tiled_vs_naive_c3_slice(built):
    """Ns=16 half-filled sector is too big for the oracle in seconds: compare the tiled kernels with the
    one-thread-per-element kernel on device (both pinned to the oracle at smaller sizes)."""
    import torch
    import hxv
    from hxv import models

    m = models.hm_2dsquare(Nbath=3)
    sec = hxv.HxvSector.from_model(m, 8, 8, rank=3, nranks=64)  # a 201/202-column slab of the full sector
    g = torch.Generator(device="cuda").manual_seed(1)
    n = sec.fullElems
    v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g) + 1j * torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    sec.set_option("kernel", 0)
    a = sec.apply_device(v).clone()
    sec.set_option("kernel", 1)
    b = sec.apply_device(v)
    torch.cuda.synchronize()
    assert (a - b).abs().max().item() / a.abs().max().item() <= TOL


def _slab_reference(orc, v_full):
    """Hv on the slab of an OracleSector opened with (rank,size), from the oracle's own matrices:
    D.v + H_up V[:,slab] + V H_dw[slab,:]^T  (ED_HAMILTONIAN_SPARSE_HxV.f90:250-295 without the transposes)."""
    import scipy.sparse as sp

    du, dd = orc.DimUp, orc.DimDw
    rp, cols, vals = orc.csr("up")
    Hup = sp.csr_matrix((vals, cols - 1, rp), shape=(du, du))
    rp, cols, vals = orc.csr("dw")
    Hdw = sp.csr_matrix((vals, cols - 1, rp), shape=(dd, dd))
    V = v_full.reshape((du, dd), order="F")
    c0 = orc.mpiIshift // du
    sl = slice(c0, c0 + orc.mpiQdw)
    out = orc.diag().reshape((du, orc.mpiQdw), order="F") * V[:, sl] + Hup @ V[:, sl] + (Hdw[sl, :] @ V.T).T
    out = np.asarray(out).reshape(-1, order="F")
    m = orc.model
    if m.Norb > 1 and (m.Jx != 0 or m.Jp != 0):   # spH0nd: local rows, GLOBAL columns (H_non_local.f90:46,84)
        rp, cols, vals = orc.csr("nd")
        out = out + sp.csr_matrix((vals, cols - 1, rp), shape=(orc.vecDim, orc.Dim)) @ v_full
    return out


@pytest.mark.parametrize("bits", [(2, 3), (4, 1), (0, 0), (5, 6)])
@pytest.mark.parametrize("shard", [(0, 1), (1, 3), (4, 5)])
def test_tiled_outer_path_and_shards_match_oracle(built, bits, shard):
    """Force small prefix blocks so the out-of-block (global gather) path and partially local dw blocks
    are exercised; compare each rank's slab with the oracle."""
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    rank, size = shard
    for m, (nup, ndw) in ((models.hm_2dsquare(Nbath=1), (4, 3)), (models.bhz_2d(Nbath=0, Ust=0.4, Jh=0.1), (3, 5)),
                          (models.bhz_2d(Nbath=0, Ust=0.4, Jh=0.1, Jx=0.25, Jp=-0.1), (4, 3))):
        sec = hxv.HxvSector.from_model(m, nup, ndw, rank=rank, nranks=size)
        orc = OracleSector(m, nup, ndw, rank, size)
        assert (sec.vecDim, sec.mpiQdw, sec.mpiIshift) == (orc.vecDim, orc.mpiQdw, orc.mpiIshift)
        v = models.deterministic_vector(sec.Dim)
        ref = _slab_reference(orc, v)
        dv = torch.from_numpy(sec.to_gather_layout(v, size)).cuda()
        for cols, rows, threads, srt in ((2, 2, 256, 0), (4, 4, 512, 1), (8, 8, 1024, 2)):
            sec.set_option("tile_bits_up", bits[0])
            sec.set_option("tile_bits_dw", bits[1])
            sec.set_option("cols_per_tile", cols)
            sec.set_option("rows_per_tile", rows)
            sec.set_option("threads_up", threads)
            sec.set_option("threads_dw", threads)
            sec.set_option("sort_mode", srt)
            for kernel in (0, 1):
                sec.set_option("kernel", kernel)
                hv = sec.unpad(sec.apply_device(dv))
                torch.cuda.synchronize()
                assert _rel(hv.cpu().numpy(), ref) <= TOL, (m.name, bits, shard, cols, rows, kernel)
        sec.close()


@pytest.mark.parametrize("config", ["C3", "C4", "C4_kanamori"])
def test_full_size_slab_matches_oracle_matrices(built, config):
    """BASELINE C3 / C4 (Ns=16, Dim=1.66e8; C4 = BHZ: complex H, Norb=2, Nspin=2 so H_up != H_dw) are out of reach
    of the serial oracle in seconds; a 1/64 slab is not: its Hv is rebuilt on the host from the oracle's H_up,
    H_dw and diagonal."""
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    m = {"C3": models.hm_2dsquare(Nbath=3), "C4": models.bhz_2d(Nbath=1, Ust=0.5, Jh=0.1),
         "C4_kanamori": models.bhz_2d(Nbath=1, Ust=0.5, Jh=0.1, Jx=0.1, Jp=0.1)}[config]
    sec = hxv.HxvSector.from_model(m, 8, 8, rank=3, nranks=64)
    assert sec.stats()["real_h"] == (1 if config == "C3" else 0)
    orc = OracleSector(m, 8, 8, 3, 64)
    rng = np.random.default_rng(7)
    v = rng.standard_normal(sec.Dim) + 1j * rng.standard_normal(sec.Dim)
    ref = _slab_reference(orc, v)
    dv = torch.from_numpy(sec.to_gather_layout(v, 64)).cuda()
    for kernel in (0, 1):
        sec.set_option("kernel", kernel)
        hv = sec.unpad(sec.apply_device(dv))
        torch.cuda.synchronize()
        assert _rel(hv.cpu().numpy(), ref) <= TOL, kernel


def test_full_size_hermiticity_and_linearity(built):
    """Size-independent properties at BASELINE's full C3 size (Dim = 165 636 900): <x|H y> = conj(<y|H x>) and
    H(a x + b y) = a H x + b H y, through the tiled kernels."""
    import torch
    import hxv
    from hxv import models

    sec = hxv.HxvSector.from_model(models.hm_2dsquare(Nbath=3), 8, 8)
    g = torch.Generator(device="cuda").manual_seed(11)
    mk = lambda: torch.randn(sec.Dim, dtype=torch.float64, device="cuda", generator=g) + 1j * torch.randn(sec.Dim, dtype=torch.float64, device="cuda", generator=g)
    x, y = mk(), mk()
    hx, hy = sec.apply_device(x).clone(), sec.apply_device(y).clone()
    a = torch.vdot(x, hy).item()
    b = torch.vdot(y, hx).item()
    scale = (x.norm() * hy.norm()).item()
    assert abs(a - b.conjugate()) <= 1e-12 * scale
    al, be = 0.3 - 1.1j, -0.7 + 0.2j
    hz = sec.apply_device(al * x + be * y)
    torch.cuda.synchronize()
    assert (hz - (al * hx + be * hy)).abs().max().item() <= 1e-12 * hz.abs().max().item()
    # REAL-vector product at the same size: the real part of the complex product (to rounding: the complex product's
    # pass A runs as pipelined jobs with another summation order; bit for bit against the same kernels), and symmetric
    xr, yr = x.real.contiguous(), y.real.contiguous()
    hxr = sec.apply_device_real(xr)
    ref_r = sec.apply_device(xr.to(torch.complex128)).real
    assert (hxr - ref_r).abs().max().item() <= 1e-14 * ref_r.abs().max().item()
    sec.set_option("job_up", 0)
    assert torch.equal(hxr, sec.apply_device(xr.to(torch.complex128)).real)
    sec.set_option("job_up", 1)
    hyr = sec.apply_device_real(yr)
    assert abs(torch.dot(xr, hyr).item() - torch.dot(yr, hxr).item()) <= 1e-12 * (xr.norm() * hyr.norm()).item()


@pytest.mark.parametrize("shard", [(0, 1), (2, 3)])
def test_create_from_csr_matches_oracle(built, shard):
    """hxv_create_from_csr: the engine fed with the reference's own stored matrices (spH0ups, spH0dws, spH0d as the
    oracle builds them): stored-diagonal mode, index-chunk tiles instead of prefix blocks."""
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    rank, size = shard
    for m, (nup, ndw) in ((models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6]), (3, 2)), (models.bhz_2d(Nbath=0, Ust=0.3), (4, 3)),
                          (models.hm_2dsquare(Nbath=2), (6, 6))):
        orc = OracleSector(m, nup, ndw, rank, size)
        sec = hxv.HxvSector.from_csr(orc.DimUp, orc.DimDw, orc.csr("up"), orc.csr("dw"), orc.diag(), rank=rank, nranks=size)
        assert (sec.vecDim, sec.mpiQdw, sec.mpiIshift) == (orc.vecDim, orc.mpiQdw, orc.mpiIshift)
        v = models.deterministic_vector(sec.Dim)
        ref = _slab_reference(orc, v)
        dv = torch.from_numpy(sec.to_gather_layout(v, size)).cuda()
        for kernel, kb in ((0, 64), (1, 64), (1, 8)):
            sec.set_option("lds_budget_kb", kb)
            sec.set_option("kernel", kernel)
            hv = sec.unpad(sec.apply_device(dv))
            torch.cuda.synchronize()
            assert _rel(hv.cpu().numpy(), ref) <= TOL, (m.name, shard, kernel, kb)
        # introspection round trip: the engine hands back the matrices it was given
        rp, cols, vals = sec.csr("up")
        rp0, cols0, vals0 = orc.csr("up")
        assert np.array_equal(rp, rp0) and np.array_equal(cols, cols0) and np.array_equal(vals, vals0)
        assert np.array_equal(sec.diag(), orc.diag().real)
        sec.close()
    # with the stored spH0nd block (Jx / Jp; local rows, global columns: ED_HAMILTONIAN_SPARSE_HxV.f90:217-225)
    m, (nup, ndw) = models.bhz_2d(Nbath=0, Ust=0.7, Jh=0.2, Jx=0.2, Jp=0.15), (4, 3)
    orc = OracleSector(m, nup, ndw, rank, size)
    sec = hxv.HxvSector.from_csr(orc.DimUp, orc.DimDw, orc.csr("up"), orc.csr("dw"), orc.diag(), rank=rank, nranks=size, nd=orc.csr("nd"))
    v = models.deterministic_vector(sec.Dim)
    ref = _slab_reference(orc, v)
    dv = torch.from_numpy(sec.to_gather_layout(v, size)).cuda()
    for kernel in (0, 1):
        sec.set_option("kernel", kernel)
        assert _rel(sec.unpad(sec.apply_device(dv)).cpu().numpy(), ref) <= TOL, (shard, kernel)
    assert not sec.real_vectors_available
    with pytest.raises(hxv.HxvError, match="already has"):
        sec.set_nonlocal_csr(*orc.csr("nd"))
    if size == 1:   # the Lanczos drivers see the block too
        vn = v / np.linalg.norm(v)
        a, b, n = sec.lanczos_tridiag(torch.from_numpy(vn).cuda(), 12)
        a0, b0 = orc.lanc_tridiag(vn, 12)
        assert n == len(a0) == 12
        assert np.abs(a - a0).max() <= 1e-11 * np.abs(a0).max() and np.abs(b[1:] - b0[1:]).max() <= 1e-11 * np.abs(b0).max()
    sec.close()


def test_edcontext_mirrors_reference_interface(built):
    """build_Hv_sector / spHtimesV_p / delete_Hv_sector through the Python mirror (ED_HAMILTONIAN.f90:10-26)."""
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    m = models.hm_2dsquare(Nbath=1)
    ctx = hxv.EDContext(m)
    isector = m.get_Sector(4, 4)
    assert ctx.spHtimesV_p is None
    ctx.build_Hv_sector(isector)
    assert ctx.Hstatus and ctx.Dim == m.getDim(isector) == ctx.vecDim_Hv_sector(isector)
    with pytest.raises(hxv.HxvError):
        ctx.build_Hv_sector(isector)                 # one sector open at a time
    v = models.deterministic_vector(ctx.Dim)
    ref = OracleSector(m, 4, 4).spMatVec_main(v)
    hv = np.empty_like(v)
    ctx.spHtimesV_p(ctx.Dim, v, hv)                  # host arrays, like the reference's callers
    assert _rel(hv, ref) <= TOL
    dv = torch.from_numpy(v).cuda()
    dhv = torch.empty_like(dv)
    ctx.spHtimesV_p(ctx.Dim, dv, dhv)                # device tensors
    torch.cuda.synchronize()
    assert _rel(dhv.cpu().numpy(), ref) <= TOL
    with pytest.raises(hxv.HxvError):
        ctx.spHtimesV_p(ctx.Dim - 1, v[:-1].copy(), hv[:-1].copy())   # Nloc /= Dim
    ctx.delete_Hv_sector()
    assert ctx.spHtimesV_p is None and not ctx.Hstatus


@pytest.mark.parametrize("P", [2, 3])
def test_alltoall_halves_reproduce_the_product(built, P):
    """hxv_apply_dw_panel + hxv_apply_up_add (the two halves used by the all-to-all exchange), all 'ranks' emulated on
    one GPU: row panels X_r = v[rows U_r, all columns] -> Y_r = X_r H_dw^T -> columns back to their owners ->
    hv_slab = D.v + H_up v + Y[:, slab].  Must equal the oracle's full product."""
    import torch
    import hxv
    from hxv import models, dw_split
    from oracle.oracle import OracleSector

    for m, (nup, ndw) in ((models.hm_2dsquare(Nbath=1), (4, 3)), (models.bhz_2d(Nbath=0, Ust=0.4, Jh=0.1), (3, 5))):
        orc = OracleSector(m, nup, ndw)
        du, dd = orc.DimUp, orc.DimDw
        v = models.deterministic_vector(orc.Dim)
        ref = orc.spMatVec_main(v)
        V = torch.from_numpy(v.reshape(dd, du)).cuda()          # [column][row]
        Y = torch.zeros_like(V)                                   # (v H_dw^T) assembled from the panels
        for r in range(P):
            nr, u0 = dw_split(du, r, P)                           # row range of rank r: same rule as the column split
            pan = hxv.HxvSector.dw_panel(m, nup, ndw, nr)
            assert pan.DimUp == nr and pan.localElems == dd * pan.pitch
            x = torch.zeros(dd, pan.pitch, dtype=torch.complex128, device="cuda")
            x[:, :nr] = V[:, u0:u0 + nr]
            y = pan.apply_dw_panel(x.reshape(-1)).view(dd, pan.pitch)
            Y[:, u0:u0 + nr] = y[:, :nr]
            pan.close()
        for r in range(P):
            sec = hxv.HxvSector.from_model(m, nup, ndw, rank=r, nranks=P)
            q, c0 = sec.mpiQdw, sec.mpiIshift // du
            vl = torch.zeros(q, sec.pitch, dtype=torch.complex128, device="cuda")
            wl = torch.zeros_like(vl)
            vl[:, :du] = V[c0:c0 + q]
            wl[:, :du] = Y[c0:c0 + q]
            hv = sec.apply_up_add(vl.reshape(-1), wl.reshape(-1)).view(q, sec.pitch)[:, :du].reshape(-1)
            torch.cuda.synchronize()
            assert _rel(hv.cpu().numpy(), ref[sec.mpiIshift: sec.mpiIshift + sec.vecDim]) <= TOL, (m.name, P, r)
            sec.close()


# ---- REAL-vector mode (H real): same kernels on double elements ---------------------------------------------------
@pytest.mark.parametrize("case", ["C1", "chain", "chain_odd", "kanamori_u", "C2"])
def test_real_vector_product_equals_complex_product(built, case):
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    if case == "C1":
        m, (nup, ndw) = models.plaquette_2x2_nobath(U=4.0, t=1.0, hfmode=False), (2, 2)
    elif case == "chain":
        m, (nup, ndw) = models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6]), (3, 3)
    elif case == "chain_odd":
        m, (nup, ndw) = models.hm_1dchain(Nlat=3, Nbath=2), (4, 5)          # DimUp != DimDw, odd dimensions
    elif case == "kanamori_u":
        m, (nup, ndw) = models.hm_2dsquare(Nbath=1), (4, 3)
    else:
        m, (nup, ndw) = models.hm_1dchain(), (6, 6)
    sec = hxv.HxvSector.from_model(m, nup, ndw)
    assert sec.real_vectors_available
    rng = np.random.default_rng(11)
    x = rng.standard_normal(sec.Dim)
    ref = OracleSector(m, nup, ndw).spMatVec_main(x.astype(np.complex128))
    assert np.abs(ref.imag).max() == 0.0
    xr = torch.from_numpy(x).cuda()
    hr = sec.apply_device_real(xr).cpu().numpy()
    hc = sec.apply_device(xr.to(torch.complex128)).cpu().numpy()
    scale = np.abs(ref).max()
    assert np.abs(hr - ref.real).max() <= 1e-13 * scale
    assert np.abs(hr - hc.real).max() <= 1e-14 * scale and np.abs(hc.imag).max() == 0.0


def test_real_vector_mode_refused_for_complex_h_and_shards(built):
    import torch
    import hxv
    from hxv import models

    bhz = hxv.HxvSector.from_model(models.bhz_2d(Nbath=0), 4, 4)
    assert not bhz.real_vectors_available
    with pytest.raises(hxv.HxvError, match="complex amplitudes"):
        bhz.apply_device_real(torch.zeros(bhz.Dim, dtype=torch.float64, device="cuda"))
    # a slab of a split sector: the drivers exchange real slabs themselves (tests/test_gpu_ranks.py); the product entry that takes
    # the WHOLE real vector is for unsplit sectors only
    shard = hxv.HxvSector.from_model(models.hm_1dchain(Nlat=2, Nbath=2), 3, 3, rank=1, nranks=2)
    assert shard.real_vectors_available
    with pytest.raises(hxv.HxvError, match="unsplit"):
        shard.apply_device_real(torch.zeros(shard.Dim, dtype=torch.float64, device="cuda"))


def test_full_size_c4_complex_hermiticity_and_linearity(built):
    """BASELINE C4 at full size (BHZ 2x2 + 1 bath, Ns=16, Dim = 165 636 900, complex amplitudes, H_up != H_dw):
    <x|H y> = conj(<y|H x>) and linearity through the complex-coefficient tiled / job kernels, and the tiled
    product against the one-thread-per-element kernel."""
    import torch
    import hxv
    from hxv import models

    sec = hxv.HxvSector.from_model(models.bhz_2d(Nbath=1, Ust=0.5, Jh=0.1), 8, 8)
    assert sec.stats()["real_h"] == 0
    g = torch.Generator(device="cuda").manual_seed(23)
    mk = lambda: torch.randn(sec.Dim, dtype=torch.float64, device="cuda", generator=g) + 1j * torch.randn(sec.Dim, dtype=torch.float64, device="cuda", generator=g)
    x, y = mk(), mk()
    hx, hy = sec.apply_device(x).clone(), sec.apply_device(y).clone()
    a = torch.vdot(x, hy).item()
    b = torch.vdot(y, hx).item()
    scale = (x.norm() * hy.norm()).item()
    assert abs(a - b.conjugate()) <= 1e-12 * scale
    al, be = 0.3 - 1.1j, -0.7 + 0.2j
    hz = sec.apply_device(al * x + be * y)
    torch.cuda.synchronize()
    assert (hz - (al * hx + be * hy)).abs().max().item() <= 1e-12 * hz.abs().max().item()
    for opt in ("job_up", "kernel"):   # job kernels -> one-tile-per-workgroup kernels -> one thread per element
        sec.set_option(opt, 0)
        h2 = sec.apply_device(x)
        torch.cuda.synchronize()
        assert (h2 - hx).abs().max().item() <= 1e-13 * hx.abs().max().item(), opt
    sec.close()


def _fill_randn(v, seed):
    import torch

    vr = torch.view_as_real(v).view(-1)
    g = torch.Generator(device="cuda").manual_seed(seed)
    for a in range(0, vr.numel(), 1 << 28):
        b = min(a + (1 << 28), vr.numel())
        vr[a:b] = torch.randn(b - a, dtype=torch.float64, device="cuda", generator=g)


def _chunked(fn, n, step=1 << 27):
    out = []
    for a in range(0, n, step):
        out.append(fn(slice(a, min(a + step, n))))
    return out


def test_c5_ns18_full_size_single_gpu(built):
    """BASELINE config 5 (Ns=18, sector (9,9), Dim = 2 363 904 400, 37.8 GB per vector) on ONE GPU at full size:
    tiled/job kernels vs the one-thread-per-element kernel, Hermiticity and linearity (64-bit indexing: Dim > 2^31)."""
    import torch
    import hxv
    from hxv import models

    torch.cuda.empty_cache()
    hxv.pool_trim()  # (buffers cached by earlier sectors of this process)
    free, _ = torch.cuda.mem_get_info()
    if free < 210e9:
        pytest.skip("needs ~200 GB of free HBM")
    sec = hxv.HxvSector.from_model(models.hm_ring(6, 2), 9, 9)
    assert sec.Dim == 2363904400 and sec.get_option("nblocks_up") == 64
    n = sec.fullElems
    x = torch.empty(n, dtype=torch.complex128, device="cuda")
    y = torch.empty_like(x)
    _fill_randn(x, 5)
    _fill_randn(y, 6)
    # pad rows of the device layout (hxv.h) must not enter the dot products: zero them
    pitch, du = sec.pitch, sec.DimUp
    if pitch != du:
        x.view(-1, pitch)[:, du:] = 0
        y.view(-1, pitch)[:, du:] = 0
    hx = torch.empty_like(x)
    hy = torch.empty_like(x)
    sec.apply_device(x, hx)
    sec.apply_device(y, hy)
    torch.cuda.synchronize()
    if pitch != du:
        hx.view(-1, pitch)[:, du:] = 0
        hy.view(-1, pitch)[:, du:] = 0
    a = sum(_chunked(lambda s: torch.vdot(x[s], hy[s]).item(), n))
    b = sum(_chunked(lambda s: torch.vdot(y[s], hx[s]).item(), n))
    scale = (sum(_chunked(lambda s: (x[s].abs() ** 2).sum().item(), n)) * sum(_chunked(lambda s: (hy[s].abs() ** 2).sum().item(), n))) ** 0.5
    assert abs(a - b.conjugate()) <= 1e-12 * scale
    # one-thread-per-element kernel and the one-tile-per-workgroup kernels against the default path, into y's storage
    hmax = max(_chunked(lambda s: hx[s].abs().max().item(), n))
    for opt in ("job_up", "kernel"):
        sec.set_option(opt, 0)
        sec.apply_device(x, y)
        torch.cuda.synchronize()
        if pitch != du:
            y.view(-1, pitch)[:, du:] = 0
        assert max(_chunked(lambda s: (y[s] - hx[s]).abs().max().item(), n)) <= 1e-13 * hmax, opt
    sec.set_option("kernel", 1)
    sec.set_option("job_up", 1)
    # linearity: z = al x + be (old y is gone: use hx as the second vector) -> H z = al H x + be H hx
    al, be = 0.3 - 1.1j, -0.7 + 0.2j
    sec.apply_device(hx, y)                       # y := H hx
    _chunked(lambda s: x[s].mul_(al).add_(hx[s], alpha=be), n)   # x := z
    _chunked(lambda s: hx[s].mul_(al).add_(y[s], alpha=be), n)   # hx := al H x + be H hx
    sec.apply_device(x, y)                        # y := H z
    torch.cuda.synchronize()
    if pitch != du:
        y.view(-1, pitch)[:, du:] = 0
        hx.view(-1, pitch)[:, du:] = 0
    zmax = max(_chunked(lambda s: y[s].abs().max().item(), n))
    assert max(_chunked(lambda s: (y[s] - hx[s]).abs().max().item(), n)) <= 1e-12 * zmax
    sec.close()


def test_c5_ns18_slab_matches_oracle_matrices(built):
    """A 1/512 slab of the Ns=18 sector against the oracle's H_up, H_dw and diagonal (the pattern of
    test_full_size_slab_matches_oracle_matrices); only the columns H_dw couples to the slab are copied to the host."""
    import scipy.sparse as sp
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    torch.cuda.empty_cache()
    hxv.pool_trim()
    free, _ = torch.cuda.mem_get_info()
    if free < 60e9:
        pytest.skip("needs ~50 GB of free HBM")
    m = models.hm_ring(6, 2)
    P, rank = 512, 3
    sec = hxv.HxvSector.from_model(m, 9, 9, rank=rank, nranks=P)
    orc = OracleSector(m, 9, 9, rank, P)
    assert (sec.vecDim, sec.mpiQdw, sec.mpiIshift) == (orc.vecDim, orc.mpiQdw, orc.mpiIshift)
    du, dd, q = orc.DimUp, orc.DimDw, orc.mpiQdw
    c0 = orc.mpiIshift // du
    # the gathered vector on the device (all-gather layout: P slabs of cmax*pitch), seeded random numbers
    v = torch.empty(sec.fullElems, dtype=torch.complex128, device="cuda")
    _fill_randn(v, 18)
    hv = sec.unpad(sec.apply_device(v))
    torch.cuda.synchronize()
    # host reference from the oracle's matrices
    rp, cols, vals = orc.csr("up")
    Hup = sp.csr_matrix((vals, cols - 1, rp), shape=(du, du))
    rp, cols, vals = orc.csr("dw")
    Hdw = sp.csr_matrix((vals, cols - 1, rp), shape=(dd, dd))[c0:c0 + q, :].tocsc()
    src = np.flatnonzero(np.diff(Hdw.indptr))                 # dw columns the slab couples to (CSC: non-empty columns)
    pitch = sec.pitch
    cmax = -(-dd // P)
    base, rem = dd // P, dd % P
    def slot(c):                                             # column -> slot of the all-gather layout (hxv.h)
        r = c // (base + 1) if c < rem * (base + 1) else rem + (c - rem * (base + 1)) // base
        cfirst = r * base + min(r, rem)
        return r * cmax + (c - cfirst)
    def col(c):
        s = slot(int(c))
        x = v[s * pitch:s * pitch + du].cpu().numpy()
        if sec.row_perm is not None:                         # device row order (hxv.h): reference row i sits at device row perm[i], times sign[i]
            x = x[sec.row_perm] * sec.row_sign
        return x
    Vloc = np.stack([col(c0 + k) for k in range(q)], axis=1)  # (du, q)
    Vsrc = np.stack([col(c) for c in src], axis=1)            # (du, len(src))
    out = orc.diag().reshape((du, q), order="F") * Vloc + Hup @ Vloc + (Hdw[:, src] @ Vsrc.T).T
    ref = np.asarray(out).reshape(-1, order="F")
    assert _rel(hv.cpu().numpy(), ref) <= TOL
    sec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shard", [(0, 1), (1, 2)])
def test_spH0nd_inside_pass_a_and_as_its_own_pass_match_the_oracle(built, shard):
    """The Jx / Jp block (sparse/H_non_local.f90:23-98, ED_HAMILTONIAN_SPARSE_HxV.f90:217-225) is added inside pass A of the tiled
    product (option "fold_nd" = 1, default) or by its own pass over hv afterwards (0): both against the oracle's spH0nd CSR, for real
    and complex one-body amplitudes, every tile width, unsplit and split."""
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    rank, size = shard
    cases = ((models.bhz_2d(Nbath=0, Ust=0.7, Jh=0.2, Jx=0.2, Jp=0.15), (4, 4)),        # complex amplitudes
             (models.bhz_2d(Nbath=1, Ust=0.5, Jh=0.1, Jx=0.0, Jp=0.3), (5, 6)),         # pair hopping only, 12 orbitals
             (models.bhz_2d(Nbath=0, Ust=0.3, Jh=0.1, Jx=-0.4, Jp=0.0, lam=0.0), (3, 4)))  # spin exchange only
    for m, (nup, ndw) in cases:
        sec = hxv.HxvSector.from_model(m, nup, ndw, rank=rank, nranks=size)
        orc = OracleSector(m, nup, ndw, rank, size)
        v = models.deterministic_vector(sec.Dim)
        ref = _slab_reference(orc, v)
        dv = torch.from_numpy(sec.to_gather_layout(v, size)).cuda()
        outs = []
        for fold in (1, 0):
            for cols in (2, 4):
                sec.set_option("fold_nd", fold)
                sec.set_option("cols_per_tile", cols)
                got = sec.unpad(sec.apply_device(dv)).cpu().numpy()
                assert _rel(got, ref) <= TOL, (m.name, fold, cols)
                outs.append(got)
        assert np.abs(outs[0] - outs[2]).max() <= 1e-14 * max(1.0, np.abs(ref).max())
        sec.close()
