"""DEVICE ROW ORDER (round 6; include/hxv.h): a sector opened from a model may store the rows of its device vectors -- the up index -- in the
order the up configurations take when the orbitals are renumbered, with the sign the reordered creation operators give every basis vector,
so that pass A's out-of-block gathers are contiguous runs.  Everything the reference sees keeps ITS order.  Checked against the CPU oracle
(the reference's order throughout) on sectors small enough for it, with the test hooks that switch the order on below its size threshold:
the product through host arrays and through device vectors, both kernels, the Lanczos drivers, the ladder operators between two sectors,
the spH0nd block, real vectors, the introspection calls; and that the order is a relabelling (a permutation, +-1 signs, H_dev = S P H P^T S)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-13


@pytest.fixture
def row_order(monkeypatch):
    """switch the device row order on for small sectors: 8 block bits at Ns = 12 (16 blocks)"""
    import hxv

    monkeypatch.setenv("HXV_ROW_ORDER_MIN_DIMUP", "16")
    monkeypatch.setenv("HXV_ROW_ORDER_BITS", "8")
    hxv.sector_cache_clear()
    yield 8
    hxv.sector_cache_clear()


def _open(m, nup, ndw, bits, **kw):
    import hxv

    sec = hxv.HxvSector.from_model(m, nup, ndw, **kw)
    sec.set_option("tile_bits_up", bits)
    return sec


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("name", ["C2", "star", "bhz", "kanamori"])
def test_product_in_device_row_order_matches_the_oracle(built, row_order, name):
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    bits = row_order
    if name == "C2":
        m, (nup, ndw) = models.hm_1dchain(eps_bath=[0.3, -0.2]), (6, 6)
    elif name == "star":
        m, (nup, ndw) = models.hm_2dsquare(Nbath=2, xmu=0.2), (5, 7)            # Ns = 12: the C3 geometry with two replicas
    elif name == "bhz":
        m, (nup, ndw) = models.bhz_2d(Nbath=0, Ust=0.4, Jh=0.1), (4, 4)
        bits = 5
    else:
        m, (nup, ndw) = models.bhz_2d(Nbath=0, Ust=0.7, Jh=0.2, Jx=0.2, Jp=0.15), (3, 5)
        bits = 5
    if bits != 8:
        import os
        os.environ["HXV_ROW_ORDER_BITS"] = str(bits)
        hxv.sector_cache_clear()
    sec = _open(m, nup, ndw, bits)
    orc = OracleSector(m, nup, ndw)
    if name in ("C2", "star"):
        assert sec.row_perm is not None, "the hook did not switch the row order on"
    if sec.row_perm is not None:
        # a relabelling: a permutation of the rows and a sign per basis vector
        assert sorted(sec.row_perm.tolist()) == list(range(sec.DimUp)) and set(np.unique(sec.row_sign).tolist()) <= {-1, 1}
        assert not np.array_equal(sec.row_perm, np.arange(sec.DimUp))
    # what the reference sees keeps its order: maps, stored matrices, diagonal
    mu, md = sec.maps()
    assert np.array_equal(mu, orc.map_up()) and np.array_equal(md, orc.map_dw())
    for which in ("up", "dw"):
        rp, cols, vals = sec.csr(which)
        rpo, colso, valso = orc.csr(which)
        assert np.array_equal(rp, rpo) and np.array_equal(cols, colso) and np.array_equal(vals, valso)
    assert np.abs(sec.diag() - np.real(orc.diag())).max() < 1e-12
    v = models.deterministic_vector(sec.Dim)
    v /= np.linalg.norm(v)
    ref = orc.spMatVec_main(v)
    assert _rel(sec.apply_host(v), ref) < TOL                                  # host arrays: converted at the boundary
    dv = sec.pad(torch.from_numpy(v).cuda())
    for kern in (1, 0):
        sec.set_option("kernel", kern)
        hv = sec.apply_device(dv)
        assert _rel(sec.unpad(hv).cpu().numpy(), ref) < TOL, kern              # device vectors: pad / unpad carry the order and the signs
    sec.set_option("kernel", 1)
    # library-owned vectors and the host <-> device copies
    back = sec.vector_to_host(sec.vector_from_host(v))
    assert np.array_equal(back, v)
    if sec.row_perm is not None:
        raw = sec.vector_from_host(v).view(-1, sec.pitch).cpu().numpy()
        V = v.reshape(sec.DimDw, sec.DimUp)
        assert np.array_equal(raw[:, sec.row_perm], V * sec.row_sign)           # d_vec[k*pitch + perm[i]] = sign[i] * v[k*DimUp + i]
    # Lanczos drivers: every returned pair is an eigenpair of the ORACLE's operator (residual through its product, in the reference's order)
    e0, vec, _ = sec.lanczos_eigh(600, 1e-13)
    x = vec.cpu().numpy()
    assert abs(np.linalg.norm(x) - 1.0) < 1e-10 and np.linalg.norm(orc.spMatVec_main(x) - e0 * x) < 1e-6
    ev, X, nc, _ = sec.eigh_lowest(2, 20)
    assert nc == 2 and abs(ev[0] - e0) < 1e-9
    for k in range(2):
        xk = X[k].cpu().numpy()
        assert np.linalg.norm(orc.spMatVec_main(xk) - ev[k] * xk) < 1e-6
    e0h, xh, _ = sec.lanczos_eigh_host(600, 1e-13)
    assert abs(e0h - e0) < 1e-10 and np.linalg.norm(orc.spMatVec_main(xh) - e0h * xh) < 1e-6
    # the host-array forms of the other two drivers (what the Fortran glue's gpu_sp_eigh / gpu_sp_lanc_tridiag_pair forward to): eig_basis(:, k)
    # comes back in the reference's order, the two real start vectors go in in it
    evh, Xh, nch, _ = sec.eigh_lowest_host(2, 20)
    assert nch == 2 and np.abs(evh - ev).max() < 1e-9
    for k in range(2):
        xk = np.ascontiguousarray(Xh[:, k])          # eig_basis(:, k), Fortran shape (Dim, neigen)
        assert np.linalg.norm(orc.spMatVec_main(xk) - evh[k] * xk) < 1e-6
    if sec.real_vectors_available:
        va = np.real(v).astype(np.complex128)
        vb = np.real(np.roll(v, 7)).astype(np.complex128)
        (aa, ba, na), (ab, bb, nb) = sec.lanczos_tridiag_pair_host(va / np.linalg.norm(va), vb / np.linalg.norm(vb), 16)
        a_o, b_o = orc.lanc_tridiag(va / np.linalg.norm(va), 16)
        a_p, b_p = orc.lanc_tridiag(vb / np.linalg.norm(vb), 16)
        assert np.abs(aa[:8] - a_o[:8]).max() < 1e-10 and np.abs(ab[:8] - a_p[:8]).max() < 1e-10
        assert np.abs(ba[:8] - b_o[:8]).max() < 1e-10 and np.abs(bb[:8] - b_p[:8]).max() < 1e-10
    a, b, n = sec.lanczos_tridiag(sec.pad(torch.from_numpy(v).cuda()), 20)
    ao, bo = orc.lanc_tridiag(v, 20)
    assert np.abs(a[:10] - ao[:10]).max() < 1e-10 and np.abs(b[:10] - bo[:10]).max() < 1e-10
    ah, bh, _ = sec.lanczos_tridiag_host(v, 20)
    assert np.abs(ah[:10] - ao[:10]).max() < 1e-10
    if sec.real_vectors_available:
        xr = np.real(v).copy()
        hr = sec.apply_device_real(torch.from_numpy(xr).cuda())
        assert _rel(hr.cpu().numpy(), np.real(orc.spMatVec_main(xr.astype(np.complex128)))) < TOL
    sec.close()


@pytest.mark.parametrize("spin,create", [(0, True), (0, False), (1, True), (1, False)])
def test_ladder_operators_between_sectors_in_device_row_order(built, row_order, spin, create):
    """c / c^dagger between two sectors that both store their rows in a device order (ED_GF_NORMAL.f90:180-199): through pad / unpad the result
    is the reference's -- checked against the same operator with the row order switched off, and against the definition on the maps."""
    import os
    import torch
    import hxv
    from hxv import models

    m = models.hm_2dsquare(Nbath=2)                     # Ns = 12
    nup, ndw = 6, 5
    d = 1 if create else -1
    tnup, tndw = (nup + d, ndw) if spin == 0 else (nup, ndw + d)
    sa, sb = _open(m, nup, ndw, row_order), _open(m, tnup, tndw, row_order)
    assert sa.row_perm is not None and sb.row_perm is not None
    rng = np.random.default_rng(3)
    psi = rng.standard_normal(sa.Dim) + 1j * rng.standard_normal(sa.Dim)
    orb = 2
    out, n2 = sa.apply_ladder(sb, orb, spin, create, torch.from_numpy(psi).cuda())
    got = out.cpu().numpy()
    # the definition (c / cdg, ED_SETUP.f90:807-833) on the reference's maps
    mua, mda = sa.maps()
    mub, mdb = sb.maps()
    ia = {int(x): k for k, x in enumerate(mua if spin == 0 else mda)}
    exp = np.zeros(sb.Dim, dtype=np.complex128)
    P = psi.reshape(sa.DimDw, sa.DimUp)
    E = exp.reshape(sb.DimDw, sb.DimUp)
    bit = 1 << orb
    for k, mt in enumerate((mub if spin == 0 else mdb).tolist()):
        if bool(mt & bit) != create:
            continue
        mf = mt ^ bit
        sg = -1.0 if bin(mf & (bit - 1)).count("1") & 1 else 1.0
        if spin == 0:
            E[:, k] = sg * P[:, ia[mf]]
        else:
            E[k, :] = sg * P[ia[mf], :]
    assert np.abs(got - exp).max() < 1e-14 and abs(n2 - np.vdot(exp, exp).real) < 1e-12 * n2
    sa.close()
    sb.close()


def test_row_order_off_is_the_reference_order_and_the_same_numbers(built, row_order, monkeypatch):
    """HXV_ROW_ORDER=0: rows in the reference's order (hxv_row_order reports identity); products and Lanczos coefficients of the two orders
    agree to rounding (another summation order inside pass A's tiles, nothing else)."""
    import torch
    import hxv
    from hxv import models

    m, (nup, ndw) = models.hm_2dsquare(Nbath=2), (6, 6)
    v = models.deterministic_vector(924 * 924)
    v /= np.linalg.norm(v)
    on = _open(m, nup, ndw, row_order)
    assert on.row_perm is not None
    h_on = on.unpad(on.apply_device(on.pad(torch.from_numpy(v).cuda()))).cpu().numpy()
    a_on, b_on, _ = on.lanczos_tridiag(on.pad(torch.from_numpy(v).cuda()), 30)
    e_on, _, _ = on.lanczos_eigh(300, 1e-13, want_vector=False)
    on.close()
    monkeypatch.setenv("HXV_ROW_ORDER", "0")
    off = _open(m, nup, ndw, row_order)
    assert off.row_perm is None
    h_off = off.unpad(off.apply_device(off.pad(torch.from_numpy(v).cuda()))).cpu().numpy()
    a_off, b_off, _ = off.lanczos_tridiag(off.pad(torch.from_numpy(v).cuda()), 30)
    e_off, _, _ = off.lanczos_eigh(300, 1e-13, want_vector=False)
    off.close()
    assert _rel(h_on, h_off) < TOL and np.abs(a_on[:12] - a_off[:12]).max() < 1e-10 and np.abs(b_on[:12] - b_off[:12]).max() < 1e-10
    assert abs(e_on - e_off) < 1e-10


def test_headline_sector_takes_the_row_order_by_itself(built):
    """C3, sector (8,8): the default plan has 16 prefix blocks of 12 low orbitals, high = replica 3; its four cluster partners move to bits
    8..11 (the hook-free path).  Plan statistics say what the order buys: as many out-of-block entries, fewer row-slot gathers."""
    import hxv
    from hxv import models

    hxv.sector_cache_clear()
    sec = hxv.HxvSector.from_model(models.hm_2dsquare(Nbath=3), 8, 8)
    assert sec.row_perm is not None and sec.get_option("tile_bits_up") == 12 and sec.get_option("nblocks_up") == 16
    assert sec.get_option("bh_up_x100") == 213 and sec.get_option("rs_up_x100") == 400
    sec.close()
    hxv.sector_cache_clear()


@pytest.mark.parametrize("bits", [3, 5, 7, 10])
@pytest.mark.parametrize("nup,ndw", [(4, 6), (7, 5)])
def test_row_order_for_other_block_sizes_and_fillings(built, monkeypatch, bits, nup, ndw):
    """The order depends on which orbitals are high (the block bits) and on the filling: products through both vector surfaces and one ladder
    operator against the oracle / the definition for a spread of them (Ns = 12 star geometry, Nspin = 1; complex amplitudes via a twisted bath)."""
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    monkeypatch.setenv("HXV_ROW_ORDER_MIN_DIMUP", "16")
    monkeypatch.setenv("HXV_ROW_ORDER_BITS", str(bits))
    hxv.sector_cache_clear()
    m = models.hm_2dsquare(Nbath=2, xmu=-0.3)
    m.Hbath[0, 1, 0, 0, 0, 0, 1] *= np.exp(0.4j)       # a complex hop and its conjugate in replica 2: complex H, Hermitian
    m.Hbath[1, 0, 0, 0, 0, 0, 1] *= np.exp(-0.4j)
    sec = _open(m, nup, ndw, bits)
    orc = OracleSector(m, nup, ndw)
    v = models.deterministic_vector(sec.Dim)
    v /= np.linalg.norm(v)
    ref = orc.spMatVec_main(v)
    assert _rel(sec.apply_host(v), ref) < TOL
    assert _rel(sec.unpad(sec.apply_device(sec.pad(torch.from_numpy(v).cuda()))).cpu().numpy(), ref) < TOL
    if sec.row_perm is not None:
        assert sorted(sec.row_perm.tolist()) == list(range(sec.DimUp))
    # c_{2,up} into the sector below: through pad / unpad the reference's result
    tgt = _open(m, nup - 1, ndw, bits)
    # (native form on both sides: where DimUp is a multiple of 8 -- (7,5): 792 -- a contiguous vector has the length of a device vector and
    #  the mirror's convenience form cannot tell them apart; pad / unpad say which is meant)
    out = torch.zeros(tgt.localElems, dtype=torch.complex128, device="cuda")
    out, n2 = sec.apply_ladder(tgt, 2, 0, False, sec.pad(torch.from_numpy(v).cuda()), out=out)
    got = tgt.unpad(out).cpu().numpy()
    mua, _ = sec.maps()
    mub, _ = tgt.maps()
    ia = {int(x): k for k, x in enumerate(mua)}
    V = v.reshape(sec.DimDw, sec.DimUp)
    E = np.zeros((tgt.DimDw, tgt.DimUp), dtype=np.complex128)
    for k, mt in enumerate(mub.tolist()):
        if mt & 4:
            continue
        mf = mt | 4
        E[:, k] = (-1.0 if bin(mf & 3).count("1") & 1 else 1.0) * V[:, ia[mf]]
    assert np.abs(got - E.reshape(-1)).max() < 1e-14
    sec.close()
    tgt.close()
    hxv.sector_cache_clear()


@pytest.mark.parametrize("exchange", ["allgather", "halo", "alltoall"])
def test_row_order_on_split_sectors(built, row_order, exchange, monkeypatch):
    """The device row order with the DimDw split (three thread ranks through the RCCL branches -- the process's communicator serves the three
    sectors a rank holds at once --, all three exchanges): rows are permuted the same way on every rank, columns
    keep the reference's order and split (ED_HAMILTONIAN.f90:93-105).  Every rank's slab of the product against the oracle's, one Lanczos
    run against the unsplit sector, and both ladder operators (up: local; dw: one column exchange) against the unsplit result."""
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    monkeypatch.setenv("HXV_RCCL_LIB", str(built.build_rccl_double()))
    m, (nup, ndw), P = models.hm_2dsquare(Nbath=2, xmu=0.1), (6, 5), 3
    orc = OracleSector(m, nup, ndw)
    v = models.deterministic_vector(orc.Dim)
    v /= np.linalg.norm(v)
    ref = orc.spMatVec_main(v)
    full = _open(m, nup, ndw, row_order)
    assert full.row_perm is not None
    a_ref, b_ref, _ = full.lanczos_tridiag(full.pad(torch.from_numpy(v).cuda()), 12)
    tup, tdw = _open(m, nup + 1, ndw, row_order), _open(m, nup, ndw + 1, row_order)
    lad_up = tup.unpad(full.apply_ladder(tup, 1, 0, True, full.pad(torch.from_numpy(v).cuda()), out=torch.zeros(tup.localElems, dtype=torch.complex128, device="cuda"))[0]).cpu().numpy()
    lad_dw = tdw.unpad(full.apply_ladder(tdw, 3, 1, True, full.pad(torch.from_numpy(v).cuda()), out=torch.zeros(tdw.localElems, dtype=torch.complex128, device="cuda"))[0]).cpu().numpy()
    for s in (full, tup, tdw):
        s.close()

    def rank(r, group):
        hxv.set_exchange_default(exchange)
        try:
            sec = hxv.HxvSector.from_model(m, nup, ndw, rank=r, nranks=P)
            su = hxv.HxvSector.from_model(m, nup + 1, ndw, rank=r, nranks=P)
            sd = hxv.HxvSector.from_model(m, nup, ndw + 1, rank=r, nranks=P)
        finally:
            hxv.set_exchange_default("allgather")
        for s in (sec, su, sd):
            s.set_option("tile_bits_up", row_order)
            group.join(s)
        assert sec.row_perm is not None
        lo, hi = sec.mpiIshift, sec.mpiIshift + sec.vecDim
        slab = sec.pad(torch.from_numpy(v[lo:hi].copy()).cuda(), sec.mpiQdw)
        hv = sec.unpad(sec.apply_device_slab(slab)).cpu().numpy()
        a, b, _ = sec.lanczos_tridiag(slab, 12)
        ou = su.unpad(sec.apply_ladder(su, 1, 0, True, slab, out=torch.zeros(su.localElems, dtype=torch.complex128, device="cuda"))[0]).cpu().numpy()
        od = sd.unpad(sec.apply_ladder(sd, 3, 1, True, slab, out=torch.zeros(sd.localElems, dtype=torch.complex128, device="cuda"))[0]).cpu().numpy()
        out = (lo, hi, hv, a, b, (su.mpiIshift, su.vecDim, ou), (sd.mpiIshift, sd.vecDim, od))
        for s in (sec, su, sd):
            s.close()
        return out

    for lo, hi, hv, a, b, (ulo, un, ou), (dlo, dn, od) in hxv.run_ranks(P, rank, transport="rccl"):
        assert _rel(hv, ref[lo:hi]) < TOL
        assert np.abs(a[:8] - a_ref[:8]).max() < 1e-10 and np.abs(b[:8] - b_ref[:8]).max() < 1e-10
        assert np.abs(ou - lad_up[ulo:ulo + un]).max() < 1e-14 and np.abs(od - lad_dw[dlo:dlo + dn]).max() < 1e-14
