"""SEVERAL ranks of a split sector on ONE GPU: the thread-rank transport of csrc/hxv_comm.cpp (hxv_comm_init_local) runs the
multi-rank code of the C-ABI for real -- slab copies into the all-gather layout, uneven splits, halo lists and offsets, the
drivers' all-reduces, the fused recurrence and the REAL-vector mode on slabs, the collective error agreement -- everything but
RCCL's own transport (RCCL refuses two ranks on one device; librccl has run with one rank only: tests/test_gpu_comm.py).
Every test runs a second time with transport "rccl_double": the handles join through hxv_comm_unique_id / hxv_comm_init, the path
one process per GPU takes, with HXV_RCCL_LIB pointing at tests/rccl_double (RCCL's ten entry points for thread ranks of one process),
so the RCCL branches of hxv_comm.cpp -- the counts, offsets and pointers of the grouped ncclSend / ncclRecv in the halo exchange, both
transposes and the ladder operators, the in-place ncclAllGather, the ncclSum all-reduces, the ncclMax error agreement -- execute
with 2-4 ranks against the same references.
Reference semantics: spMatVec_MPI_main (ED_HAMILTONIAN_SPARSE_HxV.f90:230-315), the DimDw split of ED_HAMILTONIAN.f90:93-105,
sp_lanc_tridiag / sp_lanc_eigh / sp_eigh called with MpiComm (ED_GF_NORMAL.f90:215, ED_DIAG.f90:152-156,176-177)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-13


@pytest.fixture(params=["local", "rccl_double"])
def transport(request, built, monkeypatch):
    """-> the `transport` argument of hxv.run_ranks"""
    if request.param == "local":
        return "local"
    monkeypatch.setenv("HXV_RCCL_LIB", str(built.build_rccl_double()))
    return "rccl"


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _model(name):
    from hxv import models

    if name == "chain":
        return models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6]), (3, 3)      # DimDw = 20: uneven for 3 ranks
    if name == "C2":
        return models.hm_1dchain(eps_bath=[0.3, 0.6]), (6, 6)                         # DimDw = 924
    if name == "bhz":
        return models.bhz_2d(Nbath=0), (4, 4)                                         # complex H, DimDw = 70
    if name == "kanamori":
        return models.bhz_2d(Nbath=0, Ust=0.7, Jh=0.2, Jx=0.2, Jp=0.15), (4, 4)       # + the spH0nd block (Jx, Jp)
    raise KeyError(name)


@pytest.mark.parametrize("name,nranks,exchange", [("chain", 2, "allgather"), ("chain", 3, "allgather"), ("chain", 3, "halo"), ("C2", 4, "halo"),
                                                  ("C2", 3, "allgather"), ("bhz", 3, "allgather"), ("bhz", 4, "halo"),
                                                  ("chain", 3, "alltoall"), ("C2", 4, "alltoall"), ("bhz", 3, "alltoall"), ("chain", 2, "alltoall"),
                                                  ("kanamori", 3, "allgather"), ("kanamori", 3, "halo"), ("kanamori", 4, "halo")])
def test_product_of_every_rank_through_the_exchange(built, transport, name, nranks, exchange):
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    m, (nup, ndw) = _model(name)
    orc = OracleSector(m, nup, ndw)
    v = models.deterministic_vector(orc.Dim)
    v /= np.linalg.norm(v)
    ref = orc.spMatVec_main(v)
    hxv.set_exchange_default(exchange)

    def rank(r, group):
        sec = hxv.HxvSector.from_model(m, nup, ndw, rank=r, nranks=nranks)
        assert sec.exchange_mode == exchange
        group.join(sec)
        lo, hi = sec.mpiIshift, sec.mpiIshift + sec.vecDim
        got_host = sec.apply_host(v[lo:hi])                        # spMatVec_MPI_main on this rank's slab, host arrays
        dv = sec.pad(torch.from_numpy(v[lo:hi].copy()).cuda(), sec.mpiQdw)
        got_dev = sec.unpad(sec.apply_device_slab(dv)).cpu().numpy()
        n_ex = sec.exchange_count
        if exchange != "alltoall":
            # the slab built where the exchange wants it (hxv_slab_home): same product, no slab copy
            home = sec.slab_home()
            assert home.numel() == sec.localElems
            home.copy_(dv)
            got_home = sec.unpad(sec.apply_device_slab(home)).cpu().numpy()
            assert np.array_equal(got_home, got_dev)
            assert torch.equal(home, dv)                           # (the product reads the slab, it does not change it)
        else:
            with pytest.raises(hxv.HxvError, match="no gathered vector"):
                sec.slab_home()
            # the OVERLAPPED form of the two-transposes exchange (diagonal + up hops on a second stream during the transposes, the dw
            # part added at the end -- the reference's own order, ED_HAMILTONIAN_SPARSE_HxV.f90:250-296): the same product
            sec.set_option("exchange_overlap", 1)
            for _ in range(2):   # (twice: the second product's second stream must wait for the first one's readers)
                got_ov = sec.unpad(sec.apply_device_slab(dv)).cpu().numpy()
            assert np.abs(got_ov - got_dev).max() <= 1e-14 * max(np.abs(got_dev).max(), 1e-300)
            sec.set_option("exchange_overlap", 0)
            n_ex = sec.exchange_count - 4
        sec.close()
        return lo, hi, got_host, got_dev, n_ex

    try:
        res = hxv.run_ranks(nranks, rank, transport=transport)
    finally:
        hxv.set_exchange_default("allgather")
    scale = np.abs(ref).max()
    for lo, hi, gh, gd, n_ex in res:
        assert np.abs(gh - ref[lo:hi]).max() <= TOL * scale and np.abs(gd - ref[lo:hi]).max() <= TOL * scale
        assert n_ex == (4 if exchange == "alltoall" else 2)      # (two transposes per product)


@pytest.mark.parametrize("name,nranks,exchange,real_vectors,fused", [("C2", 3, "allgather", 0, 1), ("C2", 3, "halo", 1, 1), ("C2", 2, "allgather", 1, 0),
                                                                     ("chain", 3, "halo", 0, 0), ("bhz", 3, "allgather", 0, 1),
                                                                     ("C2", 3, "alltoall", 0, 1), ("bhz", 2, "alltoall", 0, 0),
                                                                     ("C2", 4, "alltoall", 1, 1), ("chain", 3, "alltoall", 1, 0),
                                                                     ("kanamori", 3, "halo", 0, 1)])
def test_lanczos_drivers_on_split_sector_equal_the_serial_ones(built, transport, name, nranks, exchange, real_vectors, fused):
    """tridiag, eigh and eigh_lowest with slabs per rank: the same Krylov space as the unsplit sector (the start vectors hash the
    GLOBAL index), alpha/beta/E equal to rounding; every rank returns the same numbers; real vectors and the fused recurrence run
    on slabs as well."""
    import torch
    import hxv
    from hxv import models

    m, (nup, ndw) = _model(name)
    ser = hxv.HxvSector.from_model(m, nup, ndw)
    ser.set_option("real_vectors", real_vectors)
    ser.set_option("lanczos_fused", fused)
    rng = np.random.default_rng(5)
    v = rng.standard_normal(ser.Dim) + (0.0 if real_vectors else 1j * rng.standard_normal(ser.Dim))
    v = (v / np.linalg.norm(v)).astype(np.complex128)
    nl = 40 if name != "chain" else 25
    a0, b0, n0 = ser.lanczos_tridiag(torch.from_numpy(v).cuda(), nl)
    e0, vec0, _ = ser.lanczos_eigh(400, 1e-14)
    ev0, _, nc0, _ = ser.eigh_lowest(2, 16, 200, 0.0)
    want_real = bool(real_vectors) and ser.real_vectors_available
    assert ser.get_option("lanczos_real_last") == (1 if want_real else 0)
    hxv.set_exchange_default(exchange)

    def rank(r, group):
        sec = hxv.HxvSector.from_model(m, nup, ndw, rank=r, nranks=nranks)
        sec.set_option("real_vectors", real_vectors)
        sec.set_option("lanczos_fused", fused)
        group.join(sec)
        lo, hi = sec.mpiIshift, sec.mpiIshift + sec.vecDim
        a, b, n = sec.lanczos_tridiag(torch.from_numpy(v[lo:hi].copy()).cuda(), nl)
        was_real = sec.get_option("lanczos_real_last")
        # the Lanczos vectors of a split sector live in their slot of a gather buffer: no slab copy in any of the nl exchanges ...
        # (the all-to-all exchange has no gathered vector at all)
        assert sec.get_option("slab_copies") == 0 and sec.exchange_count >= n
        # ... and the same numbers, bit for bit, when they live in slab buffers and are copied before every exchange
        sec.set_option("lanczos_inplace", 0)
        a_c, b_c, n_c = sec.lanczos_tridiag(torch.from_numpy(v[lo:hi].copy()).cuda(), nl)
        assert n_c == n and np.array_equal(a_c, a) and np.array_equal(b_c, b)
        assert sec.get_option("slab_copies") >= (0 if exchange == "alltoall" else n)
        sec.set_option("lanczos_inplace", 1)
        e, vec, _ = sec.lanczos_eigh(400, 1e-14)
        ev, vecs, nc, _ = sec.eigh_lowest(2, 16, 200, 0.0)
        out = (lo, hi, a, b, n, e, vec.cpu().numpy(), ev, nc, was_real, vecs[0].cpu().numpy())
        sec.close()
        return out

    try:
        res = hxv.run_ranks(nranks, rank, transport=transport)
    finally:
        hxv.set_exchange_default("allgather")
    gs = np.zeros(ser.Dim, dtype=np.complex128)
    g2 = np.zeros(ser.Dim, dtype=np.complex128)
    for lo, hi, a, b, n, e, vec, ev, nc, was_real, v2 in res:
        # (the early steps entry by entry; later ones amplify rounding differences once Ritz values converge: there the lowest
        #  eigenvalue of the tridiagonal matrix is the meaningful comparison)
        assert n == n0 and np.abs(a[:12] - a0[:12]).max() < 1e-10 and np.abs(b[:12] - b0[:12]).max() < 1e-10
        t_split = np.linalg.eigvalsh(np.diag(a[:n]) + np.diag(b[1:n], 1) + np.diag(b[1:n], -1))[0]
        t_ser = np.linalg.eigvalsh(np.diag(a0[:n0]) + np.diag(b0[1:n0], 1) + np.diag(b0[1:n0], -1))[0]
        assert abs(t_split - t_ser) < 1e-9
        assert np.array_equal(a, res[0][2]) and np.array_equal(b, res[0][3])          # every rank holds the same numbers, bit for bit
        assert abs(e - e0) < 1e-10 and nc == nc0 == 2 and np.abs(ev - ev0).max() < 1e-9
        assert was_real == (1 if want_real else 0)
        gs[lo:hi] = vec
        g2[lo:hi] = v2
    # the slabs of the eigenvector assemble to the serial one (up to a phase), and to an eigenvector of H
    ov = abs(np.vdot(gs, vec0.cpu().numpy()))
    assert abs(ov - 1.0) < 1e-8 and abs(np.linalg.norm(gs) - 1.0) < 1e-10
    hg = ser.apply_device(torch.from_numpy(g2).cuda()).cpu().numpy()
    assert np.linalg.norm(hg - ev0[0] * g2) < 1e-7
    ser.close()


def test_a_failing_rank_stops_all_ranks_before_the_collective(built, transport):
    """comm_agree: one rank is asked for more HBM than the device has; every rank returns an error instead of waiting inside an
    all-reduce for a peer that has already left (rank-local early exits)."""
    import hxv
    from hxv import models

    m, (nup, ndw) = _model("chain")

    def rank(r, group):
        sec = hxv.HxvSector.from_model(m, nup, ndw, rank=r, nranks=2)
        group.join(sec)
        try:
            # ncv beyond the engine's limit is an argument error on every rank: harmless.  The collective one: neigen > Dim on nobody,
            # but a bad ncv only on rank 1 -> rank 0 must not hang
            sec.eigh_lowest(2, 16 if r == 0 else 9999, 50, 0.0)
            return "ok"
        except hxv.HxvError as e:
            return str(e)
        finally:
            sec.close()

    import threading

    done = []
    t = threading.Thread(target=lambda: done.append(hxv.run_ranks(2, rank, transport=transport)))
    t.start()
    t.join(60)
    assert not t.is_alive(), "a rank is still waiting for its peer: the failure was not made collective"
    assert all(x != "ok" for x in done[0]), done


@pytest.mark.parametrize("nranks", [2, 3])
@pytest.mark.parametrize("spin,create", [(0, True), (0, False), (1, True), (1, False)])
def test_ladder_operators_on_split_sectors(built, transport, nranks, spin, create):
    """c / c^dagger on the slabs of a split sector (the reference: master-only loop + scatter, ED_GF_NORMAL.f90:174-214): every rank
    builds its slab of the new vector; assembled, it equals the serial device ladder, norm included; mixed channels accumulate
    ((c^dagger_i + xi c^dagger_j)|gs>, ED_GF_NORMAL.f90:746-780)."""
    import torch
    import hxv
    from hxv import models

    m = models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6])   # Ns = 6
    nup, ndw = 3, 3
    d = 1 if create else -1
    tnup, tndw = (nup + d, ndw) if spin == 0 else (nup, ndw + d)
    sa = hxv.HxvSector.from_model(m, nup, ndw)
    sb = hxv.HxvSector.from_model(m, tnup, tndw)
    rng = np.random.default_rng(11)
    psi = rng.standard_normal(sa.Dim) + 1j * rng.standard_normal(sa.Dim)
    psi /= np.linalg.norm(psi)
    dpsi = torch.from_numpy(psi).cuda()
    orb_i, orb_j, xi = 1, 4, 0.3 - 0.8j
    ref1, n1 = sa.apply_ladder(sb, orb_i, spin, create, dpsi)
    acc = sb.pad(ref1.clone())
    ref2, n2 = sa.apply_ladder(sb, orb_j, spin, create, dpsi, coef=xi, out=acc)
    ref1 = ref1.cpu().numpy()
    ref2 = sb.unpad(ref2).cpu().numpy()

    def rank(r, group):
        fa = hxv.HxvSector.from_model(m, nup, ndw, rank=r, nranks=nranks)
        fb = hxv.HxvSector.from_model(m, tnup, tndw, rank=r, nranks=nranks)
        group.join(fb)
        slab = fa.pad(torch.from_numpy(psi[fa.mpiIshift: fa.mpiIshift + fa.vecDim].copy()).cuda(), fa.mpiQdw)
        o1, m1 = fa.apply_ladder(fb, orb_i, spin, create, slab)
        got1 = fb.unpad(o1).cpu().numpy()
        o2, m2 = fa.apply_ladder(fb, orb_j, spin, create, slab, coef=xi, out=o1)
        got2 = fb.unpad(o2).cpu().numpy()
        lo, hi = fb.mpiIshift, fb.mpiIshift + fb.vecDim
        fa.close()
        fb.close()
        return lo, hi, got1, m1, got2, m2

    res = hxv.run_ranks(nranks, rank, transport=transport)
    for lo, hi, g1, m1, g2, m2 in res:
        assert np.abs(g1 - ref1[lo:hi]).max() < 1e-14 and np.abs(g2 - ref2[lo:hi]).max() < 1e-14
        assert abs(m1 - n1) < 1e-13 and abs(m2 - n2) < 1e-13      # the GLOBAL norms, on every rank
    sa.close()
    sb.close()


@pytest.mark.parametrize("exchange", ["allgather", "halo", "alltoall"])
def test_sectors_opened_from_stored_matrices_on_three_ranks(built, transport, exchange):
    """hxv_create_from_csr handles (the reference's own spH0ups / spH0dws / spH0d, ED_VARS_GLOBAL.f90:142-144) of a split sector with each
    of the three exchanges: every rank's slab of the product against the oracle; a stored spH0nd block needs the all-gather."""
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    m, (nup, ndw) = models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6]), (3, 2)
    nranks = 3
    full = OracleSector(m, nup, ndw)
    v = models.deterministic_vector(full.Dim)
    ref = full.spMatVec_main(v)
    hxv.set_exchange_default(exchange)

    def rank(r, group):
        orc = OracleSector(m, nup, ndw, r, nranks)
        sec = hxv.HxvSector.from_csr(orc.DimUp, orc.DimDw, orc.csr("up"), orc.csr("dw"), orc.diag(), rank=r, nranks=nranks)
        assert sec.exchange_mode == exchange
        group.join(sec)
        lo, hi = sec.mpiIshift, sec.mpiIshift + sec.vecDim
        got = sec.unpad(sec.apply_device_slab(sec.pad(torch.from_numpy(v[lo:hi].copy()).cuda(), sec.mpiQdw))).cpu().numpy()
        refused = None
        if exchange != "allgather":
            try:
                sec.set_nonlocal_csr(np.zeros(sec.vecDim + 1, dtype=np.int64), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.complex128))
                refused = False
            except hxv.HxvError:
                refused = True
        sec.close()
        return lo, hi, got, refused

    try:
        res = hxv.run_ranks(nranks, rank, transport=transport)
    finally:
        hxv.set_exchange_default("allgather")
    for lo, hi, got, refused in res:
        assert np.abs(got - ref[lo:hi]).max() <= TOL * np.abs(ref).max()
        assert refused in (None, True)


@pytest.mark.parametrize("nranks,exchange", [(3, "allgather"), (2, "alltoall"), (4, "halo")])
def test_paired_tridiagonalisation_on_a_split_sector(built, transport, nranks, exchange):
    """hxv_lanczos_tridiag_pair with slabs per rank: two Green's-function channels on one product of a split sector -- the same numbers
    as the serial paired run to rounding, identical on every rank."""
    import torch
    import hxv

    m, (nup, ndw) = _model("C2")
    ser = hxv.HxvSector.from_model(m, nup, ndw)
    rng = np.random.default_rng(11)
    va = rng.standard_normal(ser.Dim).astype(np.complex128)
    vb = rng.standard_normal(ser.Dim).astype(np.complex128)
    va /= np.linalg.norm(va)
    vb *= 2.5 / np.linalg.norm(vb)          # (not normalised: the driver carries the norm)
    nl = 30
    (a0, b0, n0), (a1, b1, n1) = ser.lanczos_tridiag_pair(torch.from_numpy(va).cuda(), torch.from_numpy(vb).cuda(), nl)
    ser.close()
    hxv.set_exchange_default(exchange)

    def rank(r, group):
        sec = hxv.HxvSector.from_model(m, nup, ndw, rank=r, nranks=nranks)
        group.join(sec)
        lo, hi = sec.mpiIshift, sec.mpiIshift + sec.vecDim
        out = sec.lanczos_tridiag_pair(torch.from_numpy(va[lo:hi].copy()).cuda(), torch.from_numpy(vb[lo:hi].copy()).cuda(), nl)
        sec.close()
        return out

    try:
        res = hxv.run_ranks(nranks, rank, transport=transport)
    finally:
        hxv.set_exchange_default("allgather")
    for (a, b, n), (a2, b2, n2) in res:
        assert n == n0 and n2 == n1
        assert np.abs(a[:12] - a0[:12]).max() < 1e-10 and np.abs(b[:12] - b0[:12]).max() < 1e-10
        assert np.abs(a2[:12] - a1[:12]).max() < 1e-10 and np.abs(b2[:12] - b1[:12]).max() < 1e-10
        assert np.array_equal(a, res[0][0][0]) and np.array_equal(b2, res[0][1][1])      # every rank holds the same numbers


def test_a_rank_thread_that_fails_outside_the_library_wakes_its_peers(built):
    """a Python error in one rank's thread (here: before it joins) must not leave the peers inside cv.wait forever --
    hxv.run_ranks aborts the group (hxv_comm_local_abort), the peers' collective returns HXV_ERR_STATE, the ORIGINAL error is raised."""
    import threading
    import hxv

    m, (nup, ndw) = _model("chain")
    seen = []

    def rank(r, group):
        sec = hxv.HxvSector.from_model(m, nup, ndw, rank=r, nranks=3)
        try:
            if r == 1:
                raise RuntimeError("rank 1 failed before joining")
            try:
                group.join(sec)
            except hxv.HxvError as e:
                seen.append(str(e))
                raise
        finally:
            sec.close()

    out = []

    def runner():
        try:
            hxv.run_ranks(3, rank)
        except BaseException as e:  # noqa: BLE001
            out.append(e)

    t = threading.Thread(target=runner)
    t.start()
    t.join(60)
    assert not t.is_alive(), "the peers of the failed rank are still waiting"
    assert len(out) == 1 and isinstance(out[0], RuntimeError) and "rank 1 failed" in str(out[0]), out
    assert len(seen) == 2 and all("dropped out" in s for s in seen), seen


def test_a_rank_lost_after_joining_an_rccl_communicator_is_aborted(built):
    """over RCCL a rank whose thread fails BETWEEN collectives used to leave its peers inside ncclAllReduce until the library's
    own timeout (the double: 120 s; librccl: forever).  hxv.run_ranks now calls hxv_comm_abort (ncclCommAbort) on every joined handle:
    the peer's collective returns an error at once, the original error is raised."""
    import os
    import threading
    import time
    import hxv

    os.environ["HXV_RCCL_LIB"] = str(built.build_rccl_double())
    try:
        m, (nup, ndw) = _model("chain")
        seen, took = [], []

        def rank(r, group):
            sec = hxv.HxvSector.from_model(m, nup, ndw, rank=r, nranks=2)
            try:
                group.join(sec)
                assert sec.comm_library.endswith("librccl_double.so")
                if r == 1:
                    time.sleep(0.5)                                  # (rank 0 is inside its first all-reduce by now)
                    raise RuntimeError("rank 1 failed after joining")
                t0 = time.time()
                try:
                    sec.lanczos_eigh(64, 1e-10, want_vector=False)   # collective: waits for rank 1
                except hxv.HxvError as e:
                    seen.append(str(e))
                    took.append(time.time() - t0)
                    raise
            finally:
                sec.close()

        out = []

        def runner():
            try:
                hxv.run_ranks(2, rank, transport="rccl")
            except BaseException as e:  # noqa: BLE001
                out.append(e)

        t = threading.Thread(target=runner)
        t.start()
        t.join(60)
        assert not t.is_alive(), "the peer of the failed rank is still waiting"
        assert len(out) == 1 and isinstance(out[0], RuntimeError) and "rank 1 failed" in str(out[0]), out
        assert len(seen) == 1 and took[0] < 30.0, (seen, took)
    finally:
        os.environ.pop("HXV_RCCL_LIB", None)


@pytest.mark.parametrize("exchange,real_vectors", [("allgather", 0), ("halo", 1)])
def test_start_vector_built_at_the_slab_home(built, transport, exchange, real_vectors):
    """the drivers clear the slab's place in their three gather buffers before they read their input; a start vector that
    was built exactly there (hxv_slab_home) is staged first instead of being turned into zeros."""
    import torch
    import hxv

    m, (nup, ndw) = _model("C2")
    ser = hxv.HxvSector.from_model(m, nup, ndw)
    ser.set_option("real_vectors", real_vectors)
    rng = np.random.default_rng(3)
    v = rng.standard_normal(ser.Dim) + (0.0 if real_vectors else 1j * rng.standard_normal(ser.Dim))
    v = (v / np.linalg.norm(v)).astype(np.complex128)
    a0, b0, n0 = ser.lanczos_tridiag(torch.from_numpy(v).cuda(), 30)
    ser.close()
    hxv.set_exchange_default(exchange)

    def rank(r, group):
        sec = hxv.HxvSector.from_model(m, nup, ndw, rank=r, nranks=3)
        sec.set_option("real_vectors", real_vectors)
        group.join(sec)
        lo, hi = sec.mpiIshift, sec.mpiIshift + sec.vecDim
        home = sec.slab_home()
        home.copy_(sec.pad(torch.from_numpy(v[lo:hi].copy()).cuda(), sec.mpiQdw))
        a, b, n = sec.lanczos_tridiag(home, 30)
        (aa, ba, na), (ab, bb, nb) = ((a, b, n), (a, b, n))
        if real_vectors:  # the paired driver reads two start vectors: both at home (refilled: nothing left there survives a driver call)
            home.copy_(sec.pad(torch.from_numpy(v[lo:hi].copy()).cuda(), sec.mpiQdw))
            (aa, ba, na), (ab, bb, nb) = sec.lanczos_tridiag_pair(home, home, 30)
        sec.close()
        return a, b, n, aa, ba, ab, bb

    try:
        res = hxv.run_ranks(3, rank, transport=transport)
    finally:
        hxv.set_exchange_default("allgather")
    for a, b, n, aa, ba, ab, bb in res:
        assert n == n0
        for x, y in ((a, a0), (b, b0), (aa, a0), (ba, b0), (ab, a0), (bb, b0)):
            assert np.abs(x - y).max() <= 1e-11 * max(np.abs(y).max(), 1.0)


def _double_counters(built):
    import ctypes

    lib = ctypes.CDLL(str(built.build_rccl_double()))
    lib.rccl_double_comm_inits.restype = ctypes.c_longlong
    lib.rccl_double_comm_destroys.restype = ctypes.c_longlong
    return lib


def test_one_communicator_per_process_serves_a_sweep_of_sectors(built, monkeypatch):
    """VERDICT r5 item 2.  The reference sets MpiComm once per solve (ED_VARS_GLOBAL.f90:365-380) while its callers open a sector around every
    Lanczos run (ED_DIAG.f90:142-190: 289 per sweep; ED_GF_NORMAL.f90:208-222: 56 per Green's-function stage).  Two ranks through the RCCL
    branches (tests/rccl_double): 20 sectors opened -> joined -> product + sp_lanc_eigh -> closed, then 8 Green's-function channels (ground
    state kept, c^dagger into the N+1 sector opened and closed around each tridiagonalisation).  ONE ncclCommInitRank per rank (the double
    counts them) although hxv_comm_init is called 29 times per rank; with HXV_COMM_CACHE=0 every hxv_comm_init builds one, and every number
    of the sweep -- products, E0, alanc / blanc -- is bit-identical between the two."""
    import threading
    import torch
    import hxv
    from hxv import models

    monkeypatch.setenv("HXV_RCCL_LIB", str(built.build_rccl_double()))
    dbl = _double_counters(built)
    m = models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6])   # Ns = 6
    sectors = [(nu, nd) for nu in range(1, 6) for nd in range(1, 6)][:20]
    chans = [(orb, spin) for spin in (0, 1) for orb in (0, 1, 3, 4)]
    nranks = 2
    nid = len(sectors) + 1 + len(chans)

    def sweep():
        ids = [hxv.HxvSector.comm_unique_id() for _ in range(nid)]   # what rank 0 would draw and MPI_Bcast, one per hxv_comm_init
        bar = threading.Barrier(nranks)

        def rank(r, group):
            out = []
            k = 0
            for nup, ndw in sectors:
                sec = hxv.HxvSector.from_model(m, nup, ndw, rank=r, nranks=nranks)
                sec.comm_init(ids[k])
                k += 1
                v = models.deterministic_vector(sec.vecDim, offset=sec.mpiIshift)
                hv = sec.apply_device_slab(sec.pad(torch.from_numpy(v).cuda(), sec.mpiQdw))
                e0, _, nit = sec.lanczos_eigh(64, 1e-12, want_vector=False, native=True) if sec.Dim > 4 else (0.0, None, 0)
                out.append((sec.unpad(hv).cpu().numpy(), e0, nit))
                sec.close()
                bar.wait()
            gs = hxv.HxvSector.from_model(m, 3, 3, rank=r, nranks=nranks)
            gs.comm_init(ids[k])
            k += 1
            e0, psi, _ = gs.lanczos_eigh(128, 1e-13, native=True)
            out.append(e0)
            for orb, spin in chans:
                tgt = hxv.HxvSector.from_model(m, 3 + (spin == 0), 3 + (spin == 1), rank=r, nranks=nranks)
                tgt.comm_init(ids[k])
                k += 1
                vv, n2 = gs.apply_ladder(tgt, orb, spin, True, psi)
                a, b, n = tgt.lanczos_tridiag(vv, 12)
                out.append((n2, a.copy(), b.copy(), n))
                tgt.close()
                bar.wait()
            gs.close()
            return out

        return hxv.run_ranks(nranks, rank, transport="rccl")

    assert hxv.comm_cache_clear() >= 0                      # (communicators earlier tests left behind)
    lh0 = hxv.live_handles()
    st0, i0, d0 = hxv.comm_cache_stats(), dbl.rccl_double_comm_inits(), dbl.rccl_double_comm_destroys()
    cached = sweep()
    st1, i1 = hxv.comm_cache_stats(), dbl.rccl_double_comm_inits()
    assert i1 - i0 == nranks, (i0, i1)                      # ONE ncclCommInitRank per rank for 29 hxv_comm_init calls each
    assert st1["inits"] - st0["inits"] == nranks and st1["reuses"] - st0["reuses"] == nranks * (nid - 1) and st1["entries"] == nranks
    assert hxv.comm_cache_clear() == nranks and dbl.rccl_double_comm_destroys() - d0 == nranks and hxv.comm_cache_stats()["entries"] == 0
    monkeypatch.setenv("HXV_COMM_CACHE", "0")
    i2 = dbl.rccl_double_comm_inits()
    fresh = sweep()
    assert dbl.rccl_double_comm_inits() - i2 == nranks * nid          # the behaviour before round 6: one communicator per opened sector
    assert hxv.comm_cache_stats()["entries"] == 0
    for ra, rb in zip(cached, fresh):
        for xa, xb in zip(ra, rb):
            if isinstance(xa, tuple):
                for ya, yb in zip(xa, xb):
                    assert np.array_equal(np.asarray(ya), np.asarray(yb))
            else:
                assert xa == xb
    assert hxv.live_handles() == lh0                        # every sector of the two sweeps was closed
