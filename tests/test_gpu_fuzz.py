"""Seeded random parity sweep: random Hermitian one-body matrices (spin-diagonal, as the N_up (x) N_dw path requires),
random bath levels / hybridisations / interaction constants, random sectors, random rank splits and random tile options,
HIP product (tiled kernels, C-ABI) vs the CPU oracle.  Shakes out corner cases of the plan builder (tiny blocks, empty
in-block lists, one-column shards, odd dimensions) that the hand-picked cases do not hit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _random_model(rng):
    from hxv.models import Model

    Nlat, Norb = [(1, 1), (2, 1), (3, 1), (2, 2), (1, 2), (4, 1), (1, 3)][rng.integers(7)]
    Nspin = int(rng.integers(1, 3))
    Nbath = int(rng.integers(0, 3))
    while Nlat * Norb * (Nbath + 1) > 9:
        Nbath -= 1
    cplx = rng.random() < 0.5

    def herm_block():  # Hermitian in the (lat,orb) index, spin-diagonal
        n = Nlat * Norb
        A = rng.standard_normal((n, n)) * (rng.random((n, n)) < 0.6)
        if cplx:
            A = A + 1j * rng.standard_normal((n, n)) * (rng.random((n, n)) < 0.4)
        A = (A + A.conj().T) / 2
        return A

    def to6(blocks):  # (Nlat,Nlat,Nspin,Nspin,Norb,Norb), index (ilat,iorb) -> iorb + ilat*Norb  (ED_SETUP.f90 imp_state_index)
        h = np.zeros((Nlat, Nlat, Nspin, Nspin, Norb, Norb), dtype=np.complex128)
        for s in range(Nspin):
            A = blocks[s]
            for il in range(Nlat):
                for jl in range(Nlat):
                    for io in range(Norb):
                        for jo in range(Norb):
                            h[il, jl, s, s, io, jo] = A[io + il * Norb, jo + jl * Norb]
        return h

    hloc = to6([herm_block() for _ in range(Nspin)])
    B = max(Nbath, 1)
    hb = np.zeros((Nlat, Nlat, Nspin, Nspin, Norb, Norb, B), dtype=np.complex128)
    vb = np.zeros((Nlat, Nspin, Norb, B))
    for ib in range(Nbath):
        hb[..., ib] = to6([herm_block() for _ in range(Nspin)])
        vb[..., ib] = rng.standard_normal((Nlat, Nspin, Norb)) * (rng.random((Nlat, Nspin, Norb)) < 0.8)
    U = np.zeros(5)
    U[:Norb] = rng.random(Norb) * 3
    multi = Norb > 1
    return Model(Nlat, Norb, Nspin, Nbath, hloc, hb[..., :B] if Nbath else hb[..., :0].reshape(Nlat, Nlat, Nspin, Nspin, Norb, Norb, 0),
                 vb[..., :B] if Nbath else vb[..., :0], Uloc=U, Ust=float(rng.random()) if multi else 0.0, Jh=float(rng.random() * 0.5) if multi else 0.0,
                 Jx=float(rng.random() * 0.4) if multi and rng.random() < 0.4 else 0.0, Jp=float(rng.random() * 0.4) if multi and rng.random() < 0.4 else 0.0,
                 xmu=float(rng.standard_normal() * 0.3), hfmode=bool(rng.integers(2)), name="fuzz")


@pytest.mark.parametrize("seed", range(64))
def test_random_models_sectors_shards_and_tile_options(built, seed):
    import torch
    import hxv
    from oracle.oracle import OracleSector, spMatVec_mpi_main

    rng = np.random.default_rng(1000 + seed)
    m = _random_model(rng)
    Ns = m.Ns
    if rng.random() < 0.6:   # mostly sectors near half filling (the big ones), sometimes anything incl. empty / full
        nup, ndw = int(np.clip(Ns // 2 + rng.integers(-1, 2), 0, Ns)), int(np.clip(Ns // 2 + rng.integers(-1, 2), 0, Ns))
    else:
        nup, ndw = int(rng.integers(0, Ns + 1)), int(rng.integers(0, Ns + 1))
    full = OracleSector(m, nup, ndw)
    v = rng.standard_normal(full.Dim) + 1j * rng.standard_normal(full.Dim)
    ref = full.spMatVec_main(v)
    scale = max(np.abs(ref).max(), 1e-300)
    size = int(rng.integers(1, min(4, full.DimDw) + 1))
    for rank in range(size):
        sec = hxv.HxvSector.from_model(m, nup, ndw, rank=rank, nranks=size)
        assert (sec.Dim, sec.DimUp, sec.DimDw) == (full.Dim, full.DimUp, full.DimDw)
        dv = torch.from_numpy(sec.to_gather_layout(v, size)).cuda()
        want = ref[sec.mpiIshift: sec.mpiIshift + sec.vecDim]
        opts = [{}, {"lds_budget_kb": int(rng.choice([8, 16, 32])), "cols_per_tile": int(rng.choice([2, 4, 8])), "rows_per_tile": int(rng.choice([2, 4, 8])),
                     "threads_up": int(rng.choice([256, 512, 1024])), "threads_dw": int(rng.choice([256, 512, 1024])), "sort_mode": int(rng.integers(3)),
                     "wt_cols": int(rng.choice([2, 4, 8, 16])), "job_cols": int(rng.choice([1, 2])), "pair_rows": int(rng.choice([0, 1])),
                     "job_groups": int(rng.choice([1, 3, 100])), "job_max_blocks": int(rng.choice([0, 32])), "block_order": int(rng.choice([-1, 0, 1, 2]))},
                {"kernel": 0}]
        for o in opts:
            try:
                for k, val in o.items():
                    sec.set_option(k, val)
            except hxv.HxvError as e:      # an option set the plan cannot honour is refused loudly, never silently wrong
                assert "block larger" in str(e) or "does not fit" in str(e) or "must be" in str(e), str(e)
                continue
            got = sec.unpad(sec.apply_device(dv)).cpu().numpy() if sec.vecDim else np.zeros(0, complex)
            assert got.shape == want.shape
            if want.size:
                assert np.abs(got - want).max() <= 2e-13 * scale, (seed, rank, size, o, m.Nlat, m.Norb, m.Nspin, m.Nbath, nup, ndw)
        if size == 1 and sec.real_vectors_available:     # REAL-vector product on the real part
            sec.set_option("kernel", 1)
            hr = sec.apply_device_real(torch.from_numpy(np.ascontiguousarray(v.real)).cuda()).cpu().numpy()
            want_r = full.spMatVec_main(v.real.astype(np.complex128))
            assert np.abs(want_r.imag).max() == 0.0 and np.abs(hr - want_r.real).max() <= 2e-13 * scale
        if size == 1 and 12 <= full.Dim <= 3100:       # the Lanczos drivers on the same random operator
            sec.set_option("kernel", 1)
            w = np.linalg.eigvalsh(full.dense())
            e0 = sec.lanczos_eigh(600, 1e-13, want_vector=False)[0]
            assert abs(e0 - w[0]) <= 1e-9 * max(1.0, abs(w[0])), (seed, e0, w[0])
            ev, X, nconv, _ = sec.eigh_lowest(1, 16)
            assert abs(ev[0] - w[0]) <= 1e-9 * max(1.0, abs(w[0]))
            x = X[0].cpu().numpy()
            assert np.linalg.norm(full.dense() @ x - ev[0] * x) <= 1e-7 * max(1.0, np.abs(w).max())
        sec.close()
    if size > 1:   # the oracle's own MPI emulation agrees with its serial product (keeps the checker honest)
        chk, _ = spMatVec_mpi_main(m, nup, ndw, size, v)
        assert np.abs(chk - ref).max() <= 1e-13 * scale


@pytest.mark.parametrize("shape,sector,seed", [((4, 1, 2, 2), (6, 6), 1), ((2, 2, 1, 2), (6, 5), 2), ((3, 1, 2, 3), (6, 7), 3), ((2, 3, 1, 1), (5, 6), 4),
                                                ((1, 2, 2, 5), (6, 6), 5), ((2, 2, 2, 2), (7, 6), 6)])
def test_random_ns12_models_multi_block_plans(built, shape, sector, seed):
    """Ns = 12 with random (dense-ish, partly complex) one-body matrices and Kanamori terms: several prefix blocks per spin,
    many distinct amplitudes (LDS coefficient tables, the > 255 amplitudes fallback), shards -- vs the oracle's product."""
    import torch
    import hxv
    from hxv.models import Model
    from oracle.oracle import OracleSector

    rng = np.random.default_rng(7000 + seed)
    Nlat, Norb, Nspin, Nbath = shape
    n = Nlat * Norb

    def blocks():
        out = []
        for _ in range(Nspin):
            A = rng.standard_normal((n, n)) * (rng.random((n, n)) < 0.7)
            if seed % 2:
                A = A + 1j * rng.standard_normal((n, n)) * (rng.random((n, n)) < 0.5)
            out.append((A + A.conj().T) / 2)
        return out

    def to6(bl):
        h = np.zeros((Nlat, Nlat, Nspin, Nspin, Norb, Norb), dtype=np.complex128)
        for s in range(Nspin):
            for il in range(Nlat):
                for jl in range(Nlat):
                    for io in range(Norb):
                        for jo in range(Norb):
                            h[il, jl, s, s, io, jo] = bl[s][io + il * Norb, jo + jl * Norb]
        return h

    hb = np.stack([to6(blocks()) for _ in range(Nbath)], axis=-1)
    vb = rng.standard_normal((Nlat, Nspin, Norb, Nbath))
    U = np.zeros(5)
    U[:Norb] = 1.0 + rng.random(Norb)
    m = Model(Nlat, Norb, Nspin, Nbath, to6(blocks()), hb, vb, Uloc=U, Ust=0.7 if Norb > 1 else 0.0, Jh=0.2 if Norb > 1 else 0.0,
              Jx=0.15 if Norb > 1 and seed in (2, 6) else 0.0, Jp=0.1 if Norb > 1 and seed == 6 else 0.0, xmu=0.1, hfmode=bool(seed % 2), name="fuzz12")
    assert m.Ns == 12
    nup, ndw = sector
    full = OracleSector(m, nup, ndw)
    v = rng.standard_normal(full.Dim) + 1j * rng.standard_normal(full.Dim)
    ref = full.spMatVec_main(v)
    scale = np.abs(ref).max()
    for rank, size in ((0, 1), (1, 3)):
        sec = hxv.HxvSector.from_model(m, nup, ndw, rank=rank, nranks=size)
        dv = torch.from_numpy(sec.to_gather_layout(v, size)).cuda()
        want = ref[sec.mpiIshift: sec.mpiIshift + sec.vecDim]
        for o in ({}, {"lds_budget_kb": 16, "block_order": 0}, {"lds_budget_kb": 8, "cols_per_tile": 2, "rows_per_tile": 8, "block_order": 1}, {"block_order": 2},
                  {"kernel": 0}):
            for k, val in o.items():
                sec.set_option(k, val)
            got = sec.unpad(sec.apply_device(dv)).cpu().numpy()
            assert np.abs(got - want).max() <= 2e-13 * scale, (o, rank, size, sec.stats())
        sec.close()


def test_from_csr_with_more_distinct_amplitudes_than_the_lds_table_holds(built):
    """hxv_create_from_csr with random matrix values: > 255 distinct |amplitudes| per spin -> the tiled kernels' LDS
    coefficient table is not used and the engine falls back to its one-thread-per-element GPU kernel (never to the CPU);
    > 1023 is refused loudly."""
    import scipy.sparse as sp
    import torch
    import hxv

    rng = np.random.default_rng(77)
    du, dd = 150, 130

    def rand_herm(n, nnz_per_row):
        A = sp.random(n, n, density=nnz_per_row / n, random_state=rng, data_rvs=rng.standard_normal).tocsr()
        A = A + 1j * sp.random(n, n, density=nnz_per_row / n, random_state=rng, data_rvs=rng.standard_normal).tocsr()
        H = (A + A.conj().T).tocsr()
        H.setdiag(0)
        H.eliminate_zeros()
        H.sort_indices()
        return H

    Hup, Hdw = rand_herm(du, 3), rand_herm(dd, 3)
    assert len(np.unique(np.round(np.abs(Hup.data), 12))) > 255
    diag = rng.standard_normal(du * dd)
    csr = lambda H: (H.indptr.astype(np.int64), (H.indices + 1).astype(np.int32), H.data.astype(np.complex128))
    sec = hxv.HxvSector.from_csr(du, dd, csr(Hup), csr(Hdw), diag)
    v = rng.standard_normal(du * dd) + 1j * rng.standard_normal(du * dd)
    V = v.reshape(dd, du).T                       # V[iup, idw]
    ref = (diag.reshape(dd, du).T * V + Hup @ V + (Hdw @ V.T).T).T.reshape(-1)
    got = sec.apply_device(torch.from_numpy(v).cuda()).cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()
    e0 = sec.lanczos_eigh(600, 1e-12, want_vector=False)[0]
    Hfull = sp.diags(diag) + sp.kron(sp.identity(dd), Hup) + sp.kron(Hdw, sp.identity(du))
    import scipy.sparse.linalg as sla
    assert abs(e0 - sla.eigsh(Hfull.tocsr(), k=1, which="SA", tol=1e-12)[0][0]) < 1e-8
    big = rand_herm(700, 4)
    if len(np.unique(np.round(np.abs(big.data), 12))) > 1023:
        with pytest.raises(hxv.HxvError):
            hxv.HxvSector.from_csr(700, dd, csr(big), csr(Hdw), rng.standard_normal(700 * dd))
