"""N>1 path on CPU: world_size 2 and 3 with the gloo backend.  The exchange (equal-count all-gather of
padded slabs, the DimDw split, the gather layout) is the product code under test; the per-rank slab
product is a CPU stand-in built from the oracle's matrices (the HIP kernels need a GPU)."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nup, ndw, out):
    import torch
    import torch.distributed as dist
    import scipy.sparse as sp
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6])
    orc = OracleSector(m, nup, ndw, rank, world)
    du, dd = orc.DimUp, orc.DimDw
    rp, cols, vals = orc.csr("up")
    Hup = sp.csr_matrix((vals, cols - 1, rp), shape=(du, du))
    rp, cols, vals = orc.csr("dw")
    Hdw = sp.csr_matrix((vals, cols - 1, rp), shape=(dd, dd))

    sh = None

    def apply_local(v_gathered, hv_local):  # CPU stand-in for HxvSector.apply_device (same contract)
        V = sh.unpad(v_gathered).numpy().reshape((du, dd), order="F")
        c0 = orc.mpiIshift // du
        sl = slice(c0, c0 + orc.mpiQdw)
        res = orc.diag().reshape((du, orc.mpiQdw), order="F") * V[:, sl] + Hup @ V[:, sl] + (Hdw[sl, :] @ V.T).T
        hv_local.copy_(torch.from_numpy(np.asarray(res).reshape(-1, order="F")))
        return hv_local

    sh = hxv.ShardedHxv(du, dd, rank, world, apply_local)
    assert (sh.qdw, sh.dw0 * du, sh.Nloc) == (orc.mpiQdw, orc.mpiIshift, orc.vecDim)
    v_full = models.deterministic_vector(orc.Dim)
    v_local = torch.from_numpy(v_full[orc.mpiIshift: orc.mpiIshift + orc.vecDim].copy())
    hv_local = torch.empty(orc.vecDim, dtype=torch.complex128)
    sh(sh.Nloc, v_local, hv_local)
    # every rank must also see the same gathered vector
    g = sh.unpad(sh.gather(v_local)).numpy()
    assert np.array_equal(g, v_full)
    np.save(os.path.join(out, f"hv_{rank}.npy"), hv_local.numpy())
    with pytest.raises(ValueError):
        sh(sh.Nloc + 1, v_local, hv_local)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,sector", [(2, (3, 3)), (3, (3, 3)), (3, (2, 4)), (2, (3, 2))])
def test_sharded_product_gloo(world, sector, tmp_path):
    import torch.multiprocessing as mp
    from hxv import models
    from oracle.oracle import OracleSector

    nup, ndw = sector
    mp.spawn(_worker, args=(world, _free_port(), nup, ndw, str(tmp_path)), nprocs=world, join=True)
    m = models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6])
    s = OracleSector(m, nup, ndw)
    ref = s.spMatVec_main(models.deterministic_vector(s.Dim))
    got = np.concatenate([np.load(tmp_path / f"hv_{r}.npy") for r in range(world)])
    assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()


def _worker_a2a(rank, world, port, nup, ndw, out):
    import torch
    import torch.distributed as dist
    import scipy.sparse as sp
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6])
    orc = OracleSector(m, nup, ndw, rank, world)
    du, dd = orc.DimUp, orc.DimDw
    rp, cols, vals = orc.csr("up")
    Hup = sp.csr_matrix((vals, cols - 1, rp), shape=(du, du))
    rp, cols, vals = orc.csr("dw")
    Hdw = sp.csr_matrix((vals, cols - 1, rp), shape=(dd, dd))
    c0 = orc.mpiIshift // du

    def apply_panel(x):      # CPU stand-in for HxvSector.apply_dw_panel: Y = X H_dw^T on [DimDw x nrows]
        X = x.numpy().reshape(dd, -1)
        return torch.from_numpy(np.ascontiguousarray(Hdw @ X).reshape(-1))

    def apply_up_add(v_local, w, hv_local):   # CPU stand-in for HxvSector.apply_up_add
        Vl = v_local.numpy().reshape(orc.mpiQdw, du)
        res = orc.diag().reshape(orc.mpiQdw, du) * Vl + (Hup @ Vl.T).T + w.numpy().reshape(orc.mpiQdw, du)
        hv_local.copy_(torch.from_numpy(np.ascontiguousarray(res).reshape(-1)))
        return hv_local

    th = hxv.TransposedHxv(du, dd, rank, world, apply_panel, apply_up_add)
    v_full = models.deterministic_vector(orc.Dim)
    v_local = torch.from_numpy(v_full[orc.mpiIshift: orc.mpiIshift + orc.vecDim].copy())
    hv_local = torch.empty(orc.vecDim, dtype=torch.complex128)
    th(th.Nloc, v_local, hv_local)
    np.save(os.path.join(out, f"hv_{rank}.npy"), hv_local.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,sector", [(2, (3, 3)), (3, (3, 2)), (3, (2, 4))])
def test_transposed_exchange_gloo(world, sector, tmp_path):
    """The all-to-all exchange (the reference's own scheme) with uneven row and column splits."""
    import torch.multiprocessing as mp
    from hxv import models
    from oracle.oracle import OracleSector

    nup, ndw = sector
    mp.spawn(_worker_a2a, args=(world, _free_port(), nup, ndw, str(tmp_path)), nprocs=world, join=True)
    m = models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6])
    s = OracleSector(m, nup, ndw)
    ref = s.spMatVec_main(models.deterministic_vector(s.Dim))
    got = np.concatenate([np.load(tmp_path / f"hv_{r}.npy") for r in range(world)])
    assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()


def _worker_lanczos(rank, world, port, nup, ndw, out):
    import torch
    import torch.distributed as dist
    import scipy.sparse as sp
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6])
    orc = OracleSector(m, nup, ndw, rank, world)
    du, dd = orc.DimUp, orc.DimDw
    rp, cols, vals = orc.csr("up")
    Hup = sp.csr_matrix((vals, cols - 1, rp), shape=(du, du))
    rp, cols, vals = orc.csr("dw")
    Hdw = sp.csr_matrix((vals, cols - 1, rp), shape=(dd, dd))
    sh = None

    def apply_local(v_gathered, hv_local):
        V = sh.unpad(v_gathered).numpy().reshape((du, dd), order="F")
        c0 = orc.mpiIshift // du
        sl = slice(c0, c0 + orc.mpiQdw)
        res = orc.diag().reshape((du, orc.mpiQdw), order="F") * V[:, sl] + Hup @ V[:, sl] + (Hdw[sl, :] @ V.T).T
        hv_local.copy_(torch.from_numpy(np.asarray(res).reshape(-1, order="F")))
        return hv_local

    sh = hxv.ShardedHxv(du, dd, rank, world, apply_local)
    lz = hxv.ShardedLanczos(sh)
    v_full = models.deterministic_vector(orc.Dim)
    v_full /= np.linalg.norm(v_full)
    v_local = torch.from_numpy(v_full[orc.mpiIshift: orc.mpiIshift + orc.vecDim].copy())
    a, b, n = lz.tridiag(v_local, 25)
    e0, vec, nit = lz.eigh(300, 1e-13)
    ev, X, nconv, nmv = hxv.sharded_eigh_lowest(sh, 3, 16)
    np.savez(os.path.join(out, f"lz_{rank}.npz"), a=a, b=b, n=n, e0=e0, vec=vec.numpy(), nit=nit, start=lz.start_slab().numpy(),
             ev=ev, X=X.numpy(), nconv=nconv, nmv=nmv)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,sector", [(2, (3, 3)), (3, (3, 2))])
def test_sharded_lanczos_gloo(world, sector, tmp_path):
    """sp_lanc_tridiag / sp_lanc_eigh with a communicator (MpiStatus=T): slabs of the Lanczos vectors per rank, all-reduced
    dots, one exchange per product -- against the serial oracle recurrence and LAPACK."""
    import torch.multiprocessing as mp
    from hxv import models
    from oracle.oracle import OracleSector
    from trlan_numpy import start_vector

    nup, ndw = sector
    mp.spawn(_worker_lanczos, args=(world, _free_port(), nup, ndw, str(tmp_path)), nprocs=world, join=True)
    m = models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6])
    s = OracleSector(m, nup, ndw)
    v = models.deterministic_vector(s.Dim)
    v /= np.linalg.norm(v)
    a_ref, b_ref = s.lanc_tridiag(v, 25)
    res = [np.load(tmp_path / f"lz_{r}.npz") for r in range(world)]
    for r in res:                                           # every rank holds the same scalars
        assert int(r["n"]) == 25
        assert np.abs(r["a"] - a_ref).max() <= 1e-10 * np.abs(a_ref).max() and np.abs(r["b"] - b_ref).max() <= 1e-10 * np.abs(b_ref).max()
        assert r["b"][0] == 0.0
    H = s.dense()
    w = np.linalg.eigvalsh(H)
    assert abs(float(res[0]["e0"]) - w[0]) <= 1e-10
    x = np.concatenate([r["vec"] for r in res])
    assert abs(np.linalg.norm(x) - 1) < 1e-12 and np.linalg.norm(H @ x - float(res[0]["e0"]) * x) < 1e-8
    # sp_eigh with a communicator: three lowest pairs, same algorithm and start vector as the single-GPU hxv_eigh_lowest
    from trlan_numpy import trlan_lowest
    import scipy.sparse as sp
    ev_np, _, _, nmv_np, _ = trlan_lowest(lambda y: H @ y, s.Dim, 3, 16)
    for r in res:
        assert int(r["nconv"]) == 3 and np.abs(r["ev"] - w[:3]).max() < 1e-10 and np.abs(r["ev"] - ev_np).max() < 1e-11
        assert abs(int(r["nmv"]) - nmv_np) <= 0.25 * nmv_np
    Xf = np.concatenate([r["X"] for r in res], axis=1).T          # (Dim, 3)
    assert np.abs(Xf.conj().T @ Xf - np.eye(3)).max() < 1e-11 and np.linalg.norm(H @ Xf - Xf * res[0]["ev"], axis=0).max() < 1e-9
    # the slabs of the start vector are the slabs of the single-GPU driver's deterministic start vector
    assert np.array_equal(np.concatenate([r["start"] for r in res]), start_vector(s.Dim))


def _worker_halo(rank, world, port, nup, ndw, out):
    import torch
    import torch.distributed as dist
    import scipy.sparse as sp
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6])
    orc = OracleSector(m, nup, ndw, rank, world)
    du, dd, q = orc.DimUp, orc.DimDw, orc.mpiQdw
    rp, cols, vals = orc.csr("up")
    Hup = sp.csr_matrix((vals, cols - 1, rp), shape=(du, du))
    rp, cols, vals = orc.csr("dw")
    Hdw = sp.csr_matrix((vals, cols - 1, rp), shape=(dd, dd))
    need, send = hxv.halo_plan(rp, cols - 1, dd, world)
    c0 = orc.mpiIshift // du
    have = np.concatenate([np.arange(c0, c0 + q), need[rank]])          # global column of every slot of the halo layout

    def apply_local(v_halo, hv_local):  # CPU stand-in for HxvSector.apply_device of a halo-mode handle (same contract)
        Vh = v_halo.numpy().reshape(len(have), du)                      # [slot][row]
        sub = Hdw[c0:c0 + q, :][:, have]                                # every referenced column must be present
        assert abs(Hdw[c0:c0 + q, :]).sum() == abs(sub).sum()
        res = orc.diag().reshape(q, du) * Vh[:q] + (Hup @ Vh[:q].T).T + sub @ Vh
        hv_local.copy_(torch.from_numpy(np.ascontiguousarray(res).reshape(-1)))
        return hv_local

    hx = hxv.HaloHxv(du, dd, rank, world, need, send, apply_local)
    assert (hx.qdw, hx.dw0 * du, hx.Nloc) == (q, orc.mpiIshift, orc.vecDim)
    v_full = models.deterministic_vector(orc.Dim)
    v_local = torch.from_numpy(v_full[orc.mpiIshift: orc.mpiIshift + orc.vecDim].copy())
    hv_local = torch.empty(orc.vecDim, dtype=torch.complex128)
    hx(hx.Nloc, v_local, hv_local)
    # the halo buffer holds exactly the columns it claims to
    got = hx.exchange(v_local).numpy().reshape(len(have), du)
    assert np.array_equal(got, v_full.reshape(dd, du)[have])
    assert hx.ingest_columns < (world - 1) * (-(-dd // world)) or world == 2
    np.save(os.path.join(out, f"hv_{rank}.npy"), hv_local.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,sector", [(2, (3, 3)), (4, (3, 3)), (4, (2, 4)), (3, (3, 2))])
def test_halo_exchange_gloo(world, sector, tmp_path):
    """The halo exchange (only the columns H_dw couples across ranks travel) at world sizes 2, 3 and 4 against the oracle."""
    import torch.multiprocessing as mp
    from hxv import models
    from oracle.oracle import OracleSector

    nup, ndw = sector
    mp.spawn(_worker_halo, args=(world, _free_port(), nup, ndw, str(tmp_path)), nprocs=world, join=True)
    m = models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6])
    s = OracleSector(m, nup, ndw)
    ref = s.spMatVec_main(models.deterministic_vector(s.Dim))
    got = np.concatenate([np.load(tmp_path / f"hv_{r}.npy") for r in range(world)])
    assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()


def test_halo_plan_volume_c3():
    """Exchange volume of the three schemes for BASELINE C3 at 8 ranks (what bench.py reports as exchange_ingest_bytes_per_gpu)."""
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    orc = OracleSector(models.hm_2dsquare(Nbath=3), 8, 8)
    rp, cols, _ = orc.csr("dw")
    need, send = hxv.halo_plan(rp, cols - 1, orc.DimDw, 8)
    b = hxv.exchange_ingest_bytes(orc.DimUp, orc.DimDw, 8, need)
    assert b["halo"] < 0.62 * b["allgather"] and b["alltoall"] < b["halo"]
    # what a rank sends is what the others need from it
    for r in range(8):
        for p in range(8):
            if p != r:
                q, c0 = hxv.dw_split(orc.DimDw, r, 8)
                assert np.array_equal(np.sort(send[r][p] + c0), need[p][(need[p] >= c0) & (need[p] < c0 + q)])
