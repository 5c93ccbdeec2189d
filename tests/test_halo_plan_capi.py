"""The halo plan behind the C-ABI, checked without any transport (and without a GPU): hxv_halo_plan_from_csr computes, from the
one-spin matrix H_dw alone, what every rank of a DimDw split (ED_HAMILTONIAN.f90:93-105) receives from and sends to every peer.
For every pair (p, q): the columns p sends to q ARE the columns q receives from p, in the same order; what q receives is exactly
the set of foreign columns its rows of H_dw reference (sparse/H_dw.f90: the dw hops of spMatVec_MPI_main,
ED_HAMILTONIAN_SPARSE_HxV.f90:281-292); nothing is sent that is not received."""
import numpy as np
import pytest


def _dw_csr(model, nup, ndw):
    from oracle.oracle import OracleSector

    orc = OracleSector(model, nup, ndw)
    rp, cols, _ = orc.csr("dw")
    return orc.DimDw, rp, cols


@pytest.mark.parametrize("case,nranks", [("chain", 2), ("chain", 3), ("chain", 7), ("C2", 2), ("C2", 3), ("C2", 4), ("C2", 8), ("bhz", 3), ("bhz", 5)])
def test_send_lists_equal_receive_lists_for_every_pair(built, case, nranks):
    import hxv
    from hxv import models
    from hxv.distributed import dw_split, halo_plan

    model, (nup, ndw) = {"chain": (models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6]), (3, 3)),
                         "C2": (models.hm_1dchain(eps_bath=[0.3, 0.6]), (6, 6)),
                         "bhz": (models.bhz_2d(Nbath=0), (4, 4))}[case]
    dimdw, rp, cols = _dw_csr(model, nup, ndw)
    plans = [hxv.halo_plan_from_csr(dimdw, rp, cols, r, nranks) for r in range(nranks)]
    first = [dw_split(dimdw, r, nranks)[1] for r in range(nranks)] + [dimdw]
    owner = np.empty(dimdw, dtype=np.int64)
    for r in range(nranks):
        owner[first[r]:first[r + 1]] = r
    need_np, send_np = halo_plan(rp, cols - 1, dimdw, nranks)      # the numpy twin used by the gloo tests
    for q in range(nranks):
        rc_q, sc_q, rcols_q, scols_q = plans[q]
        assert rc_q[q] == 0 and sc_q[q] == 0
        # what q receives = the foreign columns its rows reference, ascending (= grouped by owner)
        mine = np.unique(cols[rp[first[q]]:rp[first[q + 1]]] - 1)
        want = mine[owner[mine] != q]
        assert np.array_equal(rcols_q, want) and np.array_equal(rcols_q, need_np[q])
        assert np.array_equal(np.bincount(owner[rcols_q], minlength=nranks), rc_q)
        off_q = np.concatenate([[0], np.cumsum(rc_q)])
        for p in range(nranks):
            if p == q:
                continue
            rc_p, sc_p, rcols_p, scols_p = plans[p]
            off_p = np.concatenate([[0], np.cumsum(sc_p)])
            sent = scols_p[off_p[q]:off_p[q + 1]]                  # global columns p sends to q, in sending order
            recv = rcols_q[off_q[p]:off_q[p + 1]]                  # global columns q expects from p, in slot order
            assert np.array_equal(sent, recv), (p, q)
            assert np.all(owner[sent] == p)
            assert np.array_equal(sent - first[p], send_np[p][q])  # the numpy twin sends the same LOCAL columns


def test_plan_refuses_bad_input(built):
    import hxv

    rp = np.array([0, 1, 2], dtype=np.int64)
    with pytest.raises(hxv.HxvError):
        hxv.halo_plan_from_csr(2, rp, np.array([0, 1], dtype=np.int32), 0, 2)     # 0-based columns: out of range
    with pytest.raises(hxv.HxvError):
        hxv.halo_plan_from_csr(2, rp, np.array([1, 2], dtype=np.int32), 2, 2)     # rank >= nranks
