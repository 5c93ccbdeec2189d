"""CPU check of the thick-restart Lanczos ALGORITHM (tests/trlan_numpy.py, the host-side restatement of
csrc/hxv_eigh.hip) against LAPACK on the oracle's sector matrices: what the engine's sp_eigh replacement must
deliver at ED_DIAG.f90:152-160 (lowest Neigen eigenpairs, E within 1e-10)."""
import numpy as np
import pytest

from helpers_matrix import oracle_full_matrix
from trlan_numpy import keep_count, start_vector, trlan_lowest


@pytest.mark.parametrize("case,neigen,ncv", [("C1", 1, 10), ("C1", 2, 20), ("C1", 4, 36), ("chain", 2, 20), ("chain", 3, 12), ("bhz", 2, 20)])
def test_trlan_lowest_vs_lapack(built, case, neigen, ncv):
    from hxv import models
    from oracle.oracle import OracleSector

    if case == "C1":
        m, (nup, ndw) = models.plaquette_2x2_nobath(U=4.0, t=1.0, hfmode=False), (2, 2)
    elif case == "chain":
        m, (nup, ndw) = models.hm_1dchain(Nlat=2, Nbath=2, eps_bath=[0.3, 0.6]), (3, 3)
    else:
        m, (nup, ndw) = models.bhz_2d(Nbath=0), (4, 4)
    orc = OracleSector(m, nup, ndw)
    H = oracle_full_matrix(orc)
    assert np.abs(H.toarray() - orc.dense()).max() == 0.0
    ref = np.linalg.eigvalsh(orc.dense())
    ev, X, nconv, nmv, nrestart = trlan_lowest(lambda v: H @ v, orc.Dim, neigen, ncv)
    assert nconv == neigen
    assert np.abs(ev - ref[:neigen]).max() < 1e-11
    assert np.abs(X.conj().T @ X - np.eye(neigen)).max() < 1e-12
    assert np.linalg.norm(H @ X - X * ev, axis=0).max() < 1e-10
    if case == "C1" and neigen == 1:
        assert abs(ev[0] - (-2.10274848)) < 5e-9          # survey-recorded reference value


def test_trlan_exhausts_tiny_space(built):
    """ncv >= Dim: the Krylov space closes (beta -> 0) and every Ritz pair is exact."""
    rng = np.random.default_rng(3)
    A = rng.standard_normal((6, 6)) + 1j * rng.standard_normal((6, 6))
    A = A + A.conj().T
    ev, X, nconv, nmv, _ = trlan_lowest(lambda v: A @ v, 6, 3, 20)
    assert np.abs(ev - np.linalg.eigvalsh(A)[:3]).max() < 1e-12 and nmv <= 6


def test_keep_count_and_start_vector():
    for m in range(2, 65):
        for ne in range(1, m):
            for nc in range(0, ne + 1):
                assert 1 <= keep_count(m, ne, nc) <= m - 1
    v = start_vector(1000)
    assert np.abs(v.real).max() <= 0.5 and np.abs(v.imag).max() <= 0.5 and abs(v.mean()) < 0.05
