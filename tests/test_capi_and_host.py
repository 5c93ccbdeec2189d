"""CPU-side checks: the C-ABI library loads and exports every symbol include/hxv.h declares, fails loudly
without a GPU (no CPU fallback), and the host-side mirror of the reference interface does its bookkeeping
like the reference (DimDw split, vecDim, sector ids)."""
import ctypes
import re
from math import comb
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol(built):
    import hxv

    hdr = (ROOT / "include" / "hxv.h").read_text()
    declared = sorted(set(re.findall(r"\b(hxv_[a-z_0-9]+)\s*\(", hdr)))
    assert set(declared) == set(hxv.EXPORTS), (set(declared) ^ set(hxv.EXPORTS))
    lib = ctypes.CDLL(str(hxv.LIB_PATH))
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in hxv.load_library().hxv_version()


def test_no_cpu_fallback(built):
    import torch
    import hxv
    from hxv import models

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hxv.HxvError, match="no HIP device|no CPU fallback"):
        hxv.HxvSector.from_model(models.plaquette_2x2_nobath(), 2, 2)


def test_bad_arguments_are_rejected_before_touching_the_device(built):
    import hxv
    from hxv import models

    m = models.plaquette_2x2_nobath()
    with pytest.raises(hxv.HxvError, match="nup/ndw"):
        hxv.HxvSector.from_model(m, 9, 2)
    with pytest.raises(hxv.HxvError, match="rank"):
        hxv.HxvSector.from_model(m, 2, 2, rank=3, nranks=2)
    with pytest.raises(hxv.HxvError, match="Norb"):
        hxv.HxvSector.from_model(models.Model(1, 6, 1, 0, np.zeros((1, 1, 1, 1, 6, 6)), np.zeros((1, 1, 1, 1, 6, 6, 0)), np.zeros((1, 1, 6, 0))), 1, 1)


@pytest.mark.parametrize("DimDw,P", [(6, 1), (6, 4), (924, 8), (12870, 8), (12870, 7), (48620, 8), (5, 5)])
def test_dw_split_matches_reference_rule(DimDw, P):
    """ED_HAMILTONIAN.f90:93-105, restated independently in the oracle."""
    from hxv import dw_split

    tot = 0
    for r in range(P):
        q, c0 = dw_split(DimDw, r, P)
        Q = DimDw // P
        R = DimDw % P
        if r < R:
            R, Q = 0, Q + 1
        ishift_cols = r * Q + R          # mpiIshift / DimUp
        assert (q, c0) == (Q, ishift_cols)
        tot += q
    assert tot == DimDw


def test_split_matches_oracle_sector():
    from hxv import dw_split, models
    from oracle.oracle import OracleSector

    m = models.hm_1dchain(Nlat=2, Nbath=2)
    for P in (1, 2, 3, 7):
        for r in range(P):
            s = OracleSector(m, 3, 2, r, P)
            q, c0 = dw_split(s.DimDw, r, P)
            assert (s.mpiQdw, s.mpiIshift, s.vecDim) == (q, c0 * s.DimUp, q * s.DimUp)


def test_edcontext_bookkeeping():
    import hxv
    from hxv import models

    m = models.hm_2dsquare(Nbath=3)
    assert (m.Nimp, m.Ns, m.Nsectors) == (4, 16, 289)          # ED_SETUP.f90:111-120
    isec = m.get_Sector(8, 8)
    assert isec == 1 + 8 + 8 * 17 and (m.get_Nup(isec), m.get_Ndw(isec)) == (8, 8)
    assert m.getDim(isec) == 12870**2
    ctx = hxv.EDContext(m, MpiRank=5, MpiSize=8)
    assert ctx.vecDim_Hv_sector(isec) == 12870 * 1609          # 12870 = 8*1608 + 6 -> ranks 0..5 own 1609
    assert hxv.EDContext(m, MpiRank=7, MpiSize=8).vecDim_Hv_sector(isec) == 12870 * 1608
    # communicator shrink (ED_HAMILTONIAN.f90:63-89): DimDw < MpiSize
    tiny = m.get_Sector(3, 0)                                   # DimDw = 1
    assert hxv.EDContext(m, MpiRank=0, MpiSize=4).vecDim_Hv_sector(tiny) == comb(16, 3)
    assert hxv.EDContext(m, MpiRank=2, MpiSize=4).vecDim_Hv_sector(tiny) == 0
    with pytest.raises(hxv.HxvError):
        ctx._spHtimesV(10, np.zeros(10, complex), np.zeros(10, complex))  # no sector open


def test_model_arrays_follow_reference_layout():
    from hxv import models

    m = models.bhz_2d(Nbath=1)
    assert m.impHloc.shape == (4, 4, 2, 2, 2, 2) and m.Hbath.shape == (4, 4, 2, 2, 2, 2, 1) and m.Vbath.shape == (4, 2, 2, 1)
    assert m.impHloc.flags.f_contiguous
    h = m.impHloc
    for s in range(2):  # Hermitian one-body part
        hl = h[:, :, s, s].transpose(0, 2, 1, 3).reshape(8, 8)
        assert np.abs(hl - hl.conj().T).max() == 0
    v = models.deterministic_vector(5, offset=3)
    assert np.allclose(v[0], np.sin(0.37 * 3 + 0.11) + 1j * np.cos(0.23 * 3 + 0.05))


def test_create_from_csr_validates_before_touching_the_device(built):
    import hxv

    rp = np.array([0, 1, 2], dtype=np.int64)
    ok_cols = np.array([2, 1], dtype=np.int32)
    vals = np.array([1.0 + 0j, 1.0 + 0j])
    diag = np.zeros(4, dtype=np.complex128)
    with pytest.raises(hxv.HxvError, match=r"column index outside"):
        hxv.HxvSector.from_csr(2, 2, (rp, np.array([3, 1], dtype=np.int32), vals), (rp, ok_cols, vals), diag)
    with pytest.raises(hxv.HxvError, match=r"rowptr"):
        hxv.HxvSector.from_csr(2, 2, (np.array([1, 1, 2], dtype=np.int64), ok_cols, vals), (rp, ok_cols, vals), diag)
    with pytest.raises(hxv.HxvError, match=r"complex diagonal"):
        hxv.HxvSector.from_csr(2, 2, (rp, ok_cols, vals), (rp, ok_cols, vals), diag + 1j)
    with pytest.raises(hxv.HxvError, match=r"rank"):
        hxv.HxvSector.from_csr(2, 2, (rp, ok_cols, vals), (rp, ok_cols, vals), diag, rank=0, nranks=3)


def test_model_validation_messages(built):
    import hxv
    from hxv import models

    m = models.plaquette_2x2_nobath()
    m2 = models.Model(4, 1, 1, 0, m.impHloc + 1j * np.eye(4).reshape(4, 4, 1, 1, 1, 1), m.Hbath, m.Vbath)
    with pytest.raises(hxv.HxvError, match="complex diagonal"):
        hxv.HxvSector.from_model(m2, 2, 2)
    with pytest.raises(hxv.HxvError, match="nranks > DimDw"):
        hxv.HxvSector.from_model(m, 2, 4, rank=0, nranks=2)      # DimDw = 1: shrink the communicator first
    with pytest.raises(hxv.HxvError, match="panel_rows|nrows"):
        hxv.HxvSector.dw_panel(m, 2, 2, 100)                      # more rows than DimUp
