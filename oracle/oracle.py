"""ctypes loader for the CPU oracle (oracle/hxv_oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = _HERE / "_build" / "libhxv_oracle.so"


def build(force: bool = False) -> Path:
    if force or not _LIB.exists() or _LIB.stat().st_mtime < (_HERE / "hxv_oracle.c").stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(_LIB))
        dp = C.POINTER(C.c_double)
        L.orc_open.restype = C.c_void_p
        L.orc_open.argtypes = [C.c_int] * 6 + [dp] + [C.c_double] * 5 + [C.c_int, dp, dp, dp, C.c_int, C.c_int]
        L.orc_close.argtypes = [C.c_void_p]
        for f in ("orc_dim_up", "orc_dim_dw", "orc_qdw"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("orc_dim", "orc_ishift", "orc_vecdim"):
            getattr(L, f).restype = C.c_int64
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_map_up.restype = C.POINTER(C.c_int)
        L.orc_map_up.argtypes = [C.c_void_p]
        L.orc_map_dw.restype = C.POINTER(C.c_int)
        L.orc_map_dw.argtypes = [C.c_void_p]
        L.orc_diag_ri.restype = dp
        L.orc_diag_ri.argtypes = [C.c_void_p]
        L.orc_nnz.restype = C.c_int64
        L.orc_nnz.argtypes = [C.c_void_p, C.c_int]
        L.orc_dump_csr.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int), dp]
        L.orc_spmatvec_main.restype = C.c_int
        L.orc_spmatvec_main.argtypes = [C.c_void_p, C.c_int64, dp, dp]
        L.orc_dense.restype = C.c_int
        L.orc_dense.argtypes = [C.c_void_p, dp]
        L.orc_spmatvec_mpi_main.restype = C.c_int
        L.orc_spmatvec_mpi_main.argtypes = [C.POINTER(C.c_void_p), C.c_int, dp, dp, dp]
        L.orc_lanc_tridiag.restype = C.c_int
        L.orc_lanc_tridiag.argtypes = [C.c_void_p, dp, C.c_int, dp, dp, C.c_double]
        _lib = L
    return _lib


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class OracleSector:
    """build_Hv_sector(isector) of the reference on the CPU (ED_HAMILTONIAN.f90:39-143)."""

    def __init__(self, model, nup: int, ndw: int, rank: int = 0, size: int = 1):
        L = lib()
        self.model = model
        uloc = np.ascontiguousarray(model.Uloc, dtype=np.float64)
        h = np.ascontiguousarray(model.impHloc.ravel(order="F")).view(np.float64)
        hb = np.ascontiguousarray(model.Hbath.ravel(order="F")).view(np.float64) if model.Nbath > 0 else np.zeros(2)
        vb = np.ascontiguousarray(model.Vbath.ravel(order="F")) if model.Nbath > 0 else np.zeros(1)
        self._keep = (uloc, h, hb, vb)
        self.h = L.orc_open(model.Nlat, model.Norb, model.Nspin, model.Nbath, nup, ndw, _dp(uloc), model.Ust, model.Jh, model.Jx,
                            model.Jp, model.xmu, int(model.hfmode), _dp(h), _dp(hb), _dp(vb), rank, size)
        if not self.h:
            raise RuntimeError("orc_open failed")
        self.DimUp = L.orc_dim_up(self.h)
        self.DimDw = L.orc_dim_dw(self.h)
        self.Dim = L.orc_dim(self.h)
        self.mpiQdw = L.orc_qdw(self.h)
        self.mpiIshift = L.orc_ishift(self.h)
        self.vecDim = L.orc_vecdim(self.h)

    def close(self):
        if self.h:
            lib().orc_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def map_up(self):
        return np.ctypeslib.as_array(lib().orc_map_up(self.h), shape=(self.DimUp,)).copy()

    def map_dw(self):
        return np.ctypeslib.as_array(lib().orc_map_dw(self.h), shape=(self.DimDw,)).copy()

    def diag(self):
        return np.ctypeslib.as_array(lib().orc_diag_ri(self.h), shape=(2 * self.vecDim,)).copy().view(np.complex128)

    def csr(self, which: str):
        """('up'|'dw'|'nd') -> (rowptr int64 [n+1], cols int32 1-based, vals complex128), row-list order."""
        w = {"up": 0, "dw": 1, "nd": 2}[which]
        n = {0: self.DimUp, 1: self.DimDw, 2: self.vecDim}[w]
        nnz = lib().orc_nnz(self.h, w)
        rp = np.zeros(n + 1, dtype=np.int64)
        cols = np.zeros(max(nnz, 1), dtype=np.int32)
        vals = np.zeros(2 * max(nnz, 1))
        lib().orc_dump_csr(self.h, w, rp.ctypes.data_as(C.POINTER(C.c_int64)), cols.ctypes.data_as(C.POINTER(C.c_int)), _dp(vals))
        return rp, cols[:nnz], vals.view(np.complex128)[:nnz]

    def spMatVec_main(self, v: np.ndarray) -> np.ndarray:
        v = np.ascontiguousarray(v, dtype=np.complex128)
        hv = np.empty_like(v)
        rc = lib().orc_spmatvec_main(self.h, v.size, _dp(v.view(np.float64)), _dp(hv.view(np.float64)))
        if rc:
            raise RuntimeError("orc_spmatvec_main: bad Nloc / not serial")
        return hv

    def dense(self) -> np.ndarray:
        buf = np.zeros(self.Dim * self.Dim, dtype=np.complex128)
        rc = lib().orc_dense(self.h, _dp(buf.view(np.float64)))
        if rc:
            raise RuntimeError("orc_dense failed")
        return buf.reshape((self.Dim, self.Dim), order="F")

    def lanc_tridiag(self, vin: np.ndarray, nlanc: int, threshold: float = 1e-12):
        vin = np.ascontiguousarray(vin, dtype=np.complex128)
        a = np.zeros(nlanc)
        b = np.zeros(nlanc)
        n = lib().orc_lanc_tridiag(self.h, _dp(vin.view(np.float64)), nlanc, _dp(a), _dp(b), threshold)
        return a[:n], b[:n]


def spMatVec_mpi_main(model, nup, ndw, P: int, v: np.ndarray, repeat: int = 1, sectors=None):
    """Reference MPI product with P ranks emulated by threads; returns (Hv, sectors)."""
    L = lib()
    secs = sectors or [OracleSector(model, nup, ndw, r, P) for r in range(P)]
    arr = (C.c_void_p * P)(*[s.h for s in secs])
    v = np.ascontiguousarray(v, dtype=np.complex128)
    hv = np.zeros_like(v)
    work = np.zeros(2 * v.size, dtype=np.complex128)
    for _ in range(repeat):
        rc = L.orc_spmatvec_mpi_main(arr, P, _dp(v.view(np.float64)), _dp(hv.view(np.float64)), _dp(work.view(np.float64)))
        if rc:
            raise RuntimeError("orc_spmatvec_mpi_main failed")
    return hv, secs
