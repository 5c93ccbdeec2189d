"""ctypes binding of libhxv.so (include/hxv.h) -- the only compute path of this package.

There is deliberately no CPU fallback: if the HIP library is missing or no GPU is present,
every entry point raises.  PyTorch is used only as plumbing (device buffers, streams,
torch.distributed); the arithmetic lives in csrc/*.hip.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent.parent
LIB_PATH = Path(os.environ.get("HXV_LIB", str(_PKG / "lib" / "libhxv.so")))  # HXV_LIB: A/B experiments only


class HxvError(RuntimeError):
    pass


class _Model(C.Structure):
    _fields_ = [
        ("nlat", C.c_int32), ("norb", C.c_int32), ("nspin", C.c_int32), ("nbath", C.c_int32),
        ("hfmode", C.c_int32), ("reserved", C.c_int32),
        ("uloc", C.c_double * 5),
        ("ust", C.c_double), ("jh", C.c_double), ("jx", C.c_double), ("jp", C.c_double), ("xmu", C.c_double),
        ("imphloc", C.c_void_p), ("hbath", C.c_void_p), ("vbath", C.c_void_p),
    ]


class _Stats(C.Structure):
    _fields_ = [
        ("n_apply", C.c_int64), ("algorithmic_bytes", C.c_int64), ("device_bytes", C.c_int64),
        ("kernel", C.c_int32), ("real_h", C.c_int32), ("k_up", C.c_int32), ("k_dw", C.c_int32),
        ("n_hops_up", C.c_int32), ("n_hops_dw", C.c_int32), ("h2d_bytes", C.c_int64), ("d2h_bytes", C.c_int64),
    ]


# every symbol include/hxv.h declares (tests check that the library exports all of them)
EXPORTS = [
    "hxv_create_from_model", "hxv_create_from_csr", "hxv_set_nonlocal_csr", "hxv_slab_home", "hxv_create_dw_panel", "hxv_apply_dw_panel", "hxv_apply_up_add", "hxv_destroy", "hxv_vecdim", "hxv_dims", "hxv_apply_host",
    "hxv_apply_device", "hxv_apply_device_real", "hxv_real_vectors_available", "hxv_pitch_real", "hxv_realvec_elems", "hxv_fullvec_elems", "hxv_localvec_elems", "hxv_pitch", "hxv_time_apply", "hxv_lanczos_tridiag", "hxv_lanczos_eigh", "hxv_lanczos_tridiag_host", "hxv_lanczos_eigh_host", "hxv_lanczos_tridiag_pair", "hxv_lanczos_tridiag_pair_host", "hxv_eigh_lowest", "hxv_eigh_lowest_host", "hxv_time_lanczos", "hxv_apply_ladder", "hxv_apply_ladder_axpy", "hxv_live_handles", "hxv_row_order", "hxv_get_maps",
    "hxv_nnz", "hxv_get_csr", "hxv_get_diag", "hxv_set_option", "hxv_get_option", "hxv_get_stats", "hxv_pool_trim", "hxv_pool_stats", "hxv_last_error",
    "hxv_version", "hxv_comm_unique_id", "hxv_comm_init", "hxv_comm_free", "hxv_apply_device_slab", "hxv_exchange_count",
    "hxv_set_exchange_default", "hxv_exchange_mode", "hxv_halo_counts", "hxv_halo_lists", "hxv_halo_plan_from_csr",
    "hxv_comm_local_create", "hxv_comm_init_local", "hxv_comm_local_destroy", "hxv_comm_local_abort", "hxv_time_apply_slab",
    "hxv_vector_alloc", "hxv_vector_alloc_many", "hxv_vector_free", "hxv_vector_from_host", "hxv_vector_to_host",
    "hxv_sector_cache_clear", "hxv_sector_cache_stats", "hxv_comm_abort", "hxv_comm_library", "hxv_comm_cache_stats", "hxv_comm_cache_clear", "hxv_host_register", "hxv_host_unregister",
]

_lib = None


def load_library():
    """Load libhxv.so; raises HxvError (never falls back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it bundles its own ROCm runtime (same SONAMEs as /opt/rocm); loading it first
    # guarantees one HIP runtime per process whichever order the caller imports things in.
    import torch  # noqa: F401

    if not LIB_PATH.exists():
        raise HxvError(f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950); "
                       "this package has no CPU fallback")
    L = C.CDLL(str(LIB_PATH))
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    pd, pi32, pi64 = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    L.hxv_create_from_model.argtypes = [C.POINTER(_Model), i32, i32, i32, i32, i32, C.POINTER(vp)]
    L.hxv_create_from_csr.argtypes = [i32, i32, pi64, pi32, pd, pi64, pi32, pd, pd, i32, i32, i32, C.POINTER(vp)]
    L.hxv_set_nonlocal_csr.argtypes = [vp, pi64, pi32, pd]
    L.hxv_slab_home.argtypes = [vp, C.POINTER(vp)]
    L.hxv_create_dw_panel.argtypes = [C.POINTER(_Model), i32, i32, i32, i32, C.POINTER(vp)]
    L.hxv_apply_dw_panel.argtypes = [vp, vp, vp, vp]
    L.hxv_apply_up_add.argtypes = [vp, vp, vp, vp, vp]
    L.hxv_destroy.argtypes = [vp]
    L.hxv_vecdim.argtypes = [vp]
    L.hxv_vecdim.restype = i64
    L.hxv_dims.argtypes = [vp, pi32, pi32, pi64, pi32, pi64]
    L.hxv_fullvec_elems.argtypes = [vp]
    L.hxv_fullvec_elems.restype = i64
    L.hxv_localvec_elems.argtypes = [vp]
    L.hxv_localvec_elems.restype = i64
    L.hxv_pitch.argtypes = [vp]
    L.hxv_pitch.restype = i32
    L.hxv_apply_host.argtypes = [vp, i64, vp, vp]
    L.hxv_apply_device.argtypes = [vp, vp, vp, vp]
    L.hxv_apply_device_real.argtypes = [vp, vp, vp, vp]
    L.hxv_real_vectors_available.argtypes = [vp]
    L.hxv_pitch_real.argtypes = [vp]
    L.hxv_realvec_elems.argtypes = [vp]
    L.hxv_realvec_elems.restype = i64
    L.hxv_time_apply.argtypes = [vp, vp, vp, i32, C.POINTER(C.c_float)]
    L.hxv_lanczos_tridiag.argtypes = [vp, vp, i32, pd, pd, dbl, pi32]
    L.hxv_lanczos_eigh.argtypes = [vp, i32, dbl, pd, vp, pi32]
    L.hxv_lanczos_tridiag_host.argtypes = [vp, vp, i32, pd, pd, dbl, pi32]
    L.hxv_lanczos_eigh_host.argtypes = [vp, i32, dbl, pd, vp, pi32]
    L.hxv_lanczos_tridiag_pair.argtypes = [vp, vp, vp, i32, pd, pd, pd, pd, dbl, pi32, pi32]
    L.hxv_lanczos_tridiag_pair_host.argtypes = [vp, vp, vp, i32, pd, pd, pd, pd, dbl, pi32, pi32]
    L.hxv_eigh_lowest.argtypes = [vp, i32, i32, i32, dbl, pd, vp, pi32, pi32]
    L.hxv_eigh_lowest_host.argtypes = [vp, i32, i32, i32, dbl, pd, vp, pi32, pi32]
    L.hxv_time_lanczos.argtypes = [vp, vp, i32, C.POINTER(C.c_float)]
    L.hxv_apply_ladder.argtypes = [vp, vp, i32, i32, i32, vp, vp, pd]
    L.hxv_apply_ladder_axpy.argtypes = [vp, vp, i32, i32, i32, dbl, dbl, i32, vp, vp, pd]
    L.hxv_get_maps.argtypes = [vp, pi32, pi32]
    L.hxv_nnz.argtypes = [vp, i32]
    L.hxv_nnz.restype = i64
    L.hxv_get_csr.argtypes = [vp, i32, pi64, pi32, pd]
    L.hxv_get_diag.argtypes = [vp, pd]
    L.hxv_set_option.argtypes = [vp, C.c_char_p, i64]
    L.hxv_get_option.argtypes = [vp, C.c_char_p]
    L.hxv_get_option.restype = i64
    L.hxv_get_stats.argtypes = [vp, C.POINTER(_Stats)]
    L.hxv_pool_trim.argtypes = [i32]
    L.hxv_pool_stats.argtypes = [i32, pi64, pi64, pi64]
    L.hxv_last_error.restype = C.c_char_p
    L.hxv_version.restype = C.c_char_p
    L.hxv_comm_unique_id.argtypes = [vp]
    L.hxv_comm_init.argtypes = [vp, vp]
    L.hxv_comm_free.argtypes = [vp]
    L.hxv_apply_device_slab.argtypes = [vp, vp, vp, vp]
    L.hxv_vector_alloc.argtypes = [vp, C.POINTER(vp)]
    L.hxv_vector_alloc_many.argtypes = [vp, i32, C.POINTER(vp)]
    L.hxv_vector_free.argtypes = [vp, vp]
    L.hxv_vector_from_host.argtypes = [vp, vp, vp]
    L.hxv_vector_to_host.argtypes = [vp, vp, vp]
    L.hxv_time_apply_slab.argtypes = [vp, vp, vp, i32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.hxv_exchange_count.argtypes = [vp]
    L.hxv_exchange_count.restype = i64
    L.hxv_set_exchange_default.argtypes = [i32]
    L.hxv_exchange_mode.argtypes = [vp]
    L.hxv_exchange_mode.restype = i32
    L.hxv_halo_counts.argtypes = [vp, pi32, pi32]
    L.hxv_halo_lists.argtypes = [vp, pi32, pi32]
    L.hxv_halo_plan_from_csr.argtypes = [i32, pi64, pi32, i32, i32, pi32, pi32, pi32, pi32, pi32, pi32]
    L.hxv_comm_local_create.argtypes = [i32, C.POINTER(vp)]
    L.hxv_comm_init_local.argtypes = [vp, vp]
    L.hxv_comm_local_destroy.argtypes = [vp]
    L.hxv_comm_local_abort.argtypes = [vp]
    L.hxv_comm_abort.argtypes = [vp]
    L.hxv_comm_library.argtypes = [vp]
    L.hxv_comm_library.restype = C.c_char_p
    L.hxv_sector_cache_clear.argtypes = []
    L.hxv_sector_cache_stats.argtypes = [pi64, pi64, pi64, pi64]
    L.hxv_comm_cache_stats.argtypes = [pi64, pi64, pi64]
    L.hxv_comm_cache_clear.argtypes = [pi64]
    L.hxv_row_order.argtypes = [vp, pi32, C.POINTER(C.c_int8)]
    L.hxv_row_order.restype = i32
    L.hxv_host_register.argtypes = [vp, i64]
    L.hxv_host_unregister.argtypes = [vp]
    L.hxv_live_handles.argtypes = []
    L.hxv_live_handles.restype = i64
    _lib = L
    return L


def set_exchange_default(mode: str | int):
    """Exchange of the split sectors created from now on: "allgather" / 0 [default], "halo" / 1 or "alltoall" / 2 (the reference's two
    transposes) (include/hxv.h)."""
    m = {"allgather": 0, "halo": 1, "alltoall": 2}.get(mode, mode)
    _chk(load_library().hxv_set_exchange_default(int(m)), "hxv_set_exchange_default")


def pool_trim(device: int = -1):
    """Return the engine's cached device buffers to the driver (include/hxv.h, device-buffer cache)."""
    load_library().hxv_pool_trim(device)


def pool_stats(device: int = 0) -> dict:
    c, h, m = C.c_int64(), C.c_int64(), C.c_int64()
    load_library().hxv_pool_stats(device, C.byref(c), C.byref(h), C.byref(m))
    return {"cached_bytes": c.value, "hits": h.value, "misses": m.value}


def sector_cache_clear():
    """Drop the images of closed sectors (include/hxv.h, sector-image cache)."""
    load_library().hxv_sector_cache_clear()


def sector_cache_stats() -> dict:
    e, b, h, m = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    load_library().hxv_sector_cache_stats(C.byref(e), C.byref(b), C.byref(h), C.byref(m))
    return {"entries": e.value, "bytes": b.value, "hits": h.value, "misses": m.value}


def comm_cache_stats() -> dict:
    """The process-level RCCL communicators (include/hxv.h: one ncclCommInitRank per (library, nranks, rank, device), shared by every sector)."""
    e, i, r = C.c_int64(), C.c_int64(), C.c_int64()
    load_library().hxv_comm_cache_stats(C.byref(e), C.byref(i), C.byref(r))
    return {"entries": e.value, "inits": i.value, "reuses": r.value}


def comm_cache_clear() -> int:
    """ncclCommDestroy of the cached communicators no sector is bound to (every rank, at the same point of the program); returns how many."""
    n = C.c_int64()
    load_library().hxv_comm_cache_clear(C.byref(n))
    return n.value


def host_register(a: np.ndarray):
    """Page-lock a numpy array that is passed to apply_host / the *_host drivers again and again (include/hxv.h: hxv_host_register)."""
    _chk(load_library().hxv_host_register(a.ctypes.data, a.nbytes), "hxv_host_register")


def host_unregister(a: np.ndarray):
    _chk(load_library().hxv_host_unregister(a.ctypes.data), "hxv_host_unregister")


def live_handles() -> int:
    return load_library().hxv_live_handles()


def halo_plan_from_csr(dimdw: int, rowptr, cols, rank: int, nranks: int):
    """(recv_counts, send_counts, recv_cols, send_cols) of one rank of a split, global 0-based columns, from H_dw alone (CSR with 1-based
    columns as HxvSector.csr / the oracle return it).  Host only: works without a GPU."""
    L = load_library()
    rp = np.ascontiguousarray(rowptr, dtype=np.int64)
    cl = np.ascontiguousarray(cols, dtype=np.int32)
    rc = np.zeros(nranks, dtype=np.int32)
    sc = np.zeros(nranks, dtype=np.int32)
    nr, ns = C.c_int32(), C.c_int32()
    _chk(L.hxv_halo_plan_from_csr(dimdw, _p(rp, C.c_int64), _p(cl, C.c_int32), rank, nranks, _p(rc, C.c_int32), _p(sc, C.c_int32), None, None,
                                  C.byref(nr), C.byref(ns)), "hxv_halo_plan_from_csr")
    rcols = np.zeros(max(nr.value, 1), dtype=np.int32)
    scols = np.zeros(max(ns.value, 1), dtype=np.int32)
    _chk(L.hxv_halo_plan_from_csr(dimdw, _p(rp, C.c_int64), _p(cl, C.c_int32), rank, nranks, _p(rc, C.c_int32), _p(sc, C.c_int32),
                                  _p(rcols, C.c_int32), _p(scols, C.c_int32), C.byref(nr), C.byref(ns)), "hxv_halo_plan_from_csr")
    return rc, sc, rcols[: nr.value], scols[: ns.value]


class LocalGroup:
    """A communicator whose ranks are host threads of this process (hxv_comm_local_create): how several ranks of a split sector
    run on one GPU (RCCL refuses two ranks on one device) or on the GPUs of a node under one process."""

    def __init__(self, nranks: int):
        g = C.c_void_p()
        _chk(load_library().hxv_comm_local_create(nranks, C.byref(g)), "hxv_comm_local_create")
        self._g = g
        self.nranks = nranks

    def join(self, sec: "HxvSector"):
        """Collective: the handle of this thread's rank joins the group (hxv_comm_init_local)."""
        sec.comm_init_local(self)

    def abort(self):
        """A rank's thread failed outside the library: wake every peer that waits in a collective (hxv_comm_local_abort)."""
        if getattr(self, "_g", None):
            load_library().hxv_comm_local_abort(self._g)

    def close(self):
        if getattr(self, "_g", None):
            _chk(load_library().hxv_comm_local_destroy(self._g), "hxv_comm_local_destroy")
            self._g = None


class RcclGroup:
    """The RCCL transport for ranks that are host threads: one unique id (hxv_comm_unique_id), every rank's thread calls
    hxv_comm_init.  With the real librccl this needs one GPU per rank; the test suite points HXV_RCCL_LIB at its double
    (tests/rccl_double) to run the engine's RCCL branches with several ranks on one GPU."""

    def __init__(self, nranks: int):
        import threading

        self.nranks = nranks
        self.id = HxvSector.comm_unique_id()
        self._joined = []
        self._lock = threading.Lock()

    def join(self, sec: "HxvSector"):
        sec.comm_init(self.id)
        with self._lock:
            self._joined.append(sec)

    def abort(self):
        """A rank's thread failed outside the library: ncclCommAbort on every communicator that has joined (hxv_comm_abort), so peers
        blocked in a collective return an error.  (A rank lost BEFORE ncclCommInitRank completed cannot be helped: no communicator yet.)"""
        with self._lock:
            self._joined = [s for s in self._joined if getattr(s, "_h", None)]   # (closed sectors leave the list)
            secs = list(self._joined)
        for s in secs:
            # serialised against close() of the same sector: hxv_comm_abort must not run while -- or after -- hxv_destroy frees
            # the handle on the peer's thread; a peer blocked in a collective does not hold the lock, so the abort that wakes it never waits
            with s._life:
                if getattr(s, "_h", None):
                    load_library().hxv_comm_abort(s._h)

    def close(self):
        pass


def run_ranks(nranks: int, fn, transport: str = "local"):
    """Run fn(rank, group) on one host thread per rank of a fresh group (transport "local": thread-rank transport; "rccl": the RCCL
    branches, see RcclGroup); fn joins with group.join(sector).  Returns the list of results (re-raises the first error).  A rank that
    fails aborts the group, so its peers return from their collectives instead of waiting for it."""
    import threading

    group = LocalGroup(nranks) if transport == "local" else RcclGroup(nranks)
    out, err = [None] * nranks, [None] * nranks
    order, lock = [], threading.Lock()   # errors in the order they happened: the first one is the cause, the rest are its peers waking up

    def work(r):
        try:
            out[r] = fn(r, group)
        except BaseException as e:  # noqa: BLE001 (reported to the caller below)
            err[r] = e
            with lock:
                order.append(e)
            group.abort()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    try:
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    finally:
        try:
            group.close()
        except HxvError:
            if not any(err):
                raise
    if order:
        raise order[0]
    return out


def _chk(rc: int, what: str):
    if rc != 0:
        msg = load_library().hxv_last_error().decode()
        raise HxvError(f"{what} failed (status {rc}): {msg}")


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(C.POINTER(t))


class HxvSector:
    """An open sector on one GPU: the device-side replacement of the state that
    build_Hv_sector leaves in ED_HAMILTONIAN_COMMON.f90:11-20 + spH0d/spH0ups/spH0dws."""

    def __init__(self, handle, keep=None, device_index: int = 0):
        import threading

        self._h = handle
        self._keep = keep
        self._life = threading.Lock()   # close() against RcclGroup.abort() from another rank's thread
        L = load_library()
        du, dd, q = C.c_int32(), C.c_int32(), C.c_int32()
        dim, ish = C.c_int64(), C.c_int64()
        _chk(L.hxv_dims(self._h, C.byref(du), C.byref(dd), C.byref(dim), C.byref(q), C.byref(ish)), "hxv_dims")
        self.DimUp, self.DimDw, self.Dim, self.mpiQdw, self.mpiIshift = du.value, dd.value, dim.value, q.value, ish.value
        self.vecDim = L.hxv_vecdim(self._h)
        # device layout: every column padded to `pitch` elements (include/hxv.h, DEVICE VECTOR LAYOUT)
        self.pitch = L.hxv_pitch(self._h)
        self.fullElems = L.hxv_fullvec_elems(self._h)   # length of the (padded) all-gather layout handed to apply_device
        self.localElems = L.hxv_localvec_elems(self._h)  # length of the (padded) local result / of a Lanczos vector
        self.ncolsFull = self.fullElems // self.pitch
        self.device_index = int(device_index)
        # device row order (include/hxv.h, DEVICE ROW ORDER): d_vec[k*pitch + perm[iup]] = sign[iup] * v_ref[k*DimUp + iup]; None = the reference's
        self.row_perm = self.row_sign = None
        self._row_dev = {}
        if L.hxv_row_order(self._h, None, None) == 1:
            perm = np.zeros(self.DimUp, dtype=np.int32)
            sign = np.zeros(self.DimUp, dtype=np.int8)
            L.hxv_row_order(self._h, _p(perm, C.c_int32), _p(sign, C.c_int8))
            self.row_perm, self.row_sign = perm, sign

    def _row_tables(self, device, dtype):
        """(perm as an index tensor, sign as a tensor of `dtype`) on `device`, cached."""
        import torch

        key = (str(device), dtype)
        if key not in self._row_dev:
            self._row_dev[key] = (torch.from_numpy(self.row_perm.astype(np.int64)).to(device), torch.from_numpy(self.row_sign.astype(np.float64)).to(device=device, dtype=dtype))
        return self._row_dev[key]

    # -- constructors ---------------------------------------------------------------------
    @staticmethod
    def _model_struct(model):
        m = _Model()
        m.nlat, m.norb, m.nspin, m.nbath = model.Nlat, model.Norb, model.Nspin, model.Nbath
        m.hfmode = int(bool(model.hfmode))
        for i in range(5):
            m.uloc[i] = float(model.Uloc[i])
        m.ust, m.jh, m.jx, m.jp, m.xmu = float(model.Ust), float(model.Jh), float(model.Jx), float(model.Jp), float(model.xmu)
        h = np.ascontiguousarray(model.impHloc.ravel(order="F")).view(np.float64)
        hb = np.ascontiguousarray(model.Hbath.ravel(order="F")).view(np.float64)
        vb = np.ascontiguousarray(model.Vbath.ravel(order="F"))
        m.imphloc = h.ctypes.data
        m.hbath = hb.ctypes.data if model.Nbath > 0 else None
        m.vbath = vb.ctypes.data if model.Nbath > 0 else None
        return m, (h, hb, vb)

    @classmethod
    def dw_panel(cls, model, nup: int, ndw: int, nrows: int, device: int = 0) -> "HxvSector":
        """Handle for a row panel [nrows x DimDw] (dw hops only): the middle step of the all-to-all exchange."""
        L = load_library()
        m, keep = cls._model_struct(model)
        out = C.c_void_p()
        _chk(L.hxv_create_dw_panel(C.byref(m), nup, ndw, nrows, device, C.byref(out)), "hxv_create_dw_panel")
        return cls(out, keep=keep, device_index=device)

    def apply_dw_panel(self, x, y=None):
        """y = x H_dw^T on a panel handle; x, y: [DimDw columns][pitch] torch complex128 CUDA tensors."""
        import torch

        assert x.is_cuda and x.dtype == torch.complex128 and x.is_contiguous() and x.numel() == self.localElems
        if y is None:
            y = torch.zeros(self.localElems, dtype=torch.complex128, device=x.device)
        st = torch.cuda.current_stream(x.device).cuda_stream
        _chk(load_library().hxv_apply_dw_panel(self._h, x.data_ptr(), y.data_ptr(), st), "hxv_apply_dw_panel")
        return y

    def apply_up_add(self, v_local, w, hv_local=None):
        """hv_local = D.v + H_up v + w on the local slab; all three [qdw columns][pitch]."""
        import torch

        assert v_local.is_cuda and v_local.numel() == self.localElems and w.numel() == self.localElems
        if hv_local is None:
            hv_local = torch.zeros(self.localElems, dtype=torch.complex128, device=v_local.device)
        st = torch.cuda.current_stream(v_local.device).cuda_stream
        _chk(load_library().hxv_apply_up_add(self._h, v_local.data_ptr(), w.data_ptr(), hv_local.data_ptr(), st), "hxv_apply_up_add")
        return hv_local

    @classmethod
    def from_model(cls, model, nup: int, ndw: int, rank: int = 0, nranks: int = 1, device: int = 0) -> "HxvSector":
        L = load_library()
        m = _Model()
        m.nlat, m.norb, m.nspin, m.nbath = model.Nlat, model.Norb, model.Nspin, model.Nbath
        m.hfmode = int(bool(model.hfmode))
        for i in range(5):
            m.uloc[i] = float(model.Uloc[i])
        m.ust, m.jh, m.jx, m.jp, m.xmu = float(model.Ust), float(model.Jh), float(model.Jx), float(model.Jp), float(model.xmu)
        h = np.ascontiguousarray(model.impHloc.ravel(order="F")).view(np.float64)
        hb = np.ascontiguousarray(model.Hbath.ravel(order="F")).view(np.float64)
        vb = np.ascontiguousarray(model.Vbath.ravel(order="F"))
        m.imphloc = h.ctypes.data
        m.hbath = hb.ctypes.data if model.Nbath > 0 else None
        m.vbath = vb.ctypes.data if model.Nbath > 0 else None
        out = C.c_void_p()
        _chk(L.hxv_create_from_model(C.byref(m), nup, ndw, rank, nranks, device, C.byref(out)), "hxv_create_from_model")
        return cls(out, keep=(h, hb, vb), device_index=device)

    @classmethod
    def from_csr(cls, DimUp, DimDw, up, dw, diag, rank: int = 0, nranks: int = 1, device: int = 0, nd=None) -> "HxvSector":
        """up/dw = (rowptr int64, cols int32 1-based, vals complex128) as dumped from spH0ups(1)/spH0dws(1);
        diag = local rows of spH0d (complex128); nd = spH0nd likewise (local rows, GLOBAL 1-based columns) or None."""
        L = load_library()
        arrs = []
        for rp, cols, vals in (up, dw):
            arrs += [np.ascontiguousarray(rp, dtype=np.int64), np.ascontiguousarray(cols, dtype=np.int32),
                     np.ascontiguousarray(vals, dtype=np.complex128).view(np.float64)]
        dg = np.ascontiguousarray(diag, dtype=np.complex128).view(np.float64)
        out = C.c_void_p()
        _chk(L.hxv_create_from_csr(DimUp, DimDw, _p(arrs[0], C.c_int64), _p(arrs[1], C.c_int32), _p(arrs[2], C.c_double),
                                   _p(arrs[3], C.c_int64), _p(arrs[4], C.c_int32), _p(arrs[5], C.c_double), _p(dg, C.c_double),
                                   rank, nranks, device, C.byref(out)), "hxv_create_from_csr")
        sec = cls(out, device_index=device)
        if nd is not None:
            sec.set_nonlocal_csr(*nd)
        return sec

    def set_nonlocal_csr(self, rowptr, cols, vals):
        """spH0nd as stored by the reference (ED_HAMILTONIAN_SPARSE_HxV.f90:217-225): local rows, global 1-based columns."""
        rp = np.ascontiguousarray(rowptr, dtype=np.int64)
        cc = np.ascontiguousarray(cols, dtype=np.int32)
        vv = np.ascontiguousarray(vals, dtype=np.complex128).view(np.float64)
        _chk(load_library().hxv_set_nonlocal_csr(self._h, _p(rp, C.c_int64), _p(cc, C.c_int32), _p(vv, C.c_double)), "hxv_set_nonlocal_csr")

    def _dev(self):
        import torch

        return torch.device("cuda", self.device_index)

    def close(self):
        life = getattr(self, "_life", None)
        if life is None:
            return
        with life:
            if getattr(self, "_h", None):
                load_library().hxv_destroy(self._h)
                self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the product ----------------------------------------------------------------------
    def apply_host(self, v: np.ndarray, hv: np.ndarray | None = None) -> np.ndarray:
        """spHtimesV_p(Nloc,v,Hv) on host arrays (two PCIe copies per call)."""
        v = np.ascontiguousarray(v, dtype=np.complex128)
        if hv is None:
            hv = np.empty_like(v)
        assert hv.dtype == np.complex128 and hv.flags.c_contiguous
        _chk(load_library().hxv_apply_host(self._h, v.size, v.ctypes.data, hv.ctypes.data), "hxv_apply_host")
        return hv

    # -- padded device layout helpers -------------------------------------------------------
    def pad(self, x, ncols=None):
        """contiguous [ncols x DimUp] torch vector in the REFERENCE's order -> device layout [ncols x pitch] (pads zero; with a device row
        order -- include/hxv.h -- row i of every column goes to row perm[i] with the sign of its basis vector).  The convenience forms of
        apply_device / lanczos_tridiag / apply_ladder recognise a contiguous vector by its LENGTH; where DimUp is a multiple of 8 the two
        layouts are equally long and the vector is taken as a device vector: call pad / unpad explicitly there."""
        import torch

        if self.row_perm is not None:
            # device row order: reference row i of every column goes to device row perm[i], with the sign of its basis vector
            ncols = x.numel() // self.DimUp if ncols is None else ncols
            perm, sign = self._row_tables(x.device, x.dtype)
            out = torch.zeros(ncols, self.pitch, dtype=x.dtype, device=x.device)
            out[:, perm] = x.view(ncols, self.DimUp) * sign
            return out.view(-1)
        if self.pitch == self.DimUp:
            return x
        ncols = x.numel() // self.DimUp if ncols is None else ncols
        out = torch.zeros(ncols * self.pitch, dtype=x.dtype, device=x.device)
        out.view(ncols, self.pitch)[:, : self.DimUp] = x.view(ncols, self.DimUp)
        return out

    def unpad(self, x):
        """padded device layout -> contiguous vector in the reference's order (the inverse of pad)."""
        if self.row_perm is not None:
            perm, sign = self._row_tables(x.device, x.dtype)
            return (x.view(-1, self.pitch)[:, perm] * sign).contiguous().view(-1)
        if self.pitch == self.DimUp:
            return x
        return x.view(-1, self.pitch)[:, : self.DimUp].contiguous().view(-1)

    def apply_device(self, v_full, hv_local=None, stream=None):
        """Device-resident product on torch complex128 CUDA tensors, on torch's current stream.
        Native form: v_full in the padded all-gather layout (fullElems), hv_local padded (localElems).
        Convenience form (tests): a contiguous vector of ncolsFull*DimUp elements is padded on the fly and the
        result comes back contiguous (vecDim elements)."""
        import torch

        assert v_full.is_cuda and v_full.dtype == torch.complex128 and v_full.is_contiguous()
        convenience = v_full.numel() != self.fullElems
        if convenience:
            assert v_full.numel() == self.ncolsFull * self.DimUp and hv_local is None or (hv_local is not None and hv_local.numel() == self.vecDim)
            v_full = self.pad(v_full)
            out = hv_local
            hv_local = torch.zeros(self.localElems, dtype=torch.complex128, device=v_full.device)
        elif hv_local is None:
            hv_local = torch.zeros(self.localElems, dtype=torch.complex128, device=v_full.device)
        assert hv_local.is_cuda and hv_local.dtype == torch.complex128 and hv_local.is_contiguous() and hv_local.numel() == self.localElems
        st = torch.cuda.current_stream(v_full.device).cuda_stream if stream is None else stream
        _chk(load_library().hxv_apply_device(self._h, v_full.data_ptr(), hv_local.data_ptr(), st), "hxv_apply_device")
        if convenience:
            res = self.unpad(hv_local)
            if out is not None:
                out.copy_(res)
                return out
            return res
        return hv_local

    # -- slab exchange behind the C-ABI (split sector, MpiStatus=T; include/hxv.h) ------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        """128-byte id of a new communicator (rank 0); broadcast it with the host program's own means."""
        buf = C.create_string_buffer(128)
        _chk(load_library().hxv_comm_unique_id(buf), "hxv_comm_unique_id")
        return buf.raw

    def comm_init(self, id128: bytes):
        """Collective over the nranks handles of this sector (one process per GPU)."""
        assert len(id128) == 128
        _chk(load_library().hxv_comm_init(self._h, C.create_string_buffer(id128, 128)), "hxv_comm_init")

    def comm_init_local(self, group: "LocalGroup"):
        """Join a group of THREAD ranks (include/hxv.h): collective over the group's nranks handles, one host thread per rank."""
        _chk(load_library().hxv_comm_init_local(self._h, group._g), "hxv_comm_init_local")

    def comm_free(self):
        _chk(load_library().hxv_comm_free(self._h), "hxv_comm_free")

    def comm_abort(self):
        _chk(load_library().hxv_comm_abort(self._h), "hxv_comm_abort")

    @property
    def comm_library(self) -> str:
        """File the RCCL entry points of this handle's communicator came from (hxv_comm_library)."""
        return load_library().hxv_comm_library(self._h).decode()

    def slab_home(self):
        """This rank's slot of the engine's gather buffer as a torch complex128 CUDA tensor of localElems elements: a vector built
        THERE and passed to apply_device_slab saves the slab copy of every product (hxv_slab_home)."""
        import torch

        p = C.c_void_p()
        _chk(load_library().hxv_slab_home(self._h, C.byref(p)), "hxv_slab_home")

        class _View:
            pass

        view = _View()
        view.__cuda_array_interface__ = {"shape": (int(self.localElems),), "typestr": "<c16", "data": (int(p.value), False), "version": 2,
                                         "strides": None}
        return torch.as_tensor(view, device=self._dev())

    def apply_device_slab(self, v_local, hv_local=None, stream=None):
        """(H v)|slab from this rank's slab: exchange (ncclAllGather on the stream) + product.  Padded layout, localElems each."""
        import torch

        assert v_local.is_cuda and v_local.dtype == torch.complex128 and v_local.is_contiguous() and v_local.numel() == self.localElems
        if hv_local is None:
            hv_local = torch.zeros(self.localElems, dtype=torch.complex128, device=v_local.device)
        st = torch.cuda.current_stream(v_local.device).cuda_stream if stream is None else stream
        _chk(load_library().hxv_apply_device_slab(self._h, v_local.data_ptr(), hv_local.data_ptr(), st), "hxv_apply_device_slab")
        return hv_local

    def time_apply_slab(self, v_local, hv_local, nrep: int):
        """-> (ms per slab product incl. the exchange, ms of its kernels alone), HIP events on the handle's stream; collective."""
        import torch

        torch.cuda.synchronize()
        a, b = C.c_float(), C.c_float()
        _chk(load_library().hxv_time_apply_slab(self._h, v_local.data_ptr(), hv_local.data_ptr(), nrep, C.byref(a), C.byref(b)), "hxv_time_apply_slab")
        return a.value, b.value

    @property
    def exchange_count(self) -> int:
        return load_library().hxv_exchange_count(self._h)

    @property
    def exchange_mode(self) -> str:
        return {0: "allgather", 1: "halo", 2: "alltoall"}[load_library().hxv_exchange_mode(self._h)]

    def halo_lists(self, nranks: int):
        """(recv_counts[nranks], send_counts[nranks], recv_cols (global, slot order), send_cols (local, by destination))."""
        rc = np.zeros(nranks, dtype=np.int32)
        sc = np.zeros(nranks, dtype=np.int32)
        _chk(load_library().hxv_halo_counts(self._h, _p(rc, C.c_int32), _p(sc, C.c_int32)), "hxv_halo_counts")
        rcols = np.zeros(max(int(rc.sum()), 1), dtype=np.int32)
        scols = np.zeros(max(int(sc.sum()), 1), dtype=np.int32)
        _chk(load_library().hxv_halo_lists(self._h, _p(rcols, C.c_int32), _p(scols, C.c_int32)), "hxv_halo_lists")
        return rc, sc, rcols[: int(rc.sum())], scols[: int(sc.sum())]

    # -- REAL-vector mode (H real; include/hxv.h) ----------------------------------------------
    @property
    def real_vectors_available(self) -> bool:
        return bool(load_library().hxv_real_vectors_available(self._h))

    def pad_real(self, x):
        """contiguous real [DimDw*DimUp] -> device layout double[DimDw][pitch_real]."""
        import torch

        pr = load_library().hxv_pitch_real(self._h)
        out = torch.zeros(self.DimDw, pr, dtype=torch.float64, device=x.device)
        if self.row_perm is not None:
            perm, sign = self._row_tables(x.device, torch.float64)
            out[:, perm] = x.view(self.DimDw, self.DimUp) * sign
        else:
            out[:, : self.DimUp] = x.view(self.DimDw, self.DimUp)
        return out.view(-1)

    def unpad_real(self, x):
        import torch

        pr = load_library().hxv_pitch_real(self._h)
        if self.row_perm is not None:
            perm, sign = self._row_tables(x.device, torch.float64)
            return (x.view(self.DimDw, pr)[:, perm] * sign).reshape(-1)
        return x.view(self.DimDw, pr)[:, : self.DimUp].reshape(-1)

    def apply_device_real(self, v, hv=None, stream=None):
        """Hv on a REAL vector (float64 CUDA tensor, contiguous Dim or padded hxv_realvec_elems)."""
        import torch

        assert v.is_cuda and v.dtype == torch.float64
        n = load_library().hxv_realvec_elems(self._h)
        contiguous = v.numel() != n
        if contiguous:
            assert v.numel() == self.Dim
            v = self.pad_real(v)
        out = hv if hv is not None else torch.zeros(n, dtype=torch.float64, device=v.device)
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        _chk(load_library().hxv_apply_device_real(self._h, v.data_ptr(), out.data_ptr(), st), "hxv_apply_device_real")
        return self.unpad_real(out) if (contiguous and hv is None) else out

    def to_gather_layout(self, v: np.ndarray, nranks: int) -> np.ndarray:
        """Contiguous full vector (Dim) -> the padded all-gather layout hxv_apply_device expects."""
        from .distributed import dw_split

        cmax = -(-self.DimDw // nranks)
        out = np.zeros((nranks * cmax, self.pitch), dtype=np.complex128)
        V = v.reshape(self.DimDw, self.DimUp)
        for r in range(nranks):
            q, c0 = dw_split(self.DimDw, r, nranks)
            if self.row_perm is not None:
                out[r * cmax: r * cmax + q][:, self.row_perm] = V[c0: c0 + q] * self.row_sign
            else:
                out[r * cmax: r * cmax + q, : self.DimUp] = V[c0: c0 + q]
        return out.reshape(-1)

    def time_apply(self, v_full, hv_local, nrep: int) -> float:
        """Mean ms per product over nrep launches, HIP events on the launch stream."""
        import torch

        torch.cuda.synchronize()
        ms = C.c_float()
        _chk(load_library().hxv_time_apply(self._h, v_full.data_ptr(), hv_local.data_ptr(), nrep, C.byref(ms)), "hxv_time_apply")
        return ms.value

    # -- Lanczos ---------------------------------------------------------------------------
    def lanczos_tridiag(self, vin, nlanc: int, threshold: float = 1e-12):
        """sp_lanc_tridiag(MatVec, vin, alanc, blanc): vin = normalised torch CUDA vector (contiguous or padded)."""
        import torch

        assert vin.is_cuda and vin.dtype == torch.complex128
        if vin.numel() != self.localElems:
            assert vin.numel() == self.vecDim
            vin = self.pad(vin, self.mpiQdw)
        torch.cuda.synchronize()
        a = np.zeros(nlanc)
        b = np.zeros(nlanc)
        n = C.c_int32()
        _chk(load_library().hxv_lanczos_tridiag(self._h, vin.data_ptr(), nlanc, _p(a, C.c_double), _p(b, C.c_double), threshold,
                                                C.byref(n)), "hxv_lanczos_tridiag")
        return a, b, n.value

    def lanczos_tridiag_pair(self, vin_a, vin_b, nlanc: int, threshold: float = 1e-12):
        """Two sp_lanc_tridiag runs (two Green's-function channels) on one product, real H: vin_a, vin_b = REAL start vectors in
        the complex layout (torch CUDA, contiguous or padded).  -> (alanc_a, blanc_a, n_a), (alanc_b, blanc_b, n_b)."""
        import torch

        vs = []
        for vin in (vin_a, vin_b):
            assert vin.is_cuda and vin.dtype == torch.complex128
            if vin.numel() != self.localElems:
                assert vin.numel() == self.vecDim
                vin = self.pad(vin, self.mpiQdw)
            vs.append(vin)
        torch.cuda.synchronize()
        aa, ba, ab, bb = (np.zeros(nlanc) for _ in range(4))
        na, nb = C.c_int32(), C.c_int32()
        _chk(load_library().hxv_lanczos_tridiag_pair(self._h, vs[0].data_ptr(), vs[1].data_ptr(), nlanc, _p(aa, C.c_double), _p(ba, C.c_double),
                                                     _p(ab, C.c_double), _p(bb, C.c_double), threshold, C.byref(na), C.byref(nb)),
             "hxv_lanczos_tridiag_pair")
        return (aa, ba, na.value), (ab, bb, nb.value)

    def lanczos_tridiag_pair_host(self, vin_a: np.ndarray, vin_b: np.ndarray, nlanc: int, threshold: float = 1e-12):
        va = np.ascontiguousarray(vin_a, dtype=np.complex128)
        vb = np.ascontiguousarray(vin_b, dtype=np.complex128)
        assert va.size == self.vecDim and vb.size == self.vecDim
        aa, ba, ab, bb = (np.zeros(nlanc) for _ in range(4))
        na, nb = C.c_int32(), C.c_int32()
        _chk(load_library().hxv_lanczos_tridiag_pair_host(self._h, va.ctypes.data, vb.ctypes.data, nlanc, _p(aa, C.c_double), _p(ba, C.c_double),
                                                          _p(ab, C.c_double), _p(bb, C.c_double), threshold, C.byref(na), C.byref(nb)),
             "hxv_lanczos_tridiag_pair_host")
        return (aa, ba, na.value), (ab, bb, nb.value)

    def lanczos_eigh(self, nitermax: int = 512, threshold: float = 1e-12, want_vector: bool = True, native: bool = False):
        """sp_lanc_eigh(MatVec, egs, vect, Nitermax, threshold): lowest eigenpair; the vector comes back
        contiguous (Dim) unless native=True (padded device layout)."""
        import torch

        torch.cuda.synchronize()
        e = C.c_double()
        n = C.c_int32()
        vec = torch.zeros(self.localElems, dtype=torch.complex128, device=self._dev()) if want_vector else None
        _chk(load_library().hxv_lanczos_eigh(self._h, nitermax, threshold, C.byref(e), vec.data_ptr() if want_vector else None,
                                             C.byref(n)), "hxv_lanczos_eigh")
        if want_vector and not native:
            vec = self.unpad(vec)
        return e.value, vec, n.value

    def lanczos_tridiag_host(self, vin: np.ndarray, nlanc: int, threshold: float = 1e-12):
        """sp_lanc_tridiag on a HOST start vector (reference layout): one PCIe copy per run."""
        vin = np.ascontiguousarray(vin, dtype=np.complex128)
        assert vin.size == self.vecDim          # this rank's slab (the whole vector on an unsplit sector)
        a = np.zeros(nlanc)
        b = np.zeros(nlanc)
        n = C.c_int32()
        _chk(load_library().hxv_lanczos_tridiag_host(self._h, vin.ctypes.data, nlanc, _p(a, C.c_double), _p(b, C.c_double), threshold,
                                                     C.byref(n)), "hxv_lanczos_tridiag_host")
        return a, b, n.value

    def lanczos_eigh_host(self, nitermax: int = 512, threshold: float = 1e-12):
        """sp_lanc_eigh with the eigenvector returned in a HOST array (reference layout)."""
        e = C.c_double()
        n = C.c_int32()
        vec = np.zeros(self.vecDim, dtype=np.complex128)
        _chk(load_library().hxv_lanczos_eigh_host(self._h, nitermax, threshold, C.byref(e), vec.ctypes.data, C.byref(n)),
             "hxv_lanczos_eigh_host")
        return e.value, vec, n.value

    def eigh_lowest(self, neigen: int = 1, ncv: int = 0, maxrestart: int = 512, tol: float = 0.0, want_vectors: bool = True,
                    native: bool = False):
        """sp_eigh(MatVec, eig_values, eig_basis, Nblock, Nitermax, tol) on the device (ED_DIAG.f90:152-160): the `neigen`
        lowest eigenpairs by thick-restart Lanczos.  -> (evals[neigen], evecs [neigen, Dim] (or padded if native), nconv, nmatvec)."""
        import torch

        torch.cuda.synchronize()
        ev = np.zeros(neigen)
        nc, nmv = C.c_int32(), C.c_int32()
        vecs = torch.zeros(neigen * self.localElems, dtype=torch.complex128, device=self._dev()) if want_vectors else None
        _chk(load_library().hxv_eigh_lowest(self._h, neigen, ncv, maxrestart, tol, _p(ev, C.c_double),
                                            vecs.data_ptr() if want_vectors else None, C.byref(nc), C.byref(nmv)), "hxv_eigh_lowest")
        if want_vectors:
            vecs = vecs.view(neigen, self.localElems)
            if not native:
                vecs = torch.stack([self.unpad(vecs[i]) for i in range(neigen)])
        return ev, vecs, nc.value, nmv.value

    def eigh_lowest_host(self, neigen: int = 1, ncv: int = 0, maxrestart: int = 512, tol: float = 0.0):
        """Same with eig_basis returned in a HOST array, Fortran shape (Dim, neigen)."""
        ev = np.zeros(neigen)
        nc, nmv = C.c_int32(), C.c_int32()
        basis = np.zeros((self.vecDim, neigen), dtype=np.complex128, order="F")
        _chk(load_library().hxv_eigh_lowest_host(self._h, neigen, ncv, maxrestart, tol, _p(ev, C.c_double), basis.ctypes.data,
                                                 C.byref(nc), C.byref(nmv)), "hxv_eigh_lowest_host")
        return ev, basis, nc.value, nmv.value

    def apply_ladder(self, to: "HxvSector", orbital: int, spin: int, create: bool, psi, coef: complex = 1.0, out=None):
        """c / c^dagger on (orbital, spin) from this sector into `to` (ED_GF_NORMAL.f90:180-199); returns (vector, norm2).
        With `out` (a PADDED vector of `to`, e.g. a previous result with native layout) the term coef * c^(dagger) psi is
        ADDED to it: the mixed channels (c^dagger_i + c^dagger_j)|gs>, (c^dagger_i + xi c^dagger_j)|gs> (:370-406, :746-780)."""
        import torch

        assert psi.is_cuda and psi.dtype == torch.complex128
        contiguous = psi.numel() != self.localElems
        if contiguous:
            assert psi.numel() == self.Dim
            psi = self.pad(psi)
        accumulate = out is not None
        if accumulate:
            assert out.is_cuda and out.dtype == torch.complex128 and out.numel() == to.localElems and out.is_contiguous()
        else:
            out = torch.empty(to.localElems, dtype=torch.complex128, device=psi.device)
        torch.cuda.synchronize()
        n2 = C.c_double()
        cf = complex(coef)
        _chk(load_library().hxv_apply_ladder_axpy(self._h, to._h, orbital, spin, int(bool(create)), cf.real, cf.imag, int(accumulate),
                                                  psi.data_ptr(), out.data_ptr(), C.byref(n2)), "hxv_apply_ladder_axpy")
        return (to.unpad(out) if (contiguous and not accumulate) else out), n2.value

    def time_lanczos(self, nrep: int) -> float:
        import torch

        work = torch.empty(3 * self.localElems, dtype=torch.complex128, device=self._dev())
        torch.cuda.synchronize()
        ms = C.c_float()
        _chk(load_library().hxv_time_lanczos(self._h, work.data_ptr(), nrep, C.byref(ms)), "hxv_time_lanczos")
        return ms.value

    # -- host <-> device copies of a local vector (include/hxv.h: hxv_vector_from_host / _to_host; the device buffer is a torch tensor here)
    def vector_from_host(self, v: np.ndarray):
        """this rank's slab in the reference's host layout -> a device vector in the padded device layout (device row order included)."""
        import torch

        v = np.ascontiguousarray(v, dtype=np.complex128)
        assert v.size == self.vecDim
        d = torch.zeros(self.localElems, dtype=torch.complex128, device=self._dev())
        torch.cuda.synchronize()
        _chk(load_library().hxv_vector_from_host(self._h, v.ctypes.data, d.data_ptr()), "hxv_vector_from_host")
        return d

    def vector_to_host(self, d) -> np.ndarray:
        import torch

        assert d.is_cuda and d.dtype == torch.complex128 and d.numel() == self.localElems
        out = np.empty(self.vecDim, dtype=np.complex128)
        torch.cuda.synchronize()
        _chk(load_library().hxv_vector_to_host(self._h, d.data_ptr(), out.ctypes.data), "hxv_vector_to_host")
        return out

    # -- introspection ---------------------------------------------------------------------
    def maps(self):
        mu = np.zeros(self.DimUp, dtype=np.int32)
        md = np.zeros(self.DimDw, dtype=np.int32)
        _chk(load_library().hxv_get_maps(self._h, _p(mu, C.c_int32), _p(md, C.c_int32)), "hxv_get_maps")
        return mu, md

    def csr(self, which: str):
        w = {"up": 0, "dw": 1}[which]
        n = self.DimUp if w == 0 else self.DimDw
        nnz = load_library().hxv_nnz(self._h, w)
        rp = np.zeros(n + 1, dtype=np.int64)
        cols = np.zeros(max(nnz, 1), dtype=np.int32)
        vals = np.zeros(2 * max(nnz, 1))
        _chk(load_library().hxv_get_csr(self._h, w, _p(rp, C.c_int64), _p(cols, C.c_int32), _p(vals, C.c_double)), "hxv_get_csr")
        return rp, cols[:nnz], vals.view(np.complex128)[:nnz]

    def diag(self) -> np.ndarray:
        d = np.zeros(self.vecDim)
        _chk(load_library().hxv_get_diag(self._h, _p(d, C.c_double)), "hxv_get_diag")
        return d

    def set_option(self, name: str, value: int):
        _chk(load_library().hxv_set_option(self._h, name.encode(), int(value)), f"hxv_set_option({name})")

    def get_option(self, name: str) -> int:
        return load_library().hxv_get_option(self._h, name.encode())

    def stats(self) -> dict:
        s = _Stats()
        _chk(load_library().hxv_get_stats(self._h, C.byref(s)), "hxv_get_stats")
        return {f: getattr(s, f) for f, _ in _Stats._fields_}
