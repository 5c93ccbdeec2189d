"""torch.distributed TWIN of the engine's three slab exchanges -- REHEARSAL CODE, not the product's N>1 path.

Since round 3 the exchanges live behind the C-ABI (csrc/hxv_comm.cpp: hxv_comm_init + hxv_apply_device_slab, RCCL or thread ranks), and
since round 4 `bench.py --gpus N` uses nothing else on its default `--backend nccl` path.  This module stays for `bench.py --backend gloo`
and tests/test_distributed_gloo.py: the partitioning, the equal-count all-gather layout, the halo plan and the two transposes exercised at
world sizes 2-4 on the CPU, where no GPU (and no RCCL) is needed.  A change of the exchange logic is made in hxv_comm.cpp and mirrored here.

DimDw-sharded HxV across the GPUs of one node: one process per GPU, torch.distributed as the plumbing.

Partition = the reference's own (ED_HAMILTONIAN.f90:93-105): rank r owns mpiQdw consecutive dw
columns.  The reference reassembles with two MPI_AllToAllV transposes per product
(ED_HAMILTONIAN_SPARSE_HxV.f90:279,294); here every product all-gathers the slabs into a full
replica of v on each GPU (the exchange BASELINE.json mandates) and then computes its slab of Hv
locally with no second exchange.  Unequal slabs (DimDw % P != 0) are gathered straight into the
contiguous full vector through per-rank views of different length."""
from __future__ import annotations

from math import comb


def dw_split(DimDw: int, rank: int, size: int):
    """(mpiQdw, first column) of `rank` -- ED_HAMILTONIAN.f90:93-105."""
    q, rem = divmod(DimDw, size)
    return q + (1 if rank < rem else 0), rank * q + min(rank, rem)


class ShardedHxv:
    """spHtimesV_p for MpiStatus=T on device tensors: Hv_local = (H v)_slab.

    Every product all-gathers the slabs with ONE equal-count collective (all_gather_into_tensor ->
    ncclAllGather on RCCL): each rank contributes cmax = ceil(DimDw/P) columns, ranks that own one
    column less pad with one unused column.  The gathered buffer is used as is -- the kernels address
    it through the engine's column->slot table (include/hxv.h, hxv_apply_device), so there is no
    compaction copy.  apply_local(v_gathered, hv_local) is the per-rank slab product
    (HxvSector.apply_device on the GPU box; CPU tests inject a stand-in to exercise the exchange
    with gloo)."""

    def __init__(self, DimUp: int, DimDw: int, rank: int, size: int, apply_local, group=None, pitch: int | None = None):
        """pitch = device column pitch (HxvSector.pitch); vectors are [columns x pitch] (default: DimUp, unpadded)."""
        import torch.distributed as dist

        self.dist = dist
        self.DimUp, self.DimDw, self.rank, self.size, self.group = DimUp, DimDw, rank, size, group
        self.pitch = DimUp if pitch is None else pitch
        self.qdw, self.dw0 = dw_split(DimDw, rank, size)
        self.cmax = -(-DimDw // size)
        self.Nloc = self.qdw * self.pitch        # local vector length in the device layout
        self.slab = self.cmax * self.pitch       # elements every rank contributes
        self.apply_local = apply_local
        self._vfull = None
        self._send = None

    def gather(self, v_local):
        """allgather_vector_MPI (ED_SETUP.f90:672-708), equal counts, into the padded layout."""
        import torch

        assert v_local.numel() == self.Nloc
        if self.size == 1:
            return v_local  # the slab is the whole vector: no exchange, no copy
        if self._vfull is None or self._vfull.device != v_local.device or self._vfull.dtype != v_local.dtype:
            self._vfull = torch.zeros(self.size * self.slab, dtype=v_local.dtype, device=v_local.device)
        send = v_local
        if self.Nloc != self.slab:                # this rank owns one column less: pad the send buffer
            if self._send is None or self._send.device != v_local.device:
                self._send = torch.zeros(self.slab, dtype=v_local.dtype, device=v_local.device)
            self._send[: self.Nloc].copy_(v_local)
            send = self._send
        # the collective runs on the float64 view: RCCL/NCCL has no complex datatype, and a plain byte-for-byte
        # gather is all that is needed
        out_r = torch.view_as_real(self._vfull).view(-1) if self._vfull.is_complex() else self._vfull
        in_r = torch.view_as_real(send.contiguous()).view(-1) if send.is_complex() else send.contiguous()
        self.dist.all_gather_into_tensor(out_r, in_r, group=self.group)
        return self._vfull

    def unpad(self, v_gathered):
        """padded all-gather layout -> contiguous full vector (tests / debugging only)."""
        import torch

        if self.size == 1:
            return v_gathered
        parts = []
        for r in range(self.size):
            q, _ = dw_split(self.DimDw, r, self.size)
            part = v_gathered[r * self.slab: r * self.slab + q * self.pitch]
            parts.append(part.view(q, self.pitch)[:, : self.DimUp].reshape(-1))
        return torch.cat(parts)

    def __call__(self, Nloc: int, v_local, hv_local):
        if Nloc != self.Nloc:
            raise ValueError("spMatVec_mpi_cc ERROR: Nloc /= DimUp*mpiQdw")
        return self.apply_local(self.gather(v_local), hv_local)


def halo_plan(rowptr, cols, DimDw: int, size: int):
    """Halo exchange plan from the one-spin matrix H_dw (CSR, 0-based cols) and the reference's DimDw split:
    need[r] = sorted global columns of OTHER ranks that the rows owned by rank r reference (what r receives, in slot
    order; ascending = grouped by owner), send[r][p] = LOCAL column indices rank r sends to rank p.  Pure numpy: the same
    plan the engine derives in C++ (hxv_halo_lists), used by the CPU tests and for the exchange-volume report."""
    import numpy as np

    rowptr = np.asarray(rowptr)
    cols = np.asarray(cols)
    owner = np.empty(DimDw, dtype=np.int64)
    first = []
    for r in range(size):
        q, c0 = dw_split(DimDw, r, size)
        owner[c0:c0 + q] = r
        first.append(c0)
    first.append(DimDw)
    need = []
    for r in range(size):
        src = np.unique(cols[rowptr[first[r]]:rowptr[first[r + 1]]])
        need.append(src[owner[src] != r])
    send = [[(need[p][owner[need[p]] == r] - first[r]) for p in range(size)] for r in range(size)]
    return need, send


def exchange_ingest_bytes(DimUp: int, DimDw: int, size: int, need=None, elem_bytes: int = 16):
    """Bytes one GPU receives per product (largest over the ranks) for the three exchanges of DESIGN.md section 4."""
    slab = -(-DimDw // size)
    out = {"allgather": (size - 1) * slab * DimUp * elem_bytes,
           "alltoall": 2 * (size - 1) * slab * (-(-DimUp // size)) * elem_bytes}
    if need is not None:
        out["halo"] = max(len(n) for n in need) * DimUp * elem_bytes
    return out


class HaloHxv:
    """spHtimesV_p for MpiStatus=T with the HALO exchange: each product moves only the columns of other ranks that
    H_dw couples to this rank's rows (one all_to_all_single with per-peer counts; RCCL send/recv underneath), instead
    of all-gathering every slab.  The gathered buffer has the engine's halo layout (include/hxv.h): the local slab
    first, then the received columns in ascending order; apply_local(v_halo, hv_local) is the per-rank product
    (HxvSector.apply_device of a handle created in halo mode; CPU tests inject a stand-in)."""

    def __init__(self, DimUp: int, DimDw: int, rank: int, size: int, need, send, apply_local, group=None, pitch: int | None = None,
                 stage_on_host: bool = False):
        import torch.distributed as dist

        self.dist, self.group = dist, group
        self.stage_on_host = stage_on_host   # rehearsals with gloo on CUDA tensors: run the collective on host copies
        self.DimUp, self.DimDw, self.rank, self.size = DimUp, DimDw, rank, size
        self.pitch = DimUp if pitch is None else pitch
        self.qdw, self.dw0 = dw_split(DimDw, rank, size)
        self.Nloc = self.qdw * self.pitch
        self.need = need[rank]                       # global columns received, slot order
        self.send = send[rank]                       # per destination: local column indices
        owner_first = [dw_split(DimDw, r, size)[1] for r in range(size)] + [DimDw]
        self.recv_counts = [int(((self.need >= owner_first[r]) & (self.need < owner_first[r + 1])).sum()) for r in range(size)]
        self.send_counts = [len(self.send[p]) for p in range(size)]
        self.apply_local = apply_local
        self._full = None
        self._idx = None

    @property
    def ingest_columns(self) -> int:
        return len(self.need)

    def exchange(self, v_local):
        import torch

        assert v_local.numel() == self.Nloc
        nfull = (self.qdw + len(self.need)) * self.pitch
        if self._full is None or self._full.device != v_local.device:
            self._full = torch.zeros(nfull, dtype=v_local.dtype, device=v_local.device)
            cat = [torch.as_tensor(x, dtype=torch.long) for x in self.send if len(x)]
            self._idx = (torch.cat(cat) if cat else torch.zeros(0, dtype=torch.long)).to(v_local.device)
        self._full[: self.Nloc].copy_(v_local)
        if self.size > 1:
            packed = v_local.view(self.qdw, self.pitch)[self._idx].contiguous().view(-1)
            out = self._full[self.Nloc:]
            o = torch.view_as_real(out).view(-1) if out.is_complex() else out
            i = torch.view_as_real(packed).view(-1) if packed.is_complex() else packed
            k = 2 * self.pitch if out.is_complex() else self.pitch
            osp, isp = [c * k for c in self.recv_counts], [c * k for c in self.send_counts]
            if self.stage_on_host and o.is_cuda:
                oc = torch.empty(o.shape, dtype=o.dtype)
                self.dist.all_to_all_single(oc, i.cpu(), osp, isp, group=self.group)
                o.copy_(oc)
            else:
                self.dist.all_to_all_single(o, i, osp, isp, group=self.group)
        return self._full

    def __call__(self, Nloc: int, v_local, hv_local):
        if Nloc != self.Nloc:
            raise ValueError("spMatVec_mpi_cc ERROR: Nloc /= DimUp*mpiQdw")
        return self.apply_local(self.exchange(v_local), hv_local)


class TransposedHxv:
    """The reference's own exchange (two all-to-all transposes per product, ED_HAMILTONIAN_SPARSE_HxV.f90:272-296,
    ED_HAMILTONIAN_COMMON.f90:30-94) on device tensors -- the lower-traffic alternative to ShardedHxv's all-gather:
    each transpose moves (P-1)/P of ONE slab per rank instead of (P-1) slabs.

      1. all-to-all #1: rank r receives the row panel  X = v[rows U_r, ALL columns]   (rows split like the columns)
      2. Y = X H_dw^T                                   (apply_panel: HxvSector.apply_dw_panel on a dw_panel handle)
      3. all-to-all #2: Y's column ranges go back to their owners -> W = (v H_dw^T)[all rows, my columns]
      4. hv = D.v + H_up v + W on the local slab        (apply_up_add: HxvSector.apply_up_add)

    Vectors are [columns x pitch]; panels are [DimDw x pitch_panel].  The collectives run on float64 views."""

    def __init__(self, DimUp, DimDw, rank, size, apply_panel, apply_up_add, pitch=None, pitch_panel=None, group=None,
                 stage_on_host=False):
        import torch.distributed as dist

        self.dist, self.group = dist, group
        self.DimUp, self.DimDw, self.rank, self.size = DimUp, DimDw, rank, size
        self.pitch = DimUp if pitch is None else pitch
        self.cols = [dw_split(DimDw, r, size) for r in range(size)]     # (q_r, c0_r)
        self.rows = [dw_split(DimUp, r, size) for r in range(size)]     # (n_r, u0_r): mpiQup rule, :274-275
        self.qdw, self.dw0 = self.cols[rank]
        self.nrows, self.u0 = self.rows[rank]
        self.pitch_panel = self.nrows if pitch_panel is None else pitch_panel
        self.Nloc = self.qdw * self.pitch
        self.apply_panel, self.apply_up_add = apply_panel, apply_up_add
        self.stage_on_host = stage_on_host   # rehearsals with gloo on CUDA tensors: run the collective on host copies

    def _a2a(self, out, inp, out_split, in_split):
        import torch

        o = torch.view_as_real(out).view(-1)
        i = torch.view_as_real(inp).view(-1)
        os_, is_ = [2 * x for x in out_split], [2 * x for x in in_split]
        if self.stage_on_host and o.is_cuda:
            oc, ic = torch.empty(o.shape, dtype=o.dtype), i.cpu()
            self.dist.all_to_all_single(oc, ic, os_, is_, group=self.group)
            o.copy_(oc)
        else:
            self.dist.all_to_all_single(o, i, os_, is_, group=self.group)

    def __call__(self, Nloc, v_local, hv_local):
        import torch

        if Nloc != self.Nloc:
            raise ValueError("spMatVec_mpi_cc ERROR: Nloc /= DimUp*mpiQdw")
        P, q, n = self.size, self.qdw, self.nrows
        v2 = v_local.view(q, self.pitch)
        # 1. my slab, cut by the row ranges of the receivers
        send = torch.cat([v2[:, u0:u0 + ns].reshape(-1) for ns, u0 in self.rows])
        recv = torch.empty(self.DimDw * n, dtype=v_local.dtype, device=v_local.device)
        self._a2a(recv, send, [qs * n for qs, _ in self.cols], [q * ns for ns, _ in self.rows])
        x = torch.zeros(self.DimDw, self.pitch_panel, dtype=v_local.dtype, device=v_local.device)
        off = 0
        for qs, c0 in self.cols:
            x[c0:c0 + qs, :n] = recv[off:off + qs * n].view(qs, n)
            off += qs * n
        # 2. dw hops on the row panel
        y = self.apply_panel(x.view(-1)).view(self.DimDw, self.pitch_panel)
        # 3. back to the column owners
        send = torch.cat([y[c0:c0 + qs, :n].reshape(-1) for qs, c0 in self.cols])
        recv = torch.empty(q * self.DimUp, dtype=v_local.dtype, device=v_local.device)
        self._a2a(recv, send, [q * ns for ns, _ in self.rows], [qs * n for qs, _ in self.cols])
        w = torch.zeros(q, self.pitch, dtype=v_local.dtype, device=v_local.device)
        off = 0
        for ns, u0 in self.rows:
            w[:, u0:u0 + ns] = recv[off:off + q * ns].view(q, ns)
            off += q * ns
        # 4. diagonal + up hops + dw part on the local slab
        return self.apply_up_add(v_local, w.view(-1), hv_local)


def start_vector_slab(DimUp: int, qdw: int, dw0: int, pitch: int, device, seed: int = 0x5EED5EED):
    """This rank's slab of the engine's deterministic Lanczos start vector (csrc/hxv_lanczos.hip lz_init): splitmix64 of
    the GLOBAL element index -> uniform(-0.5,0.5) re and im, pad rows zero.  Every split of the sector therefore
    starts from the same vector as the single-GPU driver."""
    import torch

    M64 = (1 << 64) - 1

    def as_i64(x):  # python int (mod 2^64) -> the int64 with the same bits
        x &= M64
        return x - (1 << 64) if x >= (1 << 63) else x

    def lsr(x, s):  # logical shift right on int64 tensors
        return (x >> s) & ((1 << (64 - s)) - 1)

    col = torch.arange(dw0, dw0 + qdw, dtype=torch.int64, device=device).view(-1, 1)
    row = torch.arange(DimUp, dtype=torch.int64, device=device).view(1, -1)
    z = (col * DimUp + row) * 2 + seed
    out = torch.zeros(qdw, pitch, dtype=torch.complex128, device=device)
    parts = []
    for k in range(2):
        x = z + k + as_i64(0x9E3779B97F4A7C15)
        x = (x ^ lsr(x, 30)) * as_i64(0xBF58476D1CE4E5B9)
        x = (x ^ lsr(x, 27)) * as_i64(0x94D049BB133111EB)
        x = x ^ lsr(x, 31)
        parts.append(lsr(x, 11).to(torch.float64) * (1.0 / 9007199254740992.0) - 0.5)
    out[:, :DimUp] = torch.complex(parts[0], parts[1])
    return out.view(-1)


class ShardedLanczos:
    """The Lanczos recurrences for a DimDw-split sector: what SciFortran's MPI flavours
    sp_lanc_tridiag(MpiComm, MatVec, vin, alanc, blanc) / sp_lanc_eigh(MpiComm, MatVec, egs, vect, Nitermax, ...) do at
    ED_GF_NORMAL.f90:215 / ED_DIAG.f90:176 when MpiStatus=T: every rank holds its slab of each Lanczos vector, the product
    is `matvec` (ShardedHxv or TransposedHxv: one exchange per product), the two dot products per iteration are
    all-reduced over the group (RCCL on GPUs).  Vector arithmetic is plain torch on the slabs (pad rows stay zero)."""

    def __init__(self, matvec, group=None):
        import torch.distributed as dist

        self.mv, self.dist, self.group = matvec, dist, group
        self.Nloc = matvec.Nloc

    def _allsum(self, x: float, like) -> float:
        import torch

        t = torch.tensor([x], dtype=torch.float64, device=like.device)
        if self.mv.size > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return float(t.item())

    def _dot(self, a, b) -> float:
        import torch

        return self._allsum(torch.vdot(a, b).real.item(), a)

    def _recurrence(self, q, nmax, threshold, on_step=None):
        """three-term recurrence from the normalised slab q -> (alpha_1..n, [0, beta_2..n]); on_step(k, q_k) sees every
        Lanczos vector (then the last product is not needed and is skipped)"""
        import torch

        qm = torch.zeros_like(q)
        w = torch.zeros_like(q)      # pad rows must be zero: the product never writes them, the dots read them
        al, be = [], [0.0]
        for k in range(nmax):
            if on_step is not None:
                on_step(k, q)
                if k + 1 == nmax:
                    break
            self.mv(self.Nloc, q, w)
            if k > 0:
                w.sub_(qm, alpha=be[-1])
            a = self._dot(q, w)
            w.sub_(q, alpha=a)
            b = self._dot(w, w) ** 0.5
            al.append(a)
            if b < threshold or k + 1 == nmax:
                break
            be.append(b)
            qm, q, w = q, w.div_(b), qm
        return al, be

    def tridiag(self, v_local, nlanc: int, threshold: float = 1e-12):
        """-> (alanc[nlanc], blanc[nlanc] with blanc[0] unused, nsteps): vin = this rank's slab of a GLOBALLY normalised vector."""
        import numpy as np

        al, be = self._recurrence(v_local.clone(), nlanc, threshold)
        a = np.zeros(nlanc)
        b = np.zeros(nlanc)
        a[: len(al)] = al
        b[: len(be)] = be
        return a, b, len(al)

    def eigh(self, nitermax: int = 512, threshold: float = 1e-12, want_vector: bool = True, device="cpu"):
        """-> (E0, slab of the normalised ground-state vector or None, iterations): two passes like the single-GPU driver
        (recurrence, then re-run to accumulate the Ritz vector), same deterministic start vector, same stopping rule."""
        import numpy as np
        import torch
        from scipy.linalg import eigh_tridiagonal

        mv = self.mv
        q0 = self.start_slab(device)
        nrm = self._dot(q0, q0) ** 0.5
        q0 = q0 / nrm
        state = {"e_old": 1e300}
        # pass 1: the recurrence with the driver's stopping rule evaluated after every step
        nmax = int(min(nitermax, mv.DimUp * mv.DimDw))
        al, be, e_new = [], [0.0], 0.0
        qm = torch.zeros_like(q0)
        q = q0.clone()
        w = torch.zeros_like(q0)     # (pad rows zero, see _recurrence)
        k = 0
        for k in range(nmax):
            mv(self.Nloc, q, w)
            if k > 0:
                w.sub_(qm, alpha=be[-1])
            a = self._dot(q, w)
            w.sub_(q, alpha=a)
            b = self._dot(w, w) ** 0.5
            al.append(a)
            d = np.array(al)
            e = np.array(be[1:])
            ev, Z = eigh_tridiagonal(d, e) if len(al) > 1 else (d.copy(), np.ones((1, 1)))
            e_new = float(ev[0])
            conv = abs(e_new - state["e_old"]) < threshold
            state["e_old"] = e_new
            if conv and want_vector:
                conv = abs(b * Z[-1, 0]) < 1e-11 * max(1.0, abs(e_new))
            if conv or b < 1e-14 or k + 1 == nmax:
                break
            be.append(b)
            qm, q, w = q, w.div_(b), qm
        niter = len(al)
        if not want_vector:
            return e_new, None, niter
        d = np.array(al)
        ev, Z = eigh_tridiagonal(d, np.array(be[1:])) if niter > 1 else (d.copy(), np.ones((1, 1)))
        y = Z[:, 0]
        out = torch.zeros_like(q0)

        def acc(kk, qk):
            out.add_(qk, alpha=float(y[kk]))

        self._recurrence(q0.clone(), niter, 0.0, acc)
        out.div_(self._dot(out, out) ** 0.5)
        return e_new, out, niter

    def start_slab(self, device="cpu"):
        mv = self.mv
        return start_vector_slab(mv.DimUp, mv.qdw, mv.dw0, mv.pitch, device)


def _keep_count(m: int, neigen: int, nconv: int) -> int:
    """Ritz vectors kept at a thick restart (same rule as csrc/hxv_eigh.hip)."""
    k = neigen + min(nconv, (m - neigen) // 2) + max(1, (m - neigen) // 4)
    return max(1, min(k, m - 1))


def sharded_eigh_lowest(matvec, neigen: int = 1, ncv: int = 0, maxrestart: int = 512, tol: float = 0.0, device="cpu", group=None):
    """sp_eigh(MpiComm, MatVec, eval, evec, Nblock, Nitermax, tol) for a DimDw-split sector (the P-ARPACK call of
    ED_DIAG.f90:152-156 when MpiStatus=T): the thick-restart Lanczos of csrc/hxv_eigh.hip on slabs -- every rank keeps its
    slab of each basis vector, projections are all-reduced, the small projected problem is solved redundantly on every
    rank.  `matvec` is a ShardedHxv / TransposedHxv.  -> (evals[neigen], slabs [neigen, Nloc], nconv, nmatvec)."""
    import numpy as np
    import torch
    import torch.distributed as dist

    mv = matvec
    dim = mv.DimUp * mv.DimDw
    neigen = min(neigen, dim)
    if ncv <= 0:
        ncv = 10 * neigen
    m = int(min(max(ncv, neigen + 1), dim))
    eps = float(np.finfo(float).eps)
    tol = max(tol, eps)
    eps23 = eps ** (2.0 / 3.0)

    def allsum(t):
        if mv.size > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t

    def dots(Vj, w):      # <V_i, w> for the stacked slabs, all-reduced (complex via its float64 view)
        c = Vj.conj() @ w
        allsum(torch.view_as_real(c))
        return c

    V = torch.zeros(m + 1, mv.Nloc, dtype=torch.complex128, device=device)
    q0 = start_vector_slab(mv.DimUp, mv.qdw, mv.dw0, mv.pitch, device)
    V[0] = q0 / float(allsum(torch.vdot(q0, q0).real.reshape(1).clone()).item()) ** 0.5
    T = np.zeros((m, m))
    w = torch.zeros(mv.Nloc, dtype=torch.complex128, device=device)
    k, nmv, nconv, meff, beta_last = 0, 0, 0, m, 0.0
    theta = S = None
    for it in range(maxrestart + 1):
        meff, beta_last = m, 0.0
        for j in range(k, m):
            mv(mv.Nloc, V[j], w)
            nmv += 1
            c = dots(V[: j + 1], w)
            ch = c.cpu().numpy()
            T[j, j] = ch[j].real
            sel = np.abs(ch) > 1e-13 * np.sqrt(np.vdot(ch, ch).real)      # measured Gram-Schmidt, selective update
            sel[max(j - 1, 0):] = True
            if j == k:
                sel[:] = True
            cs = torch.from_numpy(ch * sel).to(device)
            w.sub_(cs @ V[: j + 1])
            nrm = float(allsum(torch.vdot(w, w).real.reshape(1).clone()).item()) ** 0.5
            if nrm * nrm < 0.01 * (np.vdot(ch, ch).real + nrm * nrm):         # norm dropped 10x: one refinement pass
                c2 = dots(V[: j + 1], w)
                T[j, j] += float(c2[j].real.item())
                w.sub_(c2 @ V[: j + 1])
                nrm2 = float(allsum(torch.vdot(w, w).real.reshape(1).clone()).item()) ** 0.5
                nrm = 0.0 if nrm2 < 0.5 * nrm else nrm2
            if nrm <= 1e-13 * max(1.0, np.abs(T[: j + 1, : j + 1]).max()):
                meff, beta_last = j + 1, 0.0
                break
            if j + 1 < m:
                T[j + 1, j] = T[j, j + 1] = nrm
            beta_last = nrm
            V[j + 1] = w / nrm
        theta, S = np.linalg.eigh(T[:meff, :meff])
        ne = min(neigen, meff)
        res = np.abs(beta_last * S[meff - 1, :])
        nconv = int((res[:ne] <= tol * np.maximum(eps23, np.abs(theta[:ne]))).sum())
        if nconv == ne or meff < m or it == maxrestart:
            break
        k = _keep_count(m, neigen, nconv)
        St = torch.from_numpy(np.ascontiguousarray(S[:, :k].T)).to(device=device, dtype=torch.complex128)
        V[:k] = St @ V[:m]
        V[k] = V[m]
        T[:] = 0.0
        for i in range(k):
            T[i, i] = theta[i]
            T[k, i] = T[i, k] = beta_last * S[m - 1, i]
    ne = min(neigen, meff)
    St = torch.from_numpy(np.ascontiguousarray(S[:, :ne].T)).to(device=device, dtype=torch.complex128)
    X = St @ V[:meff]
    return theta[:ne].copy(), X, nconv, nmv
