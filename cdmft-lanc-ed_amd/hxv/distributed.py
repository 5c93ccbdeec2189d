"""DimDw-sharded HxV across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI) as the plumbing.

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
