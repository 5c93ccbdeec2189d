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
