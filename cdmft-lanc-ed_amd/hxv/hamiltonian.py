"""Host-side mirror of the reference's ED_HAMILTONIAN public interface for the HxV path
(ED_HAMILTONIAN.f90:10-26): build_Hv_sector / delete_Hv_sector / vecDim_Hv_sector and the
procedure pointer spHtimesV_p.  Same names, argument meaning and error behaviour; the module
globals the reference keeps in ED_VARS_GLOBAL / ED_HAMILTONIAN_COMMON live on an EDContext.
"""
from __future__ import annotations

from math import comb

import numpy as np

from .engine import HxvError, HxvSector
from .models import Model


class EDContext:
    """The slice of module-global state the HxV path touches (ED_HAMILTONIAN_COMMON.f90:11-20,
    ED_VARS_GLOBAL.f90:142-146, :286-306).  One sector open at a time, like the reference."""

    def __init__(self, model: Model, MpiRank: int = 0, MpiSize: int = 1, device: int = 0):
        self.model = model
        self.MpiRank, self.MpiSize = MpiRank, MpiSize
        self.MpiStatus = MpiSize > 1
        self.device = device
        self.Hstatus = False          # ED_HAMILTONIAN_COMMON.f90:18
        self.Hsector = 0              # ED_HAMILTONIAN_COMMON.f90:17
        self.spHtimesV_p = None       # ED_VARS_GLOBAL.f90:146
        self._sector: HxvSector | None = None
        self.Dim = self.DimUp = self.DimDw = 0
        self.mpiQdw = self.mpiIshift = 0

    # ED_HAMILTONIAN.f90:39-143
    def build_Hv_sector(self, isector: int) -> None:
        if self.Hstatus:
            raise HxvError("build_Hv_sector ERROR: a sector is already open (delete_Hv_sector first)")
        m = self.model
        if not (1 <= isector <= m.Nsectors):
            raise HxvError("build_Hv_sector ERROR: isector out of range")
        nup, ndw = m.get_Nup(isector), m.get_Ndw(isector)
        dimdw = comb(m.Ns, ndw)
        rank, size = self.MpiRank, self.MpiSize
        if self.MpiStatus and dimdw < size:
            # communicator shrink, ED_HAMILTONIAN.f90:63-89: excess ranks sit this sector out
            if rank >= dimdw:
                self.Hsector, self.Hstatus = isector, True
                self._sector, self.spHtimesV_p = None, None
                self.Dim = comb(m.Ns, nup) * dimdw
                self.DimUp, self.DimDw, self.mpiQdw, self.mpiIshift = comb(m.Ns, nup), dimdw, 0, 0
                return
            size = dimdw
        self._sector = HxvSector.from_model(m, nup, ndw, rank, size, self.device)
        s = self._sector
        self.Hsector, self.Hstatus = isector, True
        self.Dim, self.DimUp, self.DimDw = s.Dim, s.DimUp, s.DimDw
        self.mpiQdw, self.mpiIshift = s.mpiQdw, s.mpiIshift
        self.spHtimesV_p = self._spHtimesV        # ED_HAMILTONIAN.f90:129-141 (pointer binding)

    # ED_HAMILTONIAN.f90:149-190
    def delete_Hv_sector(self) -> None:
        if self._sector is not None:
            self._sector.close()
        self._sector = None
        self.Hsector, self.Hstatus = 0, False
        self.spHtimesV_p = None

    # ED_HAMILTONIAN.f90:197-221
    def vecDim_Hv_sector(self, isector: int) -> int:
        m = self.model
        dimup, dimdw = comb(m.Ns, m.get_Nup(isector)), comb(m.Ns, m.get_Ndw(isector))
        if self.MpiStatus:
            size = min(self.MpiSize, dimdw)
            if self.MpiRank >= size:
                return 0
            q = dimdw // size + (1 if self.MpiRank < dimdw % size else 0)
        else:
            q = dimdw
        return dimup * q

    @property
    def sector(self) -> HxvSector:
        if self._sector is None:
            raise HxvError("no sector open on this rank")
        return self._sector

    # cc_sparse_HxV (ED_VARS_GLOBAL.f90:72-78): Hv is overwritten
    def _spHtimesV(self, Nloc: int, v, Hv):
        if not self.Hstatus or self._sector is None:
            raise HxvError("spHtimesV_p ERROR: Hsector NOT set")      # ED_HAMILTONIAN_SPARSE_HxV.f90:57
        if isinstance(v, np.ndarray):
            if self.MpiStatus:
                raise HxvError("spMatVec_mpi_cc ERROR: host arrays with MpiStatus=T; use device tensors + allgather")
            if Nloc != self._sector.Dim:
                raise HxvError("spHtimesV_p ERROR: Nloc /= Dim")
            self._sector.apply_host(v, Hv)
        else:
            self._sector.apply_device(v, Hv)
        return Hv
