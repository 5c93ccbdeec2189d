"""Synthetic operator inputs for the sector HxV path.

These are *inputs* (one-body matrices, bath parameters, interaction constants) in the
reference's own array conventions, so the same arrays can be handed to the HIP engine
(through the C-ABI) and to the CPU oracle.  They mirror the model definitions of the
reference drivers; nothing here computes the Hamiltonian.

Array conventions (reference, Fortran order):
  impHloc (Nlat,Nlat,Nspin,Nspin,Norb,Norb)        ED_VARS_GLOBAL.f90:119
  Hbath   (Nlat,Nlat,Nspin,Nspin,Norb,Norb,Nbath)  ED_HAMILTONIAN_SPARSE_HxV.f90:50,65
  Vbath   (Nlat,Nspin,Norb,Nbath)   = diag_hybr    ED_HAMILTONIAN_SPARSE_HxV.f90:48,70
"""
from __future__ import annotations

from dataclasses import dataclass, field
from math import comb, sqrt

import numpy as np


@dataclass
class Model:
    """The module-global operator inputs visible to build_Hv_sector at bind time
    (SURVEY.md 8b): ED_INPUT_VARS.f90:13-16,129-135,164 + impHloc + dmft_bath."""

    Nlat: int
    Norb: int
    Nspin: int
    Nbath: int
    impHloc: np.ndarray
    Hbath: np.ndarray
    Vbath: np.ndarray
    Uloc: np.ndarray = field(default_factory=lambda: np.array([2.0, 0, 0, 0, 0]))
    Ust: float = 0.0
    Jh: float = 0.0
    Jx: float = 0.0
    Jp: float = 0.0
    xmu: float = 0.0
    hfmode: bool = True
    name: str = "model"

    def __post_init__(self):
        L, S, O, B = self.Nlat, self.Nspin, self.Norb, self.Nbath
        self.impHloc = np.asfortranarray(np.asarray(self.impHloc, dtype=np.complex128).reshape((L, L, S, S, O, O), order="F"))
        self.Hbath = np.asfortranarray(np.asarray(self.Hbath, dtype=np.complex128).reshape((L, L, S, S, O, O, B), order="F"))
        self.Vbath = np.asfortranarray(np.asarray(self.Vbath, dtype=np.float64).reshape((L, S, O, B), order="F"))
        u = np.zeros(5)
        u[: len(np.atleast_1d(self.Uloc))] = np.atleast_1d(self.Uloc)[:5]
        self.Uloc = u

    # ED_SETUP.f90:111-120
    @property
    def Nimp(self) -> int:
        return self.Nlat * self.Norb

    @property
    def Ns(self) -> int:
        return self.Nimp * (self.Nbath + 1)

    @property
    def Nsectors(self) -> int:
        return (self.Ns + 1) ** 2

    # ED_SETUP.f90:446-457 get_Sector([nup,ndw],Ns,isector)
    def get_Sector(self, nup: int, ndw: int) -> int:
        return 1 + ndw + nup * (self.Ns + 1)

    # ED_SETUP.f90:477-500
    def get_Nup(self, isector: int) -> int:
        return (isector - 1) // (self.Ns + 1)

    def get_Ndw(self, isector: int) -> int:
        return (isector - 1) % (self.Ns + 1)

    def getDim(self, isector: int) -> int:
        return comb(self.Ns, self.get_Nup(isector)) * comb(self.Ns, self.get_Ndw(isector))


def _zeros(Nlat, Nspin, Norb, Nbath):
    h = np.zeros((Nlat, Nlat, Nspin, Nspin, Norb, Norb), dtype=np.complex128, order="F")
    hb = np.zeros((Nlat, Nlat, Nspin, Nspin, Norb, Norb, Nbath), dtype=np.complex128, order="F")
    v = np.zeros((Nlat, Nspin, Norb, Nbath), order="F")
    return h, hb, v


def _square_hloc(Nx, Ny, Nspin, Norb, ts):
    """drivers/cdn_hm_2dsquare.f90:221-259 hloc_model; site index :311 indices2N."""
    Nlat = Nx * Ny
    h = np.zeros((Nlat, Nlat, Nspin, Nspin, Norb, Norb), dtype=np.complex128, order="F")
    i2n = lambda ix, iy: Nx * (iy - 1) + ix - 1
    for s in range(Nspin):
        for o in range(Norb):
            for ix in range(1, Nx + 1):
                for iy in range(1, Ny + 1):
                    a = i2n(ix, iy)
                    if ix < Nx:
                        h[a, i2n(ix + 1, iy), s, s, o, o] = -ts
                    if ix > 1:
                        h[a, i2n(ix - 1, iy), s, s, o, o] = -ts
                    if iy < Ny:
                        h[a, i2n(ix, iy + 1), s, s, o, o] = -ts
                    if iy > 1:
                        h[a, i2n(ix, iy - 1), s, s, o, o] = -ts
    return h


def _lso_eye(Nlat, Nspin, Norb):
    """lso2nnn(zeye(Nlso)): identity in (lat,spin,orb)."""
    e = np.zeros((Nlat, Nlat, Nspin, Nspin, Norb, Norb), dtype=np.complex128, order="F")
    for l in range(Nlat):
        for s in range(Nspin):
            for o in range(Norb):
                e[l, l, s, s, o, o] = 1.0
    return e


def plaquette_2x2_nobath(U=4.0, t=1.0, hfmode=False, xmu=0.0) -> Model:
    """BASELINE config C1: 2x2 Hubbard plaquette, no bath.
    Known answer: sector (2,2) E0 = -2.10274848 at U=4,t=1,hfmode=F."""
    h, hb, v = _zeros(4, 1, 1, 0)
    h[...] = _square_hloc(2, 2, 1, 1, t)
    return Model(4, 1, 1, 0, h, hb, v, Uloc=np.array([U]), hfmode=hfmode, xmu=xmu, name="C1_plaquette_2x2_nobath")


def hm_1dchain(Nlat=4, Nbath=2, ts=0.25, U=2.0, hfmode=True, xmu=0.0, eps_bath=None) -> Model:
    """BASELINE config C2 (cdn_hm_1dchain): open chain drivers/cdn_hm_1dchain.f90:162-181;
    bath basis :76-83: Hsym1=|Hloc| lambda=-1, Hsym2=1 lambda=0; V=1/sqrt(Nbath)
    (ED_BATH/dmft_aux.f90:70).  eps_bath (optional, per replica) sets non-zero bath
    on-site energies to exercise the bath diagonal."""
    h, hb, v = _zeros(Nlat, 1, 1, Nbath)
    for i in range(Nlat):
        if i > 0:
            h[i, i - 1, 0, 0, 0, 0] = -ts
        if i < Nlat - 1:
            h[i, i + 1, 0, 0, 0, 0] = -ts
    eye = _lso_eye(Nlat, 1, 1)
    for ib in range(Nbath):
        e = 0.0 if eps_bath is None else float(eps_bath[ib])
        hb[..., ib] = -1.0 * np.abs(h) + e * eye
        v[..., ib] = max(0.1, 1.0 / sqrt(Nbath))
    return Model(Nlat, 1, 1, Nbath, h, hb, v, Uloc=np.array([U]), hfmode=hfmode, xmu=xmu, name=f"C2_hm_1dchain_L{Nlat}_B{Nbath}")


def hm_2dsquare(Nx=2, Ny=2, Nbath=3, ts=0.25, U=2.0, hwband=2.0, hfmode=True, xmu=0.0) -> Model:
    """BASELINE config C3 (cdn_hm_2dsquare): drivers/cdn_hm_2dsquare.f90:94-108:
    Hsym1 = identity with lambda equispaced in [-HWBAND,HWBAND], Hsym2=|Hloc| lambda=1."""
    Nlat = Nx * Ny
    h, hb, v = _zeros(Nlat, 1, 1, Nbath)
    h[...] = _square_hloc(Nx, Ny, 1, 1, ts)
    eye = _lso_eye(Nlat, 1, 1)
    lam1 = np.zeros(Nbath)
    for ir in range(1, Nbath + 1):
        onsite = ir - 1 - (Nbath - 1) / 2.0
        lam1[ir - 1] = onsite * 2 * hwband / (Nbath - 1) if Nbath > 1 else 0.0
    if Nbath % 2 == 0 and Nbath > 0:
        lam1[Nbath // 2 - 1] = -0.1
        lam1[Nbath // 2] = 0.1
    for ib in range(Nbath):
        hb[..., ib] = lam1[ib] * eye + 1.0 * np.abs(h)
        v[..., ib] = max(0.1, 1.0 / sqrt(Nbath))
    return Model(Nlat, 1, 1, Nbath, h, hb, v, Uloc=np.array([U]), hfmode=hfmode, xmu=xmu, name=f"C3_hm_2dsquare_{Nx}x{Ny}_B{Nbath}")


def _bhz_hloc(Nx, Ny, Mh, ts, lam):
    """drivers/cdn_bhz_2d.f90:213-310: Hloc_model + t_m, t_x, t_y (Norb=2, Nspin=2)."""
    Nlat, Nspin, Norb = Nx * Ny, 2, 2
    sz = np.array([[1, 0], [0, -1]], dtype=np.complex128)
    sx = np.array([[0, 1], [1, 0]], dtype=np.complex128)
    t_m = Mh * sz
    t_x = lambda sp: -ts * sz + 0.5 * ((-1.0) ** (sp + 1)) * 1j * lam * sx  # sp = 1,2
    t_y = -ts * sz + np.array([[0, -0.5 * lam], [0.5 * lam, 0]], dtype=np.complex128)
    h = np.zeros((Nlat, Nlat, Nspin, Nspin, Norb, Norb), dtype=np.complex128, order="F")
    i2n = lambda ix, iy: Nx * (iy - 1) + ix - 1
    for s in range(Nspin):
        for ix in range(1, Nx + 1):
            for iy in range(1, Ny + 1):
                a = i2n(ix, iy)
                h[a, a, s, s] = t_m
                if ix < Nx:
                    h[i2n(ix + 1, iy), a, s, s] = t_x(s + 1)
                if ix > 1:
                    h[i2n(ix - 1, iy), a, s, s] = np.conj(t_x(s + 1).T)
                if iy < Ny:
                    h[i2n(ix, iy + 1), a, s, s] = t_y
                if iy > 1:
                    h[i2n(ix, iy - 1), a, s, s] = t_y.T
    return h


def bhz_2d(Nx=2, Ny=2, Nbath=1, Mh=1.0, ts=0.25, lam=0.3, U=2.0, Ust=0.0, Jh=0.0, Jx=0.0, Jp=0.0, hfmode=True, xmu=0.0) -> Model:
    """BASELINE config C4 (cdn_bhz_2d): complex H, Nspin=2 so H_up != H_dw.
    Bath (drivers/cdn_bhz_2d.f90:106-119): three Hsym = Hloc_model(1,0,0),(0,1,0),(0,0,1)
    with lambda=(Mh,ts,lam); init_dmft_bath zeroes a diagonal Hsym whose lambda is the same
    on every replica through rescale=linspace(HWBAND/Nbath,HWBAND,Nbath) (=0 at Nbath=1)
    (ED_BATH/dmft_aux.f90:61-65,77-99): here HWBAND-scaled for Nbath>1 with HWBAND=2."""
    Nlat = Nx * Ny
    h = _bhz_hloc(Nx, Ny, Mh, ts, lam)
    _, hb, v = _zeros(Nlat, 2, 2, Nbath)
    rescale = np.linspace(2.0 / Nbath, 2.0, Nbath) if Nbath > 1 else np.zeros(max(Nbath, 1))
    for ib in range(Nbath):
        hb[..., ib] = rescale[ib] * Mh * _bhz_hloc(Nx, Ny, 1.0, 0.0, 0.0) + ts * _bhz_hloc(Nx, Ny, 0.0, 1.0, 0.0) + lam * _bhz_hloc(Nx, Ny, 0.0, 0.0, 1.0)
        v[..., ib] = max(0.1, 1.0 / sqrt(Nbath))
    return Model(Nlat, 2, 2, Nbath, h, hb, v, Uloc=np.array([U, U]), Ust=Ust, Jh=Jh, Jx=Jx, Jp=Jp, hfmode=hfmode, xmu=xmu, name=f"C4_bhz_2d_{Nx}x{Ny}_B{Nbath}")


def hm_ring(Nlat=6, Nbath=2, ts=0.25, U=2.0, hwband=2.0, hfmode=True) -> Model:
    """BASELINE config C5 (Ns=18): SURVEY.md 8d 'any Ns=18 one-body graph': Nlat-site ring,
    replicas built like cdn_hm_2dsquare (identity + |Hloc|)."""
    h, hb, v = _zeros(Nlat, 1, 1, Nbath)
    for i in range(Nlat):
        h[i, (i + 1) % Nlat, 0, 0, 0, 0] = -ts
        h[(i + 1) % Nlat, i, 0, 0, 0, 0] = -ts
    eye = _lso_eye(Nlat, 1, 1)
    for ib in range(Nbath):
        onsite = (ib - (Nbath - 1) / 2.0) * (2 * hwband / (Nbath - 1) if Nbath > 1 else 0.0)
        if Nbath % 2 == 0:
            onsite = -0.1 if ib == Nbath // 2 - 1 else (0.1 if ib == Nbath // 2 else onsite)
        hb[..., ib] = onsite * eye + np.abs(h)
        v[..., ib] = max(0.1, 1.0 / sqrt(Nbath))
    return Model(Nlat, 1, 1, Nbath, h, hb, v, Uloc=np.array([U]), hfmode=hfmode, name=f"C5_hm_ring_L{Nlat}_B{Nbath}")


def deterministic_vector(n: int, offset: int = 0) -> np.ndarray:
    """v_k = (sin(0.37k+0.11), cos(0.23k+0.05)), k = 0-based GLOBAL index;
    not normalised here (normalise on the full vector)."""
    k = np.arange(offset, offset + n, dtype=np.float64)
    return (np.sin(0.37 * k + 0.11) + 1j * np.cos(0.23 * k + 0.05)).astype(np.complex128)
