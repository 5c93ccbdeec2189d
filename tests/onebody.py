"""Closed forms of the NON-INTERACTING limit, shared by the CPU and GPU tests (test infrastructure; nothing here computes the Hamiltonian of
the engine or of the oracle): the one-body matrix of a spin assembled straight from the model arrays, and the amplitudes of a Slater
determinant of its eigenstates in the reference's basis convention."""
import numpy as np


def one_body_matrix(m, spin):
    """h[a, b] of spin `spin` in the reference's orbital numbering (impurity is = iorb + ilat*Norb, ED_SETUP.f90:563-568; replica ib:
    Nimp + is + ib*Nimp, :367-375): impHloc, the replicas' blocks (their diagonal enters as its REAL part,
    ED_HAMILTONIAN_SPARSE_HxV.f90:71), the hybridisation V between an impurity orbital and its partner in every replica (H_up.f90:60-87),
    -xmu on the impurity (H_local.f90:22-28).  Nspin = 1: the dw parameters alias the up ones."""
    L, O, B, S = m.Nlat, m.Norb, m.Nbath, m.Nspin
    nimp, ns = L * O, L * O * (B + 1)
    s = spin if S > 1 else 0
    h = np.zeros((ns, ns), dtype=np.complex128)
    for il in range(L):
        for jl in range(L):
            for io in range(O):
                for jo in range(O):
                    a, b = io + il * O, jo + jl * O
                    h[a, b] += m.impHloc[il, jl, s, s, io, jo]
                    for ib in range(B):
                        x = m.Hbath[il, jl, s, s, io, jo, ib]
                        h[nimp + a + ib * nimp, nimp + b + ib * nimp] += x.real if a == b else x
    for ib in range(B):
        for il in range(L):
            for io in range(O):
                a = io + il * O
                h[a, nimp + a + ib * nimp] += m.Vbath[il, s, io, ib]
                h[nimp + a + ib * nimp, a] += m.Vbath[il, s, io, ib]
    for a in range(nimp):
        h[a, a] -= m.xmu
    assert np.abs(h - h.conj().T).max() == 0.0
    return h


def slater_vector(m, map_up, map_dw, levels_up, levels_dw):
    """(v, E): the normalised many-body eigenvector |levels_up> x |levels_dw> of the non-interacting model in the sector whose maps are given,
    v[idw*DimUp + iup], and its eigenvalue.  A basis state is the product of creation operators in ASCENDING orbital order (c / cdg carry
    (-1)^(occupied orbitals below), ED_SETUP.f90:807-833; no cross-spin sign), so its amplitude in prod_k d^dagger_k |0>, d^dagger_k =
    sum_o Phi[o, k] c^dagger_o, is the determinant of Phi restricted to its occupied orbitals (rows, ascending) and the chosen levels."""
    amps, E = [], 0.0
    for spin, mp, lev in ((0, map_up, levels_up), (m.Nspin - 1, map_dw, levels_dw)):
        eps, phi = np.linalg.eigh(one_body_matrix(m, spin))
        occ = np.array([[o for o in range(m.Ns) if (int(x) >> o) & 1] for x in mp])
        amps.append(np.linalg.det(phi[occ][:, :, list(lev)]) if len(lev) else np.ones(len(mp)))
        E += eps[list(lev)].sum()
    v = np.outer(amps[1], amps[0]).reshape(-1).astype(np.complex128)
    return v / np.linalg.norm(v), float(E)
