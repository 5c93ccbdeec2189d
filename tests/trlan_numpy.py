"""NumPy restatement of the thick-restart Lanczos the engine runs on the device (csrc/hxv_eigh.hip,
hxv_eigh_lowest) -- test infrastructure: the CPU suite checks the ALGORITHM here against LAPACK, the GPU
suite checks the device implementation against LAPACK/ARPACK and against this.

It stands where the reference calls SciFortran's sp_eigh (P-ARPACK, implicitly restarted Lanczos) at
ED_DIAG.f90:152-160: lowest `neigen` eigenpairs of a Hermitian operator given only as MatVec, with a
Krylov basis of ncv vectors.  Thick restart (Wu & Simon 2000) is the explicit-restart form of the same
method for Hermitian operators."""
from __future__ import annotations

import numpy as np


def start_vector(dim: int, seed: int = 0x5EED5EED) -> np.ndarray:
    """splitmix64 hash of the global index -> uniform(-0.5,0.5) re and im: the engine's deterministic start vector."""
    z = (np.arange(dim, dtype=np.uint64) * np.uint64(2) + np.uint64(seed))
    out = np.empty((2, dim))
    with np.errstate(over="ignore"):
        for k in range(2):
            x = z + np.uint64(k) + np.uint64(0x9E3779B97F4A7C15)
            x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            x = x ^ (x >> np.uint64(31))
            out[k] = (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0) - 0.5
    return out[0] + 1j * out[1]


def keep_count(m: int, neigen: int, nconv: int) -> int:
    """Ritz vectors kept at a restart (the rule of csrc/hxv_eigh.hip at its default eigh_keep_pct = 20)."""
    k = neigen + min(nconv, (m - neigen) // 2) + max(1, (m - neigen) * 20 // 100)
    return max(1, min(k, m - 1))


def trlan_lowest(matvec, dim: int, neigen: int, ncv: int, maxrestart: int = 512, tol: float = 0.0, v0=None):
    """-> (evals[neigen], evecs[dim, neigen], nconv, nmatvec, nrestart)."""
    eps = np.finfo(float).eps
    tol = max(tol, eps)
    eps23 = eps ** (2.0 / 3.0)
    m = min(max(ncv, neigen + 1), dim)
    neigen = min(neigen, dim)
    V = np.zeros((m + 1, dim), dtype=np.complex128)
    v = start_vector(dim) if v0 is None else np.asarray(v0, dtype=np.complex128)
    V[0] = v / np.linalg.norm(v)
    T = np.zeros((m, m))
    k = 0
    nmv = 0
    theta = S = None
    for it in range(maxrestart + 1):
        meff, beta_last = m, 0.0
        for j in range(k, m):
            w = matvec(V[j])
            nmv += 1
            c = V[: j + 1].conj() @ w
            T[j, j] = c[j].real
            # all projections are measured; subtracted are the two local ones, everything right after a restart, and
            # whatever exceeds 1e-13*|w| (rounding noise is left alone) -- same rule as csrc/hxv_eigh.hip gs_pass
            sel = np.abs(c) > 1e-13 * np.sqrt(np.vdot(c, c).real)
            sel[max(j - 1, 0):] = True
            if j == k:
                sel[:] = True
            w = w - (c * sel) @ V[: j + 1]
            nrm = np.linalg.norm(w)
            if nrm * nrm < 0.01 * (np.vdot(c, c).real + nrm * nrm):     # norm dropped 10x: one refinement pass (DGKS)
                c2 = V[: j + 1].conj() @ w
                T[j, j] += c2[j].real
                w = w - c2 @ V[: j + 1]
                nrm2 = np.linalg.norm(w)
                if nrm2 < 0.5 * nrm:
                    nrm2 = 0.0                                          # w lies in span(V): invariant subspace
                nrm = nrm2
            scale = max(1.0, np.abs(T[: j + 1, : j + 1]).max())
            if nrm <= 1e-13 * scale:
                meff, beta_last = j + 1, 0.0
                break
            if j + 1 < m:
                T[j + 1, j] = T[j, j + 1] = nrm
            beta_last = nrm
            V[j + 1] = w / nrm
        theta, S = np.linalg.eigh(T[:meff, :meff])
        res = np.abs(beta_last * S[meff - 1, :])
        ne = min(neigen, meff)
        conv = res[:ne] <= tol * np.maximum(eps23, np.abs(theta[:ne]))
        nconv = int(conv.sum())
        if nconv == ne or meff < m or it == maxrestart:
            break
        k = keep_count(m, neigen, nconv)
        V[:k] = S[:, :k].T @ V[:m]
        V[k] = V[m]
        T[:] = 0.0
        for i in range(k):
            T[i, i] = theta[i]
            T[k, i] = T[i, k] = beta_last * S[m - 1, i]
    ne = min(neigen, meff)
    X = (S[:, :ne].T @ V[:meff]).T
    return theta[:ne].copy(), X, nconv, nmv, it
