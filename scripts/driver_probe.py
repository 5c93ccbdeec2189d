"""Exploratory probe of the device Lanczos drivers in regimes the suite does not pin: long fixed-length runs (graph against host loop,
real against complex), long paired runs, eigh_lowest with a large basis on thread ranks."""
import os, sys
ROOT = os.path.join(os.path.dirname(__file__), "..")
sys.path.insert(0, os.path.join(ROOT, "cdmft-lanc-ed_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, hxv
from hxv import models
from scipy.linalg import eigh_tridiagonal

bad = 0
m = models.hm_1dchain(eps_bath=[0.3, 0.6])
sec = hxv.HxvSector.from_model(m, 6, 6)
rng = np.random.default_rng(1)
v = rng.standard_normal(sec.Dim); v /= np.linalg.norm(v)
vin = torch.from_numpy(v.astype(np.complex128)).cuda()
res = {}
for graph in (1, 0):
    for real in (1, 0):
        sec.set_option("lanczos_graph", graph); sec.set_option("real_vectors", real)
        a, b, n = sec.lanczos_tridiag(vin, 600)
        res[(graph, real)] = (a.copy(), b.copy(), n)
        e0 = eigh_tridiagonal(a[:n], b[1:n], select="i", select_range=(0, 0))[0][0]
        print(f"C2 tridiag 600 steps graph={graph} real={real}: n={n} lowest Ritz {e0:.12f} alpha[599] {a[n-1]:.6e} finite={np.isfinite(a).all() and np.isfinite(b).all()}", flush=True)
for real in (1, 0):
    same = np.array_equal(res[(1, real)][0], res[(0, real)][0]) and np.array_equal(res[(1, real)][1], res[(0, real)][1])
    print(f"  graph vs host loop bit-identical (real={real}): {same}")
    bad += 0 if same else 1
d = np.abs(res[(1, 1)][0][:30] - res[(1, 0)][0][:30]).max()
print(f"  real vs complex vectors, first 30 alphas: max diff {d:.1e}"); bad += 0 if d < 1e-10 else 1
vb = rng.standard_normal(sec.Dim); vb /= np.linalg.norm(vb)
for job in (2, 0):
    sec.set_option("job_up", job); sec.set_option("real_vectors", 1)
    (aa, ba, na), (ab, bb, nb) = sec.lanczos_tridiag_pair(vin, torch.from_numpy(vb.astype(np.complex128)).cuda(), 600)
    sec.set_option("real_vectors", 0)
    a1, b1, n1 = sec.lanczos_tridiag(vin, 600)
    same = np.array_equal(aa, a1) and np.array_equal(ba, b1)
    print(f"  paired 600 steps (job_up={job}): channel a bit-identical to its single run through the same kernels (real_vectors=0): {same}; both finite: {np.isfinite(ab).all()}")
    bad += 0 if same else 1
sec.set_option("job_up", 2); sec.set_option("real_vectors", 1)
ev_ser, _, nc_ser, _ = sec.eigh_lowest(4, 40, want_vectors=False)
sec.close()

def rank(r, group):
    s = hxv.HxvSector.from_model(m, 6, 6, rank=r, nranks=3)
    group.join(s)
    ev, _, nc, nmv = s.eigh_lowest(4, 40, want_vectors=False)
    s.close()
    return ev, nc, nmv
for ex in ("allgather", "halo", "alltoall"):
    hxv.set_exchange_default(ex)
    try:
        out = hxv.run_ranks(3, rank)
    finally:
        hxv.set_exchange_default("allgather")
    d = max(np.abs(ev - ev_ser).max() for ev, _, _ in out)
    print(f"eigh_lowest(4, 40) on 3 thread ranks, {ex}: max |E - serial| {d:.1e}, nconv {[nc for _, nc, _ in out]}, products {out[0][2]}")
    bad += 0 if d < 1e-9 and all(nc == 4 for _, nc, _ in out) else 1
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
