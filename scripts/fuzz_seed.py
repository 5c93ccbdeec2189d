"""Replay one seed of tests/test_gpu_fuzz.py::test_random_models_sectors_shards_and_tile_options and compare the kernels."""
import os, sys
ROOT = os.path.join(os.path.dirname(__file__), "..")
sys.path.insert(0, os.path.join(ROOT, "cdmft-lanc-ed_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, hxv
import test_gpu_fuzz as tf
from oracle.oracle import OracleSector
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(1000 + seed)
m = tf._random_model(rng)
Ns = m.Ns
if rng.random() < 0.6:
    nup, ndw = int(np.clip(Ns // 2 + rng.integers(-1, 2), 0, Ns)), int(np.clip(Ns // 2 + rng.integers(-1, 2), 0, Ns))
else:
    nup, ndw = int(rng.integers(0, Ns + 1)), int(rng.integers(0, Ns + 1))
full = OracleSector(m, nup, ndw)
v = rng.standard_normal(full.Dim) + 1j * rng.standard_normal(full.Dim)
size = int(rng.integers(1, min(4, full.DimDw) + 1))
o = {"lds_budget_kb": int(rng.choice([8, 16, 32])), "cols_per_tile": int(rng.choice([2, 4, 8])), "rows_per_tile": int(rng.choice([2, 4, 8])),
     "threads_up": int(rng.choice([256, 512, 1024])), "threads_dw": int(rng.choice([256, 512, 1024])), "sort_mode": int(rng.integers(3)),
     "wt_cols": int(rng.choice([2, 4, 8, 16])), "job_cols": int(rng.choice([1, 2])), "pair_rows": int(rng.choice([0, 1])),
     "job_groups": int(rng.choice([1, 3, 100])), "job_max_blocks": int(rng.choice([0, 32]))}
print("model Ns", Ns, "sector", nup, ndw, "Dim", full.Dim, "size", size, "opts", o, flush=True)
sec = hxv.HxvSector.from_model(m, nup, ndw)
for k, val in o.items():
    sec.set_option(k, val)
print("real_h available", sec.real_vectors_available, "blocks up", sec.get_option("nblocks_up"), flush=True)
w = np.linalg.eigvalsh(full.dense())
for job in (2, 1, 0):
    sec.set_option("job_up", job)
    for rv in (1, 0):
        sec.set_option("real_vectors", rv)
        for fused in (1, 0):
            sec.set_option("lanczos_fused", fused)
            e0 = sec.lanczos_eigh(600, 1e-13, want_vector=False)[0]
            print(f"job_up={job} real_vectors={rv} fused={fused}: E0 {e0:.10f} exact {w[0]:.10f} {'OK' if abs(e0 - w[0]) < 1e-8 else 'WRONG'}", flush=True)
