"""Soak: tests/test_gpu_fuzz.py::test_random_models_sectors_shards_and_tile_options over seeds beyond the 64 of the suite.
usage: fuzz_soak.py [first] [last]"""
import os, sys, time, traceback
ROOT = os.path.join(os.path.dirname(__file__), "..")
sys.path.insert(0, os.path.join(ROOT, "cdmft-lanc-ed_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as tf
a, b = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), (int(sys.argv[2]) if len(sys.argv) > 2 else 400)
bad, t0 = [], time.time()
for seed in range(a, b):
    try:
        tf.test_random_models_sectors_shards_and_tile_options(None, seed)
    except Exception:
        bad.append(seed)
        print("seed", seed, "FAILED\n" + traceback.format_exc()[-1500:], flush=True)
    if seed % 50 == 0:
        print(f"... seed {seed}, {time.time() - t0:.0f} s, failures so far {bad}", flush=True)
print(f"seeds {a}..{b - 1}: {len(bad)} failures {bad} in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
