#!/usr/bin/env python3
"""The callers' usage pattern, measured: scripts/harness.py driven over
  (a) every sector of C2 (Ns=12) in ED_DIAG's order: open -> sp_eigh(2, 20) -> close, against tests/golden/c2_sector_sweep.json;
  (b) the 56 Green's-function channels of one default solve at C3 (Ns=16, ground state in (8,8)), the target sector opened and closed
      around every channel as ED_GF_NORMAL.f90:208-222 does -- and the same with ed_gf_symmetric (32 real channels);
  (c) every sector of C3 (289) through ED_DIAG's loop;
  (d) Ns=18 (C5, Dim = 2.36e9, 37.8 GB per complex vector): ground state of (9,9) by the three-vector Lanczos (a Krylov basis of 21 real
      vectors would need 397 GB), then the first channels of the list, paired, NLANC_C5 steps (default 40).
Prints per-sector / per-channel open, solve and close times.   python scripts/solve_sweep.py > profiles/r05_solve_sweep.txt
Environment: PARTS=abcd (default ab), NLANC (200), SECTOR_CACHE is the library's HXV_SECTOR_CACHE."""
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "cdmft-lanc-ed_amd"))
sys.path.insert(0, str(ROOT / "scripts"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import hxv  # noqa: E402
from hxv import models  # noqa: E402
from harness import diag_sweep, gf_solve  # noqa: E402

parts = os.environ.get("PARTS", "ab")
nlanc = int(os.environ.get("NLANC", 200))
print(f"# solve_sweep on {torch.cuda.get_device_name(0)}; HXV_SECTOR_CACHE={os.environ.get('HXV_SECTOR_CACHE', '1')}", flush=True)


def table(recs, title):
    print(f"\n## {title}")
    print(f"{'(nup,ndw)':>10} {'dim':>10} {'path':>8} {'open ms':>8} {'host':>6} {'plan':>6} {'upl':>6} {'solve ms':>9} {'close ms':>8} {'matvec':>6}  E0")
    for r in recs:
        ou = r["open_us"]
        print(f"({r['nup']:3d},{r['ndw']:3d}) {r['dim']:10d} {'lanczos' if r['lanczos'] else 'small':>8} {r['open_ms']:8.2f} {ou['host'] / 1e3:6.1f} {ou['plan'] / 1e3:6.1f} "
              f"{ou['upload'] / 1e3:6.1f} {r.get('solve_ms', 0.0):9.1f} {r['close_ms']:8.2f} {r.get('nmatvec', 0):6d}  {r['evals'][0] if 'evals' in r else float('nan'):.10f}")
    lz = [r for r in recs if r["lanczos"]]
    print(f"# {len(recs)} sectors, {len(lz)} above lanc_dim_threshold; open total {sum(r['open_ms'] for r in recs):.0f} ms (mean {np.mean([r['open_ms'] for r in recs]):.2f}, "
          f"max {max(r['open_ms'] for r in recs):.1f}), solve total {sum(r.get('solve_ms', 0) for r in recs) / 1e3:.2f} s, close total {sum(r['close_ms'] for r in recs):.0f} ms")


if "a" in parts:
    m = models.hm_1dchain()
    t0 = time.time()
    recs = diag_sweep(m)
    dt = time.time() - t0
    gold = {(g["nup"], g["ndw"]): g for g in json.loads((ROOT / "tests" / "golden" / "c2_sector_sweep.json").read_text())["sectors"]}
    worst = max(np.abs(np.array(r["evals"][:min(2, r["dim"])]) - np.array(gold[(r["nup"], r["ndw"])]["lowest"][:min(2, r["dim"])])).max() for r in recs)
    table(recs, f"(a) C2 cdn_hm_1dchain Ns=12: ED_DIAG's sector loop, every sector, {dt:.1f} s; worst |E - oracle| over E0, E1 of 169 sectors = {worst:.2e}")

if "b" in parts:
    m = models.hm_2dsquare(Nbath=3)
    hxv.sector_cache_clear()
    for symmetric in (False, True):
        recs, s = gf_solve(m, 8, 8, nlanc=nlanc, symmetric=symmetric)
        print(f"\n## (b) C3 cdn_hm_2dsquare Ns=16, ground state of (8,8), ed_gf_symmetric={'T' if symmetric else 'F'}: {s['channels']} channels "
              f"({s['channels_real']} real of which {s['channels_paired']} paired, {s['channels_complex']} complex), nlanc {nlanc}")
        print(f"# ground state: open {s['gs_open_ms']:.1f} ms, sp_eigh(2,20) {s['gs_ms'] / 1e3:.2f} s ({s['gs_nmatvec']} products), E0 = {s['e0']:.12f}")
        print(f"# sector opens {s['sector_opens']} (cache hits {s['sector_open_cache_hits']}): first {s['sector_open_ms_first']:.1f} ms, mean {s['sector_open_ms_mean']:.2f} ms, max {s['sector_open_ms_max']:.1f} ms")
        print(f"# tridiagonalisations: real channels {s['real_channels_s']:.2f} s, complex channels {s['complex_channels_s']:.2f} s;  gf_solve_s {s['gf_solve_s']:.2f}")
        print(f"{'kind':>7} {'op':>3} {'terms':>22} {'sector':>8} {'open ms':>8} {'hit':>4} {'start ms':>8} {'tridiag ms':>10} {'ms/step':>8} {'close ms':>8} {'paired':>6} {'real':>5}")
        for r in recs:
            terms = "+".join(f"{'i' if complex(c).imag > 0 else '-i' if complex(c).imag < 0 else ''}c{o}" for o, c in r["terms"])
            print(f"{r['kind']:>7} {'c+' if r['create'] else 'c':>3} {terms:>22} {str(r['sector']):>8} {r['open_ms']:8.2f} {int(r['open_cache_hit']):4d} {r['start_ms']:8.2f} "
                  f"{r['tridiag_ms']:10.1f} {r['tridiag_ms'] / max(r['nsteps'], 1):8.3f} {r['close_ms']:8.2f} {int(r['paired']):6d} {int(r['real_vectors']):5d}")

if "c" in parts:
    m = models.hm_2dsquare(Nbath=3)
    hxv.sector_cache_clear()
    t0 = time.time()
    recs = diag_sweep(m, small_too=False)
    table(recs, f"(c) C3 cdn_hm_2dsquare Ns=16: ED_DIAG's sector loop, {time.time() - t0:.1f} s (sectors at or below the threshold are opened and closed only)")

if "d" in parts:
    m = models.hm_ring(6, 2)
    hxv.sector_cache_clear()
    from harness import gf_channels
    nl5 = int(os.environ.get("NLANC_C5", 40))
    ch = [c for c in gf_channels(m, symmetric=True) if c["create"]][:4]      # c+_0, (c+_0 + c+_1), (c+_0 + c+_2), (c+_0 + c+_3): two pairs in sector (10,9)
    recs, s = gf_solve(m, 9, 9, nlanc=nl5, symmetric=True, gs_method="lanczos", channels=ch)
    print(f"\n## (d) C5 hm_ring Ns=18, ground state of (9,9) Dim=2363904400 by hxv_lanczos_eigh, then {len(ch)} real channels paired, nlanc {nl5}")
    print(f"# ground state: open {s['gs_open_ms']:.1f} ms, {s['gs_nmatvec']} iterations (two passes) {s['gs_ms'] / 1e3:.2f} s, E0 = {s['e0']:.12f}")
    for r in recs:
        print(f"# channel {r['kind']} {'c+' if r['create'] else 'c'} {r['terms']} sector {r['sector']} dim {r['dim']}: open {r['open_ms']:.1f} ms (hit {int(r['open_cache_hit'])}), start vector {r['start_ms']:.1f} ms, "
              f"{r['nsteps']} steps {r['tridiag_ms'] / 1e3:.2f} s = {r['tridiag_ms'] / max(r['nsteps'], 1):.2f} ms per channel-step (paired {int(r['paired'])})")
    print(f"# total {s['gf_solve_s']:.1f} s")
