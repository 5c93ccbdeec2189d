"""One Green's-function channel of the Ns=16 problem, device-resident end to end (ED_GF_NORMAL.f90:160-230):
ground state of sector (8,8) -> c^dagger_{site 1, up}|gs> in sector (9,8) -> 200-step tridiagonalisation there."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cdmft-lanc-ed_amd"))
import numpy as np, torch, hxv
from hxv import models

m = models.hm_2dsquare(Nbath=3)
t0 = time.time(); gs = hxv.HxvSector.from_model(m, 8, 8); up = hxv.HxvSector.from_model(m, 9, 8); tb = time.time() - t0
t0 = time.time(); e0, psi, nit = gs.lanczos_eigh(512, 1e-12, native=True); torch.cuda.synchronize(); tg = time.time() - t0
t0 = time.time(); vv, n2 = gs.apply_ladder(up, 0, 0, True, psi); torch.cuda.synchronize(); tl = time.time() - t0
vv = vv / np.sqrt(n2)
nl = int(os.environ.get("NLANC", 200))      # lanc_ngfiter default (ED_INPUT_VARS.f90)
t0 = time.time(); a, b, n = up.lanczos_tridiag(vv, nl); torch.cuda.synchronize(); tt = time.time() - t0
print(f"sectors built {tb:.2f}s | ground state E0={e0:.10f} ({nit} it) {tg:.2f}s | c^dagger|gs> norm2={n2:.6f} {tl*1e3:.1f} ms | "
      f"tridiag {n} steps in sector (9,8) Dim={up.Dim} {tt:.2f}s ({tt/n*1e3:.2f} ms/step, real vectors={up.get_option('lanczos_real_last')})", flush=True)
wm = np.pi / 100.0 * (2 * np.arange(1, 9) - 1)
ev, Z = np.linalg.eigh(np.diag(a[:n]) + np.diag(b[1:n], 1) + np.diag(b[1:n], -1))
G = (n2 * Z[0, :] ** 2 / (1j * wm[:, None] - (ev[None, :] - e0))).sum(axis=1)
print("particle part of G_11(i w_n), n=1..4:", np.array2string(G[:4], precision=6), flush=True)
# two channels of the same sector at once (hxv_lanczos_tridiag_pair): c^dagger_{site 1, up}|gs> and c^dagger_{site 2, up}|gs>
v2, n22 = gs.apply_ladder(up, 1, 0, True, psi)
torch.cuda.synchronize(); t0 = time.time()
(aa, ba, na), (ab, bb, nb) = up.lanczos_tridiag_pair(vv, v2 / np.sqrt(n22), nl); torch.cuda.synchronize(); tp = time.time() - t0
print(f"two channels on one product: {na}+{nb} steps {tp:.2f}s ({tp/na*1e3/2:.2f} ms per channel-step); channel a vs its single run: max|d alanc|(20) {np.abs(aa[:20]-a[:20]).max():.1e}", flush=True)
