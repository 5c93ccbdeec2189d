"""the skeleton (loads + stores, no arithmetic) of pass B's out-of-block half as an LDS-free whole-line kernel,
measured BEFORE building it.  Source lists = the real out-of-block structure of H_dw at C3 (12-orbital prefix blocks).  Builds
scripts/oob_skeleton.hip on the GPU box.  Kill criterion: <= 0.6 ms (the phase costs 0.88 ms inside pass B today)."""
import ctypes as C, os, subprocess, sys
import numpy as np
sys.path.insert(0, "cdmft-lanc-ed_amd")
import torch, hxv
from hxv import models

os.makedirs("gpurun_out", exist_ok=True)
so = "gpurun_out/liboob_skeleton.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, "scripts/oob_skeleton.hip"])
L = C.CDLL(so)
L.oob_skeleton_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.POINTER(C.c_float)]
sec = hxv.HxvSector.from_model(models.hm_2dsquare(Nbath=3), 8, 8)
rp, cols, _ = sec.csr("dw")
mu, md = sec.maps()
lowbits = sec.get_option("tile_bits_dw")
blk = md >> lowbits
dimdw, dimup, pitch = sec.DimDw, sec.DimUp, sec.pitch
lists = [[int(c) - 1 for c in cols[rp[i]:rp[i + 1]] if blk[c - 1] != blk[i]] for i in range(dimdw)]
nsrc = max(len(x) for x in lists)
src = -np.ones((nsrc, dimdw), dtype=np.int32)
for i, x in enumerate(lists):
    src[: len(x), i] = x
print(f"C3 H_dw out-of-block sources per column: mean {np.mean([len(x) for x in lists]):.2f}, max {nsrc} (blocks of {lowbits} low orbitals)")
sec.close()
d_src = torch.from_numpy(src).cuda()
v = torch.randn(pitch * dimdw, dtype=torch.complex128, device="cuda")
wt = torch.zeros(dimup * (dimdw + 8), dtype=torch.complex128, device="cuda")
ms = C.c_float()
for rows in (16, 8):
    for mode, name in ((1, "gathers only"), (2, "stores only"), (3, "gathers + one store (the skeleton)")):
        L.oob_skeleton_run(v.data_ptr(), wt.data_ptr(), d_src.data_ptr(), nsrc, dimup, dimdw, pitch, rows, mode, 2, C.byref(ms))
        rc = L.oob_skeleton_run(v.data_ptr(), wt.data_ptr(), d_src.data_ptr(), nsrc, dimup, dimdw, pitch, rows, mode, 10, C.byref(ms))
        print(f"wave footprint {rows:2d} rows x {64 // rows} columns, {name:36s}: {ms.value:.3f} ms (rc {rc})", flush=True)
