"""What HBM gives for short column segments (pass B's access pattern).  Builds scripts/strided_bench.hip on the GPU box."""
import ctypes as C, os, subprocess, sys
sys.path.insert(0, "cdmft-lanc-ed_amd")
import torch
os.makedirs("gpurun_out", exist_ok=True)
so = "gpurun_out/libstrided_bench.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, "scripts/strided_bench.hip"])
L = C.CDLL(so)
L.strided_bench.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.POINTER(C.c_float)]
dimup = int(os.environ.get("DIMUP", "12870")); pitch = (dimup + 7) & ~7; ncols = dimup
v = torch.randn(pitch * ncols, dtype=torch.float64, device="cuda") + 1j * torch.randn(pitch * ncols, dtype=torch.float64, device="cuda")
out = torch.zeros_like(v)
torch.cuda.synchronize()
ms = C.c_float()
names = {0: "read", 1: "read+lds", 2: "read+lds+store", 3: "read+lds+nt-store", 4: "read+lds+RMW", 5: "read+lds+nt-RMW"}
for R, n in ((4, 924), (8, 924), (16, 462)):
    for mode in range(6):
        L.strided_bench(v.data_ptr(), out.data_ptr(), pitch, ncols, R, n, mode, 1, C.byref(ms))
        rc = L.strided_bench(v.data_ptr(), out.data_ptr(), pitch, ncols, R, n, mode, 3, C.byref(ms))
        print(f"R={R:3d} n={n:5d} {names[mode]:20s}: {ms.value:.3f} ms rc={rc}", flush=True)
