import sys, ctypes as C
sys.path.insert(0, "cdmft-lanc-ed_amd")
import torch, hxv
from hxv import models
m = models.hm_2dsquare(Nbath=3)
import os
NUP, NDW = map(int, os.environ.get("SECTOR", "8,8").split(","))
sec = hxv.HxvSector.from_model(m, NUP, NDW)
print("sector", NUP, NDW, sec.DimUp)
L = hxv.load_library()
L.hxv_debug_strided_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float)]
v = torch.randn(sec.fullElems, dtype=torch.float64, device="cuda") + 1j * torch.randn(sec.fullElems, dtype=torch.float64, device="cuda")
out = torch.zeros(sec.fullElems, dtype=torch.complex128, device="cuda")
torch.cuda.synchronize()
ms = C.c_float()
names = {0: "read", 1: "read+lds", 2: "read+lds+store", 3: "read+lds+nt-store", 4: "read+lds+RMW", 5: "read+lds+nt-RMW"}
for R, n in ((4, 924), (8, 924), (16, 462)):
    for mode in range(6):
        L.hxv_debug_strided_read(sec._h, v.data_ptr(), out.data_ptr(), R, n, mode, 1, C.byref(ms))
        rc = L.hxv_debug_strided_read(sec._h, v.data_ptr(), out.data_ptr(), R, n, mode, 3, C.byref(ms))
        print(f"R={R:3d} n={n:5d} {names[mode]:20s}: {ms.value:.3f} ms rc={rc}", flush=True)
