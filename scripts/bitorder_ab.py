"""Same-box A/B of the orbital -> bit assignment of a spin's basis (TIMING ONLY: the relabelled model is another
Hamiltonian with the same hop graph, the device layout it implies is not converted at the boundaries).  WORKLOAD=C3|C4|C5.
usage: bitorder_ab.py "name:up=p0,p1,...;dw=p0,p1,..." ...   (pos lists = bit of orbital 0, 1, ...; omitted spin = reference order)

HISTORY: the `HXV_EXP_UP_ORDER` / `HXV_EXP_DW_ORDER` hooks this script sets lived in hxv_sector.cpp at commit d4f54db only (they relabelled the
hops before the tables were built); the log it produced is profiles/r06_bitorder_c3.log.  The up-spin result became the engine's device row
order (commit e2da8cd; `HXV_ROW_ORDER=0|1`, tests/test_gpu_row_order.py), which converts at the boundaries and replaced the hooks; on later
commits this script times the shipped order only (the env variables are ignored) -- use HXV_ROW_ORDER=0/1 for the up-spin A/B."""
import os, sys
os.environ["HXV_EXPERIMENTS"] = "1"
os.environ["HXV_SECTOR_CACHE"] = "0"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cdmft-lanc-ed_amd"))
import torch, hxv
from hxv import models

wl = os.environ.get("WORKLOAD", "C3")
m, (nup, ndw) = {"C3": (models.hm_2dsquare(Nbath=3), (8, 8)), "C4": (models.bhz_2d(Nbath=1), (8, 8)), "C5": (models.hm_ring(6, 2), (9, 9))}[wl]
v = hv = None
for spec in sys.argv[1:] or ["ref:"]:
    name, _, rest = spec.partition(":")
    os.environ.pop("HXV_EXP_UP_ORDER", None)
    os.environ.pop("HXV_EXP_DW_ORDER", None)
    for part in rest.split(";"):
        if part.startswith("up="):
            os.environ["HXV_EXP_UP_ORDER"] = part[3:]
        if part.startswith("dw="):
            os.environ["HXV_EXP_DW_ORDER"] = part[3:]
    sec = hxv.HxvSector.from_model(m, nup, ndw)
    if v is None:
        n = sec.fullElems
        v = torch.empty(n, dtype=torch.complex128, device="cuda")
        vr = torch.view_as_real(v).view(-1)
        g = torch.Generator(device="cuda").manual_seed(5)
        for a in range(0, 2 * n, 1 << 28):
            b = min(a + (1 << 28), 2 * n)
            vr[a:b] = torch.randn(b - a, dtype=torch.float64, device="cuda", generator=g)
        hv = torch.empty(sec.localElems, dtype=torch.complex128, device="cuda")
    nrep = 3 if wl == "C5" else 8
    sec.time_apply(v, hv, 2)
    full = min(sec.time_apply(v, hv, nrep) for _ in range(3))
    sec.set_option("passes", 2)
    sec.time_apply(v, hv, 1)
    tb = min(sec.time_apply(v, hv, nrep) for _ in range(3))
    sec.set_option("passes", 1)
    sec.time_apply(v, hv, 1)
    ta = min(sec.time_apply(v, hv, nrep) for _ in range(3))
    sec.set_option("passes", 3)
    line = f"{wl} [{name:24s}] full {full:8.3f} ms  passB {tb:8.3f}  passA {ta:8.3f} |"
    for k in ("bh_up_x100", "rs_up_x100", "slots_out_up_x100", "max_outer_up", "table_classes_up", "rs_tables_up", "bh_dw_x100", "rs_dw_x100", "max_outer_dw", "table_classes_dw"):
        try:
            line += f" {k}={sec.get_option(k)}"
        except Exception:
            pass
    if sec.real_vectors_available and wl != "C5":
        line += f" | real Lanczos it {sec.time_lanczos(10):.3f} ms"
        sec.set_option("real_vectors", 0)
        line += f" complex it {sec.time_lanczos(10):.3f} ms"
    print(line, flush=True)
    sec.close()
    del sec
