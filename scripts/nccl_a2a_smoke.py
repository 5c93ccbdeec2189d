"""RCCL smoke on one GPU (world_size 1): the all-to-all exchange path (TransposedHxv, device tensors, backend nccl)
against the plain product."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cdmft-lanc-ed_amd"))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
import hxv
from hxv import models
m, (nup, ndw) = models.hm_1dchain(), (6, 6)
sec = hxv.HxvSector.from_model(m, nup, ndw)
panel = hxv.HxvSector.dw_panel(m, nup, ndw, sec.DimUp)
th = hxv.TransposedHxv(sec.DimUp, sec.DimDw, 0, 1, panel.apply_dw_panel, sec.apply_up_add, pitch=sec.pitch, pitch_panel=panel.pitch)
v = sec.pad(torch.randn(sec.Dim, dtype=torch.complex128, device=dev))
hv = torch.zeros_like(v)
th(th.Nloc, v, hv)
ref = sec.apply_device(v)
torch.cuda.synchronize()
print("nccl all-to-all path ok, rel err", ((hv - ref).abs().max() / ref.abs().max()).item())
t = torch.ones(4, dtype=torch.float64, device=dev); o = torch.empty_like(t)
dist.all_to_all_single(o, t, [4], [4]); torch.cuda.synchronize(); print("all_to_all_single with split lists ok", bool(torch.equal(o, t)))
dist.destroy_process_group()
