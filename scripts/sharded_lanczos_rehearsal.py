"""Rehearsal of the N>1 Lanczos on ONE GPU (gloo, several ranks sharing the device): ShardedLanczos over ShardedHxv /
TransposedHxv reproduces the single-GPU ground state.
  python -m torch.distributed.run --nproc-per-node 3 --master-addr 127.0.0.1 scripts/sharded_lanczos_rehearsal.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cdmft-lanc-ed_amd"))
import torch, torch.distributed as dist
import hxv
from hxv import models

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
m, (nup, ndw) = models.hm_1dchain(), (6, 6)            # C2
sec = hxv.HxvSector.from_model(m, nup, ndw, rank=rank, nranks=world)
sh = hxv.ShardedHxv(sec.DimUp, sec.DimDw, rank, world, sec.apply_device, pitch=sec.pitch)
e0, vec, nit = hxv.ShardedLanczos(sh).eigh(512, 1e-13, device="cuda")
nrows = hxv.dw_split(sec.DimUp, rank, world)[0]
panel = hxv.HxvSector.dw_panel(m, nup, ndw, nrows)
th = hxv.TransposedHxv(sec.DimUp, sec.DimDw, rank, world, panel.apply_dw_panel, sec.apply_up_add, pitch=sec.pitch, pitch_panel=panel.pitch,
                       stage_on_host=True)
e0t, _, nitt = hxv.ShardedLanczos(th).eigh(512, 1e-13, want_vector=False, device="cuda")
ev, X, nconv, nmv = hxv.sharded_eigh_lowest(sh, 2, 20, device="cuda")
if rank == 0:
    full = hxv.HxvSector.from_model(m, nup, ndw)
    full.set_option("real_vectors", 0)
    ref, _, nref = full.lanczos_eigh(512, 1e-13, want_vector=False)
    print(f"world={world}: E0 all-gather {e0:.12f} ({nit} it)  all-to-all {e0t:.12f} ({nitt} it)  single GPU {ref:.12f} ({nref} it)", flush=True)
    assert abs(e0 - ref) < 1e-10 and abs(e0t - ref) < 1e-10
    evs, _, ncs, nmvs = full.eigh_lowest(2, 20, want_vectors=False)
    print(f"  sp_eigh flavour: sharded {ev} ({nmv} products, nconv={nconv})  single GPU {evs} ({nmvs} products)", flush=True)
    assert nconv == 2 and abs(ev - evs).max() < 1e-10 and abs(nmv - nmvs) <= 0.25 * nmvs
dist.barrier()
dist.destroy_process_group()
