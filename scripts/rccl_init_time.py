"""What ONE ncclCommInitRank + ncclCommDestroy of the real librccl costs: one rank on one GPU -- the only contact with librccl
a one-GPU box allows; on a node the ring / channel setup over xGMI comes on top.  HXV_COMM_CACHE=0 makes every hxv_comm_init build a
communicator and every hxv_comm_free destroy it (the behaviour before round 6: 345 of these per solve); with the cache the same loop builds one."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cdmft-lanc-ed_amd"))
import torch, hxv
from hxv import models

m = models.hm_1dchain(Nlat=2, Nbath=2)
for cache in ("0", "1"):
    os.environ["HXV_COMM_CACHE"] = cache
    t_init, t_free, t_open = [], [], []
    for k in range(8):
        t0 = time.perf_counter()
        sec = hxv.HxvSector.from_model(m, 3, 3)
        t1 = time.perf_counter()
        sec.comm_init(hxv.HxvSector.comm_unique_id())
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        sec.close()
        t3 = time.perf_counter()
        t_open.append((t1 - t0) * 1e3), t_init.append((t2 - t1) * 1e3), t_free.append((t3 - t2) * 1e3)
    st = hxv.comm_cache_stats()
    print(f"HXV_COMM_CACHE={cache}: hxv_comm_init ms first {t_init[0]:.1f}, then {['%.2f' % x for x in t_init[1:]]}; close (incl. ncclCommDestroy when not cached) ms {['%.2f' % x for x in t_free]}; "
          f"sector open ms {['%.2f' % x for x in t_open[1:]]}; cache {st}", flush=True)
    hxv.comm_cache_clear()
