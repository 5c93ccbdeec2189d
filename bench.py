#!/usr/bin/env python3
"""bench.py -- sector HxV throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" = one full product Hv = H v of the BASELINE C3 sector (cdn_hm_2dsquare: 2x2 cluster +
3 replica baths, Ns=16, sector (8,8), Dim = 165 636 900, complex fp64), vectors resident in HBM.
With N>1 ranks the sector is split along DimDw exactly like the reference (ED_HAMILTONIAN.f90:93-105):
each step exchanges the slabs over RCCL THROUGH THE C-ABI (hxv_comm_unique_id -> broadcast -> hxv_comm_init ->
hxv_apply_device_slab: what a Fortran rank of the reference would run; --exchange allgather [default] | halo | alltoall = the
reference's own two transposes) and every rank computes its slab (strong scaling: the sector is fixed).  --backend gloo goes
through torch.distributed instead (hxv/distributed.py: the rehearsal twin).  Before the warm-up ONE product of a split run is checked on every
rank against the unsplit sector (1e-13; "checked" / "check_rel_err" in the line; the run aborts otherwise; --no-check skips).  value = algorithmic GB/s of the whole job = 32 B x Dim / step time
(read v once + write Hv once per basis state).

Rank 0 prints ONE JSON line.  It also carries
  roofline     : achieved/peak HBM GB/s of the product's kernels, timed with HIP events on the launch stream
  cpu_baseline : the reference algorithm (oracle spMatVec_mpi_main, thread-ranks) on the host cores, N=1 only
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "cdmft-lanc-ed_amd"))
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "scripts"))  # harness.py: the callers' call order (measurement scaffolding, not the package)

KERNELS_STAMP = "r06-a"   # bumped whenever the product kernels OR what they run on (tables, device row order) change: profiles/traffic.json and
                          # profiles/kernel_trace.json are quoted only for the same stamp (r06-a: round 5's kernels on the device row order)
HBM_COPY_GBS = 6290.0     # what a float4 copy measures on MI355X (MI355X_MICROARCH.md, "HBM3E peak BW ... 6.29 TB/s measured"): the FIXED yardstick
                          # of the two-pass design's floor; the box's own copy rate is reported beside it, not used for it
XGMI_LINK_GBS = 153.0  # one xGMI link, per direction (7 links per GPU: SURVEY.md 8e)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def host_cpu():
    """(cpu model, logical CPUs of the host, CPUs this process may run on, PHYSICAL cores among them) from /proc/cpuinfo."""
    model, phys = "unknown", {}
    try:
        cur = {}
        for line in open("/proc/cpuinfo"):
            if ":" not in line:
                if "processor" in cur:
                    phys[int(cur["processor"])] = (cur.get("physical id", "0"), cur.get("core id", cur["processor"]))
                cur = {}
                continue
            k, v = (x.strip() for x in line.split(":", 1))
            cur[k] = v
            if k == "model name":
                model = v
        if "processor" in cur:
            phys[int(cur["processor"])] = (cur.get("physical id", "0"), cur.get("core id", cur["processor"]))
    except OSError:
        pass
    allowed = sorted(os.sched_getaffinity(0))
    ncores = len({phys.get(c, ("0", str(c))) for c in allowed})
    return model, os.cpu_count() or len(allowed), len(allowed), max(1, ncores)


def cgroup_cpu_limit():
    """CPUs the container's cgroup grants this process (cpu.max of cgroup v2 / cfs quota of v1), or None without a quota: the GPU boxes
    show all of the host's CPUs in the affinity mask but schedule a share of them."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(float(q) / float(per) + 0.999))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, int(q / per + 0.999))
    except (OSError, ValueError):
        pass
    return None


def cpu_baseline(model, nup, ndw, budget_s=25.0):
    """Reference algorithm on the host: oracle spMatVec_mpi_main with one thread-rank per PHYSICAL core this process may
    use -- no fixed cap; the core count that ran, the CPU model and the host's totals are stated.  Where the affinity mask shows more
    cores than the box schedules (a CPU share without a visible cgroup quota) a second sample with 16 thread-ranks is taken and the FASTER
    of the two is the value; both are listed."""
    import numpy as np
    from oracle.oracle import OracleSector, spMatVec_mpi_main

    cpu_model, host_cpus, allowed, phys_cores = host_cpu()
    quota = cgroup_cpu_limit()
    # one thread-rank per physical core this process may use; a cgroup CPU quota below that is the share the box really schedules (more
    # thread-ranks than that only time-slice: 128 ranks on the pool's 16-CPU share measured HALF the rate of 16); HXV_CPU_BASELINE_RANKS overrides
    P0 = phys_cores if quota is None else max(1, min(phys_cores, quota))
    why = "" if P0 == phys_cores else f"; capped by the container's cgroup CPU quota of {quota}"
    cands = [P0] + ([16] if quota is None and phys_cores > 16 else [])
    if os.environ.get("HXV_CPU_BASELINE_RANKS"):
        cands = [max(1, int(os.environ["HXV_CPU_BASELINE_RANKS"]))]
        why = "; HXV_CPU_BASELINE_RANKS"
    rng = np.random.default_rng(0)
    samples, v = [], None
    for P in cands:
        t0 = time.time()
        secs = [OracleSector(model, nup, ndw, r, P) for r in range(P)]
        build_s = time.time() - t0
        dim = secs[0].Dim
        if v is None:
            v = rng.standard_normal(dim) + 1j * rng.standard_normal(dim)
        spMatVec_mpi_main(model, nup, ndw, P, v, repeat=1, sectors=secs)  # warm-up
        t0 = time.time()
        n = 0
        while True:
            spMatVec_mpi_main(model, nup, ndw, P, v, repeat=1, sectors=secs)
            n += 1
            if time.time() - t0 > budget_s / (2 * len(cands)) or n >= 5:
                break
        dt = (time.time() - t0) / n
        for s in secs:
            s.close()
        samples.append({"thread_ranks": P, "products": n, "s_per_matvec": dt, "GBs": 32.0 * dim / dt / 1e9, "matrix_build_s": round(build_s, 1)})
    best = max(samples, key=lambda x: x["GBs"])
    # SURVEY 8d asks for the C2 figure beside C3's: the same algorithm on the Ns=12 sector (6,6), same thread-rank count
    c2 = None
    try:
        from hxv import models as _models

        m2, P2 = _models.hm_1dchain(), min(best["thread_ranks"], 64)
        secs = [OracleSector(m2, 6, 6, r, P2) for r in range(P2)]
        d2 = secs[0].Dim
        v2 = rng.standard_normal(d2) + 1j * rng.standard_normal(d2)
        spMatVec_mpi_main(m2, 6, 6, P2, v2, repeat=2, sectors=secs)
        t0 = time.time()
        spMatVec_mpi_main(m2, 6, 6, P2, v2, repeat=20, sectors=secs)
        dt2 = (time.time() - t0) / 20
        for s in secs:
            s.close()
        c2 = {"workload": f"C2: {m2.name} sector (6,6) Dim={d2}", "thread_ranks": P2, "s_per_matvec": dt2, "GBs": 32.0 * d2 / dt2 / 1e9}
    except Exception as e:  # (context only)
        c2 = {"failed": str(e)}
    return {"value": best["GBs"], "unit": "GB/s", "cores": best["thread_ranks"], "kind": "port", "cpu_model": cpu_model, "host_cores": host_cpus, "c2": c2,
            "cpus_allowed": allowed, "physical_cores_allowed": phys_cores, "cgroup_cpu_quota": quota, "samples": samples,
            "sample": f"{best['products']} full products of the same sector, {best['thread_ranks']} thread-ranks (physical cores among the {allowed} CPUs of the "
                      f"affinity mask: {phys_cores}{why}; reference spMatVec_mpi_main restated in oracle/hxv_oracle.c; matrix build {best['matrix_build_s']}s untimed)",
            "s_per_matvec": best["s_per_matvec"]}


def copy_rate_gbs(dev, nbytes=1 << 30, reps=5):
    """Measured copy rate of this box (read + written bytes per second of a plain device copy): the yardstick of the two-pass
    design's own floor (80 B per basis state)."""
    import torch

    x = torch.empty(nbytes // 8, dtype=torch.float64, device=dev).normal_()
    y = torch.empty_like(x)
    y.copy_(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def time_other_workload(name, dev, reps):
    """One of the other full-size BASELINE configs on this GPU: mean ms per product (HIP events on the launch stream)."""
    import torch
    import hxv
    from hxv import models

    model, (nup, ndw) = {"C4": (models.bhz_2d(Nbath=1), (8, 8)), "C5": (models.hm_ring(6, 2), (9, 9))}[name]
    sec = hxv.HxvSector.from_model(model, nup, ndw, device=dev.index or 0)
    n = sec.fullElems
    v = torch.empty(n, dtype=torch.complex128, device=dev)
    vr = torch.view_as_real(v).view(-1)
    g = torch.Generator(device=dev).manual_seed(7)
    for a in range(0, 2 * n, 1 << 28):   # (chunks: torch.randn on 75 GB at once would need a second buffer of that size)
        b = min(a + (1 << 28), 2 * n)
        vr[a:b] = torch.randn(b - a, dtype=torch.float64, device=dev, generator=g)
    hv = torch.empty(sec.localElems, dtype=torch.complex128, device=dev)
    sec.time_apply(v, hv, 1)
    ms = sec.time_apply(v, hv, reps)
    out = {"workload": f"{name}: {model.name} sector ({nup},{ndw}) Dim={sec.Dim}", "ms_per_product": round(ms, 4),
           "achieved_GBs": round(32.0 * sec.Dim / (ms * 1e-3) / 1e9, 1), "frac": round(32.0 * sec.Dim / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    sec.close()
    del v, hv, vr
    torch.cuda.empty_cache()
    hxv.pool_trim(dev.index or 0)
    return out


def exchange_link_figures(sec, world, exchange):
    """(bytes one GPU receives per product, link-bound ms) of an open split sector under `exchange` (xGMI is point to point --
    every peer's bytes arrive over that peer's own link, 153 GB/s each way, 7 links per GPU -- so the bound is the LARGEST per-peer ingest
    over one link's rate).  all-gather: one slab per peer; two transposes: 2 x slab/world per peer; halo: the busiest peer's columns."""
    slab = 16 * (-(-sec.DimDw // world)) * sec.pitch
    if exchange == "halo" and sec.exchange_mode == "halo":
        cols = sec.halo_lists(world)[0]          # (the engine's own receive lists)
        ingest, per_peer = 16 * sec.pitch * int(cols.sum()), 16 * sec.pitch * int(cols.max())
    elif exchange == "alltoall":
        ingest, per_peer = 2 * (world - 1) * slab // world, 2 * slab // world
    else:
        ingest, per_peer = (world - 1) * slab, slab
    return ingest, per_peer / (XGMI_LINK_GBS * 1e9) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="C3", choices=["C2", "C3", "C4", "C5"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lanczos", action="store_true", help="skip the Lanczos-iteration timing (N=1; second half of BASELINE's metric)")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the C4 / C5 product timings reported in config.other_workloads (N=1)")
    ap.add_argument("--no-gf-solve", action="store_true",
                    help="skip the solve-shaped leg (N=1, C3): the 56 Green's-function channels of one default solve, the target sector opened and "
                         "closed around every channel as ED_GF_NORMAL.f90:208-222 does (config.gf_solve; about a minute)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the N>1 path on one GPU)")
    ap.add_argument("--exchange", default="allgather", choices=["allgather", "alltoall", "halo"],
                    help="N>1: allgather = one RCCL all-gather per product (BASELINE's mandated exchange, default); "
                         "alltoall = the reference's own two transposes per product (lower traffic); "
                         "halo = only the columns H_dw couples across ranks travel (one all-to-all with per-peer counts)")
    ap.add_argument("--parallelism", default="dimdw", choices=["dimdw", "sectors"],
                    help="N>1: dimdw = ONE sector split along DimDw with an exchange per product (BASELINE's scheme, strong scaling, default); "
                         "sectors = every GPU runs its own whole sector, no exchange (how independent sectors / Green's-function channels "
                         "of one ED solve spread over a node; weak scaling)")
    ap.add_argument("--rehearse-capi", action="store_true",
                    help="one GPU, launched through torch.distributed.run with ONE rank: run exactly the calls of the N>1 path "
                         "(process group on RCCL, hxv_comm_unique_id -> broadcast -> hxv_comm_init -> hxv_slab_home -> hxv_apply_device_slab, "
                         "barrier, max over ranks) on a one-rank communicator")
    ap.add_argument("--capi", action="store_true",
                    help="N>1 with --backend gloo: keep the C-ABI exchange (hxv_comm_init + hxv_apply_device_slab) for the data path and use gloo "
                         "for the control plane only -- with HXV_RCCL_LIB pointing at tests/rccl_double/_build/librccl_double_mp.so this is how N "
                         "processes on ONE GPU rehearse the driver's launch line through the engine's RCCL branches")
    ap.add_argument("--no-other-exchanges", action="store_true", help="N>1 through the C-ABI: do not open, check and time the two exchanges that were not asked for (config.other_exchanges)")
    ap.add_argument("--no-apply-host", action="store_true", help="skip the host-array leg (N=1): hxv_apply_host with its two PCIe copies against the box's own link rates (config.apply_host)")
    ap.add_argument("--no-check", action="store_true", help="N>1: skip the one checked product before the warm-up (split vs unsplit product on every rank)")
    ap.add_argument("--check", action="store_true", help="(kept for old command lines: the check is the default now)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import hxv
    from hxv import models

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    local_rank = local_rank % max(torch.cuda.device_count(), 1)   # (rehearsals put several ranks on one GPU)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.rehearse_capi   # the distributed code path (a rehearsal runs it with one rank)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    if args.workload == "C2":
        model, (nup, ndw) = models.hm_1dchain(), (6, 6)
    elif args.workload == "C3":
        model, (nup, ndw) = models.hm_2dsquare(Nbath=3), (8, 8)
    elif args.workload == "C4":
        model, (nup, ndw) = models.bhz_2d(Nbath=1), (8, 8)
    else:
        model, (nup, ndw) = models.hm_ring(6, 2), (9, 9)

    by_sector = world > 1 and args.parallelism == "sectors"
    # N>1 on RCCL (the default): the exchange runs behind the C-ABI, exactly what a Fortran rank of the reference would call
    # (hxv_comm_unique_id on rank 0 -> the host program's own broadcast -> hxv_comm_init -> hxv_apply_device_slab), for all three
    # exchanges.  --backend gloo (CPU rendezvous, rehearsals only) goes through the torch twin hxv/distributed.py instead.
    capi_exchange = multi and not by_sector and (args.backend == "nccl" or args.capi)
    twin = multi and not by_sector and not capi_exchange
    if multi and not by_sector and args.exchange != "allgather":
        hxv.set_exchange_default(args.exchange if capi_exchange or args.exchange == "halo" else "allgather")   # layout chosen when the sector is opened
    if by_sector:
        sec = hxv.HxvSector.from_model(model, nup, ndw, device=local_rank)      # the whole sector on every GPU
    else:
        sec = hxv.HxvSector.from_model(model, nup, ndw, rank=rank, nranks=world, device=local_rank)
    hxv.set_exchange_default("allgather")
    Dim, Nloc = sec.Dim, sec.localElems   # Nloc: local vector length in the padded device layout (include/hxv.h)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    v_local = torch.randn(Nloc, dtype=torch.float64, device=dev, generator=g) + 1j * torch.randn(Nloc, dtype=torch.float64, device=dev, generator=g)
    torch.view_as_real(v_local).view(-1, sec.pitch, 2)[:, sec.DimUp:, :] = 0.0   # (pad rows are zero in every device vector)
    hv_local = torch.empty(Nloc, dtype=torch.complex128, device=dev)
    if multi and not by_sector and world > 1 and sec.exchange_mode != args.exchange and (capi_exchange or args.exchange == "halo"):
        # (the spH0nd block keeps the all-gather exchange in mode 2; tiny sectors keep it too): report what actually runs
        if rank == 0:
            print(f"bench.py: the sector was opened with the {sec.exchange_mode} exchange; --exchange {args.exchange} does not apply", file=sys.stderr)
        args.exchange = sec.exchange_mode
    halo = twin and args.exchange == "halo"
    sh = hx = th = None
    if capi_exchange:
        ident = [hxv.HxvSector.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        sec.comm_init(ident[0])
        # the slab lives where the exchange wants it (hxv_slab_home), as a device-resident Lanczos vector can: no slab copy per product
        if sec.exchange_mode != "alltoall":
            home = sec.slab_home()
            home.copy_(v_local)
            v_local = home
    elif twin:
        # torch twin of the three exchanges (hxv/distributed.py): --backend gloo only
        sh = hxv.ShardedHxv(sec.DimUp, sec.DimDw, rank, world, sec.apply_device, pitch=sec.pitch)
        if halo:
            if model.Norb > 1 and (model.Jx != 0 or model.Jp != 0):
                raise SystemExit("bench.py: --exchange halo with Jx / Jp runs through the C-ABI only (--backend nccl)")
            rp, cols, _ = sec.csr("dw")
            need, send = hxv.halo_plan(rp, cols - 1, sec.DimDw, world)
            hx = hxv.HaloHxv(sec.DimUp, sec.DimDw, rank, world, need, send, sec.apply_device, pitch=sec.pitch, stage_on_host=True)
        if args.exchange == "alltoall":
            nrows = hxv.dw_split(sec.DimUp, rank, world)[0]
            panel = hxv.HxvSector.dw_panel(model, nup, ndw, nrows, device=local_rank)
            th = hxv.TransposedHxv(sec.DimUp, sec.DimDw, rank, world, panel.apply_dw_panel, sec.apply_up_add, pitch=sec.pitch,
                                   pitch_panel=panel.pitch, stage_on_host=True)

    def step():
        if capi_exchange:
            sec.apply_device_slab(v_local, hv_local)
        elif th is not None:
            th(Nloc, v_local, hv_local)
        elif hx is not None:
            hx(Nloc, v_local, hv_local)
        elif sh is not None:
            sh(Nloc, v_local, hv_local)
        else:
            sec.apply_device(v_local, hv_local)

    checked, check_err = False, None
    if multi and not by_sector and not args.no_check:
        # ONE product before the warm-up, checked on every rank: this rank's slab of the split product == the same slab of the
        # unsplit product of the assembled vector (the N>1 path has never met a second GPU: the first run on a node must not be blind)
        step()
        cmax = -(-sec.DimDw // world)
        mine = torch.zeros(cmax * sec.pitch, dtype=torch.complex128, device=dev)
        mine[:Nloc] = v_local
        parts = [torch.empty_like(mine) for _ in range(world)]
        if args.backend == "nccl":
            dist.all_gather(parts, mine)
        else:
            hp = [torch.empty(mine.shape, dtype=mine.dtype) for _ in range(world)]
            dist.all_gather(hp, mine.cpu())
            parts = [x.to(dev) for x in hp]
        qs = [sec.DimDw // world + (1 if r < sec.DimDw % world else 0) for r in range(world)]   # ED_HAMILTONIAN.f90:93-98
        vg = torch.cat([parts[r][: qs[r] * sec.pitch] for r in range(world)])
        del parts, mine
        full_sec = hxv.HxvSector.from_model(model, nup, ndw, device=local_rank)
        c0 = sum(qs[:rank])
        ref = full_sec.apply_device(vg)[c0 * sec.pitch: (c0 + qs[rank]) * sec.pitch]
        torch.cuda.synchronize()
        ref, got = (x.view(-1, sec.pitch)[:, : sec.DimUp] for x in (ref, hv_local))   # (pad rows are never written by the product)
        check_err = (ref - got).abs().max().item() / max(ref.abs().max().item(), 1e-300)
        del got
        full_sec.close()
        del vg, ref
        torch.cuda.empty_cache()
        hxv.pool_trim(local_rank)
        t = torch.tensor([check_err], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        check_err = float(t.item())
        checked = True
        if check_err > 1e-13:
            raise SystemExit(f"bench.py: the split product differs from the unsplit one (max rel err over ranks {check_err:.2e}); no number reported")
    for _ in range(args.warmup):
        step()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_step = dt / args.steps * 1e3
    value = (world if by_sector else 1) * 32.0 * Dim / (ms_step * 1e-3) / 1e9   # sectors: every rank finished a whole product

    # roofline of the product's kernels on this rank: HIP events on the stream they are launched on
    nk = max(5, min(args.steps, 20))
    if capi_exchange:
        step_ev_ms, k_ms = sec.time_apply_slab(v_local, hv_local, nk)      # (collective: every rank runs the same nk products)
    else:
        vfull = hx.exchange(v_local) if hx is not None else (sh.gather(v_local) if sh is not None else v_local)
        torch.cuda.synchronize()
        k_ms = sec.time_apply(vfull, hv_local, nk)
        step_ev_ms = None
        del vfull
    achieved = 32.0 * sec.vecDim / (k_ms * 1e-3) / 1e9
    # HBM-side bytes per product come from a separate rocprofv3 --pmc collection (scripts/prof_traffic.sh ->
    # profiles/traffic.json); the figure is only quoted for the kernel build it was collected on (its `kernels` stamp)
    traffic, traffic_src = None, None
    tf = ROOT / "profiles" / "traffic.json"
    if tf.exists():
        try:
            tj = json.loads(tf.read_text())
            if tj.get("workload") == args.workload and tj.get("n_gpus", 1) == world and tj.get("kernels_stamp") == KERNELS_STAMP:
                traffic = tj.get("hbm_bytes_per_product")
                traffic_src = f"profiles/traffic.json (rocprofv3 --pmc, kernels {KERNELS_STAMP})"
        except Exception:
            traffic = None
    # the two-pass design's own floor: 80 B per basis state (pass B 32, pass A 48) at the guide's measured copy rate, 6.29 TB/s -- a FIXED
    # yardstick (the box's own copy rate moved the fraction between runs); that rate is reported beside it
    copy_gbs = copy_rate_gbs(dev)
    floor_ms = 80.0 * sec.vecDim / (HBM_COPY_GBS * 1e9) * 1e3
    # the rocprofv3 --kernel-trace summary of this command, quoted only when it was collected on THIS kernel build (profiles/kernel_trace.json)
    trace = None
    kt = ROOT / "profiles" / "kernel_trace.json"
    if kt.exists():
        try:
            kj = json.loads(kt.read_text())
            if kj.get("workload") == args.workload and kj.get("n_gpus", 1) == world and kj.get("kernels_stamp") == KERNELS_STAMP:
                trace = {"file": kj.get("file"), "avg_ms": kj.get("avg_ms"), "product_ms": kj.get("product_ms")}
        except Exception:
            trace = None
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "design_floor_ms": round(floor_ms, 4), "frac_of_design_floor": round(floor_ms / k_ms, 4), "design_floor_rate_GBs": HBM_COPY_GBS,
                "copy_rate_GBs": round(copy_gbs, 1), "kernels_stamp": KERNELS_STAMP, "rocprof_kernel_trace": trace,
                "traffic": traffic, "traffic_source": traffic_src, "kernel": ("hxv_up_job" if sec.get_option("job_up_active") else "hxv_pass_up") + " + hxv_pass_dw (one product)", "kernel_ms": round(k_ms, 4),
                "algorithmic_bytes": 32 * sec.vecDim}
    if step_ev_ms is not None:
        roofline["slab_product_ms_on_stream"] = round(step_ev_ms, 4)   # exchange + kernels of this rank, HIP events (hxv_time_apply_slab)
        # what the exchange costs each rank: the slab product on the stream minus its kernels; min / max over the ranks
        # (overlapped mode 2 runs pass A on a second stream BESIDE the exchange: that part of kernel_ms is not subtracted -- ADVICE r5)
        k_ovl = sec.get_option("time_kernels_overlapped_us") * 1e-3
        if k_ovl > 0:
            roofline["kernel_ms_overlapped"] = round(k_ovl, 4)
        ex_ms = step_ev_ms - (k_ms - k_ovl)
        if multi:
            t2 = torch.tensor([ex_ms, -ex_ms], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            roofline["exchange_ms"] = {"this_rank": round(ex_ms, 4), "max": round(float(t2[0].item()), 4), "min": round(-float(t2[1].item()), 4)}
        else:
            roofline["exchange_ms"] = {"this_rank": round(ex_ms, 4), "max": round(ex_ms, 4), "min": round(ex_ms, 4)}

    # FIRST CONTACT WITH N GPUs RECORDS EVERY EXCHANGE: `value` above is the exchange asked for (default: the all-gather
    # BASELINE mandates); the other two are opened, checked and timed in the same launch -- the sector re-opened under the other layout
    # (hxv_set_exchange_default), joined through hxv_comm_init (the process's communicator serves it: no second ncclCommInitRank), ONE product
    # compared with the checked product of the main exchange on every rank, then warm-up + timed steps + the kernels' share like above.
    other_exchanges = None
    if capi_exchange and world > 1 and not args.no_other_exchanges:
        other_exchanges = {}
        cpu_or_dev = dev if args.backend == "nccl" else "cpu"
        step()                                                   # hv_local = the main exchange's product of v_local (checked above)
        torch.cuda.synchronize()
        hv_ref = hv_local.clone()
        v_ref = v_local.clone()
        def all_ok(flag: bool) -> bool:
            """every rank made it here without an error (a rank that failed alone must not leave its peers inside the next collective)"""
            t_ = torch.tensor([0.0 if flag else 1.0], dtype=torch.float64, device=cpu_or_dev)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            return float(t_.item()) == 0.0

        for mode in ("allgather", "alltoall", "halo"):
            if mode == args.exchange:
                continue
            # (the main exchange's number is measured and checked already: whatever goes wrong here is RECORDED under its name, on every rank
            #  alike, and never takes the line down)
            s2, v2, hv2, note = None, None, None, ""
            try:
                hxv.set_exchange_default(mode)
                s2 = hxv.HxvSector.from_model(model, nup, ndw, rank=rank, nranks=world, device=local_rank)
            except Exception as e:  # noqa: BLE001
                note = f"open: {e}"
            finally:
                hxv.set_exchange_default("allgather")
            if not all_ok(s2 is not None):
                other_exchanges[mode] = {"failed": note or "a peer could not open the sector"}
                if s2 is not None:
                    s2.close()
                continue
            if s2.exchange_mode != mode:     # (the same on every rank: decided by the model and the sector's size)
                other_exchanges[mode] = {"skipped": f"the sector opens with the {s2.exchange_mode} exchange (spH0nd block or a tiny sector)"}
                s2.close()
                continue
            ident2 = [hxv.HxvSector.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ident2, src=0)
            try:
                s2.comm_init(ident2[0])          # (binds to the process's communicator: no second ncclCommInitRank)
                v2 = v_ref
                if s2.exchange_mode != "alltoall":
                    v2 = s2.slab_home()
                    v2.copy_(v_ref)
                hv2 = torch.empty_like(hv_ref)
            except Exception as e:  # noqa: BLE001
                note = f"prepare: {e}"
                hv2 = None
            if not all_ok(hv2 is not None):
                other_exchanges[mode] = {"failed": note or "a peer could not prepare the exchange"}
                s2.close()
                continue
            s2.apply_device_slab(v2, hv2)
            torch.cuda.synchronize()
            a, b = (x.view(-1, s2.pitch)[:, : s2.DimUp] for x in (hv_ref, hv2))
            err = (a - b).abs().max().item() / max(a.abs().max().item(), 1e-300)
            t = torch.tensor([err], dtype=torch.float64, device=cpu_or_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            err = float(t.item())
            if not err <= 1e-13:
                other_exchanges[mode] = {"failed": f"its product differs from the checked {args.exchange} one (max rel err over ranks {err:.2e}); not timed"}
                s2.close()
                continue
            n_o = max(3, min(args.steps, 20))
            for _ in range(max(1, min(args.warmup, 5))):
                s2.apply_device_slab(v2, hv2)
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_o):
                s2.apply_device_slab(v2, hv2)
            torch.cuda.synchronize()
            dist.barrier()
            dt2 = time.perf_counter() - t0
            t = torch.tensor([dt2], dtype=torch.float64, device=cpu_or_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms2 = float(t.item()) / n_o * 1e3
            ev_ms, k2_ms = s2.time_apply_slab(v2, hv2, max(3, min(n_o, 10)))
            ex2 = ev_ms - k2_ms
            t2 = torch.tensor([ex2, -ex2], dtype=torch.float64, device=cpu_or_dev)
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            ingest2, lb2 = exchange_link_figures(s2, world, mode)
            other_exchanges[mode] = {"ms_per_step": round(ms2, 4), "GBs": round(32.0 * Dim / (ms2 * 1e-3) / 1e9, 1), "steps": n_o, "check_rel_err_vs_main": err,
                                     "link_bound_ms": round(lb2, 4), "exchange_ingest_bytes_per_gpu": ingest2, "kernel_ms": round(k2_ms, 4),
                                     "exchange_ms": {"this_rank": round(ex2, 4), "max": round(float(t2[0].item()), 4), "min": round(-float(t2[1].item()), 4)}}
            s2.close()
            del hv2, v2
            torch.cuda.empty_cache()
        del hv_ref, v_ref

    ns = {"C2": 12, "C3": 16, "C4": 16, "C5": 18}[args.workload]
    out = {"metric": f"sector-HxV achieved HBM GB/s (algorithmic 32 B x Dim per product), Ns={ns} half-filled sector", "value": round(value, 1),
           "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
           "higher_is_better": True, "scaling": "weak" if by_sector else "strong", "vs_baseline": None, "dtype": "complex128 (f64)", "data": "synthetic",
           "config": {"workload": f"{args.workload}: {model.name} sector ({nup},{ndw}) Dim={Dim}", "DimUp": sec.DimUp, "DimDw": sec.DimDw,
                      "device_row_order": sec.row_perm is not None,
                      "parallelism": f"{world} independent sectors, one per GPU, no exchange" if by_sector else f"DimDw split x{world}" + ({"allgather": " + RCCL allgather per product", "halo": " + RCCL send/recv of the columns H_dw couples across ranks",
                                                                  "alltoall": " + 2 RCCL all-to-all transposes per product"}[args.exchange] if world > 1 else ""),
                      "matvecs_per_s": round((world if by_sector else 1) * 1e3 / ms_step, 2)},
           "roofline": roofline}
    if multi and not by_sector:
        out["checked"] = checked
        out["check_rel_err"] = check_err
        out["config"]["transport"] = "C-ABI (hxv_comm_init + hxv_apply_device_slab over RCCL)" if capi_exchange else f"torch.distributed ({args.backend}) twin, rehearsal only"
        # WHICH librccl ran: the one the engine dlopen()ed for its communicator, and the one(s) mapped into this process (torch's own)
        out["config"]["rccl_lib"] = sec.comm_library if capi_exchange else None
        try:
            out["config"]["rccl_libs_mapped"] = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "rccl" in ln.lower() and "/" in ln})
        except OSError:
            out["config"]["rccl_libs_mapped"] = []
    if world > 1 and not by_sector:
        # what the exchange moves into every GPU per product: the xGMI links, not HBM, bound the N>1 product
        out["config"]["exchange"] = args.exchange
        ingest, lb_ms = exchange_link_figures(sec, world, args.exchange)
        out["config"]["exchange_ingest_bytes_per_gpu"] = ingest
        # First-contact fields ("the result JSON should carry the link-bound alongside the measurement"): exchange_link_figures
        out["config"]["link_GBs_assumed"] = XGMI_LINK_GBS
        out["config"]["link_bound_ms"] = round(lb_ms, 4)
        out["config"]["link_bound_note"] = "largest per-peer ingest of one product / one xGMI link (point to point, 7 links per GPU); compute overlaps none of it in the default exchanges"
        if other_exchanges is not None:
            out["config"]["other_exchanges"] = other_exchanges
            out["config"]["comm_cache"] = hxv.comm_cache_stats()   # one ncclCommInitRank served every sector this process opened
    if world == 1 and not args.rehearse_capi:
        # what each of the three exchanges would move into one GPU per product at 8 ranks
        rp, cols, _ = sec.csr("dw")
        need8, _ = hxv.halo_plan(rp, cols - 1, sec.DimDw, 8)
        out["config"]["exchange_ingest_bytes_per_gpu_at_8_ranks"] = hxv.exchange_ingest_bytes(sec.DimUp, sec.DimDw, 8, need8)
    if world == 1 and not args.rehearse_capi and not args.no_apply_host:
        # THE DROP-IN SURFACE ITSELF: north_star keeps spHtimesV_p on HOST arrays (cc_sparse_HxV, ED_VARS_GLOBAL.f90:72-78; caller
        # ED_DIAG.f90:145,152) -- hxv_apply_host = H2D of v, the product, D2H of Hv.  Host arrays page-locked ONCE (hxv_host_register), then three
        # products; the floor beside it = the same bytes at this box's own H2D and D2H rates (plain pinned copies of the same size) + the kernels.
        try:
            import numpy as np

            nb = 16 * sec.vecDim
            vh = np.empty(sec.vecDim, dtype=np.complex128)
            hh = np.empty(sec.vecDim, dtype=np.complex128)
            vh.real[:] = 1.0 / np.sqrt(sec.vecDim)
            vh.imag[:] = 0.0
            t0 = time.perf_counter()
            sec.apply_host(vh, hh)                                  # pageable arrays (what an unmodified host hands over), first call: staging buffers
            t_first = time.perf_counter() - t0
            t0 = time.perf_counter()
            sec.apply_host(vh, hh)
            t_page = time.perf_counter() - t0
            t0 = time.perf_counter()
            hxv.host_register(vh)
            hxv.host_register(hh)
            t_reg = time.perf_counter() - t0
            sec.apply_host(vh, hh)
            t0 = time.perf_counter()
            for _ in range(3):
                sec.apply_host(vh, hh)
            t_pin = (time.perf_counter() - t0) / 3
            # the box's own link rates for the same byte count: plain copies between a page-locked host buffer and the device, each way, twice
            dbuf = torch.empty(sec.vecDim, dtype=torch.complex128, device=dev)
            pbuf = torch.empty(sec.vecDim, dtype=torch.complex128, pin_memory=True)
            rates = {}
            for name, dst, src in (("h2d", dbuf, pbuf), ("d2h", pbuf, dbuf)):
                dst.copy_(src, non_blocking=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(2):
                    dst.copy_(src, non_blocking=True)
                torch.cuda.synchronize()
                rates[name] = nb * 2 / (time.perf_counter() - t0) / 1e9
            del pbuf
            del dbuf
            hxv.host_unregister(vh)
            hxv.host_unregister(hh)
            del vh, hh
            floor_h = nb / (rates["h2d"] * 1e9) * 1e3 + k_ms + nb / (rates["d2h"] * 1e9) * 1e3
            out["config"]["apply_host"] = {"what": "hxv_apply_host on host arrays (the spHtimesV_p surface itself): H2D of v + product + D2H of Hv, arrays page-locked once",
                                           "ms_per_product": round(t_pin * 1e3, 2), "h2d_GBs": round(rates["h2d"], 1), "d2h_GBs": round(rates["d2h"], 1),
                                           "pcie_floor_ms": round(floor_h, 2), "ratio_to_floor": round(t_pin * 1e3 / floor_h, 3),
                                           "bytes_each_way": nb, "kernels_ms": round(k_ms, 3), "ms_per_product_pageable": round(t_page * 1e3, 2),
                                           "first_call_ms": round(t_first * 1e3, 1), "host_register_ms": round(t_reg * 1e3, 1),
                                           "GBs_algorithmic": round(32.0 * sec.vecDim / t_pin / 1e9, 1)}
        except Exception as e:  # noqa: BLE001 (a leg beside the headline: reported, never fatal)
            out["config"]["apply_host"] = {"failed": str(e)}
    if not args.no_lanczos and world == 1 and not args.rehearse_capi:
        # full iterations: product + fused recurrence + 2 reductions, vectors in HBM.  Headline = complex(8) vectors, the
        # reference's data type; when H is real (C2, C3) the device drivers also run on real vectors (half the bytes).
        sec.set_option("real_vectors", 0)
        lz_ms = sec.time_lanczos(20)
        out["config"]["lanczos_ms_per_iter"] = round(lz_ms, 4)
        out["config"]["lanczos_matvecs_per_s"] = round(1e3 / lz_ms, 2)
        # the Lanczos half of the metric against the roofline.  Bytes CHARGED per iteration and basis state, complex vectors: the product
        # 32 + w -= beta x_prev 16 (read x_prev; fused into the product's write) + w -= alpha x, |w| 48 (read w, x, write w: a
        # pass of its own, alpha is a global reduction of the product) = 96; the design moves 144 (its two-pass product moves 80).
        lz_bytes = 96 * sec.vecDim
        roofline["lanczos"] = {"bytes_per_state": 96, "bytes": lz_bytes, "ms_per_iter": round(lz_ms, 4), "achieved": round(lz_bytes / (lz_ms * 1e-3) / 1e9, 1),
                               "unit": "GB/s", "frac": round(lz_bytes / (lz_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "design_bytes_per_state": 144,
                               "frac_of_design_floor": round(144.0 * sec.vecDim / (HBM_COPY_GBS * 1e9) * 1e3 / lz_ms, 4)}
        sec.set_option("real_vectors", 1)
        if sec.real_vectors_available:
            lzr_ms = sec.time_lanczos(20)
            out["config"]["lanczos_real_vectors_ms_per_iter"] = round(lzr_ms, 4)
            out["config"]["lanczos_real_vectors_matvecs_per_s"] = round(1e3 / lzr_ms, 2)
            # REAL vectors (real H): half of every byte count above -- product 16, iteration 48 charged, 72 moved by the design
            vr = torch.randn(sec.DimDw * sec.DimUp, dtype=torch.float64, device=dev, generator=g)
            vr = sec.pad_real(vr)
            hr = torch.empty_like(vr)
            for _ in range(3):
                sec.apply_device_real(vr, hr)
            e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0_.record()
            for _ in range(20):
                sec.apply_device_real(vr, hr)      # (launched on torch's current stream: the events see it)
            e1_.record()
            torch.cuda.synchronize()
            pr_ms = e0_.elapsed_time(e1_) / 20
            del vr, hr
            rb, rlz = 16 * sec.vecDim, 48 * sec.vecDim
            roofline["real_vectors"] = {"product_bytes_per_state": 16, "product_ms": round(pr_ms, 4), "product_achieved": round(rb / (pr_ms * 1e-3) / 1e9, 1),
                                        "product_frac": round(rb / (pr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                        "product_frac_of_design_floor": round(40.0 * sec.vecDim / (HBM_COPY_GBS * 1e9) * 1e3 / pr_ms, 4),
                                        "lanczos_bytes_per_state": 48, "lanczos_ms_per_iter": round(lzr_ms, 4), "lanczos_achieved": round(rlz / (lzr_ms * 1e-3) / 1e9, 1),
                                        "lanczos_frac": round(rlz / (lzr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "unit": "GB/s"}
        if sec.real_vectors_available:
            # two real tridiagonalisations per complex product (hxv_lanczos_tridiag_pair: two Green's-function channels at once)
            va = torch.zeros(sec.localElems, dtype=torch.complex128, device=dev)
            vb = torch.zeros_like(va)
            for w in (va, vb):
                x = torch.randn(sec.DimDw, sec.DimUp, dtype=torch.float64, device=dev, generator=g)
                torch.view_as_real(w).view(sec.DimDw, sec.pitch, 2)[:, : sec.DimUp, 0] = x / x.norm()
            def wall(n):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                sec.lanczos_tridiag_pair(va, vb, n)
                torch.cuda.synchronize()
                return time.perf_counter() - t0
            wall(3)
            t_short, t_long = wall(4), wall(24)
            pair_ms = (t_long - t_short) / 20 * 1e3
            out["config"]["lanczos_paired_real_ms_per_iter"] = round(pair_ms, 4)           # one iteration of BOTH channels
            out["config"]["lanczos_paired_real_ms_per_iter_per_channel"] = round(pair_ms / 2, 4)
            out["config"]["lanczos_paired_real_matvecs_per_s"] = round(2e3 / pair_ms, 2)
            del va, vb
    if world == 1 and args.workload == "C3" and not args.no_gf_solve and not args.no_lanczos and not args.rehearse_capi:
        # the callers' usage pattern as a measured whole (scripts/harness.py): ground state of (8,8) by the default spectrum call, then the
        # 56 tridiagonalisations of build_gf_normal (ED_GF_NORMAL.f90:36-110; 8 diagonal + 24 real mixed + 24 complex mixed: chan4, the
        # default), each with its sector N+-1 opened and closed around it (:208-222), device-resident, two real channels per product
        from harness import gf_solve

        hxv.sector_cache_clear()
        recs, gs_ = gf_solve(model, nup, ndw, nlanc=200, symmetric=False, device=local_rank)
        real_part = [r for r in recs if r["kind"] != "mix_xi"]
        chan2_s = (gs_["gs_open_ms"] + gs_["gs_ms"] + sum(r["open_ms"] / (2 if r["paired"] else 1) + r["start_ms"] / (2 if r["paired"] else 1) + r["tridiag_ms"]
                                                           + r["close_ms"] / (2 if r["paired"] else 1) for r in real_part)) * 1e-3
        out["config"]["gf_solve"] = {
            "what": "ground state + the 56 channels of one default solve (ed_gf_symmetric=F: chan4), nlanc 200, sector opened/closed per channel",
            "gf_solve_s": round(gs_["gf_solve_s"], 2), "ground_state_s": round(gs_["gs_ms"] * 1e-3, 3), "ground_state_products": gs_["gs_nmatvec"],
            "channels": gs_["channels"], "channels_real": gs_["channels_real"], "channels_complex": gs_["channels_complex"], "channels_paired": gs_["channels_paired"],
            "real_channels_s": round(gs_["real_channels_s"], 2), "complex_channels_s": round(gs_["complex_channels_s"], 2),
            "ms_per_channel_step_real_paired": round(gs_["real_channels_s"] * 1e3 / max(1, gs_["channels_real"]) / 200, 3),
            "ms_per_channel_step_complex": round(gs_["complex_channels_s"] * 1e3 / max(1, gs_["channels_complex"]) / 200, 3),
            "sector_opens": gs_["sector_opens"], "sector_open_cache_hits": gs_["sector_open_cache_hits"],
            "sector_open_ms": {"first": round(gs_["sector_open_ms_first"], 2), "mean": round(gs_["sector_open_ms_mean"], 3), "max": round(gs_["sector_open_ms_max"], 2)},
            "gf_solve_symmetric_s": round(chan2_s, 2),
            "gf_solve_symmetric_note": "ed_gf_symmetric=T (chan2: the 32 real channels only) = the real-channel part of the same run; for real symmetric "
                                       "H the complex channels carry nothing the real ones do not (INTEGRATION.md section 4)"}
        del recs
        torch.cuda.empty_cache()
    if world == 1 and args.workload == "C3" and not args.no_other_workloads and not args.rehearse_capi:
        # the other full-size configs, driver-timed on the same GPU (parity-test sizes of BASELINE.json, not the headline)
        sec.close()
        del v_local, hv_local
        torch.cuda.empty_cache()
        hxv.pool_trim(local_rank)
        ow = {}
        for w in ("C4", "C5"):
            try:
                ow[w] = time_other_workload(w, dev, 10 if w == "C4" else 3)
            except Exception as e:  # noqa: BLE001 (parity-test sizes beside the headline: reported, never fatal -- C5 needs ~150 GB of free HBM)
                ow[w] = {"failed": str(e)}
                torch.cuda.empty_cache()
                hxv.pool_trim(local_rank)
        out["config"]["other_workloads"] = ow
        v_local = hv_local = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.rehearse_capi:
        sec.close()
        v_local = hv_local = None
        torch.cuda.empty_cache()
        try:
            out["cpu_baseline"] = cpu_baseline(model, nup, ndw)
        except Exception as e:  # the baseline is context, never the product: report, do not fail the bench
            out["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if multi:
        # every rank at the same point: close the sector, destroy the process's RCCL communicator (hxv_comm_cache_clear), then torch's
        try:
            sec.close()
            hxv.comm_cache_clear()
        except Exception:  # noqa: BLE001 (the line is out: never fail on the way down)
            pass
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
