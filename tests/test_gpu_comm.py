"""The slab exchange behind the C-ABI (hxv_comm_*, include/hxv.h): RCCL communicator per open sector, products and device
Lanczos drivers through it.  One GPU here, so the communicator has ONE rank: ncclAllGather / ncclAllReduce run, the numbers must
equal the serial path exactly.  (N>1 on hardware is the driver's scaling run; the N>1 arithmetic of the exchange -- unequal
slabs, all-gather layout -- is covered by tests/test_gpu_parity.py and the gloo tests.)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-13


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_world_size_one_rccl_product_and_drivers(built):
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    m = models.hm_1dchain(eps_bath=[0.3, 0.6])
    ser = hxv.HxvSector.from_model(m, 6, 6)
    sec = hxv.HxvSector.from_model(m, 6, 6)
    sec.comm_init(hxv.HxvSector.comm_unique_id())
    orc = OracleSector(m, 6, 6)
    v = models.deterministic_vector(sec.Dim)
    v /= np.linalg.norm(v)
    ref = orc.spMatVec_main(v)
    # host-array product = spMatVec_MPI_main on the (one) slab
    assert _rel(sec.apply_host(v), ref) <= TOL
    n0 = sec.exchange_count
    assert n0 >= 1
    dv = sec.pad(torch.from_numpy(v).cuda())
    hv = sec.unpad(sec.apply_device_slab(dv))
    torch.cuda.synchronize()
    assert _rel(hv.cpu().numpy(), ref) <= TOL and sec.exchange_count == n0 + 1
    home = sec.slab_home()                                   # the slab in its slot of the gather buffer: the all-gather runs in place
    home.copy_(dv)
    assert _rel(sec.unpad(sec.apply_device_slab(home)).cpu().numpy(), ref) <= TOL
    # Lanczos drivers: all-reduced dots, plain recurrence -> same numbers as the serial handle to rounding
    a1, b1, n1 = sec.lanczos_tridiag(torch.from_numpy(v).cuda(), 60)
    ser.set_option("lanczos_fused", 0)
    ser.set_option("real_vectors", 0)
    a0, b0, n0s = ser.lanczos_tridiag(torch.from_numpy(v).cuda(), 60)
    assert n1 == n0s and np.abs(a1 - a0).max() < 1e-11 and np.abs(b1 - b0).max() < 1e-11
    e1, vec1, _ = sec.lanczos_eigh(400, 1e-14)
    e0, vec0, _ = ser.lanczos_eigh(400, 1e-14)
    assert abs(e1 - e0) < 1e-11
    hvec = orc.spMatVec_main(vec1.cpu().numpy())
    assert np.linalg.norm(hvec - e1 * vec1.cpu().numpy()) < 1e-8
    ev1, vecs1, nc1, _ = sec.eigh_lowest(2, 20, 200, 0.0)
    ev0, _, nc0, _ = ser.eigh_lowest(2, 20, 200, 0.0)
    assert nc1 == 2 and nc0 == 2 and np.abs(ev1 - ev0).max() < 1e-10
    # an unnormalised start vector is normalised by the driver (SciFortran's sp_lanc_tridiag does so on its first iteration)
    a2, b2, _ = sec.lanczos_tridiag(torch.from_numpy(3.7 * v).cuda(), 60)
    assert np.abs(a2 - a1).max() < 1e-10 and np.abs(b2 - b1).max() < 1e-10
    for fused in (1, 0):
        ser.set_option("lanczos_fused", fused)
        a3, b3, _ = ser.lanczos_tridiag(torch.from_numpy(3.7 * v).cuda(), 60)
        assert np.abs(a3 - a0).max() < 1e-10 and np.abs(b3 - b0).max() < 1e-10, fused
    sec.close()
    ser.close()


def test_split_sector_without_communicator_fails_loudly(built):
    import hxv
    from hxv import models

    sec = hxv.HxvSector.from_model(models.hm_1dchain(), 6, 6, rank=1, nranks=3)
    v = np.zeros(sec.vecDim, dtype=np.complex128)
    with pytest.raises(hxv.engine.HxvError, match="hxv_comm_init"):
        sec.apply_host(v)
    with pytest.raises(hxv.engine.HxvError, match="hxv_comm_init"):
        sec.lanczos_eigh(10)
    sec.close()


def test_side_stream_then_close_then_reopen(built):
    """buffers of a closed handle go to the device-buffer cache; kernels launched on a CALLER's stream must have
    finished by then (hxv_destroy synchronises the device).  Apply on a side stream, close at once, reopen, apply_host."""
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    m = models.hm_1dchain(eps_bath=[0.3, 0.6])
    orc = OracleSector(m, 6, 6)
    v = models.deterministic_vector(orc.Dim)
    ref = orc.spMatVec_main(v)
    side = torch.cuda.Stream()
    for _ in range(3):
        sec = hxv.HxvSector.from_model(m, 6, 6)
        dv = sec.pad(torch.from_numpy(v).cuda())
        out = torch.zeros(sec.localElems, dtype=torch.complex128, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(20):
                sec.apply_device(dv, out, stream=side.cuda_stream)
        sec.close()                      # no synchronisation by the caller
        sec2 = hxv.HxvSector.from_model(m, 6, 6)
        assert _rel(sec2.apply_host(v), ref) <= TOL
        sec2.close()
        assert _rel(sec.unpad(out).cpu().numpy(), ref) <= TOL


def test_experiment_options_are_gated(built):
    import hxv
    from hxv import models

    sec = hxv.HxvSector.from_model(models.hm_1dchain(), 6, 6)
    old = os.environ.pop("HXV_EXPERIMENTS", None)
    try:
        for name, val in (("debug", 1), ("passes", 1), ("job_debug", 4)):
            with pytest.raises(hxv.engine.HxvError, match="HXV_EXPERIMENTS"):
                sec.set_option(name, val)
        sec.set_option("passes", 3)      # the neutral values stay settable
        sec.set_option("debug", 0)
    finally:
        if old is not None:
            os.environ["HXV_EXPERIMENTS"] = old
    sec.close()


@pytest.mark.parametrize("world", [3, 4])
def test_halo_layout_rehearsal_one_gpu(built, world):
    """The engine's own halo layout (C++ plan, column -> slot table, tiled and job kernels) with every rank of a split sector
    opened in ONE process: each rank's halo-layout vector is filled from the global vector by the lists the engine reports,
    its slab of H v is compared with the oracle, and the C++ plan with the numpy one."""
    import torch
    import hxv
    from hxv import models
    from oracle.oracle import OracleSector

    hxv.set_exchange_default("halo")
    try:
        for m, (nup, ndw) in ((models.hm_2dsquare(Nbath=1), (4, 4)), (models.hm_1dchain(eps_bath=[0.3, 0.6]), (6, 6)), (models.bhz_2d(Nbath=0, Ust=0.4, Jh=0.1), (3, 5))):
            full = OracleSector(m, nup, ndw)
            du, dd = full.DimUp, full.DimDw
            v = models.deterministic_vector(full.Dim)
            ref = full.spMatVec_main(v)
            rp, cols, _ = full.csr("dw")
            need, send = hxv.halo_plan(rp, cols - 1, dd, world)
            V = v.reshape(dd, du)
            for rank in range(world):
                sec = hxv.HxvSector.from_model(m, nup, ndw, rank=rank, nranks=world)
                assert sec.exchange_mode == "halo"
                rc, sc, rcols, scols = sec.halo_lists(world)
                assert np.array_equal(rcols, need[rank]) and np.array_equal(scols, np.concatenate([send[rank][p] for p in range(world)]).astype(np.int32))
                q, c0 = hxv.dw_split(dd, rank, world)
                have = np.concatenate([np.arange(c0, c0 + q), rcols])
                assert sec.fullElems == len(have) * sec.pitch
                buf = np.zeros((len(have), sec.pitch), dtype=np.complex128)
                buf[:, :du] = V[have]
                for job in (1, 0):
                    sec.set_option("job_up", job)
                    hv = sec.unpad(sec.apply_device(torch.from_numpy(buf.reshape(-1)).cuda()))
                    torch.cuda.synchronize()
                    assert _rel(hv.cpu().numpy(), ref[c0 * du:(c0 + q) * du]) <= TOL, (m.name, world, rank, job)
                sec.set_option("kernel", 0)
                hv = sec.unpad(sec.apply_device(torch.from_numpy(buf.reshape(-1)).cuda()))
                assert _rel(hv.cpu().numpy(), ref[c0 * du:(c0 + q) * du]) <= TOL
                sec.close()
    finally:
        hxv.set_exchange_default("allgather")


@pytest.mark.parametrize("exchange", ["allgather", "halo", "alltoall"])
def test_bench_rehearses_the_multi_gpu_calls_with_one_rank(built, exchange):
    """`bench.py --gpus N` (N > 1) cannot run on a one-GPU box; `--rehearse-capi` runs exactly its calls -- process group on RCCL,
    hxv_comm_unique_id -> broadcast -> hxv_comm_init -> hxv_slab_home -> hxv_apply_device_slab, the checked product before the warm-up,
    barrier, max over ranks, the roofline leg through hxv_time_apply_slab -- with one rank, for every --exchange."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--rehearse-capi", "--workload", "C2", "--steps", "3",
                        "--warmup", "1", "--exchange", exchange], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["roofline"]["achieved"] > 0
    assert line["checked"] is True and line["check_rel_err"] <= 1e-13 and "C-ABI" in line["config"]["transport"]
    assert line["roofline"]["kernel_ms"] > 0 and line["roofline"]["slab_product_ms_on_stream"] >= line["roofline"]["kernel_ms"] * 0.99


@pytest.mark.parametrize("exchange,nproc", [("allgather", 2), ("halo", 3), ("alltoall", 2)])
def test_bench_n_processes_on_one_gpu_through_the_rccl_branches(built, exchange, nproc):
    """The driver's launch line -- `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`, one PROCESS per rank -- on a one-GPU
    box: librccl refuses several ranks on one device, so HXV_RCCL_LIB points the engine's RCCL entry points at the process double of
    tests/rccl_double (shared-memory staging) and gloo carries the control plane (`--capi --backend gloo`).  Everything else is the N > 1
    path of bench.py: id broadcast, hxv_comm_init per process, hxv_slab_home, the checked product against the unsplit sector on every
    rank, warm-up, timed steps, max over ranks, the roofline leg through hxv_time_apply_slab."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HXV_RCCL_LIB=str(built.build_rccl_double_mp()))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    port = {"allgather": "29581", "halo": "29582", "alltoall": "29583"}[exchange]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1", "--master-port", port,
                        os.path.join(root, "bench.py"), "--gpus", str(nproc), "--backend", "gloo", "--capi", "--workload", "C2", "--steps", "3", "--warmup", "1",
                        "--exchange", exchange], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == nproc and line["value"] > 0 and line["scaling"] == "strong"
    assert line["checked"] is True and line["check_rel_err"] <= 1e-13 and "C-ABI" in line["config"]["transport"]
    assert line["config"]["exchange"] == exchange and line["config"]["exchange_ingest_bytes_per_gpu"] > 0
    assert line["roofline"]["kernel_ms"] > 0
    # first-contact fields: the link bound next to the measurement, the RCCL library the engine resolved, the
    # exchange's share of a slab product per rank
    cfg, rl = line["config"], line["roofline"]
    assert cfg["link_GBs_assumed"] == 153.0 and 0 < cfg["link_bound_ms"] < rl["slab_product_ms_on_stream"] * 1e3
    per_peer = {"allgather": cfg["exchange_ingest_bytes_per_gpu"] / (nproc - 1), "alltoall": cfg["exchange_ingest_bytes_per_gpu"] / (nproc - 1)}.get(exchange)
    if per_peer is not None:
        assert abs(cfg["link_bound_ms"] - per_peer / 153e9 * 1e3) < 1e-3
    assert cfg["rccl_lib"].endswith("librccl_double_mp.so") and isinstance(cfg["rccl_libs_mapped"], list)
    ex = rl["exchange_ms"]
    assert ex["min"] <= ex["this_rank"] <= ex["max"] and abs(ex["this_rank"] - (rl["slab_product_ms_on_stream"] - rl["kernel_ms"])) < 1e-3
    # first contact records EVERY exchange: the two that were not asked for are opened, checked against the main one's
    # product and timed in the same launch, each with its own link bound -- and one communicator per process served all three sectors
    oth = cfg["other_exchanges"]
    assert set(oth) == {"allgather", "halo", "alltoall"} - {exchange}
    for name, o in oth.items():
        assert o["ms_per_step"] > 0 and o["GBs"] > 0 and o["link_bound_ms"] > 0 and o["check_rel_err_vs_main"] <= 1e-13, (name, o)
        assert o["exchange_ms"]["min"] <= o["exchange_ms"]["this_rank"] <= o["exchange_ms"]["max"] and o["kernel_ms"] > 0
    assert oth.get("alltoall", cfg)["link_bound_ms"] <= oth.get("allgather", cfg)["link_bound_ms"]      # the two transposes move the least
    assert cfg["comm_cache"]["inits"] == 1 and cfg["comm_cache"]["reuses"] >= 2                          # (rank 0's process)
