"""Measurement harness: drives the engine in the CALL ORDER of the reference's two callers of spHtimesV_p.

It holds no solver logic of its own -- no state lists, no bath, no Green's-function assembly beyond the numbers a test needs to
compare: it opens, solves and closes sectors the way the callers do, and records what each step cost.

  diag_sweep  ED_DIAG.f90:78-260     for EVERY sector (nup, ndw) of the model: build_Hv_sector -> sp_eigh(Neigen, Nblock) when
                                     Dim > lanc_dim_threshold (:106, default 1024; below it the reference diagonalises a dense matrix on
                                     the host and never reaches the pointer) -> delete_Hv_sector
  gf_solve    ED_GF_NORMAL.f90:36-110, 123-306, 531-903
                                     for the ground state: for every channel of build_gf_normal -- the diagonal c^+_i / c_i (:123-306), the
                                     mixed (c^+_i + c^+_j), (c_i + c_j) (:573-728) and, unless ed_gf_symmetric, the complex
                                     (c^+_i + xi c^+_j), (c_i - xi c_j) (:737-903) -- build_Hv_sector(jsector) -> start vector ->
                                     sp_lanc_tridiag(nlanc = min(jdim, lanc_ngfiter)) -> delete_Hv_sector (:208-222)
"""
from __future__ import annotations

import time

import numpy as np

from hxv.engine import HxvSector  # (scripts/ is measurement scaffolding; the package is the engine only)


def _sync():
    import torch

    torch.cuda.synchronize()


def diag_sweep(model, neigen: int = 2, ncv_factor: int = 10, ncv_add: int = 0, niter: int = 512, dim_threshold: int = 1024, tol: float = 0.0,
               device: int = 0, sectors=None, small_too: bool = True, options: dict | None = None):
    """Open -> lowest eigenpairs -> close for every sector, in the reference's sector order (isector = 1 + ndw + nup*(Ns+1),
    ED_SETUP.f90:446-457).  Returns one record per sector.  Sectors at or below the threshold are solved too when small_too
    (the reference sends them to LAPACK: they exercise the engine's edge shapes, DimUp = 1 or DimDw = 1 included)."""
    Ns = model.Nlat * model.Norb * (model.Nbath + 1)
    out = []
    todo = sectors if sectors is not None else [(nup, ndw) for nup in range(Ns + 1) for ndw in range(Ns + 1)]
    for nup, ndw in todo:
        t0 = time.perf_counter()
        sec = HxvSector.from_model(model, nup, ndw, device=device)
        t1 = time.perf_counter()
        for k, v in (options or {}).items():
            sec.set_option(k, v)
        dim = sec.Dim
        ne = min(dim, neigen)                                              # ED_DIAG.f90:94
        ncv = min(dim, ncv_factor * max(ne, neigen) + ncv_add)             # :96
        lanczos = dim > dim_threshold and ne != dim                        # :104-106
        rec = {"nup": nup, "ndw": ndw, "dim": dim, "lanczos": bool(lanczos), "open_ms": (t1 - t0) * 1e3,
               "open_cache_hit": bool(sec.get_option("open_cache_hit")), "open_us": {k: sec.get_option("open_us_" + k) for k in ("host", "plan", "upload", "total")}}
        if lanczos or small_too:
            ev, _, nconv, nmv = sec.eigh_lowest(ne, ncv, min(dim, niter), tol, want_vectors=False)
            _sync()
            rec.update(evals=[float(x) for x in ev], nconv=int(nconv), nmatvec=int(nmv), solve_ms=(time.perf_counter() - t1) * 1e3,
                       real_vectors=bool(sec.get_option("lanczos_real_last")))
        t2 = time.perf_counter()
        sec.close()
        rec["close_ms"] = (time.perf_counter() - t2) * 1e3
        out.append(rec)
    return out


def gf_channels(model, symmetric: bool = False):
    """The tridiagonalisations of build_gf_normal for ONE state of the list and spin index 1, in the reference's order
    (ED_GF_NORMAL.f90:62-86): (create, [(orbital, coefficient), ...]) with orbital = imp_state_index - 1."""
    nimp = model.Nlat * model.Norb
    chans = []
    for i in range(nimp):
        chans.append({"kind": "diag", "create": True, "terms": [(i, 1.0)]})            # :123-223
        chans.append({"kind": "diag", "create": False, "terms": [(i, 1.0)]})           # :226-306
        for j in range(nimp):
            if j == i:
                continue
            chans.append({"kind": "mix", "create": True, "terms": [(i, 1.0), (j, 1.0)]})       # :357-430 / :573-646
            chans.append({"kind": "mix", "create": False, "terms": [(i, 1.0), (j, 1.0)]})      # :439-512 / :655-728
            if not symmetric:
                chans.append({"kind": "mix_xi", "create": True, "terms": [(i, 1.0), (j, 1j)]})     # :737-809
                chans.append({"kind": "mix_xi", "create": False, "terms": [(i, 1.0), (j, -1j)]})   # :818-890
    return chans


def gf_solve(model, nup: int, ndw: int, nlanc: int = 200, symmetric: bool = False, pair: bool = True, spin: int = 0, device: int = 0,
             gs_method: str = "eigh", keep_tridiag: bool = False, keep_psi: bool = False, channels=None):
    """Ground state of sector (nup, ndw), then every channel of build_gf_normal device-resident: the target sector N+-1 is opened and
    closed around EVERY tridiagonalisation, as the reference does (:208-222) -- re-opens are what the engine's sector cache serves.
    pair: two consecutive REAL channels of one target sector share a complex product (hxv_lanczos_tridiag_pair; legal when H is real).
    Returns (records, summary)."""
    import torch

    t_all = time.perf_counter()
    t0 = time.perf_counter()
    gs = HxvSector.from_model(model, nup, ndw, device=device)
    gs_open_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    if gs_method == "eigh":
        ev, X, _, gs_nmv = gs.eigh_lowest(min(2, gs.Dim), min(gs.Dim, 20), native=True)       # ED_DIAG's default call
        e0, psi = float(ev[0]), X[0].contiguous()
        del X
    else:
        e0, psi, gs_nmv = gs.lanczos_eigh(512, 1e-13, native=True)
    _sync()
    gs_ms = (time.perf_counter() - t0) * 1e3
    # ED_DIAG closes the sector after the spectrum stage (delete_Hv_sector, ED_DIAG.f90:186); the Green's-function stage only needs its
    # basis again (build_sector, ED_GF_NORMAL.f90:165).  Here: close and re-open -- the re-open shares the cached image and holds no
    # Lanczos vectors or scratch (at Ns=18 those are 95 GB the channels need)
    real_ok = gs.real_vectors_available
    gs.close()
    from hxv.engine import pool_trim
    pool_trim(device)
    torch.cuda.empty_cache()
    gs = HxvSector.from_model(model, nup, ndw, device=device)
    chans = channels if channels is not None else gf_channels(model, symmetric)
    recs, open_ms, open_hits = [], [], []

    def target(create):
        return (nup + 1, ndw) if (create and spin == 0) else (nup - 1, ndw) if spin == 0 else (nup, ndw + 1) if create else (nup, ndw - 1)

    def start_vector(sec, ch):
        vv, n2 = None, 0.0
        for orb, cf in ch["terms"]:
            vv, n2 = gs.apply_ladder(sec, orb, spin, ch["create"], psi, coef=cf, out=vv) if vv is not None else gs.apply_ladder(sec, orb, spin, ch["create"], psi, coef=cf)
        return vv, n2

    def run(batch):
        """one open -> tridiagonalisation(s) -> close; batch = one channel, or two real ones of the same target sector"""
        tu, td = target(batch[0]["create"])
        Ns = model.Nlat * model.Norb * (model.Nbath + 1)
        if not (0 <= tu <= Ns and 0 <= td <= Ns):
            return                                                       # getCDGsector / getCsector == 0: no such sector (:170, :237)
        t0 = time.perf_counter()
        sec = HxvSector.from_model(model, tu, td, device=device)
        t1 = time.perf_counter()
        open_ms.append((t1 - t0) * 1e3)
        hit = bool(sec.get_option("open_cache_hit"))
        open_hits.append(hit)
        nl = min(sec.Dim, nlanc)                                         # :204-207
        vs = [start_vector(sec, ch) for ch in batch]
        _sync()
        t2 = time.perf_counter()
        # (the drivers normalise a start vector that is not, like SciFortran's sp_lanc_tridiag; the reference divides by sqrt(norm2)
        #  on the host first, :197-199 -- a Dim-sized pass the device run does not need)
        live = [k for k, (_, n2) in enumerate(vs) if n2 > 0.0]
        if len(live) == 2:
            res = sec.lanczos_tridiag_pair(vs[0][0], vs[1][0], nl)
        else:
            res = [(np.zeros(0), np.zeros(0), 0)] * len(batch)
            for k in live:
                res[k] = sec.lanczos_tridiag(vs[k][0], nl)
        _sync()
        t3 = time.perf_counter()
        real_last = bool(sec.get_option("lanczos_real_last"))
        sec.close()
        t4 = time.perf_counter()
        for ch, (vv, n2), (a, b, n) in zip(batch, vs, res):
            r = {"kind": ch["kind"], "create": ch["create"], "terms": ch["terms"], "sector": (tu, td), "dim": sec.Dim, "norm2": float(n2), "nsteps": int(n),
                 "paired": len(batch) == 2, "real_vectors": real_last or len(batch) == 2, "open_ms": (t1 - t0) * 1e3, "open_cache_hit": hit,
                 "start_ms": (t2 - t1) * 1e3, "tridiag_ms": (t3 - t2) * 1e3 / len(batch), "close_ms": (t4 - t3) * 1e3}
            if keep_tridiag:
                r["alanc"], r["blanc"] = a[:n].copy(), b[:n].copy()
            recs.append(r)

    # pairing keeps the reference's order of channels and holds at most one real channel back per target sector (the list alternates
    # c^+ and c channels: the partner of a c^+ channel is the next real c^+ channel)
    waiting = {}
    for ch in chans:
        is_real = real_ok and all(complex(c).imag == 0.0 for _, c in ch["terms"])
        if pair and is_real:
            tgt = target(ch["create"])
            if tgt in waiting:
                run([waiting.pop(tgt), ch])
            else:
                waiting[tgt] = ch
        else:
            run([ch])
    for ch in waiting.values():
        run([ch])
    psi_host = gs.unpad(psi).cpu().numpy() if keep_psi else None   # (before the handle is closed: unpad needs its layout)
    gs.close()
    _sync()
    total_s = time.perf_counter() - t_all
    summary = {"gs_sector": (nup, ndw), "e0": e0, "gs_open_ms": gs_open_ms, "gs_ms": gs_ms, "gs_nmatvec": int(gs_nmv), "nlanc": nlanc,
               "channels": len(recs), "channels_real": sum(1 for r in recs if r["kind"] != "mix_xi"), "channels_complex": sum(1 for r in recs if r["kind"] == "mix_xi"),
               "channels_paired": sum(1 for r in recs if r["paired"]), "sector_open_ms_mean": float(np.mean(open_ms)) if open_ms else 0.0,
               "sector_open_ms_max": float(np.max(open_ms)) if open_ms else 0.0, "sector_open_ms_first": float(open_ms[0]) if open_ms else 0.0,
               "sector_opens": len(open_ms), "sector_open_cache_hits": int(sum(open_hits)),
               "real_channels_s": sum(r["tridiag_ms"] for r in recs if r["kind"] != "mix_xi") * 1e-3,
               "complex_channels_s": sum(r["tridiag_ms"] for r in recs if r["kind"] == "mix_xi") * 1e-3, "gf_solve_s": total_s}
    if keep_psi:
        summary["psi"] = psi_host                           # (tests: the ground state in the reference's contiguous host layout)
    del psi
    torch.cuda.empty_cache()
    return recs, summary
