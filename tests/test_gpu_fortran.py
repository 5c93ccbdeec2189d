"""The Fortran host (flang, ISO_C_BINDING glue) drives the engine through the reference's procedure-pointer
surface: spHtimesV_p => gpuMatVec_main, Lanczos on host arrays."""
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_fortran_host_through_procedure_pointer(built):
    from hxv import models
    from oracle.oracle import OracleSector

    exe = built.build_fortran()
    if exe is None:
        pytest.skip("flang not available")
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    txt = out.stdout
    e0_c1 = float(re.search(r"C1 plaquette.*E0=\s*([-\d.Ee+]+)", txt).group(1))
    assert abs(e0_c1 - (-2.10274848)) < 5e-9                      # reference dense H
    ref = np.linalg.eigvalsh(OracleSector(models.plaquette_2x2_nobath(U=4.0, t=1.0, hfmode=False), 2, 2).dense())[0]
    assert abs(e0_c1 - ref) < 1e-9
    m = models.hm_1dchain(eps_bath=[0.3, 0.6])
    orc = OracleSector(m, 6, 6)
    v = models.deterministic_vector(orc.Dim)
    hv = orc.spMatVec_main(v)
    nums = re.search(r"C2 chain.*Hv\(1\),Hv\(Dim\)=\s*(.*)", txt).group(1).split()
    got = [float(x) for x in nums]
    want = [hv[0].real, hv[0].imag, hv[-1].real, hv[-1].imag]
    assert np.allclose(got, want, rtol=1e-12, atol=1e-13)
    e0_c2 = float(re.search(r"C2 chain sector\(6,6\) Dim=\s*\d+ E0=\s*([-\d.Ee+]+)", txt).group(1))
    a, b = orc.lanc_tridiag(v / np.linalg.norm(v), 200)
    T = np.diag(a) + np.diag(b[1:], 1) + np.diag(b[1:], -1)
    assert abs(e0_c2 - np.linalg.eigvalsh(T)[0]) < 1e-9
    # device Lanczos behind the SciFortran call signatures (gpu_sp_lanc_tridiag / gpu_sp_lanc_eigh)
    e0_tri = float(re.search(r"C2 device tridiag E0=\s*([-\d.Ee+]+)", txt).group(1))
    m2 = re.search(r"C2 device eigh E0=\s*([-\d.Ee+]+)\s*\|vec\|\^2-1=\s*([-\d.Ee+]+)", txt)
    e0_eig, dn = float(m2.group(1)), float(m2.group(2))
    assert abs(e0_tri - e0_c2) < 1e-9 and abs(e0_eig - e0_c2) < 1e-9 and abs(dn) < 1e-10


def test_fortran_gpu_sp_eigh_wrapper(built):
    """gpu_sp_eigh(MatVec,eval,evec,Nblock,Nitermax,tol) -- SciFortran's sp_eigh signature, ED_DIAG.f90:152-160 -- called from
    the Fortran demo host on C2: two lowest eigenvalues vs scipy ARPACK on the oracle's matrices, residual of the 2nd pair."""
    import scipy.sparse.linalg as sla
    from hxv import models
    from oracle.oracle import OracleSector
    from helpers_matrix import oracle_full_matrix

    exe = built.build_fortran()
    if exe is None:
        pytest.skip("flang not available")
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    m2 = re.search(r"C2 device sp_eigh E=\s*([-\d.Ee+]+)\s+([-\d.Ee+]+)\s+resid2=\s*([-\d.Ee+]+)", out.stdout)
    assert m2, out.stdout
    e = np.array([float(m2.group(1)), float(m2.group(2))])
    H = oracle_full_matrix(OracleSector(models.hm_1dchain(eps_bath=[0.3, 0.6]), 6, 6))
    ref = np.sort(sla.eigsh(H, k=2, which="SA", ncv=20, tol=1e-13)[0])
    assert np.abs(e - ref).max() < 1e-9 and float(m2.group(3)) < 1e-8


def test_fortran_mpi_branch_call_text(built):
    """The reference's MpiStatus=T call lines (ED_DIAG.f90:152-156,176-177; ED_GF_NORMAL.f90:215: communicator first) compile
    against the glue's generic interfaces and run through the engine's own communicator (one rank: RCCL all-gather /
    all-reduce execute); the numbers equal the serial branch."""
    exe = built.build_fortran()
    if exe is None:
        pytest.skip("flang not available")
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    txt = out.stdout
    ser = re.search(r"C2 device sp_eigh E=\s*([-\d.Ee+]+)\s+([-\d.Ee+]+)", txt)
    mpi = re.search(r"C2 MPI-branch sp_eigh E=\s*([-\d.Ee+]+)\s+([-\d.Ee+]+)", txt)
    assert ser and mpi, txt
    assert abs(float(ser.group(1)) - float(mpi.group(1))) < 1e-9 and abs(float(ser.group(2)) - float(mpi.group(2))) < 1e-9
    e_l = float(re.search(r"C2 MPI-branch sp_lanc_eigh E0=\s*([-\d.Ee+]+)", txt).group(1))
    e_t = float(re.search(r"C2 MPI-branch sp_lanc_tridiag E0=\s*([-\d.Ee+]+)", txt).group(1))
    assert abs(e_l - float(ser.group(1))) < 1e-9 and abs(e_t - float(ser.group(1))) < 1e-8


def test_fortran_paired_tridiagonalisation(built):
    """gpu_sp_lanc_tridiag_pair: two channels of ED_GF_NORMAL.f90:123-306 on one product, called from the Fortran demo host;
    both lowest Ritz values equal the ground state, and channel a equals its own single run."""
    exe = built.build_fortran()
    if exe is None:
        pytest.skip("flang not available")
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    txt = out.stdout
    mp = re.search(r"C2 device tridiag pair E0=\s*([-\d.Ee+]+)\s+([-\d.Ee+]+)", txt)
    ms = re.search(r"C2 device tridiag channel a alone E0=\s*([-\d.Ee+]+)", txt)
    e0 = float(re.search(r"C2 device tridiag E0=\s*([-\d.Ee+]+)", txt).group(1))
    assert mp and ms, txt
    assert abs(float(mp.group(1)) - float(ms.group(1))) < 1e-10
    assert abs(float(mp.group(1)) - e0) < 1e-8 and abs(float(mp.group(2)) - e0) < 1e-8


def test_fortran_device_resident_green_function_channel(built):
    """gpu_sp_lanc_eigh_dev -> gpu_keep_sector -> gpu_apply_ladder -> gpu_sp_lanc_tridiag_dev (the three-line change of
    ED_GF_NORMAL.f90:174-217 in INTEGRATION.md): ground state, c^dagger|gs> and a 100-step tridiagonalisation with NO Dim-sized PCIe
    transfer (the engine's own h2d / d2h byte counters, hxv_get_stats), alanc / blanc equal to the host-array path -- the reference's
    serial c^dagger loop restated in the demo host + gpu_sp_lanc_tridiag on the host array, which does move vectors."""
    exe = built.build_fortran()
    if exe is None:
        pytest.skip("flang not available")
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    txt = out.stdout
    m = re.search(r"GF device channel: E0=\s*([-\d.Ee+]+)\s*norm2=\s*([-\d.Ee+]+).*channel-sector=\s*(\d+)\s+(\d+)\s+(\d+)\s+(\d+)", txt)
    assert m, txt
    e0, norm2 = float(m.group(1)), float(m.group(2))
    assert [int(m.group(k)) for k in (3, 4, 5, 6)] == [0, 0, 0, 0]            # nothing vector-sized crossed PCIe
    se = re.search(r"GF device sp_eigh: E=\s*([-\d.Ee+]+)\s+([-\d.Ee+]+)\s*\|E0 - lanc_eigh E0\|=\s*([-\d.Ee+]+)", txt)
    assert se and float(se.group(3)) < 1e-9 and float(se.group(2)) > float(se.group(1))      # gpu_sp_eigh_dev: eigenvectors stay on the device
    e0_ref = float(re.search(r"C2 device eigh E0=\s*([-\d.Ee+]+)", txt).group(1))
    assert abs(e0 - e0_ref) < 1e-9 and 0.0 < norm2 < 1.0
    h = re.search(r"GF host-array channel: \|norm2 diff\|=\s*([-\d.Ee+]+)\s*max\|da\|\(8\)=\s*([-\d.Ee+]+)\s*max\|db\|\(8\)=\s*([-\d.Ee+]+)\s*"
                  r"lowest Ritz values=\s*([-\d.Ee+]+)\s+([-\d.Ee+]+)", txt)
    assert h, txt
    assert float(h.group(1)) < 1e-12 and float(h.group(2)) < 1e-10 and float(h.group(3)) < 1e-10
    assert abs(float(h.group(4)) - float(h.group(5))) < 1e-9
    pr = re.search(r"GF device pair: max\|alanc_a-single\|\(8\)=\s*([-\d.Ee+]+)\s*max\|blanc_a-single\|\(8\)=\s*([-\d.Ee+]+)\s*lowest Ritz values a,b=\s*([-\d.Ee+]+)\s+([-\d.Ee+]+)"
                   r"\s*PCIe bytes=\s*(\d+)\s+(\d+)", txt)
    assert pr, txt                                                             # gpu_sp_lanc_tridiag_pair_dev: two device channels on one product
    assert float(pr.group(1)) < 1e-10 and float(pr.group(2)) < 1e-10 and abs(float(pr.group(3)) - float(h.group(4))) < 1e-9
    assert int(pr.group(5)) == 0 and int(pr.group(6)) == 0 and float(pr.group(4)) < -5.0
    hb = re.search(r"GF host-array channel: PCIe bytes \(h2d,d2h\) channel-sector=\s*(\d+)\s+(\d+)", txt)
    assert int(hb.group(1)) == 792 * 924 * 16                                  # the host start vector of sector (7,6), once
    # vectors freed in the awkward order: the sector kept through a view, the allocation's first vector freed first, another
    # vector of the kept sector outliving the keeper -- lifetimes are counted, nothing dangles
    lt = re.search(r"GF lifetimes any order: \|norm2\(ev1\) - norm2\(gs\)\|=\s*([-\d.Ee+]+)\s*norm2\(ev2\)=\s*([-\d.Ee+]+)\s*max\|roundtrip - psi\|=\s*([-\d.Ee+]+)", txt)
    assert lt, txt
    assert float(lt.group(1)) < 1e-9 and 0.0 < float(lt.group(2)) < 1.0 and float(lt.group(3)) == 0.0
    # a sector closed UNDER a live vector, the next one opened at the same address: entries are matched by serial number, so the new
    # sector does not inherit the dead one's entry -- kept through a vector and freed, it is closed (live sectors back to where they were)
    dz = re.search(r"GF lifetimes after a sector closed under a live vector: live sectors before / kept\+open / after=\s*(\d+)\s+(\d+)\s+(\d+)\s*"
                   r"\|E0\(new\) - E0\(old\)\|=\s*([-\d.Ee+]+)\s*\|norm2 - norm2\(gs\)\|=\s*([-\d.Ee+]+)", txt)
    assert dz, txt
    before, mid, after = (int(dz.group(k)) for k in (1, 2, 3))
    assert mid == before + 2 and after == before, (before, mid, after)
    assert float(dz.group(4)) < 1e-9 and float(dz.group(5)) < 1e-9


def test_fortran_stored_matrices_binding(built):
    """gpu_build_Hv_sector_from_csr (the reference's spH0ups(1) / spH0dws(1) / spH0d handed over flattened) from the Fortran host: the
    matrices come out of a model-built sector through gpu_get_sector_csr / _diag, go back in through the binding, and the two products
    agree; nnz(H_up) = 8 568 is the survey's count for this model."""
    exe = built.build_fortran()
    if exe is None:
        pytest.skip("flang not available")
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"stored matrices: nnz\(H_up\),nnz\(H_dw\)=\s*(\d+)\s+(\d+)\s*max\|Hv\(csr\)-Hv\(model\)\|=\s*([-\d.Ee+]+)\s*max\|Hv\|=\s*([-\d.Ee+]+)", out.stdout)
    assert m, out.stdout
    assert int(m.group(1)) == 8568 and int(m.group(2)) == 8568
    assert float(m.group(3)) <= 1e-13 * float(m.group(4))
