// HIP kernels (gfx950) for the sector product  Hv = D.v + H_up v + v H_dw^T  on the
// DimUp x DimDw layout (up index fastest), complex fp64.
//
// Reference semantics: ED_HAMILTONIAN_SPARSE_HxV.f90:167-227 (serial) / :230-315 (MPI slab).
// No MFMA: the path is a sparse gather over a 2.65 GB vector, bound by HBM and by on-chip
// gather bandwidth.
#include <algorithm>

#include "hxv_device.hpp"

namespace hxv {

// ---------------------------------------------------------------------------------------
// Variant 0: one thread per output element, every gather straight from global memory.
// The simplest correct statement of the product; kept as the on-device cross-check of the
// tiled kernels and for tiny sectors.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) hxv_naive_kernel(DevSector s, const double2* __restrict__ v, double2* __restrict__ hv) {
  const int64_t nloc = (int64_t)s.qdw * s.dimup;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nloc; t += (int64_t)gridDim.x * blockDim.x) {
    const int cl = (int)(t / s.dimup);
    const int i = (int)(t - (int64_t)cl * s.dimup);
    const int c = cl + s.dw0;
    const double2* __restrict__ vcol = v + (int64_t)(s.slab0 + cl) * s.dimup;
    const double d = diag_at(s.diag, i, c, t);
    const double2 x = vcol[i];
    double2 acc = make_double2(d * x.x, d * x.y);
    for (int k = 0; k < s.up.K; ++k) {
      const uint32_t e = s.up.ell[(int64_t)k * s.up.dim + i];
      if (e == ELL_EMPTY) break;
      cfma(acc, ell_coef(s.up.coef, e), vcol[e & ELL_SRC_MASK]);
    }
    for (int k = 0; k < s.dw.K; ++k) {
      const uint32_t e = s.dw.ell[(int64_t)k * s.dw.dim + c];
      if (e == ELL_EMPTY) break;
      cfma(acc, ell_coef(s.dw.coef, e), v[(int64_t)(e & ELL_SRC_MASK) * s.dimup + i]);
    }
    hv[t] = acc;
  }
}

hipError_t launch_hxv_naive(const DevSector& s, const double2* v_full, double2* hv_local, hipStream_t st) {
  const int64_t nloc = (int64_t)s.qdw * s.dimup;
  if (nloc == 0) return hipSuccess;
  int64_t blocks = (nloc + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(hxv_naive_kernel, dim3((unsigned)blocks), dim3(256), 0, st, s, v_full, hv_local);
  return hipGetLastError();
}

}  // namespace hxv
