// Device helpers shared by the kernels (diagonal on the fly, complex FMA, ELL decode).
#pragma once
#include "hxv_internal.hpp"

namespace hxv {

__device__ __forceinline__ double diag_cross(const CrossParams& cp, uint32_t mu, uint32_t md) {
  // Non-separable part of H_local.f90:35-50:  sum_o U_o n_up n_dw  + Ust sum_site sum_{a!=b} n_up,a n_dw,b
  uint32_t both = mu & md;
  double d = 0.0;
  if (cp.norb == 1) return cp.uloc[0] * (double)__popc(both & cp.orbmask[0]);
  for (int io = 0; io < cp.norb; ++io) d += cp.uloc[io] * (double)__popc(both & cp.orbmask[io]);
  if (cp.ust != 0.0) {
    int acc = 0;
    for (int il = 0; il < cp.nlat; ++il) {
      uint32_t sm = cp.sitemask[il];
      acc += __popc(mu & sm) * __popc(md & sm) - __popc(both & sm);
    }
    d += cp.ust * (double)acc;
  }
  return d;
}

__device__ __forceinline__ double diag_at(const DevDiag& dg, int iup, int idw, int64_t iloc) {
  if (dg.mode == 1) return dg.stored[iloc];
  return dg.a_up[iup] + dg.a_dw[idw] + diag_cross(dg.cross, dg.map_up[iup], dg.map_dw[idw]);
}

__device__ __forceinline__ void cfma(double2& acc, double2 c, double2 x) {
  acc.x = fma(c.x, x.x, acc.x);
  acc.x = fma(-c.y, x.y, acc.x);
  acc.y = fma(c.x, x.y, acc.y);
  acc.y = fma(c.y, x.x, acc.y);
}

__device__ __forceinline__ double2 ell_coef(const double2* __restrict__ coef, uint32_t e) {
  double2 c = coef[(e >> ELL_SRC_BITS) & ELL_COEF_MASK];
  if (e >> 31) {
    c.x = -c.x;
    c.y = -c.y;
  }
  return c;
}

}  // namespace hxv
