// HIP kernels (gfx950) for the sector product  Hv = D.v + H_up v + v H_dw^T  on the
// DimUp x DimDw layout (up index fastest), complex fp64.
//
// Reference semantics: ED_HAMILTONIAN_SPARSE_HxV.f90:167-227 (serial) / :230-315 (MPI slab).
// No MFMA: the path is a sparse gather over a 2.65 GB vector, bound by HBM and by on-chip
// gather bandwidth.
#include <algorithm>

#include "hxv_device.hpp"
#include "hxv_tile_dev.hpp"

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
    const double2* __restrict__ vcol = v + (int64_t)(s.slab0 + cl) * s.pitch;
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
      cfma(acc, ell_coef(s.dw.coef, e), v[(int64_t)(e & ELL_SRC_MASK) * s.pitch + i]);
    }
    hv[(int64_t)cl * s.pitch + i] = acc;
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

// ---------------------------------------------------------------------------------------
// Non-local block spH0nd (spin exchange Jx, pair hopping Jp; Norb>1 only), on the fly:
//   hv(i) += sum_j H_nd(i,j) v(j),   j = state reached from i by the two-spin operator
// exactly the (symmetric, "transposed") rows the reference stores at sparse/H_non_local.f90:4-100 and
// multiplies at ED_HAMILTONIAN_SPARSE_HxV.f90:217-225 / :300-313 (there on the all-gathered vector).
// Signs are the product of the four single-spin signs (c/cdg, ED_SETUP.f90:807-833), no cross-spin sign.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int rank_in_map(const uint32_t* __restrict__ map, int dim, uint32_t value) {
  int lo = 0, hi = dim - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (map[mid] < value)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

__device__ __forceinline__ int par_below(uint32_t m, int pos) { return __popc(m & ((1u << pos) - 1u)) & 1; }

__global__ void __launch_bounds__(256) hxv_nonlocal_kernel(DevSector s, const double2* __restrict__ v, double2* __restrict__ hv) {
  const int64_t nloc = (int64_t)s.qdw * s.dimup;
  const uint32_t* __restrict__ map_up = s.diag.map_up;
  const uint32_t* __restrict__ map_dw = s.diag.map_dw;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nloc; t += (int64_t)gridDim.x * blockDim.x) {
    const int cl = (int)(t / s.dimup);
    const int i = (int)(t - (int64_t)cl * s.dimup);
    const uint32_t mu = map_up[i], md = map_dw[cl + s.dw0];
    double2 acc = make_double2(0.0, 0.0);
    bool any = false;
    for (int il = 0; il < s.nd.nlat; ++il)
      for (int io = 0; io < s.nd.norb; ++io)
        for (int jo = 0; jo < s.nd.norb; ++jo) {
          if (io == jo) continue;
          const int is = io + il * s.nd.norb, js = jo + il * s.nd.norb;  // imp_state_index, 0-based bit
          const bool nu_i = (mu >> is) & 1u, nu_j = (mu >> js) & 1u, nd_i = (md >> is) & 1u, nd_j = (md >> js) & 1u;
          // spin exchange, H_non_local.f90:26-60:  [c^+_js c_is]_dw [c^+_is c_js]_up
          if (s.nd.jx != 0.0 && nu_j && nd_i && !nd_j && !nu_i) {
            const uint32_t k1 = md & ~(1u << is), k2 = k1 | (1u << js);
            const uint32_t k3 = mu & ~(1u << js), k4 = k3 | (1u << is);
            const int sg = par_below(md, is) ^ par_below(k1, js) ^ par_below(mu, js) ^ par_below(k3, is);
            const int jdw = rank_in_map(map_dw, s.dimdw, k2), jup = rank_in_map(map_up, s.dimup, k4);
            const double2 x = v[(int64_t)s.vcol[jdw] * s.pitch + jup];
            const double c = sg ? -s.nd.jx : s.nd.jx;
            acc.x += c * x.x;
            acc.y += c * x.y;
            any = true;
          }
          // pair hopping, H_non_local.f90:65-98:  [c^+_is c_js]_dw [c^+_is c_js]_up
          if (s.nd.jp != 0.0 && nu_j && nd_j && !nd_i && !nu_i) {
            const uint32_t k1 = md & ~(1u << js), k2 = k1 | (1u << is);
            const uint32_t k3 = mu & ~(1u << js), k4 = k3 | (1u << is);
            const int sg = par_below(md, js) ^ par_below(k1, is) ^ par_below(mu, js) ^ par_below(k3, is);
            const int jdw = rank_in_map(map_dw, s.dimdw, k2), jup = rank_in_map(map_up, s.dimup, k4);
            const double2 x = v[(int64_t)s.vcol[jdw] * s.pitch + jup];
            const double c = sg ? -s.nd.jp : s.nd.jp;
            acc.x += c * x.x;
            acc.y += c * x.y;
            any = true;
          }
        }
    if (any) {
      const int64_t o = (int64_t)cl * s.pitch + i;
      double2 h = hv[o];
      h.x += acc.x;
      h.y += acc.y;
      hv[o] = h;
    }
  }
}

// Same block from the one-spin move tables (SectorHost::nd_up / nd_dw): spH0nd = sum over sites and orbital pairs of
//   Jx [c^+_j c_i]_dw [c^+_i c_j]_up + Jp [c^+_i c_j]_dw [c^+_i c_j]_up,   Kronecker products of two one-body moves,
// so an element's partner is (up-move of the row) x (dw-move of the column).  One workgroup per (column, 1024 rows): the
// dw moves of the column are wave-uniform (no work at all for the terms the column's dw state rules out), the up moves
// come from a coalesced table row, the partner values from ONE other column with rows in increasing order.  No search.
__global__ void __launch_bounds__(1024) hxv_nonlocal_tab(DevSector s, const double2* __restrict__ v, double2* __restrict__ hv, int rchunks) {
  const int cl = blockIdx.x / rchunks;                      // local column
  const int i = (blockIdx.x - cl * rchunks) * 1024 + threadIdx.x;
  const bool row_ok = i < s.dimup;
  const int ir = min(i, s.dimup - 1);
  const int c = cl + s.dw0;
  const int O = s.nd.norb;
  double2 acc = make_double2(0.0, 0.0);
  bool any = false;
  for (int il = 0; il < s.nd.nlat; ++il)
    for (int io = 0; io < O; ++io)
      for (int jo = 0; jo < O; ++jo) {
        if (io == jo) continue;
        const int q = (il * O + io) * O + jo, r = (il * O + jo) * O + io;
        // dw: spin exchange moves i -> j (entry q), pair hopping moves j -> i (entry r); wave-uniform
        const uint32_t dse = s.nd.jx != 0.0 ? __builtin_amdgcn_readfirstlane(s.nd_dw[(int64_t)q * s.dimdw + c]) : ND_INVALID;
        const uint32_t dph = s.nd.jp != 0.0 ? __builtin_amdgcn_readfirstlane(s.nd_dw[(int64_t)r * s.dimdw + c]) : ND_INVALID;
        if (dse == ND_INVALID && dph == ND_INVALID) continue;
        const uint32_t u = s.nd_up[(int64_t)r * s.dimup + ir];   // up: both terms move j -> i
        if (u == ND_INVALID) continue;
        const int jup = (int)(u & 0x7FFFFFFFu);
        if (dse != ND_INVALID) {
          const double2 x = v[(int64_t)(dse & 0x7FFFFFFFu) * s.pitch + jup];
          const double cf = ((u ^ dse) >> 31) ? -s.nd.jx : s.nd.jx;
          acc.x += cf * x.x;
          acc.y += cf * x.y;
          any = true;
        }
        if (dph != ND_INVALID) {
          const double2 x = v[(int64_t)(dph & 0x7FFFFFFFu) * s.pitch + jup];
          const double cf = ((u ^ dph) >> 31) ? -s.nd.jp : s.nd.jp;
          acc.x += cf * x.x;
          acc.y += cf * x.y;
          any = true;
        }
      }
  if (any && row_ok) {
    const int64_t o = (int64_t)cl * s.pitch + i;
    double2 h = hv[o];
    h.x += acc.x;
    h.y += acc.y;
    hv[o] = h;
  }
}

__global__ void __launch_bounds__(256) pack_columns_kernel(const double2* __restrict__ in, double2* __restrict__ out, const int32_t* __restrict__ cols,
                                                          int pitch) {
  const double2* __restrict__ src = in + (int64_t)cols[blockIdx.x] * pitch;
  double2* __restrict__ dst = out + (int64_t)blockIdx.x * pitch;
  for (int i = threadIdx.x; i < pitch; i += 256) dst[i] = src[i];
}

hipError_t launch_pack_columns(const double2* d_in, double2* d_out, const int32_t* d_cols, int ncols, int pitch, hipStream_t st) {
  if (ncols <= 0) return hipSuccess;
  hipLaunchKernelGGL(pack_columns_kernel, dim3((unsigned)ncols), dim3(256), 0, st, d_in, d_out, d_cols, pitch);
  return hipGetLastError();
}

// hv += the dw part, read where the second transpose of exchange mode 2 left it (one block per rank of origin, WtRange): the last step of
// the OVERLAPPED form of that exchange, where diagonal + up hops ran on a second stream while the transposes were under way.
template <typename VT>
__global__ void __launch_bounds__(256) add_pieces_kernel(VT* __restrict__ hv, const WtRange* __restrict__ wtr, int nwtr, int dimup, int pitch) {
  const int c = blockIdx.y;
  for (int row = blockIdx.x * 256 + threadIdx.x; row < dimup; row += gridDim.x * 256) {
    int k = 0;
    while (k + 1 < nwtr && row >= wtr[k].row1) ++k;
    const VT* __restrict__ src = reinterpret_cast<const VT*>(wtr[k].base) + (int64_t)c * wtr[k].stride + (row - wtr[k].row0);
    VT a = hv[(int64_t)c * pitch + row];
    vadd(a, *src);
    hv[(int64_t)c * pitch + row] = a;
  }
}

hipError_t launch_add_pieces(void* hv, const WtRange* wtr, int nwtr, int dimup, int pitch, int ncols, bool real, hipStream_t st) {
  if (ncols <= 0) return hipSuccess;
  const dim3 grid((unsigned)std::min((dimup + 255) / 256, 64), (unsigned)ncols);
  if (real)
    hipLaunchKernelGGL(add_pieces_kernel<double>, grid, dim3(256), 0, st, (double*)hv, wtr, nwtr, dimup, pitch);
  else
    hipLaunchKernelGGL(add_pieces_kernel<double2>, grid, dim3(256), 0, st, (double2*)hv, wtr, nwtr, dimup, pitch);
  return hipGetLastError();
}

// spH0nd handed over as stored rows (hxv_set_nonlocal_csr): hv(i) += sum_k vals(k) v(cols(k)), the loop of
// ED_HAMILTONIAN_SPARSE_HxV.f90:217-225 with one thread per local row; the global column index is split into (iup, idw) to find the
// element in the gathered layout.  A few entries per row: no tiling.
__global__ void __launch_bounds__(256) hxv_nonlocal_csr(DevSector s, const double2* __restrict__ v, double2* __restrict__ hv) {
  const int64_t nloc = (int64_t)s.qdw * s.dimup;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nloc; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k0 = s.ndcsr_rowptr[t], k1 = s.ndcsr_rowptr[t + 1];
    if (k0 == k1) continue;
    double2 acc = make_double2(0.0, 0.0);
    for (int64_t k = k0; k < k1; ++k) {
      const int64_t j = (int64_t)s.ndcsr_cols[k] - 1;  // Fortran column
      const int jdw = (int)(j / s.dimup), jup = (int)(j - (int64_t)jdw * s.dimup);
      cfma(acc, s.ndcsr_vals[k], v[(int64_t)s.vcol[jdw] * s.pitch + jup]);
    }
    const int cl = (int)(t / s.dimup);
    const int64_t o = (int64_t)cl * s.pitch + (t - (int64_t)cl * s.dimup);
    double2 h = hv[o];
    h.x += acc.x;
    h.y += acc.y;
    hv[o] = h;
  }
}

hipError_t launch_hxv_nonlocal(const DevSector& s, const double2* v_full, double2* hv_local, hipStream_t st) {
  const int64_t nloc = (int64_t)s.qdw * s.dimup;
  if (nloc == 0 || !s.nd.active) return hipSuccess;
  if (s.ndcsr_rowptr) {
    int64_t blocks = std::min<int64_t>((nloc + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(hxv_nonlocal_csr, dim3((unsigned)blocks), dim3(256), 0, st, s, v_full, hv_local);
    return hipGetLastError();
  }
  if (s.nd_up && s.nd_dw) {
    const int rchunks = (s.dimup + 1023) / 1024;
    hipLaunchKernelGGL(hxv_nonlocal_tab, dim3((unsigned)((int64_t)s.qdw * rchunks)), dim3(1024), 0, st, s, v_full, hv_local, rchunks);
    return hipGetLastError();
  }
  int64_t blocks = (nloc + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(hxv_nonlocal_kernel, dim3((unsigned)blocks), dim3(256), 0, st, s, v_full, hv_local);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// c / c^dagger on one orbital of one spin, between two sectors (ED_GF_NORMAL.f90:180-199:
// vvinit(j) = sgn * state_cvec(i)).  One thread per TARGET element: it looks its source up.
// ---------------------------------------------------------------------------------------
// Device row order (SectorHost::up_perm; spin up only -- a dw operator leaves the row where it is): map_from is then the SORTED reference
// map of the source sector and the source row found in it goes through perm_from to its device row; the two basis signs (by device row)
// multiply the operator's own.  Null pointers = the reference's order.
__global__ void __launch_bounds__(256) ladder_kernel(const uint32_t* __restrict__ map_from, int dim_from, const uint32_t* __restrict__ map_to,
                                                    int pitch_from, int dimup_to, int pitch_to, int dimdw_to, int orbital, int spin, int create,
                                                    const double2* __restrict__ psi, double2* __restrict__ out, double2 coef,
                                                    int accumulate, const int32_t* __restrict__ perm_from, const uint8_t* __restrict__ sign_from,
                                                    const uint8_t* __restrict__ sign_to) {
  // out = (accumulate ? out : 0) + coef * c^(dagger) psi : the mixed channels of the Green's function start from
  // (c^dagger_i + c^dagger_j)|gs> and (c^dagger_i + xi c^dagger_j)|gs> (ED_GF_NORMAL.f90:370-406, 746-780)
  const int64_t n = (int64_t)dimup_to * dimdw_to;
  const uint32_t bit = 1u << orbital;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(t / dimup_to), i = (int)(t - (int64_t)c * dimup_to);
    const uint32_t m_to = map_to[spin == 0 ? i : c];
    double2 r = make_double2(0.0, 0.0);
    // the target must have the orbital occupied after a creation / empty after a destruction
    if (((m_to & bit) != 0u) == (create != 0)) {
      const uint32_t m_from = m_to ^ bit;
      int j = rank_in_map(map_from, dim_from, m_from);
      int neg = par_below(m_from, orbital);
      if (spin == 0) {
        if (perm_from) j = perm_from[j];
        if (sign_from) neg ^= (int)sign_from[j];
        if (sign_to) neg ^= (int)sign_to[i];
      }
      const double sg = neg ? -1.0 : 1.0;
      const double2 x = spin == 0 ? psi[(int64_t)c * pitch_from + j] : psi[(int64_t)j * pitch_from + i];
      r = make_double2(sg * (coef.x * x.x - coef.y * x.y), sg * (coef.x * x.y + coef.y * x.x));
    }
    if (accumulate) {
      const double2 o = out[(int64_t)c * pitch_to + i];
      r.x += o.x;
      r.y += o.y;
    }
    out[(int64_t)c * pitch_to + i] = r;
  }
}

hipError_t launch_ladder(const uint32_t* map_from, int dim_from, const uint32_t* map_to, int dim_to, int pitch_from, int dimup_to,
                         int pitch_to, int dimdw_to, int orbital, int spin, int create, const double2* psi, double2* out, hipStream_t st,
                         double2 coef, int accumulate, const int32_t* perm_from, const uint8_t* sign_from, const uint8_t* sign_to) {
  (void)dim_to;
  const int64_t n = (int64_t)dimup_to * dimdw_to;
  if (n == 0) return hipSuccess;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(ladder_kernel, dim3((unsigned)blocks), dim3(256), 0, st, map_from, dim_from, map_to, pitch_from, dimup_to, pitch_to,
                     dimdw_to, orbital, spin, create, psi, out, coef, accumulate, perm_from, sign_from, sign_to);
  return hipGetLastError();
}

}  // namespace hxv
