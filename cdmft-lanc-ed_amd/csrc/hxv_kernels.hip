// HIP kernels (gfx950) for the sector product  Hv = D.v + H_up v + v H_dw^T  on the
// DimUp x DimDw layout (up index fastest), complex fp64.
//
// Reference semantics: ED_HAMILTONIAN_SPARSE_HxV.f90:167-227 (serial) / :230-315 (MPI slab).
// No MFMA: the path is a sparse gather over a 2.65 GB vector, bound by HBM and by on-chip
// gather bandwidth.
#include <algorithm>

#include "hxv_internal.hpp"
#include "hxv_tiles.hpp"

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

// ---------------------------------------------------------------------------------------
// Variant 1: two passes over HBM, every in-block gather served from LDS.
//
//   pass A (hxv_pass_up):  hv  = D.v + H_up v      tile = [up block] x [C columns]
//   pass B (hxv_pass_dw):  hv += v H_dw^T          tile = [R rows]   x [dw block]
//
// A "block" is a prefix block of the sorted spin basis (hxv_tiles.hpp): hops among its low
// orbitals stay inside the tile (LDS gathers), hops that touch a high orbital read another
// block of the same columns/rows from global memory; the workgroups that share those
// columns/rows are placed on one XCD so that these reads hit its L2.
// ---------------------------------------------------------------------------------------
struct DevTiles {
  const uint32_t* start;
  const uint32_t* ell_in;
  const uint32_t* ell_out;
  int nblocks, k_in, k_out;
  int ncoef;
};

template <bool REAL>
__device__ __forceinline__ void hfma(double2& acc, double2 c, double2 x) {
  if (REAL) {
    acc.x = fma(c.x, x.x, acc.x);
    acc.y = fma(c.x, x.y, acc.y);
  } else {
    cfma(acc, c, x);
  }
}

// Decode an ELL word against the LDS copy of the coefficient table.  An empty slot decodes to
// (offset 0, coefficient 0): the gather stays in bounds and contributes nothing, so the hop
// loops need no per-lane branch.
__device__ __forceinline__ double2 lds_coef(const double2* lcoef, uint32_t e, bool valid) {
  double2 c = lcoef[(e >> ELL_SRC_BITS) & ELL_COEF_MASK];
  const double sg = valid ? ((e >> 31) ? -1.0 : 1.0) : 0.0;
  c.x *= sg;
  c.y *= sg;
  return c;
}

constexpr int HOP_CHUNK = 4;

template <int C, bool REAL>
__global__ void __launch_bounds__(512) hxv_pass_up(DevSector s, DevTiles t, const double2* __restrict__ v, double2* __restrict__ hv,
                                                  int ngroups) {
  extern __shared__ double2 lds[];
  const int b = blockIdx.x;
  const int xcd = b & 7, j = b >> 3;
  const int g = (j / t.nblocks) * 8 + xcd;  // column group; all blocks of a group share blockIdx%8 (= one XCD)
  const int kb = j - (j / t.nblocks) * t.nblocks;
  if (g >= ngroups) return;
  const int T = blockDim.x;
  const int r0 = (int)t.start[kb];
  const int n = (int)t.start[kb + 1] - r0;
  const int c0 = g * C;  // local column
  const int nc = min(C, s.qdw - c0);
  double2* lcoef = lds + C * n;
  const double2* __restrict__ vcol0 = v + (int64_t)(s.slab0 + c0) * s.dimup;
#pragma unroll
  for (int cc = 0; cc < C; ++cc) {
    if (cc < nc)
      for (int r = threadIdx.x; r < n; r += T) lds[cc * n + r] = vcol0[(int64_t)cc * s.dimup + r0 + r];
    else
      for (int r = threadIdx.x; r < n; r += T) lds[cc * n + r] = make_double2(0.0, 0.0);
  }
  for (int q = threadIdx.x; q < t.ncoef; q += T) lcoef[q] = s.up.coef[q];
  __syncthreads();
  for (int r = threadIdx.x; r < n; r += T) {
    const int i = r0 + r;
    double2 acc[C];
    if (s.diag.mode == 0) {
      const double au = s.diag.a_up[i];
      const uint32_t mu = s.diag.map_up[i];
#pragma unroll
      for (int cc = 0; cc < C; ++cc) {
        const int c = s.dw0 + min(c0 + cc, s.qdw - 1);
        const double d = au + s.diag.a_dw[c] + diag_cross(s.diag.cross, mu, s.diag.map_dw[c]);
        const double2 x = lds[cc * n + r];
        acc[cc] = make_double2(d * x.x, d * x.y);
      }
    } else {
#pragma unroll
      for (int cc = 0; cc < C; ++cc) {
        const double d = s.diag.stored[(int64_t)min(c0 + cc, s.qdw - 1) * s.dimup + i];
        const double2 x = lds[cc * n + r];
        acc[cc] = make_double2(d * x.x, d * x.y);
      }
    }
    // hops that leave the block: gathers from global memory (L2 of this XCD)
    for (int k0 = 0; k0 < t.k_out; k0 += HOP_CHUNK) {
      uint32_t e[HOP_CHUNK];
#pragma unroll
      for (int u = 0; u < HOP_CHUNK; ++u) e[u] = (k0 + u < t.k_out) ? t.ell_out[(int64_t)(k0 + u) * s.dimup + i] : ELL_EMPTY;
      if (__all(e[0] == ELL_EMPTY)) break;
#pragma unroll
      for (int u = 0; u < HOP_CHUNK; ++u) {
        const bool valid = e[u] != ELL_EMPTY;
        const double2 cf = lds_coef(lcoef, e[u], valid);
        const double2* __restrict__ src = vcol0 + (valid ? (e[u] & ELL_SRC_MASK) : 0u);
#pragma unroll
        for (int cc = 0; cc < C; ++cc) hfma<REAL>(acc[cc], cf, src[(int64_t)min(cc, nc - 1) * s.dimup]);
      }
    }
    // hops inside the block: gathers from the LDS tile
    for (int k0 = 0; k0 < t.k_in; k0 += HOP_CHUNK) {
      uint32_t e[HOP_CHUNK];
#pragma unroll
      for (int u = 0; u < HOP_CHUNK; ++u) e[u] = (k0 + u < t.k_in) ? t.ell_in[(int64_t)(k0 + u) * s.dimup + i] : ELL_EMPTY;
      if (__all(e[0] == ELL_EMPTY)) break;
#pragma unroll
      for (int u = 0; u < HOP_CHUNK; ++u) {
        const bool valid = e[u] != ELL_EMPTY;
        const double2 cf = lds_coef(lcoef, e[u], valid);
        const int off = valid ? (int)(e[u] & ELL_SRC_MASK) : 0;
#pragma unroll
        for (int cc = 0; cc < C; ++cc) hfma<REAL>(acc[cc], cf, lds[cc * n + off]);
      }
    }
#pragma unroll
    for (int cc = 0; cc < C; ++cc)
      if (cc < nc) hv[(int64_t)(c0 + cc) * s.dimup + i] = acc[cc];
  }
}

template <int R, bool REAL>
__global__ void __launch_bounds__(512) hxv_pass_dw(DevSector s, DevTiles t, const double2* __restrict__ v, double2* __restrict__ hv,
                                                  int ngroups, int groups_per_xcd) {
  extern __shared__ double2 lds[];
  const int b = blockIdx.x;
  const int xcd = b & 7, j = b >> 3;
  const int gl = j / t.nblocks;
  const int kb = j - gl * t.nblocks;
  const int rg = xcd * groups_per_xcd + gl;  // contiguous row ranges per XCD: neighbouring row groups share cache lines
  if (gl >= groups_per_xcd || rg >= ngroups) return;
  const int cb0 = (int)t.start[kb];
  const int n = (int)t.start[kb + 1] - cb0;
  if (cb0 + n <= s.dw0 || cb0 >= s.dw0 + s.qdw) return;  // block holds no local output column
  const int i0 = rg * R;
  const int r = threadIdx.x % R;
  const int irow = min(i0 + r, s.dimup - 1);  // clamp: out-of-range lanes recompute the last row and do not store
  const bool row_ok = (i0 + r) < s.dimup;
  const int CSTEP = blockDim.x / R;
  double2* lcoef = lds + R * n;
  for (int col = threadIdx.x / R; col < n; col += CSTEP) lds[col * R + r] = v[(int64_t)s.vcol[cb0 + col] * s.dimup + irow];
  for (int q = threadIdx.x; q < t.ncoef; q += blockDim.x) lcoef[q] = s.dw.coef[q];
  __syncthreads();
  for (int col = threadIdx.x / R; col < n; col += CSTEP) {
    const int c = cb0 + col;
    if (c < s.dw0 || c >= s.dw0 + s.qdw) continue;
    const int64_t o = (int64_t)(c - s.dw0) * s.dimup + irow;
    double2 acc = hv[o];
    for (int k0 = 0; k0 < t.k_out; k0 += HOP_CHUNK) {
      uint32_t e[HOP_CHUNK];
#pragma unroll
      for (int u = 0; u < HOP_CHUNK; ++u) e[u] = (k0 + u < t.k_out) ? t.ell_out[(int64_t)(k0 + u) * s.dimdw + c] : ELL_EMPTY;
      if (__all(e[0] == ELL_EMPTY)) break;
#pragma unroll
      for (int u = 0; u < HOP_CHUNK; ++u) {
        const bool valid = e[u] != ELL_EMPTY;
        const double2 cf = lds_coef(lcoef, e[u], valid);
        hfma<REAL>(acc, cf, v[(int64_t)(valid ? (e[u] & ELL_SRC_MASK) : 0u) * s.dimup + irow]);
      }
    }
    for (int k0 = 0; k0 < t.k_in; k0 += HOP_CHUNK) {
      uint32_t e[HOP_CHUNK];
#pragma unroll
      for (int u = 0; u < HOP_CHUNK; ++u) e[u] = (k0 + u < t.k_in) ? t.ell_in[(int64_t)(k0 + u) * s.dimdw + c] : ELL_EMPTY;
      if (__all(e[0] == ELL_EMPTY)) break;
#pragma unroll
      for (int u = 0; u < HOP_CHUNK; ++u) {
        const bool valid = e[u] != ELL_EMPTY;
        const double2 cf = lds_coef(lcoef, e[u], valid);
        hfma<REAL>(acc, cf, lds[(valid ? (int)(e[u] & ELL_SRC_MASK) : 0) * R + r]);
      }
    }
    if (row_ok) hv[o] = acc;
  }
}

namespace {

int64_t binom64(int n, int k) {
  if (k < 0 || k > n) return 0;
  k = std::min(k, n - k);
  int64_t r = 1;
  for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
  return r;
}

// Split one spin sector into prefix blocks of `lowbits` low orbitals and split its ELL table.
// Works from the index structure alone when no basis map is available (from_csr): then a
// single block (lowbits = all) or fixed-size chunks are used.
void build_spin_tiles(const SpinOp& op, const std::vector<uint32_t>& map, int lowbits, int chunk, const std::vector<uint32_t>* vcol,
                      SpinTiles& t, std::vector<uint32_t>& ell_in, std::vector<uint32_t>& ell_out) {
  const int dim = op.dim;
  t.start.clear();
  if (!map.empty()) {
    uint32_t prev = 0xFFFFFFFFu;
    for (int i = 0; i < dim; ++i) {
      uint32_t hi = lowbits >= 32 ? 0u : (map[i] >> lowbits);
      if (hi != prev) {
        t.start.push_back((uint32_t)i);
        prev = hi;
      }
    }
  } else {
    for (int i = 0; i < dim; i += chunk) t.start.push_back((uint32_t)i);
  }
  t.start.push_back((uint32_t)dim);
  t.lowbits = lowbits;
  t.nblocks = (int)t.start.size() - 1;
  t.max_block = 0;
  std::vector<uint32_t> block_of(dim);
  for (int k = 0; k < t.nblocks; ++k) {
    t.max_block = std::max<int>(t.max_block, (int)(t.start[k + 1] - t.start[k]));
    for (uint32_t i = t.start[k]; i < t.start[k + 1]; ++i) block_of[i] = (uint32_t)k;
  }
  // count inner / outer entries per row
  int kin = 0, kout = 0;
  t.n_in = t.n_out = 0;
  for (int i = 0; i < dim; ++i) {
    int a = 0, b = 0;
    for (int64_t p = op.rowptr[i]; p < op.rowptr[i + 1]; ++p) (block_of[op.cols[p]] == block_of[i] ? a : b)++;
    kin = std::max(kin, a);
    kout = std::max(kout, b);
    t.n_in += a;
    t.n_out += b;
  }
  t.k_in = kin;
  t.k_out = kout;
  ell_in.assign((size_t)std::max(kin, 1) * dim, ELL_EMPTY);
  ell_out.assign((size_t)std::max(kout, 1) * dim, ELL_EMPTY);
  for (int i = 0; i < dim; ++i) {
    int a = 0, b = 0, k = 0;
    for (int64_t p = op.rowptr[i]; p < op.rowptr[i + 1]; ++p, ++k) {
      const uint32_t e = op.ell[(size_t)k * dim + i];  // same order as the CSR row
      const uint32_t src = e & ELL_SRC_MASK;
      if (block_of[src] == block_of[i])
        ell_in[(size_t)(a++) * dim + i] = (e & ~ELL_SRC_MASK) | (src - t.start[block_of[i]]);
      else
        ell_out[(size_t)(b++) * dim + i] = vcol ? ((e & ~ELL_SRC_MASK) | (*vcol)[src]) : e;
    }
  }
}

int choose_lowbits(int ns, int npart, int width, int budget_bytes) {
  // largest number of low orbitals whose biggest block still fits width x 16 B in the budget
  for (int L = ns; L >= 0; --L) {
    int64_t mx = 0;
    for (int p = 0; p <= ns - L; ++p) mx = std::max(mx, binom64(L, npart - p));
    if (mx * width * 16 <= budget_bytes) return L;
  }
  return 0;
}

template <int C>
hipError_t launch_up(const DevSector& s, const DevTiles& t, int lds_bytes, int threads, const double2* v, double2* hv, hipStream_t st) {
  const int ngroups = (s.qdw + C - 1) / C;
  const int64_t nwg = (int64_t)((ngroups + 7) / 8) * 8 * t.nblocks;
  auto kern = s.real_h ? hxv_pass_up<C, true> : hxv_pass_up<C, false>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(threads), (size_t)lds_bytes, st, s, t, v, hv, ngroups);
  return hipGetLastError();
}

template <int R>
hipError_t launch_dw(const DevSector& s, const DevTiles& t, int lds_bytes, int threads, const double2* v, double2* hv, hipStream_t st) {
  const int ngroups = (s.dimup + R - 1) / R;
  const int gpx = (ngroups + 7) / 8;
  const int64_t nwg = (int64_t)gpx * 8 * t.nblocks;
  auto kern = s.real_h ? hxv_pass_dw<R, true> : hxv_pass_dw<R, false>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(threads), (size_t)lds_bytes, st, s, t, v, hv, ngroups, gpx);
  return hipGetLastError();
}

}  // namespace

std::string make_tile_plan(const SectorHost& s, TilePlan& plan, const UploadU32& upload) {
  TileOptions& o = plan.opt;
  if (o.cols_per_tile != 2 && o.cols_per_tile != 4 && o.cols_per_tile != 8) return "cols_per_tile must be 2, 4 or 8";
  if (o.rows_per_tile != 4 && o.rows_per_tile != 8 && o.rows_per_tile != 16) return "rows_per_tile must be 4, 8 or 16";
  if (o.lds_budget_kb_up < 8 || o.lds_budget_kb_up > 144 || o.lds_budget_kb_dw < 8 || o.lds_budget_kb_dw > 144)
    return "lds_budget_kb must be in [8,144]";
  if (o.threads_up != 256 && o.threads_up != 512) return "threads_up must be 256 or 512";
  if (o.threads_dw != 256 && o.threads_dw != 512) return "threads_dw must be 256 or 512";
  plan.ncoef_up = (int)s.up.coef.size();
  plan.ncoef_dw = (int)s.dw.coef.size();
  std::vector<uint32_t> ein, eout;
  auto one = [&](const SpinOp& op, const std::vector<uint32_t>& map, int npart, int width, int force, int budget_kb,
                 const std::vector<uint32_t>* vcol, SpinTiles& t) -> std::string {
    const int budget = budget_kb * 1024 - 16 * std::max(plan.ncoef_up, plan.ncoef_dw);
    int L = 32, chunk = std::max(1, budget / (16 * width));
    if (!map.empty()) {
      L = force >= 0 ? std::min(force, s.ns) : choose_lowbits(s.ns, npart, width, budget);
    }
    build_spin_tiles(op, map, L, chunk, vcol, t, ein, eout);
    if ((int64_t)t.max_block * width * 16 + 16 * 1024 > 160 * 1024) return "tile does not fit the 160 KB LDS";
    std::vector<uint32_t> st(t.start.begin(), t.start.end());
    if (upload(st, &t.d_start) != hipSuccess) return "upload of tile table failed";
    if (upload(ein, &t.d_ell_in) != hipSuccess) return "upload of inner ELL failed";
    if (upload(eout, &t.d_ell_out) != hipSuccess) return "upload of outer ELL failed";
    return "";
  };
  std::string e = one(s.up, s.map_up, s.nup, o.cols_per_tile, o.force_bits_up, o.lds_budget_kb_up, nullptr, plan.up);
  if (!e.empty()) return e;
  return one(s.dw, s.map_dw, s.ndw, o.rows_per_tile, o.force_bits_dw, o.lds_budget_kb_dw, &s.vcol, plan.dw);
}

hipError_t launch_hxv_tiled(const DevSector& s, const TilePlan& plan, const double2* v, double2* hv, hipStream_t st) {
  if (s.qdw == 0) return hipSuccess;
  DevTiles tu{plan.up.d_start, plan.up.d_ell_in, plan.up.d_ell_out, plan.up.nblocks, plan.up.k_in, plan.up.k_out, plan.ncoef_up};
  DevTiles td{plan.dw.d_start, plan.dw.d_ell_in, plan.dw.d_ell_out, plan.dw.nblocks, plan.dw.k_in, plan.dw.k_out, plan.ncoef_dw};
  const int C = plan.opt.cols_per_tile, R = plan.opt.rows_per_tile;
  const int lds_a = (plan.up.max_block * C + plan.ncoef_up) * 16, lds_b = (plan.dw.max_block * R + plan.ncoef_dw) * 16;
  const int ta = plan.opt.threads_up, tb = plan.opt.threads_dw;
  hipError_t e = hipSuccess;
  if (plan.opt.passes & 1) switch (C) {
    case 2: e = launch_up<2>(s, tu, lds_a, ta, v, hv, st); break;
    case 4: e = launch_up<4>(s, tu, lds_a, ta, v, hv, st); break;
    default: e = launch_up<8>(s, tu, lds_a, ta, v, hv, st); break;
  }
  if (e != hipSuccess) return e;
  if (plan.opt.passes & 2) switch (R) {
    case 4: e = launch_dw<4>(s, td, lds_b, tb, v, hv, st); break;
    case 8: e = launch_dw<8>(s, td, lds_b, tb, v, hv, st); break;
    default: e = launch_dw<16>(s, td, lds_b, tb, v, hv, st); break;
  }
  return e;
}

}  // namespace hxv
