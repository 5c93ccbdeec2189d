// Tiled two-pass kernels (gfx950) and their plan builder.
//
//   pass A (hxv_pass_up):  hv  = D.v + H_up v      tile = [up prefix block] x [C columns]
//   pass B (hxv_pass_dw):  hv += v H_dw^T          tile = [R rows] x [dw prefix block]
//
// A "block" is a prefix block of the sorted spin basis (hxv_tiles.hpp): hops among its low
// orbitals stay inside the tile and are gathered from LDS; hops that touch a high orbital read
// another block of the same columns (pass A) / rows (pass B) from global memory, and the
// workgroups that share those columns/rows carry the same blockIdx%8 (one XCD) so these reads
// hit its L2.  Reference semantics: ED_HAMILTONIAN_SPARSE_HxV.f90:167-227 / :230-315.
#include <algorithm>
#include <map>
#include <mutex>
#include <thread>
#include <numeric>
#include <type_traits>

#include "hxv_tile_dev.hpp"

#ifndef HXV_PAIR_LDS
#define HXV_PAIR_LDS 1  // (0: A/B builds only -- the real-vector pass A with one 8-byte LDS element per column)
#endif

namespace hxv {

// ---------------------------------------------------------------------------------------
// pass A
// ---------------------------------------------------------------------------------------
// (Round 3 also ran this kernel with several rows per thread -- blocks of 13-14 low orbitals, or 512-thread workgroups on blocks of
//  12 -- and every such plan was slower at C3, C4 and C5 (profiles/r03_ab_mr_*.log): one row per thread it stays.)
// LZ: 0 plain product, 1 Lanczos epilogue, 2 PAIRED Lanczos epilogue (real H, complex vectors): the real and the imaginary part
// are two independent real Lanczos vectors (H(x + iy) = Hx + iHy), each with its own scalars and its own partial sums.
// ND: the spin-exchange / pair-hopping block spH0nd (Jx, Jp; sparse/H_non_local.f90:23-98, ED_HAMILTONIAN_SPARSE_HxV.f90:217-225) is
// added here instead of in a third read-modify-write pass over hv: its terms are Kronecker products of two one-body moves on a site, so
// an element's partner is (up move of the row) x (dw move of the column) -- the up moves of a row serve all C columns of the tile, the dw
// moves of a column are uniform scalars, the partner values come from one other column of the gathered vector (rows of the same block:
// the moves only touch impurity orbitals, the lowest bits).
template <int C, bool REAL, bool NORB1, int LZ, bool P16, typename VT, bool ND = false>
__global__ void __launch_bounds__(1024, 8) hxv_pass_up(DevSector s, DevTiles t, const VT* __restrict__ v,
                                                      const VT* __restrict__ wt, VT* __restrict__ hv, int ngroups,
                                                      int groups_per_xcd, int wc, LzEpilogue lz) {
  using CT = typename Coef<REAL>::type;
  extern __shared__ double2 lds_raw[];
  VT* lds = reinterpret_cast<VT*>(lds_raw);
  const int b = blockIdx.x;
  const int xcd = b & 7, j = b >> 3;
  // column group: all blocks of a group share blockIdx%8 (= one XCD), and each XCD owns a contiguous range of
  // groups because neighbouring groups share the cache lines of the transposed scratch wt
  const int gl = j / t.nblocks;
  int kb = j - gl * t.nblocks;
  // (Round 4 also ran the tiles of Kanamori sectors block-major inside chunks of 8 - 64 column groups, so that the partner columns of the
  //  folded spH0nd block's dw moves were L2-resident: 7.37 - 8.14 ms against 7.12 at C4 with Jx / Jp -- the hopping part's out-of-block
  //  gathers, then G tiles apart, lose more than the block gains; profiles/r04_ab_kanamori_chunks.log.)
  const int g = xcd * groups_per_xcd + gl;
  if (gl >= groups_per_xcd || g >= ngroups) {
    if (LZ && threadIdx.x == 0) {
      lz.partial[blockIdx.x] = 0.0;
      if (LZ == 2) lz.partial2[blockIdx.x] = 0.0;
    }
    return;
  }
  if (t.order) kb = (int)t.order[kb];  // blocks that share their in-block tables are dispatched next to each other
  const int T = blockDim.x;
  const int r0 = (int)t.start[kb];
  const int n = (int)t.start[kb + 1] - r0;
  const int tb0 = (int)t.tstart[kb];  // the block whose in-block tables this one shares
  const int c0 = g * C;  // local column
  const int nc = min(C, s.qdw - c0);
  CT* lcoef = reinterpret_cast<CT*>(lds + C * n);
  const VT* __restrict__ vcol0 = v + (int64_t)(s.slab0 + c0) * s.pitch;
  // REAL vectors (round 5): the columns of a tile are independent batch entries here (the up hops act on rows), so two adjacent columns
  // share one 16-byte LDS element: an in-block hop costs C/2 ds_read_b128 instead of C ds_read_b64 -- the LDS access pattern of the
  // complex kernel, for which the hop lists were dealt over the banks.  Same FMAs in the same order: bit-identical results.
  constexpr bool PL = HXV_PAIR_LDS && sizeof(VT) == 8 && (C % 2 == 0);
  auto lidx = [&](int cc, int r) -> int { return PL ? ((((cc >> 1) * n + r) << 1) + (cc & 1)) : cc * n + r; };
  if constexpr (PL) {
    double2* l2 = reinterpret_cast<double2*>(lds);
#pragma unroll
    for (int pc = 0; pc < C / 2; ++pc) {
      const VT* __restrict__ sa = vcol0 + (int64_t)min(2 * pc, nc - 1) * s.pitch + r0;
      const VT* __restrict__ sb = vcol0 + (int64_t)min(2 * pc + 1, nc - 1) * s.pitch + r0;
      for (int r = threadIdx.x; r < n; r += T) l2[pc * n + r] = make_double2((double)sa[r], (double)sb[r]);
    }
  } else {
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
      const VT* __restrict__ src = vcol0 + (int64_t)min(cc, nc - 1) * s.pitch + r0;
      for (int r = threadIdx.x; r < n; r += T) lds[cc * n + r] = src[r];
    }
  }
  for (int q = threadIdx.x; q < t.nscoef; q += T) lcoef[q] = Coef<REAL>::from(t.scoef[q]);
  uint32_t* lrq = reinterpret_cast<uint32_t*>(lcoef + t.nscoef);  // (ND only; the launcher adds the bytes)
  const int nvp = ND ? s.nd.nlat * s.nd.norb * (s.nd.norb - 1) : 0;  // ordered pairs of different orbitals of a site
  uint32_t* lnd = lrq + nvp;
  if constexpr (ND) {
    const int O = s.nd.norb;
    for (int idx = threadIdx.x; idx < nvp * 2 * C; idx += T) {
      const int pp = idx / (2 * C), kind = (idx / C) & 1, cc = idx % C;
      const int il = pp / (O * (O - 1)), rem = pp % (O * (O - 1)), io = rem / (O - 1);
      int jo = rem % (O - 1);
      jo += jo >= io ? 1 : 0;
      const int q = (il * O + io) * O + jo, rq = (il * O + jo) * O + io;
      const int c = s.dw0 + min(c0 + cc, s.qdw - 1);
      // dw: spin exchange moves i -> j (entry q), pair hopping j -> i (entry rq); up (both terms): j -> i (entry rq)
      uint32_t w = ND_INVALID;
      if (kind == 0 && s.nd.jx != 0.0) w = s.nd_dw[(int64_t)q * s.dimdw + c];
      if (kind == 1 && s.nd.jp != 0.0) w = s.nd_dw[(int64_t)rq * s.dimdw + c];
      lnd[idx] = w;
      if (kind == 0 && cc == 0) lrq[pp] = (uint32_t)rq;
    }
  }
  const uint32_t p16m = (1u << t.p16_bits) - 1u;  // half-size table words: (coefficient index << p16_bits) | offset
  double asum = 0.0;
  // one row per thread (the plan guarantees n <= blockDim.x)
  const int p = threadIdx.x;
  VT acc[C];
  double au = 0.0;
  uint32_t mu = 0;
  // Issued BEFORE the barrier so their latency hides behind the tile load: the dw-hop part that pass B left in the
  // column-group-blocked scratch wt[group][row][wc] (C*16 contiguous bytes per row, rows consecutive: a plain
  // streaming read) becomes the initial value of the accumulators; the diagonal's per-row inputs come along.
  auto row_inputs = [&]() {
    if (p < n) {
      if (wt && wc == 0) {
        // natural layout [local column][pitch] (the dw part assembled like hv) -- or, with t.wtr, the same in PIECES (all-to-all
        // exchange): one block per rank of origin, read where the second transpose left it (WtRange)
        const VT* wb = wt;
        int64_t wstr = s.pitch;
        int wrow = r0 + p;
        if (t.wtr) {
          int k = 0;
          while (k + 1 < t.nwtr && wrow >= t.wtr[k].row1) ++k;
          wb = reinterpret_cast<const VT*>(t.wtr[k].base);
          wstr = t.wtr[k].stride;
          wrow -= t.wtr[k].row0;
        }
        const VT* __restrict__ wcol = wb + (int64_t)c0 * wstr + wrow;
#pragma unroll
        for (int cc = 0; cc < C; ++cc) acc[cc] = wcol[(int64_t)min(cc, nc - 1) * wstr];
      } else if (wt && (wc >> 8)) {
        // blocked scratch with COLUMN-MAJOR patches (round 5): wt[group][patch of Rp rows][wcl columns][Rp rows], Rp = pass B's rows per tile in
        // this kernel's element units.  Pass B still writes Rp*wcl*16 contiguous bytes per patch; here the Rp lanes of a patch read Rp*16
        // contiguous bytes per column instead of each lane its own 64-byte stretch four (eight) times over: a quarter of the L1 accesses
        // for the same lines.  (REAL vectors whose pass B ran on row pairs arrive in the same layout: a pair of rows IS two rows of it.)
        const int wcl = wc & 0xFF, Rp = wc >> 8, row = r0 + p;
        const int dR = (s.dimup + Rp - 1) & ~(Rp - 1);
        const VT* __restrict__ wrow = wt + ((int64_t)(c0 / wcl) * dR + (row & ~(Rp - 1))) * wcl + (int64_t)(c0 % wcl) * Rp + (row & (Rp - 1));
#pragma unroll
        for (int cc = 0; cc < C; ++cc) acc[cc] = wrow[min(cc, nc - 1) * Rp];
      } else if (wt) {
        const VT* __restrict__ wrow = wt + ((int64_t)(c0 / wc) * s.dimup + r0 + p) * wc + (c0 % wc);
#pragma unroll
        for (int cc = 0; cc < C; ++cc) acc[cc] = wrow[min(cc, nc - 1)];
      } else {
#pragma unroll
        for (int cc = 0; cc < C; ++cc) acc[cc] = vzero<VT>();
      }
      if (s.diag.mode == 0) {
        au = s.diag.a_up[r0 + p];
        mu = s.diag.map_up[r0 + p];
      }
    }
  };
  row_inputs();
  __syncthreads();
  VT xq[LZ ? C : 1];  // the thread's own input elements, kept for the Lanczos epilogue
  if (p < n) {
    const uint32_t packed = __builtin_amdgcn_readfirstlane(t.gmax[t.gstart[kb] + (p >> 6)]);
    const int kin = (int)(packed & 0xFFFFu);
    if (LZ) {
#pragma unroll
      for (int cc = 0; cc < C; ++cc) xq[LZ ? cc : 0] = lds[lidx(cc, p)];
    }
    const int i = r0 + p;  // pass A visits the rows in natural order: every global access stays coalesced
    const int r = p;
    if (s.diag.mode == 0) {
#pragma unroll
      for (int cc = 0; cc < C; ++cc) {
        const double d = diag_value<NORB1>(s.diag, au, mu, s.dw0 + min(c0 + cc, s.qdw - 1));
        Coef<true>::fma(acc[cc], d, lds[lidx(cc, r)]);
      }
    } else {
#pragma unroll
      for (int cc = 0; cc < C; ++cc) {
        const double d = s.diag.stored[(int64_t)min(c0 + cc, s.qdw - 1) * s.dimup + i];
        Coef<true>::fma(acc[cc], d, lds[lidx(cc, r)]);
      }
    }
    // hops that leave the block, same columns, other rows: from global memory (L2 of this XCD)
    if (!(t.debug & 1)) {
      // block hops: the partner block is one contiguous run, lanes read consecutive rows
      for (uint32_t h = t.bh_ptr[kb]; h < ((t.debug & 256) ? t.bh_ptr[kb] : t.bh_ptr[kb + 1]); ++h) {
        const CT cf = lcoef[t.bh[2 * h + 1]];
        const VT* __restrict__ src = vcol0 + t.bh[2 * h] + r;
#pragma unroll
        for (int cc = 0; cc < C; ++cc) Coef<REAL>::fma(acc[cc], cf, src[(int64_t)min(cc, nc - 1) * s.pitch]);
      }
      // row slots: one table word per row and (block, source block) pair -- or, packed, per TWO such pairs (P16)
      const uint32_t rs1 = (t.debug & 512) ? t.rs_ptr[kb] : t.rs_ptr[kb + 1];
      // (eight-column tiles -- real vectors -- keep one word per slot: with the packed words the fused real-vector pass A went from
      //  1.67 to 1.88 ms at C3, the whole round-3 regression of the real Lanczos iteration, 3.77 -> 3.96 ms; profiles/r04_bisect_real.log)
      if (P16 && C < 8 && t.rs16) {
        const uint32_t empty16 = (uint32_t)(t.nscoef - 1) << t.p16_bits;
        for (uint32_t sl = t.rs_ptr[kb]; sl < rs1; sl += 2) {
          const uint32_t w = t.rs16[t.rs16_off[sl] + r];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            if (sl + hh < rs1) {  // (uniform)
              const uint32_t e = hh ? w >> 16 : w & 0xFFFFu;
              if (__all(e == empty16)) continue;
              CT cf = lcoef[e >> t.p16_bits];
              if (t.rs_neg[sl + hh]) cf = Coef<REAL>::neg(cf);  // (uniform: the shared table holds the other overall sign)
              const VT* __restrict__ src = vcol0 + t.rs_base[sl + hh] + (e & p16m);
#pragma unroll
              for (int cc = 0; cc < C; ++cc) Coef<REAL>::fma(acc[cc], cf, src[(int64_t)min(cc, nc - 1) * s.pitch]);
            }
          }
        }
      } else {
        const uint32_t emptyz = (uint32_t)(t.nscoef - 1) << TILE_COEF_SHIFT;
        for (uint32_t sl = t.rs_ptr[kb]; sl < rs1; ++sl) {
          const uint32_t e = t.rs_tab[t.rs_off[sl] + r];
          if (__all(e == emptyz)) continue;
          CT cf = lcoef[e >> TILE_COEF_SHIFT];
          if (t.rs_neg[sl]) cf = Coef<REAL>::neg(cf);  // (uniform: the shared table holds the other overall sign)
          const VT* __restrict__ src = vcol0 + t.rs_base[sl] + (e & TILE_OFF_MASK);
#pragma unroll
          for (int cc = 0; cc < C; ++cc) Coef<REAL>::fma(acc[cc], cf, src[(int64_t)min(cc, nc - 1) * s.pitch]);
        }
      }
    }
    if constexpr (ND) {
      // lrq / lnd (LDS, filled with the tile): per ordered orbital pair p of a site, the up-move table row and, per column of the tile, the
      // dw partner columns of the two terms -- uniform values, read back as scalars; a term the column's dw state rules out costs nothing.
      // Lanes whose up state rules the move out fetch their own row with a zero coefficient (no lane-divergent branch around a load).
      for (int p0 = 0; p0 < nvp; p0 += 8) {
        uint32_t u[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (p0 + k < nvp) u[k] = s.nd_up[(int64_t)__builtin_amdgcn_readfirstlane(lrq[p0 + k]) * s.dimup + i];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (p0 + k < nvp) {
            const bool upok = u[k] != ND_INVALID;
            const int64_t jup = upok ? (int64_t)(u[k] & 0x7FFFFFFFu) : (int64_t)r0;  // (idle lanes: one shared address)
            uint32_t dw[2 * C];  // the pair's partner columns for the C columns of the tile: one batch of LDS reads
#pragma unroll
            for (int e = 0; e < 2 * C; ++e) dw[e] = lnd[(p0 + k) * 2 * C + e];
#pragma unroll
            for (int e = 0; e < 2 * C; ++e) dw[e] = __builtin_amdgcn_readfirstlane(dw[e]);
#pragma unroll
            for (int cc = 0; cc < C; ++cc) {
              const uint32_t dse = dw[cc], dph = dw[C + cc];
              VT x1 = vzero<VT>(), x2 = vzero<VT>();
              if (dse != ND_INVALID) x1 = v[(int64_t)(dse & 0x7FFFFFFFu) * s.pitch + jup];
              if (dph != ND_INVALID) x2 = v[(int64_t)(dph & 0x7FFFFFFFu) * s.pitch + jup];
              if (dse != ND_INVALID) Coef<true>::fma(acc[cc], upok ? (((u[k] ^ dse) >> 31) ? -s.nd.jx : s.nd.jx) : 0.0, x1);
              if (dph != ND_INVALID) Coef<true>::fma(acc[cc], upok ? (((u[k] ^ dph) >> 31) ? -s.nd.jp : s.nd.jp) : 0.0, x2);
            }
          }
        }
      }
    }
    // hops inside the block: gathers from the LDS tile
    for (int k0 = 0; k0 < ((t.debug & 2) ? 0 : kin); k0 += HOP_CHUNK) {
      uint32_t e[P16 ? HOP_CHUNK / 2 : HOP_CHUNK];
      if constexpr (P16) {  // half-size table: two hops per word
#pragma unroll
        for (int u = 0; u < HOP_CHUNK / 2; ++u) e[u] = t.ell16[(int64_t)(k0 / 2 + u) * s.dimup + tb0 + p];
      } else {
#pragma unroll
        for (int u = 0; u < HOP_CHUNK; ++u) e[u] = t.ell_in[(int64_t)(k0 + u) * s.dimup + tb0 + p];
      }
#pragma unroll
      for (int u = 0; u < HOP_CHUNK; ++u) {
        if (k0 + u < kin) {  // wave-uniform: all 8 words are loaded at once, only the live slots are computed
          uint32_t ci;
          int off;
          if constexpr (P16) {
            const uint32_t hw = (u & 1) ? e[u >> 1] >> 16 : e[u >> 1] & 0xFFFFu;
            ci = hw >> t.p16_bits;
            off = (int)(hw & p16m);
          } else {
            ci = e[u] >> TILE_COEF_SHIFT;
            off = (int)(e[u] & TILE_OFF_MASK);
          }
          const CT cf = lcoef[ci];
          if constexpr (PL) {
            const double2* l2 = reinterpret_cast<const double2*>(lds);
#pragma unroll
            for (int pc = 0; pc < C / 2; ++pc) {
              const double2 x2 = l2[pc * n + off];
              Coef<REAL>::fma(acc[2 * pc], cf, x2.x);
              Coef<REAL>::fma(acc[2 * pc + 1], cf, x2.y);
            }
          } else {
#pragma unroll
            for (int cc = 0; cc < C; ++cc) Coef<REAL>::fma(acc[cc], cf, lds[cc * n + off]);
          }
        }
      }
    }
  }
  // Epilogue, same thread <-> row mapping: store hv with lanes along the rows.
  // With LZ: w = s*(H x) - c*xm and the partial sums of Re(conj(s*x) w).
  double asum2 = 0.0;  // (LZ == 2: the imaginary-part vector's sum)
  if (p < n) {
    const double sc = LZ ? lz.scal[lz.i_s] : 1.0;
    const double cm = (LZ && lz.xm) ? lz.scal[lz.i_c] : 0.0;
    const double sc2 = LZ == 2 ? lz.scal[lz.i_s2] : 1.0;
    const double cm2 = (LZ == 2 && lz.xm) ? lz.scal[lz.i_c2] : 0.0;
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
      if (cc < nc) {
        const int64_t o = (int64_t)(c0 + cc) * s.pitch + r0 + p;
        VT w = acc[cc];
        if constexpr (LZ == 2) {
          // component-wise: the same operations, in the same order, as two LZ == 1 runs on (x, 0) and (y, 0)
          const VT xo = xq[cc];
          pair_scale(w, sc, sc2);
          if (lz.xm) pair_fma(w, -cm, -cm2, reinterpret_cast<const VT*>(lz.xm)[o]);
          asum = ::fma(sc, pair_dot_re(xo, w), asum);
          asum2 = ::fma(sc2, pair_dot_im(xo, w), asum2);
        } else if (LZ) {
          vscale(w, sc);
          if (lz.xm) Coef<true>::fma(w, -cm, reinterpret_cast<const VT*>(lz.xm)[o]);
          asum = ::fma(sc, vdot(xq[LZ ? cc : 0], w), asum);
        }
        if (t.debug & 8)
          hv[o] = w;
        else  // contiguous runs of a whole block: streaming stores measured ~10 % faster here
          store_stream(&hv[o], w);
      }
    }
  }
  if (LZ) {
    // wavefront partial sums first (shuffles), one LDS word per wave afterwards: two barriers instead of a tree of eleven
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      asum += __shfl_down(asum, off, 64);
      if (LZ == 2) asum2 += __shfl_down(asum2, off, 64);
    }
    __syncthreads();  // every gather from the tile is done
    double* red = reinterpret_cast<double*>(lds);
    if ((threadIdx.x & 63) == 0) {
      red[threadIdx.x >> 6] = asum;
      if (LZ == 2) red[16 + (threadIdx.x >> 6)] = asum2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double tot = 0.0, tot2 = 0.0;
      for (int w = 0; w < (T >> 6); ++w) {
        tot += red[w];
        if (LZ == 2) tot2 += red[16 + w];
      }
      lz.partial[blockIdx.x] = tot;
      if (LZ == 2) lz.partial2[blockIdx.x] = tot2;
    }
  }
}

// ---------------------------------------------------------------------------------------
// pass B.  Tile element (column c of the block, row r) sits at pair index q = c*R + r in LDS, which is also the order the
// lanes touch global memory in (lanes along the R contiguous rows of a column: coalesced 16*R-byte segments), so the
// streaming phases address LDS linearly.  In the in-block phase a thread owns one column and its R rows: one table decode
// serves R gathers at immediate offsets.  The out-of-block hops run afterwards, lanes along the rows again, after the
// in-block sums have been parked in the tile.
// The address arithmetic is kept off the vector ALU (it was more than half of this kernel's instructions): blockDim.x is a
// multiple of R, so a thread's pairs all have the same row; every global access is a uniform 64-bit base plus a per-thread
// 32-bit byte offset that is computed once (tile load, block hops, scratch store), or one 64-bit multiply-add (row slots).
// ---------------------------------------------------------------------------------------
template <int R, int NP, bool REAL, bool P16, typename VT>
__global__ void __launch_bounds__(1024, 8) hxv_pass_dw(DevSector s, DevTiles t, const VT* __restrict__ v, VT* __restrict__ wt,
                                                      int ngroups, int groups_per_xcd, int wc) {
  // NP = (row,column) pairs of the tile per thread (plan: max_block*R <= NP*blockDim.x); all their global
  // loads are issued before the first use so a workgroup keeps NP requests per lane in flight.
  using CT = typename Coef<REAL>::type;
  extern __shared__ double2 lds_raw[];
  constexpr int VB = (int)sizeof(VT);
  constexpr int LR = R == 2 ? 1 : (R == 4 ? 2 : 3);
  static_assert(R == 2 || R == 4 || R == 8, "R");
  const int b = blockIdx.x;
  const int xcd = b & 7, j = b >> 3;
  int gl = j / t.nblocks;
  int kb = j - gl * t.nblocks;
  if (t.pair_rows) {
    // Large sectors: one row group's panel (R rows x DimDw columns, in R*16-byte pieces of 128-byte lines) no longer fits the
    // XCD's L2 next to its neighbour's, so the row group that shares its lines would find them evicted.  The two run back
    // to back instead, block by block (Ns=18: pass B 53.8 -> 47.5 ms; at Ns=16, where both panels fit, this order is 7 % slower).
    const int pr = j / (2 * t.nblocks), rem = j - pr * 2 * t.nblocks;
    kb = rem >> 1;
    gl = 2 * pr + (rem & 1);
  }
  const int rg = xcd * groups_per_xcd + gl;  // contiguous row ranges per XCD: neighbouring row groups share cache lines
  if (gl >= groups_per_xcd || rg >= ngroups) return;
  if (t.order) kb = (int)t.order[kb];  // blocks that share their in-block tables are dispatched next to each other
  const int cb0 = (int)t.start[kb];
  const int n = (int)t.start[kb + 1] - cb0;
  if (cb0 + n <= s.dw0 || cb0 >= s.dw0 + s.qdw) return;  // block holds no local output column
  const int T = blockDim.x;
  const int tid = threadIdx.x;
  const int i0 = rg * R;
  // LDS (by byte offset): the signed coefficients at 0 -- an in-block table word's coefficient field shifted down IS the
  // coefficient's address, because a column offset within a block (< 1024) leaves the bits between the fields clear --
  // and the tile behind them
  constexpr int LCB = sizeof(CT) == 8 ? 3 : 4;
  constexpr int LTB = LR + (VB == 8 ? 3 : 4);  // log2 of the bytes of one column of the tile
  const uint32_t ltile = (uint32_t)((t.nscoef << LCB) + 255) & ~255u;
  // XOR swizzle of the R row positions inside a column's chunk by the column's index among the columns that share its
  // 256-byte bank sweep: without it the R gathers of an in-block hop (all lanes at the same row position of random
  // columns) would use only 1/R of the banks.  The linear phases see a permutation inside each chunk: still conflict-free.
  constexpr int LSW = 8 - LTB;                      // log2(columns per 256 B); LTB <= 7
  constexpr int LVB = VB == 8 ? 3 : 4;
  auto swz_of = [](uint32_t col) -> uint32_t { return ((col >> LSW) & (R - 1)) << LVB; };  // byte XOR of a column
  const uint32_t tq = ltile + (((uint32_t)tid << LVB) ^ swz_of((uint32_t)tid >> LR));  // this thread's pair 0 (pair it: + it*T*VB)
  const uint32_t emptyz = (uint32_t)(t.nscoef - 1) << TILE_COEF_SHIFT;
  const uint32_t p16m = (1u << t.p16_bits) - 1u;  // half-size table words: (coefficient index << p16_bits) | column
  for (int q = tid; q < t.nscoef; q += T) lds_st<CT>(q << LCB, Coef<REAL>::from(t.scoef[q]));
  // per-thread constants
  const int r = tid & (R - 1);
  const int cstep = T >> LR;                      // columns between a thread's consecutive pairs
  const uint32_t pitchb = (uint32_t)s.pitch * VB;  // bytes per column
  const uint32_t rowb = (uint32_t)(min(i0 + r, s.dimup - 1) - i0) * VB;
  const char* __restrict__ vrows = reinterpret_cast<const char*>(v) + (int64_t)i0 * VB;  // uniform
  // column of pair `it` within the block (clamped) and its byte offset relative to (first column of a block, row i0);
  // the offsets are kept in registers when a thread has few pairs and recomputed (two instructions) when it has eight
  auto ccol = [&](int it) -> uint32_t { return (uint32_t)min((tid >> LR) + it * cstep, n - 1); };
  constexpr bool KEEP = NP <= 4 && R < 8;  // (eight-row tiles need the registers for their accumulators)
  uint32_t voff_keep[KEEP ? NP : 1];
  if constexpr (KEEP) {
#pragma unroll
    for (int it = 0; it < NP; ++it) voff_keep[it] = ccol(it) * pitchb + rowb;
  }
  auto voff = [&](int it) -> uint32_t {
    if constexpr (KEEP)
      return voff_keep[it];
    else
      return ccol(it) * pitchb + rowb;
  };
  // phase 0: tile load
  if (s.vcol_identity) {
    const char* __restrict__ src = vrows + (int64_t)cb0 * pitchb;
    VT x[NP];
#pragma unroll
    for (int it = 0; it < NP; ++it) x[it] = *reinterpret_cast<const VT*>(src + voff(it));
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      if ((tid >> LR) + it * cstep < n) lds_st<VT>(tq + it * T * VB, x[it]);
    }
  } else {
    uint32_t slot[NP];
#pragma unroll
    for (int it = 0; it < NP; ++it) slot[it] = s.vcol[cb0 + ccol(it)];
    VT x[NP];
#pragma unroll
    for (int it = 0; it < NP; ++it) x[it] = *reinterpret_cast<const VT*>(vrows + ((uint64_t)slot[it] * pitchb + rowb));
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      if ((tid >> LR) + it * cstep < n) lds_st<VT>(tq + it * T * VB, x[it]);
    }
  }
  const uint32_t rs0 = t.rs_ptr[kb], rs_end = (t.debug & 512) ? rs0 : t.rs_ptr[kb + 1];
  __syncthreads();
  // in-block hops, one column per thread (plan guarantees n <= blockDim.x)
  {
    VT acc[R];
    int col1 = 0;
    if (tid < n) {
      const uint32_t packed = __builtin_amdgcn_readfirstlane(t.gmax[t.gstart[kb] + (tid >> 6)]);
      const int kin = (t.debug & 2) ? 0 : (int)(packed & 0xFFFFu);
      const int tb0 = (int)t.tstart[kb];  // the block whose in-block tables this one shares
      col1 = (int)t.perm[tb0 + tid] - tb0;
      const uint32_t* __restrict__ ellp = (P16 ? t.ell16 : t.ell_in) + tb0 + tid;
#pragma unroll
      for (int rr = 0; rr < R; ++rr) acc[rr] = vzero<VT>();
      for (int k0 = 0; k0 < kin; k0 += HOP_CHUNK) {
        uint32_t e[P16 ? HOP_CHUNK / 2 : HOP_CHUNK];
        if constexpr (P16) {
#pragma unroll
          for (int u = 0; u < HOP_CHUNK / 2; ++u) e[u] = ellp[(int64_t)(k0 / 2 + u) * s.dimdw];
        } else {
#pragma unroll
          for (int u = 0; u < HOP_CHUNK; ++u) e[u] = ellp[(int64_t)(k0 + u) * s.dimdw];
        }
#pragma unroll
        for (int u = 0; u < HOP_CHUNK; ++u) {
          if (k0 + u < kin) {  // wave-uniform
            uint32_t cfa, col;  // coefficient's LDS address, source column within the block
            if constexpr (P16) {
              const uint32_t hw = (u & 1) ? e[u >> 1] >> 16 : e[u >> 1] & 0xFFFFu;
              cfa = (hw >> t.p16_bits) << LCB;
              col = hw & p16m;
            } else {
              cfa = e[u] >> (TILE_COEF_SHIFT - LCB);
              col = e[u] & TILE_OFF_MASK;
            }
            const CT cf = lds_ld<CT>(cfa);
            const uint32_t src = shl_add<LTB>(col, ltile), sw = swz_of(col);
#pragma unroll
            for (int rr = 0; rr < R; ++rr) Coef<REAL>::fma(acc[rr], cf, lds_ld<VT>(src + ((uint32_t)(rr * VB) ^ sw)));
          }
        }
      }
    }
    __syncthreads();  // every in-block gather is done: the tile can be overwritten by the sums
    if (tid < n) {
      const uint32_t dst = ltile + ((uint32_t)col1 << LTB), sw = swz_of((uint32_t)col1);
#pragma unroll
      for (int rr = 0; rr < R; ++rr) lds_st<VT>(dst + ((uint32_t)(rr * VB) ^ sw), acc[rr]);
    }
  }
  __syncthreads();
  // out-of-block hops: lanes along the contiguous rows again (coalesced R*16-byte segments of other columns, L2 of
  // this XCD); the sums are added into the tile, each element by the one thread that owns the (row,column) pair.
  // Pairs are handled HB at a time to bound the registers (two 1024-thread workgroups per CU need <= 64 VGPRs).
  if (!(t.debug & 1)) {
    constexpr int HB = NP > 4 ? 4 : NP;
    const char* __restrict__ vrow64 = vrows + rowb;  // per-thread 64-bit base of the row-slot gathers
#pragma unroll
    for (int base = 0; base < NP; base += HB) {
      VT osum[HB];
#pragma unroll
      for (int it = 0; it < HB; ++it) osum[it] = vzero<VT>();
      // block hops: source column slot = start + column offset, one signed coefficient for the whole block
      for (uint32_t h = t.bh_ptr[kb]; h < ((t.debug & 256) ? t.bh_ptr[kb] : t.bh_ptr[kb + 1]); ++h) {
        const CT cf = lds_ld<CT>(t.bh[2 * h + 1] << LCB);
        const char* __restrict__ src = vrows + (int64_t)t.bh[2 * h] * pitchb;
        VT x[HB];
#pragma unroll
        for (int it = 0; it < HB; ++it) x[it] = *reinterpret_cast<const VT*>(src + voff(base + it));
#pragma unroll
        for (int it = 0; it < HB; ++it) Coef<REAL>::fma(osum[it], cf, x[it]);
      }
      // row slots: one table word per column of the block and (block, source block) pair; the words of SB slots
      // are fetched together so that the gathers that depend on them follow one table round trip, not SB
      // (the half-size, two-slots-per-word copy of these tables serves pass A only: here its decode costs the registers that
      //  keep eight-pair tiles from spilling, for no measurable gain)
      constexpr int SB = 2;
      for (uint32_t sl0 = rs0; sl0 < rs_end; sl0 += SB) {
        uint32_t e[SB][HB];
#pragma unroll
        for (int jj = 0; jj < SB; ++jj) {
          if (sl0 + jj < rs_end) {  // uniform
            const uint32_t* __restrict__ tab = t.rs_tab + t.rs_off[sl0 + jj];
#pragma unroll
            for (int it = 0; it < HB; ++it) e[jj][it] = tab[ccol(base + it)];
          }
        }
#pragma unroll
        for (int jj = 0; jj < SB; ++jj) {
          if (sl0 + jj < rs_end) {
            bool none = true;
#pragma unroll
            for (int it = 0; it < HB; ++it) none = none && (e[jj][it] == emptyz);
            if (__all(none)) continue;
            VT x[HB];
            const char* __restrict__ vslot = vrow64 + (uint64_t)t.rs_base[sl0 + jj] * pitchb;  // (the slot's source block)
            const bool neg = t.rs_neg[sl0 + jj] != 0;  // (uniform: the shared table holds the other overall sign)
#pragma unroll
            for (int it = 0; it < HB; ++it)
              x[it] = *reinterpret_cast<const VT*>(vslot + (uint64_t)(e[jj][it] & TILE_OFF_MASK) * pitchb);
#pragma unroll
            for (int it = 0; it < HB; ++it) {
              CT cf = lds_ld<CT>((e[jj][it] >> TILE_COEF_SHIFT) << LCB);
              if (neg) cf = Coef<REAL>::neg(cf);
              Coef<REAL>::fma(osum[it], cf, x[it]);
            }
          }
        }
      }
#pragma unroll
      for (int it = 0; it < HB; ++it) {
        if ((tid >> LR) + (base + it) * cstep < n) {
          const uint32_t q = tq + (base + it) * T * VB;
          VT a = lds_ld<VT>(q);
          vadd(a, osum[it]);
          lds_st<VT>(q, a);
        }
      }
    }
  }
  __syncthreads();
  const int cl0 = max(cb0, s.dw0) - s.dw0, cl1 = min(cb0 + n, s.dw0 + s.qdw) - s.dw0;  // local output columns [cl0,cl1)
  if (wc == 0) {
    // natural layout [local column][pitch], lanes along the R rows (row-panel product of the all-to-all exchange:
    // 1/P of the data, so the short strided write runs do not matter)
    for (int q = tid, k = 0; q < n * R; q += T, ++k) {
      const int lc = cb0 + (q >> LR) - s.dw0;
      if (lc >= cl0 && lc < cl1 && i0 + r < s.dimup) wt[(int64_t)lc * s.pitch + i0 + r] = lds_ld<VT>(tq + k * T * VB);
    }
    return;
  }
  // store into the column-group-blocked scratch wt[group][row][wc] (wc = pass A's scratch group width, a power of two): the
  // R rows x wc columns of one group are R*wc*16 contiguous, aligned bytes -- strided WRITES need long aligned runs on this
  // memory system -- and pass A later reads its whole tile of wt as one contiguous run.  A sweep of the workgroup covers
  // T/(R*wc) groups = T/R columns; uniform base and LDS address advance by constants from sweep to sweep.
  {
    const bool cm = (wc & 0x100) != 0;       // column-major patches (pass A's tile kernel reads them with a quarter of the L1 accesses)
    wc &= 0xFF;
    const int lw = 31 - __clz(wc);
    const int g0 = cl0 >> lw, g1 = (cl1 + wc - 1) >> lw;
    const int per = R << lw;                 // elements of one group's patch
    const int gstep = T >> (LR + lw);        // groups per sweep (T >= R*wc)
    const int rem = tid & (per - 1);         // this thread's element of the patch = its place in the patch's contiguous bytes
    const int r2 = cm ? (rem & (R - 1)) : (rem >> lw), cc = cm ? (rem >> LR) : (rem & (wc - 1));
    const int gi = tid >> (LR + lw);
    const bool rowok = i0 + r2 < s.dimup;
    int lc = ((g0 + gi) << lw) + cc;         // local column of this thread in the current sweep
    const uint32_t drows = cm ? (uint32_t)((s.dimup + R - 1) & ~(R - 1)) : (uint32_t)s.dimup;  // rows of a group's stretch (whole patches when column-major)
    const uint32_t so = (uint32_t)gi * (drows * per / R * VB) + (uint32_t)rem * VB;
    // LDS byte offset of (column, row); a sweep advances the column by T/R, which leaves its swizzle bits alone
    uint32_t lo = ltile + (uint32_t)(((lc + s.dw0 - cb0) << LTB) + ((r2 << LVB) ^ (int)swz_of((uint32_t)(lc + s.dw0 - cb0))));
    char* __restrict__ dstb = reinterpret_cast<char*>(wt) + ((int64_t)g0 * drows + i0) * ((int64_t)VB << lw);
    const int64_t dstep = (int64_t)gstep * drows * ((int64_t)VB << lw);
    for (int g = g0; g < g1; g += gstep) {
      // streaming store: wt is read back once, by pass A, long after it has left L2; not letting it linger leaves the L2
      // to the tile lines that the out-of-block gathers of the neighbouring workgroups hit (-3 % on pass B, measured)
      if (lc >= cl0 && lc < cl1 && rowok) store_stream(reinterpret_cast<VT*>(dstb + so), lds_ld<VT>(lo));
      lc += gstep << lw;
      lo += T * VB;
      dstb += dstep;
    }
  }
}

// ---------------------------------------------------------------------------------------
// plan builder (host)
// ---------------------------------------------------------------------------------------
namespace {

int64_t binom64(int n, int k) {
  if (k < 0 || k > n) return 0;
  k = std::min(k, n - k);
  int64_t r = 1;
  for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
  return r;
}

int choose_lowbits(int ns, int npart, int width, int budget_bytes, int max_block) {
  // largest number of low orbitals whose biggest block fits width x 16 B in the budget
  for (int L = ns; L >= 0; --L) {
    int64_t mx = 0;
    for (int p = 0; p <= ns - L; ++p) mx = std::max(mx, binom64(L, npart - p));
    if (mx * width * 16 <= budget_bytes && mx <= max_block) return L;
  }
  return 0;
}

struct HostTiles {
  std::vector<uint32_t> start, perm, gstart, gmax, ell_in, ell16, tstart;
  std::vector<uint32_t> bh_ptr, bh, rs_ptr, rs_off, rs_tab, rs_base, rs_neg, order, order_pc, rs16, rs16_off;
};

// sorted_out: outer table indexed by sorted position (pass A) or by natural index (pass B)
// tile_rows: rows of pass B's tile (its LDS layout decides which in-block gathers collide), 0 = pass A
void build_spin_tiles(const SpinOp& op, const std::vector<uint32_t>& map, int lowbits, int chunk, const std::vector<uint32_t>* vcol,
                      int sort_mode, bool sorted_out, int tile_rows, bool spread_banks, SpinTiles& t, HostTiles& h) {
  const int dim = op.dim;
  h.start.clear();
  if (!map.empty()) {
    uint32_t prev = 0xFFFFFFFFu;
    for (int i = 0; i < dim; ++i) {
      uint32_t hi = lowbits >= 32 ? 0u : (map[i] >> lowbits);
      if (hi != prev) {
        h.start.push_back((uint32_t)i);
        prev = hi;
      }
    }
  } else {
    for (int i = 0; i < dim; i += chunk) h.start.push_back((uint32_t)i);
  }
  h.start.push_back((uint32_t)dim);
  t.start = h.start;
  t.lowbits = lowbits;
  t.nblocks = (int)h.start.size() - 1;
  t.max_block = 0;
  std::vector<uint32_t> block_of(dim);
  for (int k = 0; k < t.nblocks; ++k) {
    t.max_block = std::max<int>(t.max_block, (int)(h.start[k + 1] - h.start[k]));
    for (uint32_t i = h.start[k]; i < h.start[k + 1]; ++i) block_of[i] = (uint32_t)k;
  }
  std::vector<int> cin(dim, 0), cout(dim, 0);
  t.n_in = t.n_out = 0;
  for (int i = 0; i < dim; ++i) {
    for (int64_t p = op.rowptr[i]; p < op.rowptr[i + 1]; ++p) (block_of[op.cols[p]] == block_of[i] ? cin[i] : cout[i])++;
    t.n_in += cin[i];
    t.n_out += cout[i];
  }
  const int kin = *std::max_element(cin.begin(), cin.end()), kout = *std::max_element(cout.begin(), cout.end());
  auto pad4 = [](int k) { return std::max(HOP_CHUNK, (k + HOP_CHUNK - 1) / HOP_CHUNK * HOP_CHUNK); };
  // one extra all-empty chunk on the natural-order outer table terminates its "all lanes empty" loop
  t.k_in = pad4(kin);
  t.k_in_real = kin;
  t.k_out = pad4(kout) + (sorted_out ? 0 : HOP_CHUNK);
  // visiting order inside each block
  h.perm.resize(dim);
  std::iota(h.perm.begin(), h.perm.end(), 0u);
  for (int k = 0; k < t.nblocks; ++k) {
    auto first = h.perm.begin() + h.start[k], last = h.perm.begin() + h.start[k + 1];
    if (sort_mode == 1)
      std::stable_sort(first, last, [&](uint32_t a, uint32_t b) { return cin[a] > cin[b]; });
    else if (sort_mode == 2)
      std::stable_sort(first, last, [&](uint32_t a, uint32_t b) { return cout[a] != cout[b] ? cout[a] > cout[b] : cin[a] > cin[b]; });
  }
  // 64-position groups and their loop bounds
  h.gstart.assign(t.nblocks + 1, 0);
  h.gmax.clear();
  double sl_in = 0, sl_out = 0;
  for (int k = 0; k < t.nblocks; ++k) {
    h.gstart[k] = (uint32_t)h.gmax.size();
    for (uint32_t a = h.start[k]; a < h.start[k + 1]; a += 64) {
      int mi = 0, mo = 0;
      const uint32_t bnd = std::min<uint32_t>(a + 64, h.start[k + 1]);
      for (uint32_t q = a; q < bnd; ++q) {
        mi = std::max(mi, cin[h.perm[q]]);
        mo = std::max(mo, cout[h.perm[q]]);
      }
      h.gmax.push_back((uint32_t)mi | ((uint32_t)mo << 16));
      sl_in += (double)(bnd - a) * ((mi + HOP_CHUNK - 1) / HOP_CHUNK * HOP_CHUNK);
      sl_out += (double)(bnd - a) * ((mo + HOP_CHUNK - 1) / HOP_CHUNK * HOP_CHUNK);
    }
  }
  h.gstart[t.nblocks] = (uint32_t)h.gmax.size();
  t.slots_in = sl_in / dim;
  t.slots_out = sl_out / dim;
  // tables
  const uint32_t emptyz = (uint32_t)(2 * op.coef.size()) << TILE_COEF_SHIFT;
  h.ell_in.assign((size_t)t.k_in * dim, emptyz);
  std::vector<uint32_t> pos_of(dim);
  for (int q = 0; q < dim; ++q) pos_of[h.perm[q]] = (uint32_t)q;
  for (int i = 0; i < dim; ++i) {
    int a = 0, b = 0, k = 0;
    const size_t qi = pos_of[i];
    for (int64_t p = op.rowptr[i]; p < op.rowptr[i + 1]; ++p, ++k) {
      const uint32_t e = op.ell[(size_t)k * dim + i];  // same order as the CSR row
      const uint32_t src = e & ELL_SRC_MASK;
      const uint32_t ci = (2u * ((e >> ELL_SRC_BITS) & ELL_COEF_MASK) + (e >> 31)) << TILE_COEF_SHIFT;
      if (block_of[src] == block_of[i])
        h.ell_in[(size_t)(a++) * dim + qi] = ci | (src - h.start[block_of[i]]);
      else
        ++b;  // out-of-block entry: handled by the structured part below
    }
  }
  // LDS bank spreading.  Slot k of a wave's 64 rows mixes different hops, so its 64 gather addresses are close to random: a
  // ds_read_b128 is served 16 lanes at a time and 16 random 16-byte bank quads collide (scripts/lds_conflicts.py: half of the
  // in-block LDS cycles are conflicts; 0.68 for uniformly random addresses).  The ORDER of a row's hops is free, so each wave's
  // lists are re-dealt slot by slot: every lane takes, among its remaining hops, the one whose bank unit is least loaded within
  // its lane group (MI355X_MICROARCH.md, LDS), for the complex-vector layouts (ds_read_b128: groups {0-3,12-15,20-27},
  // {4-11,16-19,28-31} and the same +32; 16 quads of 16 bytes; pass A lds[column*n + row] -> row & 15; pass B [column][R rows]
  // with the row position XOR-swizzled).  (real_layout: ds_read_b64, groups {0-31}, {32-63}, 32 pairs of 8 bytes -- see below.)
  auto spread = [&](std::vector<uint32_t>& tab, bool real_layout) {
    auto lane_group = [&](int l) -> int {
      if (real_layout) return l >> 5;
      const int m = l & 31;
      return ((l >> 5) << 1) | (((m >= 4 && m < 12) || (m >= 16 && m < 20) || m >= 28) ? 1 : 0);
    };
    const int units = real_layout ? 32 : 16;
    const int rows = real_layout ? 8 : tile_rows;  // (real vectors run pass B on 8-row tiles of doubles)
    int lsw = 2, rmask = 3, lr = 2;
    if (rows == 2) lsw = 3, rmask = 1, lr = 1;
    if (rows == 8) lsw = real_layout ? 2 : 1, rmask = 7, lr = 3;
    auto unit = [&](uint32_t off) -> int {
      return tile_rows == 0 ? (int)(off & (uint32_t)(units - 1)) : (int)(((off << lr) + ((off >> lsw) & (uint32_t)rmask)) & (uint32_t)(units - 1));
    };
    std::vector<std::vector<uint32_t>> rem(64);
    for (int k = 0; k < t.nblocks; ++k) {
      for (uint32_t a = h.start[k]; a < h.start[k + 1]; a += 64) {
        const int nl = (int)std::min<uint32_t>(64, h.start[k + 1] - a);
        int kmax = 0;
        for (int l = 0; l < nl; ++l) {
          rem[l].clear();
          const int cnt = cin[h.perm[a + l]];
          for (int q = 0; q < cnt; ++q) rem[l].push_back(tab[(size_t)q * dim + a + l]);
          kmax = std::max(kmax, cnt);
        }
        for (int slot = 0; slot < kmax; ++slot) {
          int load[4][32] = {{0}};
          uint32_t seen[4][32][4];  // distinct addresses already on a unit (identical ones broadcast): the first few are enough
          for (int l = 0; l < nl; ++l) {
            if (rem[l].empty()) continue;
            const int g = lane_group(l);
            size_t best = 0;
            int best_load = 1 << 30;
            for (size_t j = 0; j < rem[l].size(); ++j) {
              const uint32_t off = rem[l][j] & TILE_OFF_MASK;
              const int qd = unit(off);
              int ld = load[g][qd];
              for (int u = 0; u < std::min(ld, 4); ++u)
                if (seen[g][qd][u] == off) {
                  ld = -1;  // the same address is already being read: a broadcast, free
                  break;
                }
              if (ld < best_load) best_load = ld, best = j;
            }
            const uint32_t w = rem[l][best];
            rem[l].erase(rem[l].begin() + (long)best);
            tab[(size_t)slot * dim + a + l] = w;
            const uint32_t off = w & TILE_OFF_MASK;
            const int qd = unit(off);
            if (best_load >= 0) {
              if (load[g][qd] < 4) seen[g][qd][load[g][qd]] = off;
              ++load[g][qd];
            }
          }
        }
      }
    }
  };
  // (A second deal for the real-vector layouts was built and measured: no gain on the real-vector Lanczos iteration, 4.005 against
  //  4.007 ms at C3, and it costs the bit-for-bit agreement of the real product with the real part of the complex one.  One table.)
  if (spread_banks) spread(h.ell_in, false);
  // Half-size copy of the in-block table: two hops per word, each (coefficient index << p16_bits) | offset, whenever the
  // block offsets and the signed-coefficient indices fit 16 bits together (C3: 10 + 3 bits; blocks of 14 low orbitals: 12 + 3).
  // Half the table bytes to keep in L2 and half the loads; the 32-bit table stays for the job kernel, which packs its words once per job.
  h.ell16.clear();
  t.p16_bits = 0;
  {
    int ob = 1, cb = 1;
    while ((1 << ob) < t.max_block) ++ob;
    while ((1u << cb) < 2 * op.coef.size() + 1) ++cb;
    if (ob + cb <= 16 && t.k_in % 2 == 0) {
      t.p16_bits = std::max(ob, 10);  // (10 when it fits: the split the kernels were tuned with)
      if (t.p16_bits + cb > 16) t.p16_bits = ob;
      const uint32_t om = (1u << t.p16_bits) - 1u;
      auto pack = [&](const std::vector<uint32_t>& src, std::vector<uint32_t>& dst) {
        dst.assign((size_t)(t.k_in / 2) * dim, 0u);
        for (int a = 0; a < t.k_in; ++a)
          for (int q = 0; q < dim; ++q) {
            const uint32_t e = src[(size_t)a * dim + q];
            const uint32_t half = ((e >> TILE_COEF_SHIFT) << t.p16_bits) | (e & om);
            dst[(size_t)(a / 2) * dim + q] |= half << (16 * (a & 1));
          }
      };
      pack(h.ell_in, h.ell16);
    }
  }
  // Blocks whose high orbitals hold the same NUMBER of particles contain the same low-orbital patterns, and hops among the
  // low orbitals see nothing else: their in-block tables (words, visiting order, loop bounds) are identical.  Every block
  // reads the tables of the first block that equals it (compared, not assumed), so the hot table set shrinks from one
  // slice per block to one per class (C3: 5 instead of 16) and stays in L2 between the workgroups that use it.
  h.tstart.assign(t.nblocks, 0);
  t.table_classes = 0;
  for (int k = 0; k < t.nblocks; ++k) {
    h.tstart[k] = h.start[k];
    const uint32_t nk = h.start[k + 1] - h.start[k];
    bool found = false;
    for (int c = 0; c < k && !found; ++c) {
      if (h.tstart[c] != h.start[c] || h.start[c + 1] - h.start[c] != nk) continue;  // (only first-of-class blocks are candidates)
      bool same = true;
      for (uint32_t q = 0; q < nk && same; ++q) same = h.perm[h.start[k] + q] - h.start[k] == h.perm[h.start[c] + q] - h.start[c];
      const uint32_t ng = (nk + 63) / 64;
      for (uint32_t q = 0; q < ng && same; ++q) same = (h.gmax[h.gstart[k] + q] & 0xFFFFu) == (h.gmax[h.gstart[c] + q] & 0xFFFFu);  // (the kernels read the in-block bound only)
      for (int a = 0; a < t.k_in && same; ++a)
        same = std::equal(h.ell_in.begin() + (size_t)a * dim + h.start[k], h.ell_in.begin() + (size_t)a * dim + h.start[k] + nk,
                          h.ell_in.begin() + (size_t)a * dim + h.start[c]);
      if (same) {
        h.tstart[k] = h.start[c];
        h.gstart[k] = h.gstart[c];
        found = true;
      }
    }
    if (!found) ++t.table_classes;
  }
  // ---- structured out-of-block part: group by (block, source block)
  h.bh_ptr.assign(t.nblocks + 1, 0);
  h.rs_ptr.assign(t.nblocks + 1, 0);
  h.bh.clear();
  h.rs_off.clear();
  h.rs_tab.clear();
  h.rs_base.clear();
  h.rs_neg.clear();
  double bh_rows = 0, rs_rows = 0;
  struct Ent {
    uint32_t off, src, ci;
  };
  std::vector<std::vector<Ent>> by_src(t.nblocks);
  std::vector<int> touched;
  for (int k = 0; k < t.nblocks; ++k) {
    const uint32_t b0 = h.start[k], nb = h.start[k + 1] - b0;
    touched.clear();
    for (uint32_t i = b0; i < b0 + nb; ++i) {
      int kk = 0;
      for (int64_t p = op.rowptr[i]; p < op.rowptr[i + 1]; ++p, ++kk) {
        const uint32_t e = op.ell[(size_t)kk * dim + i];
        const uint32_t src = e & ELL_SRC_MASK;
        const uint32_t sb = block_of[src];
        if ((int)sb == k) continue;
        if (by_src[sb].empty()) touched.push_back((int)sb);
        by_src[sb].push_back({i - b0, src, 2u * ((e >> ELL_SRC_BITS) & ELL_COEF_MASK) + (e >> 31)});
      }
    }
    std::sort(touched.begin(), touched.end());
    h.bh_ptr[k] = (uint32_t)(h.bh.size() / 2);
    h.rs_ptr[k] = (uint32_t)h.rs_off.size();
    for (int sb : touched) {
      auto& ents = by_src[sb];
      const uint32_t s0 = h.start[sb], ns = h.start[sb + 1] - s0;
      bool uniform = ents.size() == nb && ns == nb;
      if (uniform)
        for (const Ent& en : ents)
          if (en.src - s0 != en.off || en.ci != ents[0].ci) {
            uniform = false;
            break;
          }
      // the source run must also be contiguous in the (possibly padded) gather layout
      if (uniform && vcol)
        for (uint32_t q = 0; q < ns; ++q)
          if ((*vcol)[s0 + q] != (*vcol)[s0] + q) {
            uniform = false;
            break;
          }
      if (uniform) {
        h.bh.push_back(vcol ? (*vcol)[s0] : s0);
        h.bh.push_back(ents[0].ci);
        bh_rows += nb;
      } else {
        // as many slots as the busiest row has entries from this source block
        std::vector<int> mult(nb, 0);
        int nsl = 0;
        for (const Ent& en : ents) nsl = std::max(nsl, ++mult[en.off]);
        // Table words hold the source RELATIVE to its block when the block's slots are contiguous in the gather layout
        // (always on an unsplit sector); the slot's base is a per-slot constant.  Relative tables repeat between blocks of
        // the same class up to one overall sign, and are shared below.
        bool contig = true;
        if (vcol)
          for (uint32_t q = 0; q < ns && contig; ++q) contig = (*vcol)[s0 + q] == (*vcol)[s0] + q;
        const size_t base = h.rs_tab.size();
        h.rs_tab.resize(base + (size_t)nsl * nb, emptyz);
        std::fill(mult.begin(), mult.end(), 0);
        for (const Ent& en : ents) {
          const int sl = mult[en.off]++;
          const uint32_t where = contig ? en.src - s0 : (vcol ? (*vcol)[en.src] : en.src);
          h.rs_tab[base + (size_t)sl * nb + en.off] = (en.ci << TILE_COEF_SHIFT) | where;
        }
        for (int sl = 0; sl < nsl; ++sl) {
          h.rs_off.push_back((uint32_t)(base + (size_t)sl * nb));
          h.rs_base.push_back(contig ? (vcol ? (*vcol)[s0] : s0) : 0u);
          h.rs_neg.push_back(0u);
        }
        rs_rows += (double)nsl * nb;
      }
      ents.clear();
    }
  }
  h.bh_ptr[t.nblocks] = (uint32_t)(h.bh.size() / 2);
  h.rs_ptr[t.nblocks] = (uint32_t)h.rs_off.size();
  // share row-slot tables that are equal up to an overall sign (a coefficient index is 2*amplitude + sign): a slot reads
  // the first table that equals its own after normalising the sign of its first entry, and negates if it had to flip
  {
    std::map<std::vector<uint32_t>, uint32_t> seen;  // normalised table -> offset of its first copy
    t.rs_tables = 0;
    for (int k = 0; k < t.nblocks; ++k) {
      const uint32_t nb = h.start[k + 1] - h.start[k];
      for (uint32_t sl = h.rs_ptr[k]; sl < h.rs_ptr[k + 1]; ++sl) {
        std::vector<uint32_t> tab(h.rs_tab.begin() + h.rs_off[sl], h.rs_tab.begin() + h.rs_off[sl] + nb);
        uint32_t flip = 0;
        for (uint32_t w : tab)
          if (w != emptyz) {
            flip = (w >> TILE_COEF_SHIFT) & 1u;
            break;
          }
        if (flip)
          for (uint32_t& w : tab)
            if (w != emptyz) w ^= 1u << TILE_COEF_SHIFT;
        auto it = seen.find(tab);
        if (it == seen.end()) {
          std::copy(tab.begin(), tab.end(), h.rs_tab.begin() + h.rs_off[sl]);  // stored normalised
          seen.emplace(std::move(tab), h.rs_off[sl]);
          ++t.rs_tables;
        } else {
          h.rs_off[sl] = it->second;
        }
        h.rs_neg[sl] = flip;
      }
    }
  }
  // Half-size row-slot tables, two slots of a block per 32-bit word ((coefficient index << p16_bits) | source offset relative to
  // the slot's block, the split of the half-size in-block table): half the table loads and bytes of the out-of-block phases.
  // Needs block-relative words (contiguous gather slots: always on an unsplit sector).  Pairs are shared like the single tables.
  h.rs16.clear();
  h.rs16_off.assign(h.rs_off.size(), 0u);
  if (t.p16_bits > 0 && !h.rs_off.empty()) {
    bool ok = true;
    // (absolute words appear only with non-contiguous gather slots: their offsets do not fit the field)
    const uint32_t om = (1u << t.p16_bits) - 1u;
    for (int k = 0; k < t.nblocks && ok; ++k) {
      const uint32_t nb = h.start[k + 1] - h.start[k];
      for (uint32_t sl = h.rs_ptr[k]; sl < h.rs_ptr[k + 1] && ok; ++sl)
        for (uint32_t q = 0; q < nb && ok; ++q) ok = (h.rs_tab[h.rs_off[sl] + q] & TILE_OFF_MASK) <= om;
    }
    if (ok) {
      std::map<std::pair<uint32_t, uint32_t>, uint32_t> seen;  // (table of the first slot, table of the second slot or ~0) -> packed offset
      const uint32_t e16 = (uint32_t)(2 * op.coef.size()) << t.p16_bits;  // empty slot: (offset 0, zero coefficient)
      for (int k = 0; k < t.nblocks; ++k) {
        const uint32_t nb = h.start[k + 1] - h.start[k];
        for (uint32_t sl = h.rs_ptr[k]; sl < h.rs_ptr[k + 1]; sl += 2) {
          const bool two = sl + 1 < h.rs_ptr[k + 1];
          const auto key = std::make_pair(h.rs_off[sl], two ? h.rs_off[sl + 1] : 0xFFFFFFFFu);
          auto it = seen.find(key);
          if (it == seen.end()) {
            const uint32_t base = (uint32_t)h.rs16.size();
            h.rs16.resize(base + nb);
            for (uint32_t q = 0; q < nb; ++q) {
              const uint32_t a = h.rs_tab[h.rs_off[sl] + q];
              const uint32_t lo = ((a >> TILE_COEF_SHIFT) << t.p16_bits) | (a & om);
              uint32_t hi = e16;
              if (two) {
                const uint32_t bb = h.rs_tab[h.rs_off[sl + 1] + q];
                hi = ((bb >> TILE_COEF_SHIFT) << t.p16_bits) | (bb & om);
              }
              h.rs16[base + q] = lo | (hi << 16);
            }
            it = seen.emplace(key, base).first;
          }
          h.rs16_off[sl] = it->second;
        }
      }
    }
  }
  t.rs16_on = !h.rs16.empty();
  if (h.rs16.empty()) h.rs16.assign(1, 0u);
  if (h.rs16_off.empty()) h.rs16_off.assign(1, 0u);
  if (h.bh.empty()) h.bh.assign(2, 0);
  if (h.rs_off.empty()) {
    h.rs_off.assign(1, 0);
    h.rs_base.assign(1, 0);
    h.rs_neg.assign(1, 0);
  }
  if (h.rs_tab.empty()) h.rs_tab.assign(1, emptyz);
  t.bh_per_row = bh_rows / dim;
  t.rs_per_row = rs_rows / dim;
  t.max_outer = 0;
  for (int k = 0; k < t.nblocks; ++k) {
    t.max_outer = std::max<int>(t.max_outer, (int)(h.bh_ptr[k + 1] - h.bh_ptr[k]) + (int)(h.rs_ptr[k + 1] - h.rs_ptr[k]));
  }
  h.order.resize(t.nblocks);
  std::iota(h.order.begin(), h.order.end(), 0u);
  std::stable_sort(h.order.begin(), h.order.end(),
                   [&](uint32_t a, uint32_t b) { return h.start[a + 1] - h.start[a] > h.start[b + 1] - h.start[b]; });
  // Dispatch order of the tile kernels for LARGE sectors (TileOptions::block_order): blocks by the particle number of their high
  // orbitals, natural order inside.  Two blocks are coupled when their high patterns differ by one hop among the high orbitals (same
  // particle number) or by one particle (a hop between a low and a high orbital): sorting the patterns of a hypercube by weight, then by
  // value, is the order that keeps every such pair closest (Harper), and blocks of one weight share their in-block tables anyway.
  h.order_pc.resize(t.nblocks);
  std::iota(h.order_pc.begin(), h.order_pc.end(), 0u);
  if (!map.empty() && lowbits < 32)
    std::stable_sort(h.order_pc.begin(), h.order_pc.end(), [&](uint32_t a, uint32_t b) {
      return __builtin_popcount(map[h.start[a]] >> lowbits) < __builtin_popcount(map[h.start[b]] >> lowbits);
    });
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel and size, not per launch (small sectors are launch-bound)
hipError_t allow_dynamic_lds(const void* kern, int bytes) {
  static std::mutex mu;
  static std::map<const void*, int> granted;
  std::lock_guard<std::mutex> lk(mu);
  int& g = granted[kern];
  if (bytes <= g) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) g = bytes;
  return e;
}

// the spH0nd block is folded into pass A when its move tables exist (hxv_sector.cpp builds them up to 32 M entries per spin)
static bool fold_nd(const DevSector& s) { return nd_folds(s); }

template <int C, int LZ, typename VT>
hipError_t launch_up_lz(const DevSector& s, const DevTiles& t, int lds_bytes, int threads, bool norb1, int wc, const VT* v,
                        const VT* wt, VT* hv, const LzEpilogue& lz, hipStream_t st) {
  const int ngroups = (s.qdw + C - 1) / C;
  const int gpx = (ngroups + 7) / 8;
  const int64_t nwg = (int64_t)gpx * 8 * t.nblocks;
  void (*kern)(DevSector, DevTiles, const VT*, const VT*, VT*, int, int, int, LzEpilogue) = nullptr;
  const bool p16 = t.ell16 != nullptr;  // (the half-size in-block table exists)
  if constexpr (LZ == 2) {
    // paired epilogue: real H on complex vectors only
    if constexpr (std::is_same<VT, double2>::value && C <= 4) {
      if (!s.real_h) return hipErrorInvalidValue;
      if (p16)
        kern = norb1 ? hxv_pass_up<C, true, true, 2, true, double2> : hxv_pass_up<C, true, false, 2, true, double2>;
      else
        kern = norb1 ? hxv_pass_up<C, true, true, 2, false, double2> : hxv_pass_up<C, true, false, 2, false, double2>;
    } else {
      return hipErrorInvalidValue;
    }
  } else if constexpr (std::is_same<VT, double>::value) {  // real vectors exist for real H only
    if (p16)
      kern = norb1 ? hxv_pass_up<C, true, true, LZ, true, double> : hxv_pass_up<C, true, false, LZ, true, double>;
    else
      kern = norb1 ? hxv_pass_up<C, true, true, LZ, false, double> : hxv_pass_up<C, true, false, LZ, false, double>;
  } else if (s.real_h) {
    if (p16)
      kern = norb1 ? hxv_pass_up<C, true, true, LZ, true, double2> : hxv_pass_up<C, true, false, LZ, true, double2>;
    else
      kern = norb1 ? hxv_pass_up<C, true, true, LZ, false, double2> : hxv_pass_up<C, true, false, LZ, false, double2>;
  } else {
    if (p16)
      kern = norb1 ? hxv_pass_up<C, false, true, LZ, true, double2> : hxv_pass_up<C, false, false, LZ, true, double2>;
    else
      kern = norb1 ? hxv_pass_up<C, false, true, LZ, false, double2> : hxv_pass_up<C, false, false, LZ, false, double2>;
  }
  if constexpr (std::is_same<VT, double2>::value && C <= 4 && LZ != 2) {
    if (fold_nd(s) && !norb1) {  // the spH0nd block rides along (Norb > 1: never the one-orbital diagonal)
      if (s.real_h)
        kern = p16 ? hxv_pass_up<C, true, false, LZ, true, double2, true> : hxv_pass_up<C, true, false, LZ, false, double2, true>;
      else
        kern = p16 ? hxv_pass_up<C, false, false, LZ, true, double2, true> : hxv_pass_up<C, false, false, LZ, false, double2, true>;
    }
  }
  lds_bytes = std::max(lds_bytes, 32 * 8);  // the epilogue reduces through LDS: 16 (+16 paired) doubles
  hipError_t e = allow_dynamic_lds((const void*)kern, lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(threads), (size_t)lds_bytes, st, s, t, v, wt, hv, ngroups, gpx, wc, lz);
  return hipGetLastError();
}

template <int C, typename VT>
hipError_t launch_up(const DevSector& s, const DevTiles& t, int lds_bytes, int threads, bool norb1, int wc, const VT* v,
                     const VT* wt, VT* hv, const LzEpilogue* lz, hipStream_t st) {
  if (lz && lz->pair) return launch_up_lz<C, 2, VT>(s, t, lds_bytes, threads, norb1, wc, v, wt, hv, *lz, st);
  if (lz) return launch_up_lz<C, 1, VT>(s, t, lds_bytes, threads, norb1, wc, v, wt, hv, *lz, st);
  return launch_up_lz<C, 0, VT>(s, t, lds_bytes, threads, norb1, wc, v, wt, hv, LzEpilogue(), st);
}

template <int R, int NP, typename VT>
hipError_t launch_dw_np(const DevSector& s, const DevTiles& t, int lds_bytes, int threads, int wc, const VT* v, VT* hv,
                        hipStream_t st) {
  const int ngroups = (s.dimup + R - 1) / R;
  const int gpx = (ngroups + 7) / 8;
  const int64_t nwg = (int64_t)((gpx + 1) & ~1) * 8 * t.nblocks;  // (an even number of row groups per XCD: DevTiles::pair_rows)
  void (*kern)(DevSector, DevTiles, const VT*, VT*, int, int, int);
  const bool p16 = t.ell16 != nullptr;  // (the half-size in-block table exists: blocks <= 1024 columns, <= 64 signed coefficients)
  if constexpr (std::is_same<VT, double>::value)
    kern = p16 ? hxv_pass_dw<R, NP, true, true, double> : hxv_pass_dw<R, NP, true, false, double>;
  else if (s.real_h)
    kern = p16 ? hxv_pass_dw<R, NP, true, true, double2> : hxv_pass_dw<R, NP, true, false, double2>;
  else
    kern = p16 ? hxv_pass_dw<R, NP, false, true, double2> : hxv_pass_dw<R, NP, false, false, double2>;
  hipError_t e = allow_dynamic_lds((const void*)kern, lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(threads), (size_t)lds_bytes, st, s, t, v, hv, ngroups, gpx, wc);
  return hipGetLastError();
}

template <int R, typename VT>
hipError_t launch_dw(const DevSector& s, const DevTiles& t, int max_block, int lds_bytes, int threads, int wc, const VT* v,
                     VT* hv, hipStream_t st) {
  const int np = (max_block * R + threads - 1) / threads;  // <= R because max_block <= threads
  if (np <= 1) return launch_dw_np<R, 1, VT>(s, t, lds_bytes, threads, wc, v, hv, st);
  if (np <= 2) return launch_dw_np<R, 2, VT>(s, t, lds_bytes, threads, wc, v, hv, st);
  if (np <= 4) return launch_dw_np<R, 4, VT>(s, t, lds_bytes, threads, wc, v, hv, st);
  return launch_dw_np<R, 8, VT>(s, t, lds_bytes, threads, wc, v, hv, st);
}

std::vector<double2> signed_coefs(const SpinOp& op) {
  std::vector<double2> sc(2 * op.coef.size() + 1);
  for (size_t i = 0; i < op.coef.size(); ++i) {
    sc[2 * i] = make_double2(op.coef[i].real(), op.coef[i].imag());
    sc[2 * i + 1] = make_double2(-op.coef[i].real(), -op.coef[i].imag());
  }
  sc.back() = make_double2(0.0, 0.0);
  return sc;
}

}  // namespace

// the block bits pass A gets from the DEFAULT options (what make_tile_plan's `build` computes for the up spin): the device row order of a
// sector is chosen for them once, when the sector is built, and does not follow later option changes (device vectors outlive those)
int default_lowbits_up(int ns, int npart, int ncoef) {
  const TileOptions o;
  const int budget = o.lds_budget_kb_up * 1024 - 16 * (2 * ncoef + 1);
  return o.force_bits_up >= 0 ? std::min(o.force_bits_up, ns) : choose_lowbits(ns, npart, o.cols_per_tile, budget, o.threads_up);
}

std::string make_tile_plan(const SectorHost& s, TilePlan& plan, const PlanUploader& up) {
  TileOptions& o = plan.opt;
  if (o.cols_per_tile != 2 && o.cols_per_tile != 4 && o.cols_per_tile != 8) return "cols_per_tile must be 2, 4 or 8";
  // Large sectors (Ns=18): two neighbouring 4-row panels of DimDw columns are 128 B x DimDw = 6 MB of lines, more than an XCD's
  // L2, and nearly every out-of-block gather of pass B then leaves the XCD: eight rows per tile (whole lines; smaller blocks
  // to stay within the LDS budget) measured 40.8 ms against 47.4 ms per pass B there (Ns=16: 2.25 against 2.22 ms).
  if (o.rows_per_tile == 0) o.rows_per_tile = (int64_t)128 * s.dimdw > ((int64_t)4 << 20) ? 8 : 4;
  if (o.rows_per_tile != 2 && o.rows_per_tile != 4 && o.rows_per_tile != 8) return "rows_per_tile must be 0 (automatic), 2, 4 or 8";
  if (o.lds_budget_kb_up < 8 || o.lds_budget_kb_up > 144 || o.lds_budget_kb_dw < 8 || o.lds_budget_kb_dw > 144)
    return "lds_budget_kb must be in [8,144]";
  for (int th : {o.threads_up, o.threads_dw})
    if (th != 256 && th != 512 && th != 1024) return "threads must be 256, 512 or 1024";
  if (o.sort_mode < 0 || o.sort_mode > 2) return "sort_mode must be 0, 1 or 2";
  if (o.wt_cols != 2 && o.wt_cols != 4 && o.wt_cols != 8 && o.wt_cols != 16) return "wt_cols must be 2, 4, 8 or 16";
  if (o.job_cols != 1 && o.job_cols != 2) return "job_cols must be 1 or 2";
  if (o.job_groups < 1 || o.job_groups > 65536) return "job_groups must be in [1,65536]";
  if (o.job_stages < 2 || o.job_stages > 8) return "job_stages must be in [2,8]";
  plan.ncoef_up = (int)s.dev_up().coef.size();
  plan.ncoef_dw = (int)s.dw.coef.size();
  plan.usable = plan.ncoef_up <= TILE_MAX_COEF && plan.ncoef_dw <= TILE_MAX_COEF;
  if (!plan.usable) return "";
  // the two spins' tables are independent: built on one host thread each, handed to the uploader afterwards (it is not re-entrant)
  HostTiles hu, hd;
  auto build = [&](const SpinOp& op, const std::vector<uint32_t>& map, int npart, int width, int force, int budget_kb, int max_block,
                   const std::vector<uint32_t>* vcol, bool sorted_out, int sort_mode, SpinTiles& t, HostTiles& h) -> std::string {
    const int tile_rows = sorted_out ? 0 : width;
    const int budget = budget_kb * 1024 - 16 * (2 * (int)op.coef.size() + 1);
    int L = 32, chunk = std::max(1, std::min(budget / (16 * width), max_block));
    if (!map.empty()) L = force >= 0 ? std::min(force, s.ns) : choose_lowbits(s.ns, npart, width, budget, max_block);
    build_spin_tiles(op, map, L, chunk, vcol, sort_mode, sorted_out, tile_rows, o.spread_banks != 0, t, h);
    if ((int64_t)t.max_block * width * 16 + 16 * (2 * (int64_t)op.coef.size() + 1) > 160 * 1024) return "tile does not fit the 160 KB LDS";
    if (t.max_block > max_block) return "block larger than the workgroup (one thread per block row/column)";
    return "";
  };
  auto send = [&](HostTiles& h, SpinTiles& t) -> std::string {
    if (up.u32(h.start, &t.d_start) != hipSuccess || up.u32(h.perm, &t.d_perm) != hipSuccess ||
        up.u32(h.gstart, &t.d_gstart) != hipSuccess || up.u32(h.gmax, &t.d_gmax) != hipSuccess ||
        up.u32(h.ell_in, &t.d_ell_in) != hipSuccess || up.u32(h.tstart, &t.d_tstart) != hipSuccess ||
        (!h.ell16.empty() && up.u32(h.ell16, &t.d_ell16) != hipSuccess) ||
        up.u32(h.bh_ptr, &t.d_bh_ptr) != hipSuccess || up.u32(h.bh, &t.d_bh) != hipSuccess ||
        up.u32(h.rs_ptr, &t.d_rs_ptr) != hipSuccess || up.u32(h.rs_off, &t.d_rs_off) != hipSuccess ||
        up.u32(h.rs_tab, &t.d_rs_tab) != hipSuccess || up.u32(h.order, &t.d_order) != hipSuccess || up.u32(h.order_pc, &t.d_order_pc) != hipSuccess ||
        up.u32(h.rs_base, &t.d_rs_base) != hipSuccess || up.u32(h.rs_neg, &t.d_rs_neg) != hipSuccess ||
        up.u32(h.rs16, &t.d_rs16) != hipSuccess || up.u32(h.rs16_off, &t.d_rs16_off) != hipSuccess)
      return "upload of tile tables failed";
    return "";
  };
  static const std::vector<uint32_t> no_map;  // panel handles have no up basis: plain index chunks (pass A never runs)
  std::string e_up, e;
  GuardedThread th;   // (an exception on either side comes back as an error string; the thread is joined on every path -- ADVICE r5)
  th.run([&] {
    e_up = build(s.dev_up(), s.panel_rows > 0 ? no_map : s.dev_key_up(), s.nup, o.cols_per_tile, o.force_bits_up, o.lds_budget_kb_up, o.threads_up, nullptr, true, 0,
                 plan.up, hu);
  });
  // pass B sorts by the inner count only: its outer table is read in natural column order
  try {
    e = build(s.dw, s.map_dw, s.ndw, o.rows_per_tile, o.force_bits_dw, o.lds_budget_kb_dw, o.threads_dw, &s.vcol, false, o.sort_mode_dw ? 1 : 0, plan.dw,
              hd);
  } catch (const std::exception& ex) {
    e = std::string("tile plan of H_dw: ") + ex.what();
  }
  th.join();
  if (!th.err.empty()) return "tile plan of H_up: " + th.err;
  if (!e_up.empty()) return e_up;
  if (!e.empty()) return e;
  e = send(hu, plan.up);
  if (!e.empty()) return e;
  e = send(hd, plan.dw);
  if (!e.empty()) return e;
  if (up.d2(signed_coefs(s.dev_up()), &plan.d_scoef_up) != hipSuccess || up.d2(signed_coefs(s.dw), &plan.d_scoef_dw) != hipSuccess)
    return "upload of coefficient tables failed";
  return "";
}

int64_t tiled_wt_elems(const DevSector& s, const TilePlan& plan) {
  const int wc = std::max(plan.opt.cols_per_tile, plan.opt.wt_cols);
  return (int64_t)((s.qdw + wc - 1) / wc) * wc * ((s.dimup + 15) & ~15);  // (whole patches of up to 8 complex / 16 real rows per group)
}

// Real-vector mode runs the same plans with twice the columns (pass A) / rows (pass B) per tile: the same tile bytes.
static int real_cols(const TilePlan& plan) { return std::min(8, 2 * plan.opt.cols_per_tile); }
// Complex vectors: at most four columns per pass-A tile.  The eight-column kernels need far more than the 64 registers two
// workgroups per CU leave (they spill 30 - 140 of them, scalar registers as well), were 35 % slower when they were measured, and
// one of them (complex H, Lanczos epilogue) came out of hipcc computing wrong sums after an unrelated two-pointer growth of
// the kernel arguments: "cols_per_tile" = 8 keeps its meaning for the plan (block size) and for real vectors only.
static int cplx_cols(const TilePlan& plan) { return std::min(4, plan.opt.cols_per_tile); }
static int real_rows(const TilePlan& plan) { return std::min(8, 2 * plan.opt.rows_per_tile); }
static int real_wc(const TilePlan& plan) { return std::max(real_cols(plan), std::min(16, 2 * plan.opt.wt_cols)); }

// Pass A runs as jobs (hxv_jobs.hip) when the plan allows it and the tile ring fits the LDS; wc_out = scratch group width.
static bool use_job_up(const DevSector& s, const TilePlan& plan, bool real_vec, bool lz, bool wt_natural, int* wc_out = nullptr) {
  if (real_vec || !plan.opt.job_up || plan.opt.sort_mode != 0 || plan.opt.debug != 0 || !job_up_usable(s, plan)) return false;
  if (s.nd.active) return false;  // (the spH0nd block rides on the one-tile-per-workgroup kernel only)
  // job_up = 2 (default): jobs for the fused Lanczos product only.  With the in-block tables shared between blocks of a class
  // the one-tile-per-workgroup kernel is 2 % faster for the plain product (2.22 against 2.27 ms at C3), the job kernel 2.5 %
  // faster with the Lanczos epilogue, whose second input vector it streams through the tile ring (6.29 against 6.45 ms).
  if (plan.opt.job_up == 2 && !lz) return false;
  int wc = wt_natural ? 0 : std::max(plan.opt.job_cols, plan.opt.wt_cols);
  if (!job_up_fits(s, plan, lz, wc)) {
    // the Lanczos epilogue streams a third vector through the tile ring: narrower scratch groups (two buffers of
    // wc columns each sit beside the ring) can make room for it
    if (wt_natural || wc <= 2 || plan.opt.job_cols > 2 || !job_up_fits(s, plan, lz, 2)) return false;
    wc = 2;
  }
  if (wc_out) *wc_out = wc;
  return true;
}

int64_t tiled_pass_up_workgroups(const DevSector& s, const TilePlan& plan, bool real_vec, bool pieces) {
  // (both epilogues run on the same kernel: jobs where they apply, else one tile per workgroup; a dw part handed over in pieces --
  //  exchange mode 2 -- always goes through the tile kernel)
  if (!pieces && use_job_up(s, plan, real_vec, true, false)) return job_up_workgroups(s, plan);
  const int C = real_vec ? real_cols(plan) : cplx_cols(plan);
  const int ngroups = (s.qdw + C - 1) / C;
  return (int64_t)((ngroups + 7) / 8) * 8 * plan.up.nblocks;
}

template <typename VT>
static hipError_t launch_tiled_vt(const DevSector& s, const TilePlan& plan, const VT* v, VT* wt, VT* hv, hipStream_t st, const LzEpilogue* lz,
                                  int only_pass, bool wt_natural, const WtRange* wtr = nullptr, int nwtr = 0) {
  constexpr bool RV = std::is_same<VT, double>::value;
  // dispatch order of a group's blocks (TileOptions::block_order).  Automatic: by the particle number of the high orbitals where table
  // classes are few (it IS a class order there, and it keeps coupled blocks close: Ns=18 fabric traffic 330 -> 261 GB per product,
  // 73.1 -> 70.1 ms; Ns=16 -0.8 %; profiles/r04_ab_block_order.log), natural otherwise (11 table sets for 16 blocks, C4: 2.6 % faster).
  auto pick_order = [&](const SpinTiles& t) -> const uint32_t* {
    int bo = plan.opt.block_order;
    if (plan.opt.debug & 32) bo = 2;
    if (bo < 0) bo = 2 * t.table_classes <= t.nblocks ? 1 : 2;
    return bo == 0 ? t.d_order : (bo == 1 ? t.d_order_pc : nullptr);
  };
  DevTiles tu{plan.up.d_start, plan.up.d_tstart, plan.up.d_perm, plan.up.d_gstart, plan.up.d_gmax, plan.up.d_ell_in, plan.up.d_ell16,
              plan.d_scoef_up, plan.up.d_bh_ptr, plan.up.d_bh, plan.up.d_rs_ptr, plan.up.d_rs_off, plan.up.d_rs_tab, plan.up.d_rs_base, plan.up.d_rs_neg,
              plan.up.nblocks, 2 * plan.ncoef_up + 1, plan.opt.debug, 0, pick_order(plan.up), plan.up.p16_bits, (plan.up.rs16_on && !(plan.opt.debug & 64)) ? plan.up.d_rs16 : nullptr, plan.up.d_rs16_off, wtr, nwtr};
  DevTiles td{plan.dw.d_start, plan.dw.d_tstart, plan.dw.d_perm, plan.dw.d_gstart, plan.dw.d_gmax, plan.dw.d_ell_in, plan.dw.d_ell16,
              plan.d_scoef_dw, plan.dw.d_bh_ptr, plan.dw.d_bh, plan.dw.d_rs_ptr, plan.dw.d_rs_off, plan.dw.d_rs_tab, plan.dw.d_rs_base, plan.dw.d_rs_neg,
              plan.dw.nblocks, 2 * plan.ncoef_dw + 1, plan.opt.debug, 0, pick_order(plan.dw), plan.dw.p16_bits, (plan.dw.rs16_on && !(plan.opt.debug & 64)) ? plan.dw.d_rs16 : nullptr, plan.dw.d_rs16_off, nullptr, 0};
  // (class order only where classes are few: with 11 table sets for 16 blocks (C4) the natural order measured 2.6 % faster)
  // (decided below, once the tile's row count R is known)
  const int C = RV ? real_cols(plan) : cplx_cols(plan), R = RV ? real_rows(plan) : plan.opt.rows_per_tile;
  // columns per group of the wt scratch; 0 = natural layout
  const int passes = only_pass ? only_pass : plan.opt.passes;
  {
    // (only tiles whose column segments are half lines have a neighbour to pair with)
    const int64_t panel_pair = (int64_t)2 * R * (int)sizeof(VT) * s.dimdw;  // bytes of the lines two neighbouring row groups share
    td.pair_rows = plan.opt.pair_rows < 0 ? ((R * (int)sizeof(VT) < 128 && panel_pair > ((int64_t)4 << 20)) ? 1 : 0) : plan.opt.pair_rows;
  }
  // (with the job kernels pass A's tile width no longer constrains the scratch layout)
  int wc = wt_natural ? 0 : (RV ? real_wc(plan) : std::max(C, plan.opt.wt_cols));
  bool job_a = false;
  if constexpr (!RV) job_a = (passes & 1) && !wtr && use_job_up(s, plan, false, lz != nullptr, wt_natural, &wc);  // (a dw part in pieces: tile kernel)
  // (folded spH0nd block: one table row index and 2*C partner-column words per ordered orbital pair of a site, behind the coefficients)
  const int lds_nd = nd_folds(s) ? s.nd.nlat * s.nd.norb * (s.nd.norb - 1) * (2 * C + 1) * 4 : 0;
  const int lds_a = std::max(plan.up.max_block * C * (int)sizeof(VT) + tu.nscoef * 16 + lds_nd, plan.opt.lds_min_kb_up * 1024);
  const int lds_b = std::max(plan.dw.max_block * R * (int)sizeof(VT) + ((td.nscoef * 16 + 255) & ~255), plan.opt.lds_min_kb_dw * 1024);
  const int ta = plan.opt.threads_up, tb = plan.opt.threads_dw;
  const bool norb1 = s.diag.mode == 0 && s.diag.cross.norb == 1;
  hipError_t e = hipSuccess;
  // REAL vectors: pass B treats rows as independent batch entries (the dw hops act on columns), so a real vector IS a complex vector of
  // DimUp/2 rows with real coefficients: the complex kernel does one table decode and one 16-byte LDS gather where the double kernel
  // does two of each (round 5: 1.23 -> ~1.0 ms at C3; option "real_dw_pairs").  With column-major patches the scratch it writes is the
  // real kernel's own layout (a pair of rows = two rows of a patch); natural-layout outputs (row panels of exchange mode 2) likewise.
  bool dw_pairs = false;
  if constexpr (RV) dw_pairs = plan.opt.real_dw_pairs && (passes & 2) && (s.pitch % 2 == 0);
  // blocked scratch: column-major patches for the tile kernels (pass A reads them with a quarter of the L1 accesses); the job kernel's
  // group buffers keep the row-major patches.  Row pairs need them (their scratch would otherwise interleave the two rows of a pair).
  const bool cm = wc > 0 && !job_a && (plan.opt.wt_colmajor || dw_pairs);
  const int wc_b = cm ? (wc | 0x100) : wc;
  if (passes & 2) {
    if constexpr (RV) {
      if (dw_pairs) {
        DevSector sp = s;
        sp.dimup = (s.dimup + 1) / 2;
        sp.pitch = s.pitch / 2;
        const double2* v2 = reinterpret_cast<const double2*>(v);
        double2* w2 = reinterpret_cast<double2*>(wt);
        if (R == 4)
          e = launch_dw<2, double2>(sp, td, plan.dw.max_block, lds_b, tb, wc_b, v2, w2, st);
        else
          e = launch_dw<4, double2>(sp, td, plan.dw.max_block, lds_b, tb, wc_b, v2, w2, st);
      }
    }
    if (!dw_pairs) switch (R) {
        case 2: e = launch_dw<2, VT>(s, td, plan.dw.max_block, lds_b, tb, wc_b, v, wt, st); break;
        case 4: e = launch_dw<4, VT>(s, td, plan.dw.max_block, lds_b, tb, wc_b, v, wt, st); break;
        default: e = launch_dw<8, VT>(s, td, plan.dw.max_block, lds_b, tb, wc_b, v, wt, st); break;
      }
  }
  if (e != hipSuccess) return e;
  if (cm) wc |= R << 8;  // (pass A: patch rows in ITS element units -- R real rows also when pass B ran on R/2 row pairs)
  const VT* wta = ((passes & 2) || only_pass == 1) ? wt : nullptr;
  if constexpr (!RV) {
    if (job_a) return launch_up_job(s, plan, tu, wc, v, wta, hv, lz, st);
  }
  if (passes & 1) switch (C) {
      case 2: e = launch_up<2, VT>(s, tu, lds_a, ta, norb1, wc, v, wta, hv, lz, st); break;
      case 4: e = launch_up<4, VT>(s, tu, lds_a, ta, norb1, wc, v, wta, hv, lz, st); break;
      default:
        if constexpr (RV)
          e = launch_up<8, VT>(s, tu, lds_a, ta, norb1, wc, v, wta, hv, lz, st);
        else
          e = hipErrorInvalidValue;  // (cplx_cols)
        break;
    }
  return e;
}

hipError_t launch_hxv_tiled(const DevSector& s, const TilePlan& plan, const double2* v, double2* wt, double2* hv, hipStream_t st,
                            const LzEpilogue* lz, int only_pass, bool wt_natural, const WtRange* wtr, int nwtr) {
  // wt: scratch of tiled_wt_elems() elements (dw-hop part, column-group-blocked), owned by the handle
  if (s.qdw == 0) return hipSuccess;
  return launch_tiled_vt<double2>(s, plan, v, wt, hv, st, lz, only_pass, wt_natural, wtr, nwtr);
}

hipError_t launch_hxv_tiled_real(const DevSector& s, const TilePlan& plan, const double* v, double* wt, double* hv, hipStream_t st,
                                 const LzEpilogue* lz, int only_pass, bool wt_natural, const WtRange* wtr, int nwtr) {
  // REAL vectors (H real; s.pitch must be the real pitch, a multiple of 16): wt needs no more bytes than in complex mode
  if (s.qdw == 0) return hipSuccess;
  if (!s.real_h) return hipErrorInvalidValue;
  return launch_tiled_vt<double>(s, plan, v, wt, hv, st, lz, only_pass, wt_natural, wtr, nwtr);
}

}  // namespace hxv
