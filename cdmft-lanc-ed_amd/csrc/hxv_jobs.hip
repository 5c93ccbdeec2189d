// Pipelined "job" kernels (gfx950): the same two passes as hxv_tiled.hip, restructured so that the HBM streams, the
// LDS gathers and the out-of-block gathers of a workgroup overlap instead of adding up.
//
//   job of pass A = (up prefix block kb) x (a run of `gc` column groups of C columns).  One 1024-thread workgroup per CU:
//     * wave 15 is the LOADER: it streams the [block] x [C columns] tiles of v and of the dw-hop scratch wt into an LDS
//       ring with LDS-DMA (global_load_lds_dwordx4: no staging registers), NST-1 tiles ahead of the compute waves, paced
//       by counted s_waitcnt vmcnt(N);
//     * waves 0..14 COMPUTE one row of the block per thread.  The per-row hop tables (in-block LDS offsets, out-of-block
//       row slots and block hops) are read ONCE per job into registers, so inside the tile loop a hop costs a decode,
//       one LDS coefficient read, C ds_read_b128 and the FMAs -- no dependent table round trips;
//     * one s_barrier per tile: "tile k+1 has landed" and "tile k's buffer is free" are the same barrier.
//   The out-of-block gathers of a tile are issued first (L2 of this XCD), the in-block LDS phase runs under their
//   latency, then they are consumed.  Workgroups of one chunk of column groups are adjacent in blockIdx (one XCD each),
//   so the blocks a gather reads are being streamed by a neighbour at about the same time.
//
// Reference semantics: ED_HAMILTONIAN_SPARSE_HxV.f90:167-227 (Hv = spH0d.v + spH0ups(1).v + spH0dws(1).v).
#include <algorithm>
#include <map>
#include <mutex>
#include <type_traits>

#include "hxv_tile_dev.hpp"

namespace hxv {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// s_waitcnt vmcnt needs an immediate: one case per value (gfx9 encodes 6 bits)
#define HXV_VMC(n) \
  case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vmcnt(int n) {
  switch (n) {
    HXV_VMC(0) HXV_VMC(1) HXV_VMC(2) HXV_VMC(3) HXV_VMC(4) HXV_VMC(5) HXV_VMC(6) HXV_VMC(7) HXV_VMC(8) HXV_VMC(9)
    HXV_VMC(10) HXV_VMC(11) HXV_VMC(12) HXV_VMC(13) HXV_VMC(14) HXV_VMC(15) HXV_VMC(16) HXV_VMC(17) HXV_VMC(18) HXV_VMC(19)
    HXV_VMC(20) HXV_VMC(21) HXV_VMC(22) HXV_VMC(23) HXV_VMC(24) HXV_VMC(25) HXV_VMC(26) HXV_VMC(27) HXV_VMC(28) HXV_VMC(29)
    HXV_VMC(30) HXV_VMC(31) HXV_VMC(32) HXV_VMC(33) HXV_VMC(34) HXV_VMC(35) HXV_VMC(36) HXV_VMC(37) HXV_VMC(38) HXV_VMC(39)
    HXV_VMC(40) HXV_VMC(41) HXV_VMC(42) HXV_VMC(43) HXV_VMC(44) HXV_VMC(45) HXV_VMC(46) HXV_VMC(47) HXV_VMC(48) HXV_VMC(49)
    HXV_VMC(50) HXV_VMC(51) HXV_VMC(52) HXV_VMC(53) HXV_VMC(54) HXV_VMC(55) HXV_VMC(56) HXV_VMC(57) HXV_VMC(58) HXV_VMC(59)
    HXV_VMC(60) HXV_VMC(61) HXV_VMC(62)
    default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
  }
}
#undef HXV_VMC

// workgroup barrier that leaves vector-memory operations (LDS-DMA, gathers, stores) in flight
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// wave-uniform read-only words through the scalar cache: a vector load here would sit in front of the tile's out-of-block
// gathers in the (in-order) vmcnt queue and expose their latency before the LDS phase
__device__ __forceinline__ uint32_t sload_u32(const uint32_t* p) {
  uint32_t r;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
  return r;
}
__device__ __forceinline__ double sload_f64(const double* p) {
  double r;
  asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
  return r;
}

struct JobUp {
  const void* v;       // gathered vector (all-gather layout)
  const void* wt;      // dw-hop scratch (may be null: no dw part)
  void* hv;            // local slab
  int ngroups, gpx;    // column groups of C columns; groups per XCD
  int gc;              // groups per job
  int chunks;          // jobs per (XCD, block) = ceil(gpx / gc)
  int wc;              // wt layout: 0 natural [column][pitch], else column-group-blocked wt[group][row][wc]
  int ns;              // LDS column stride of a tile (block rows rounded up to 64)
  int nst;             // ring depth (tiles)
  int stage_bytes;     // one ring stage: the v tile (and the previous Lanczos vector's tile with the LZ epilogue)
  int wt_bytes;        // one wt group buffer: block rows x max(wc,1) columns
  int kin_rows;        // rows of the in-block table (plan k_in)
  int max_outer;       // most out-of-block slots (row slots + block hops) of any block
  int debug;           // timing experiments only (option job_debug): 1 no out-of-block gathers, 2 no in-block hops, 4 no compute at all,
                       // 8 loader skips wt, 16 loader issues nothing, 32 no hv store, 64 nt policy for the wt DMA,
                       // 128 every gather reads the thread's own row, 256 gathers scattered over the own block
  const uint32_t* order;  // [nblocks] blocks of a chunk, largest first
};

constexpr int JOB_WAVES = 16, JOB_LOADER = JOB_WAVES - 1, JOB_MAX_STAGES = 8;

// LZ: 0 plain product, 1 Lanczos epilogue, 2 PAIRED epilogue (real H: Re and Im are two independent real Lanczos vectors, hxv_tiles.hpp)
// (LDS-DMA destinations are byte offsets: see the note on the host pass's pointer width in hxv_tile_dev.hpp)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-void-pointer-cast"
template <int C, bool REALC, bool NORB1, int LZ, int KIN, int KO, typename VT>
__global__ void __launch_bounds__(1024) hxv_up_job(DevSector s, DevTiles t, JobUp jb, LzEpilogue lz) {
  using CT = typename Coef<REALC>::type;
  constexpr uint32_t OFFM = (1u << TILE_COEF_SHIFT) - 1u;
  const int b = blockIdx.x;
  const int xcd = b & 7, j = b >> 3;
  const int chunk = j / t.nblocks;
  const int kb = (int)jb.order[j - chunk * t.nblocks];
  // (the XCD's groups are cut into `chunks` runs of equal length +-1: no short last job)
  const int g0 = xcd * jb.gpx + (int)(((int64_t)chunk * jb.gpx) / jb.chunks);
  const int g1 = min(xcd * jb.gpx + (int)(((int64_t)(chunk + 1) * jb.gpx) / jb.chunks), jb.ngroups);
  const int ntile = g1 - g0;
  if (chunk >= jb.chunks || ntile <= 0) {
    if (LZ && threadIdx.x == 0) {
      lz.partial[blockIdx.x] = 0.0;
      if (LZ == 2) lz.partial2[blockIdx.x] = 0.0;
    }
    return;
  }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int r0 = (int)t.start[kb];
  const int n = (int)t.start[kb + 1] - r0;
  const int ns = jb.ns, nst = jb.nst;
  const int nch = (n + 63) >> 6;  // 1 KiB pieces per tile column
  const VT* __restrict__ v = reinterpret_cast<const VT*>(jb.v);
  const VT* __restrict__ wt = reinterpret_cast<const VT*>(jb.wt);
  // LDS by byte offset (hxv_tile_dev.hpp): signed coefficients at 0 (a shifted table word is their address), the loader's
  // bookkeeping words, the tile ring, two wt group buffers
  static_assert(sizeof(VT) == 16, "complex vectors only");
  constexpr int LCB = REALC ? 3 : 4;
  const uint32_t book = ((uint32_t)t.nscoef * 16 + 255) & ~255u;
  const uint32_t ring0 = book + 256;
  const uint32_t wtb = ring0 + (uint32_t)nst * jb.stage_bytes;
  const int wcw = max(jb.wc, 1);
  const int lw = 31 - __clz(wcw);  // wcw is a power of two

  if (wave == JOB_LOADER) {
    // ------------------------------------------------------------------ loader wave: LDS-DMA only, no register loads
    if (jb.debug & 8) wt = nullptr;
    const bool nodma = (jb.debug & 16) != 0;
    int ops = 0;              // LDS-DMA instructions issued so far (they retire in order)
    // value of `ops` after the last piece of the tile in each ring stage / of the wt group in each group buffer
    // (loader-private words in LDS: dynamic indexing without register arrays)
    const uint32_t vdone = book, wdone = book + 4 * JOB_MAX_STAGES;
    int st_issue = 0;         // ring stage of the next tile to issue
    auto issue_v = [&](int k) {
      const uint32_t base = ring0 + (uint32_t)st_issue * jb.stage_bytes;
      const int c0 = (g0 + k) * C;
      if (!nodma) {
#pragma unroll
        for (int cc = 0; cc < C; ++cc) {
          const int c = min(c0 + cc, s.qdw - 1);  // local column
          const VT* __restrict__ src = v + (int64_t)(s.slab0 + c) * s.pitch + r0;
          for (int ch = 0; ch < nch; ++ch)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(src + min(ch * 64 + lane, n - 1)),
                                             (lds_void_t*)(base + (uint32_t)(cc * ns + ch * 64) * 16), 16, 0, 0);
        }
        ops += C * nch;
        if (LZ && lz.xm) {
          const VT* __restrict__ xm = reinterpret_cast<const VT*>(lz.xm);
#pragma unroll
          for (int cc = 0; cc < C; ++cc) {
            const int c = min(c0 + cc, s.qdw - 1);
            const VT* __restrict__ src = xm + (int64_t)c * s.pitch + r0;
            for (int ch = 0; ch < nch; ++ch)
              __builtin_amdgcn_global_load_lds((glb_void_t*)(src + min(ch * 64 + lane, n - 1)),
                                               (lds_void_t*)(base + (uint32_t)((C + cc) * ns + ch * 64) * 16), 16, 0, 0);
          }
          ops += C * nch;
        }
      }
      if (lane == 0) lds_st<int>(vdone + 4 * st_issue, ops);
      st_issue = st_issue + 1 == nst ? 0 : st_issue + 1;
    };
    // wt arrives one column GROUP at a time (all wcw columns of the block's rows: one contiguous run of the
    // column-group-blocked scratch, every line read exactly once) into one of two group buffers
    const int wpieces = (n * wcw + 63) >> 6;
    auto issue_w = [&](int G) {
      if (wt && !nodma) {
        const uint32_t base = wtb + (uint32_t)(G & 1) * jb.wt_bytes;
        const VT* __restrict__ src = jb.wc ? wt + ((int64_t)G * s.dimup + r0) * wcw : wt + (int64_t)G * s.pitch + r0;
        for (int q = 0; q < wpieces; ++q)
          if (jb.debug & 64)  // (experiment: streaming policy for the once-read scratch)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(src + min(q * 64 + lane, n * wcw - 1)), (lds_void_t*)(base + (uint32_t)q * 1024), 16, 0, 2);
          else
            __builtin_amdgcn_global_load_lds((glb_void_t*)(src + min(q * 64 + lane, n * wcw - 1)), (lds_void_t*)(base + (uint32_t)q * 1024), 16, 0, 0);
        ops += wpieces;
      }
      if (lane == 0) lds_st<int>(wdone + 4 * (G & 1), ops);
    };
    for (int q = lane; q < t.nscoef; q += 64) lds_st<CT>((uint32_t)q << LCB, Coef<REALC>::from(t.scoef[q]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the coefficient loads above are ordinary loads)
    const int cfirst = g0 * C, clast = min(g1 * C, s.qdw) - 1;
    const int Gfirst = cfirst >> lw, Glast = clast >> lw;
    issue_w(Gfirst);
    int issued = 0;
    for (; issued < min(nst - 1, ntile); ++issued) issue_v(issued);
    if (Gfirst < Glast) issue_w(Gfirst + 1);
    int Gprev = Gfirst;
    int st_wait = 0;
    for (int k = 0; k < ntile; ++k) {
      const int Gk = ((g0 + k) * C) >> lw;  // (C divides wcw or wcw == 1: a tile never straddles two groups)
      const int need = __builtin_amdgcn_readfirstlane(max(lds_ld<int>(vdone + 4 * st_wait), lds_ld<int>(wdone + 4 * (Gk & 1))));
      st_wait = st_wait + 1 == nst ? 0 : st_wait + 1;
      wait_vmcnt(min(63, ops - need));  // tile k and its wt group have landed; younger pieces stay in flight
      wg_barrier();                     // A(k): publishes tile k, frees the buffer of tile k-1
      if (issued < ntile) {
        issue_v(issued);
        ++issued;
      }
      if (Gk != Gprev) {  // first tile of a group: the previous group's buffer is free now
        if (Gk < Glast) issue_w(Gk + 1);
        Gprev = Gk;
      }
    }
    return;
  }

  // -------------------------------------------------------------------- compute waves: one block row per thread
  const int p = threadIdx.x;
  const bool row_ok = p < n;
  const bool wave_on = (wave << 6) < n;
  // Table words are re-packed once per job so that a hop in the tile loop decodes in two instructions:
  //   in-block      (coefficient's LDS byte address) << 16 | (source row * 16)
  //   out-of-block  (coefficient index) << 23            | (source row of the column * 16)     [dimup < 65536: job_up_usable]
  uint32_t tin[KIN], tou[KO];
  int kin = 0, nouter = 0;
  double au = 0.0;
  uint32_t mu = 0;
  if (wave_on) {
    const int pr = min(p, n - 1);
    const int tb0 = (int)t.tstart[kb];  // the block whose in-block tables this one shares
    const uint32_t EMPTY = (uint32_t)(t.nscoef - 1) << TILE_COEF_SHIFT;
#pragma unroll
    for (int k = 0; k < KIN; ++k) {
      const uint32_t e = (row_ok && k < jb.kin_rows) ? t.ell_in[(int64_t)k * s.dimup + tb0 + pr] : EMPTY;
      tin[k] = ((e >> TILE_COEF_SHIFT) << (16 + LCB)) | ((e & OFFM) << 4);
    }
    const int rs0 = (int)t.rs_ptr[kb], nrs = (int)t.rs_ptr[kb + 1] - rs0;
    const int bh0 = (int)t.bh_ptr[kb], nbh = (int)t.bh_ptr[kb + 1] - bh0;
    nouter = __builtin_amdgcn_readfirstlane(nrs + nbh);
#pragma unroll
    for (int i = 0; i < KO; ++i) {
      uint32_t e = EMPTY;
      if (i < nrs) {
        e = t.rs_tab[t.rs_off[rs0 + i] + pr];
        if (e != EMPTY) e = ((e ^ (t.rs_neg[rs0 + i] << TILE_COEF_SHIFT)) & ~OFFM) | ((e & OFFM) + t.rs_base[rs0 + i]);  // shared table: sign, base
      } else if (i < nrs + nbh) {
        const int h = bh0 + i - nrs;
        e = (t.bh[2 * h + 1] << TILE_COEF_SHIFT) | (t.bh[2 * h] + (uint32_t)pr);
      }
      if (!row_ok) e = EMPTY;
      tou[i] = ((e >> TILE_COEF_SHIFT) << 23) | ((e & OFFM) << 4);
      if (jb.debug & 128) tou[i] = (tou[i] & ~0x7FFFFFu) | ((uint32_t)(r0 + pr) << 4);  // experiment: every gather reads the thread's own row
      if (jb.debug & 256) tou[i] = (tou[i] & ~0x7FFFFFu) | ((uint32_t)(r0 + ((pr * 37 + i * 101) % n)) << 4);  // experiment: scattered rows of the own block
    }
    kin = (int)(__builtin_amdgcn_readfirstlane(t.gmax[t.gstart[kb] + wave]) & 0xFFFFu);
    if (s.diag.mode == 0) {
      au = s.diag.a_up[r0 + pr];
      mu = s.diag.map_up[r0 + pr];
    }
  }
  double asum = 0.0, asum2 = 0.0;
  const double sc = LZ ? lz.scal[lz.i_s] : 1.0;
  const double cm = (LZ && lz.xm) ? lz.scal[lz.i_c] : 0.0;
  const double sc2 = LZ == 2 ? lz.scal[lz.i_s2] : 1.0;
  const double cm2 = (LZ == 2 && lz.xm) ? lz.scal[lz.i_c2] : 0.0;
  char* __restrict__ hvb = reinterpret_cast<char*>(jb.hv) + (int64_t)r0 * 16;
  const char* __restrict__ vb = reinterpret_cast<const char*>(v);
  const uint32_t p16 = (uint32_t)p << 4, pw16 = p16 << lw;
  const int64_t pitchb = (int64_t)s.pitch * 16;
  const bool outer_on = !(jb.debug & 1), inner_on = !(jb.debug & 2);
  VT pend[C];       // result of the previous tile, stored one tile late
  int pend_c0 = 0;
  auto flush_pending = [&](int ncols) {
    if (row_ok && !(jb.debug & 32)) {
#pragma unroll
      for (int cc = 0; cc < C; ++cc) {
        if (cc < ncols) store_stream(reinterpret_cast<VT*>(hvb + (int64_t)(pend_c0 + cc) * pitchb + p16), pend[cc]);
      }
    }
  };

  uint32_t stage_off = ring0;  // ring stage of tile k
  int stage = 0;
  // out-of-block gathers (slots 0..7) of a tile
  auto issue_outer = [&](VT (&xo)[8][C], int k) {
    const int c0 = (g0 + k) * C;
    const int nc = min(C, s.qdw - c0);
    const char* __restrict__ vcol0 = vb + (int64_t)(s.slab0 + c0) * pitchb;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      // (blocks have 4..8 slots at C3; a wave-uniform guard skips the rest)
      if (i < 4 || i < nouter) {
        const uint32_t off = tou[i] & 0x7FFFFFu;
#pragma unroll
        for (int cc = 0; cc < C; ++cc)
          xo[i][cc] = *reinterpret_cast<const VT*>((C == 1 ? vcol0 : vcol0 + (int64_t)min(cc, nc - 1) * pitchb) + off);
      }
    }
  };
  auto tile = [&](VT (&xo)[8][C], int k) {
    wg_barrier();  // A(k)
    const uint32_t tile_off = stage_off;
    stage = stage + 1 == nst ? 0 : stage + 1;
    stage_off = ring0 + (uint32_t)stage * jb.stage_bytes;
    if (!wave_on || (jb.debug & 4)) return;
    // keep the table words opaque per tile: otherwise the compiler hoists the decoded LDS address and coefficient address
    // of every entry out of the tile loop (two more registers per entry) and spills
#pragma unroll
    for (int q = 0; q < KIN; ++q) asm volatile("" : "+v"(tin[q]));
#pragma unroll
    for (int q = 0; q < KO; ++q) asm volatile("" : "+v"(tou[q]));
    const int c0 = (g0 + k) * C;
    const int nc = min(C, s.qdw - c0);
    const char* __restrict__ vcol0 = vb + (int64_t)(s.slab0 + c0) * pitchb;
    // (the previous tile's result is stored AFTER the gathers are issued: vector-memory operations retire in order, so a
    //  store issued first would put its write acknowledgement in front of them)
    if (outer_on) issue_outer(xo, k);
    if (k > 0) flush_pending(C);
    VT acc[C], xq[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
      xq[cc] = lds_ld<VT>(tile_off + (uint32_t)(cc * ns) * 16 + p16);
      if (wt) {
        const int c = c0 + min(cc, nc - 1);
        acc[cc] = lds_ld<VT>(wtb + (uint32_t)((c >> lw) & 1) * jb.wt_bytes + (uint32_t)(c & (wcw - 1)) * 16 + pw16);
      } else {
        acc[cc] = vzero<VT>();
      }
      const int cg = __builtin_amdgcn_readfirstlane(s.dw0 + min(c0 + cc, s.qdw - 1));
      const uint32_t md = sload_u32(s.diag.map_dw + cg);
      const double adw = sload_f64(s.diag.a_dw + cg);
      double d = au + adw;
      if (NORB1)
        d += s.diag.cross.uloc[0] * (double)__popc(mu & md & s.diag.cross.orbmask[0]);
      else
        d += diag_cross(s.diag.cross, mu, md);
      Coef<true>::fma(acc[cc], d, xq[cc]);
    }
    // in-block hops: gathers from the LDS tile, tables in registers
#pragma unroll
    for (int k4 = 0; k4 < KIN; k4 += 4) {
      if (k4 < kin && inner_on) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t e = tin[k4 + u];
          const CT cf = lds_ld<CT>(e >> 16);
          const uint32_t src = (e & 0xFFFFu) + tile_off;
#pragma unroll
          for (int cc = 0; cc < C; ++cc) Coef<REALC>::fma(acc[cc], cf, lds_ld<VT>(src + (uint32_t)(cc * ns) * 16));
        }
      }
    }
    if (outer_on) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i < 4 || i < nouter) {
          const CT cf = lds_ld<CT>(REALC ? tou[i] >> 20 : (tou[i] >> 23) << LCB);
#pragma unroll
          for (int cc = 0; cc < C; ++cc) Coef<REALC>::fma(acc[cc], cf, xo[i][cc]);
        }
      }
      // blocks with more than 8 out-of-block slots (wave-uniform): the rest in further batches of 8, latency exposed
#pragma unroll
      for (int b8 = 8; b8 < KO; b8 += 8) {
        if (b8 < nouter) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const uint32_t off = tou[b8 + i] & 0x7FFFFFu;
#pragma unroll
            for (int cc = 0; cc < C; ++cc) xo[i][cc] = *reinterpret_cast<const VT*>(vcol0 + (int64_t)min(cc, nc - 1) * pitchb + off);
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const CT cf = lds_ld<CT>(REALC ? tou[b8 + i] >> 20 : (tou[b8 + i] >> 23) << LCB);
#pragma unroll
            for (int cc = 0; cc < C; ++cc) Coef<REALC>::fma(acc[cc], cf, xo[i][cc]);
          }
        }
      }
    }
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
      VT w = acc[cc];
      if constexpr (LZ == 2) {
        // component-wise: the same operations, in the same order, as two LZ == 1 runs on (x, 0) and (y, 0)
        pair_scale(w, sc, sc2);
        if (lz.xm) pair_fma(w, -cm, -cm2, lds_ld<VT>(tile_off + (uint32_t)((C + cc) * ns) * 16 + p16));
        if (row_ok && cc < nc) {
          asum = ::fma(sc, pair_dot_re(xq[cc], w), asum);
          asum2 = ::fma(sc2, pair_dot_im(xq[cc], w), asum2);
        }
      } else if (LZ) {
        vscale(w, sc);
        if (lz.xm) Coef<true>::fma(w, -cm, lds_ld<VT>(tile_off + (uint32_t)((C + cc) * ns) * 16 + p16));
        if (row_ok && cc < nc) asum = ::fma(sc, vdot(xq[cc], w), asum);
      }
      pend[cc] = w;
    }
    pend_c0 = c0;
  };
  VT xa[8][C];
  for (int k = 0; k < ntile; ++k) tile(xa, k);
  if (wave_on && ntile > 0 && !(jb.debug & 4)) flush_pending(min(C, s.qdw - pend_c0));
  if (LZ) {
    // wavefront partial sums first (DPP/shuffle), one LDS word per wave afterwards
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      asum += __shfl_down(asum, off, 64);
      if (LZ == 2) asum2 += __shfl_down(asum2, off, 64);
    }
    wg_barrier();  // every tile buffer is free (the loader has left)
    if (lane == 0) {
      lds_st<double>(ring0 + 8 * wave, asum);
      if (LZ == 2) lds_st<double>(ring0 + 8 * (JOB_WAVES + wave), asum2);
    }
    wg_barrier();
    if (threadIdx.x == 0) {
      double tot = 0.0, tot2 = 0.0;
      for (int w = 0; w < JOB_LOADER; ++w) {
        tot += lds_ld<double>(ring0 + 8 * w);
        if (LZ == 2) tot2 += lds_ld<double>(ring0 + 8 * (JOB_WAVES + w));
      }
      lz.partial[blockIdx.x] = tot;
      if (LZ == 2) lz.partial2[blockIdx.x] = tot2;
    }
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
#pragma clang diagnostic pop
namespace {

hipError_t allow_lds(const void* kern, int bytes) {
  static std::mutex mu;
  static std::map<const void*, int> granted;
  std::lock_guard<std::mutex> lk(mu);
  int& g = granted[kern];
  if (bytes <= g) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) g = bytes;
  return e;
}

// register-resident tables of a job: up to 24 in-block words and 8 out-of-block slots per row.  Blocks with more slots (BHZ: up
// to 16) would need a second batch of gathers whose latency nothing covers: measured 5 % slower than one tile per
// workgroup at C4, so such plans do not run as jobs.
constexpr int JOB_KIN = 24, JOB_KO = 8;

template <int C, int LZ, int KIN, int KO>
hipError_t launch_up_job_k(const DevSector& s, const DevTiles& t, const JobUp& jb, int lds_bytes, int64_t nwg, const LzEpilogue& lz,
                           hipStream_t st) {
  const bool norb1 = s.diag.cross.norb == 1;
  void (*kern)(DevSector, DevTiles, JobUp, LzEpilogue) = nullptr;
  if constexpr (LZ == 2) {  // the paired epilogue exists for real H only
    if (!s.real_h) return hipErrorInvalidValue;
    kern = norb1 ? hxv_up_job<C, true, true, 2, KIN, KO, double2> : hxv_up_job<C, true, false, 2, KIN, KO, double2>;
  } else if (s.real_h)
    kern = norb1 ? hxv_up_job<C, true, true, LZ, KIN, KO, double2> : hxv_up_job<C, true, false, LZ, KIN, KO, double2>;
  else
    kern = norb1 ? hxv_up_job<C, false, true, LZ, KIN, KO, double2> : hxv_up_job<C, false, false, LZ, KIN, KO, double2>;
  hipError_t e = allow_lds((const void*)kern, lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(64 * JOB_WAVES), (size_t)lds_bytes, st, s, t, jb, lz);
  return hipGetLastError();
}

template <int C, int LZ>
hipError_t launch_up_job_c(const DevSector& s, const DevTiles& t, const JobUp& jb, int lds_bytes, int64_t nwg, const LzEpilogue& lz,
                           hipStream_t st) {
  // fewer table registers when the longest in-block list / the out-of-block slot count allow it
  if (jb.max_outer > JOB_KO) return hipErrorInvalidValue;  // (job_up_usable)
  if (jb.kin_rows <= 20) return launch_up_job_k<C, LZ, 20, JOB_KO>(s, t, jb, lds_bytes, nwg, lz, st);
  return launch_up_job_k<C, LZ, JOB_KIN, JOB_KO>(s, t, jb, lds_bytes, nwg, lz, st);
}

}  // namespace

bool job_up_usable(const DevSector& s, const TilePlan& plan) {
  const SpinTiles& u = plan.up;
  return plan.usable && s.diag.mode == 0 && u.max_block <= 64 * JOB_LOADER && u.k_in <= JOB_KIN && u.max_outer <= JOB_KO &&
         u.d_order != nullptr && s.dimup < 65536 &&  // (packed out-of-block words: hxv_up_job)
         u.nblocks <= plan.opt.job_max_blocks && plan.opt.job_cols == 1;
}

static void job_up_geometry(const DevSector& s, const TilePlan& plan, bool lz_xm, int wc, JobUp& jb, int& lds_bytes, int64_t& nwg) {
  const int C = plan.opt.job_cols;
  jb.ngroups = (s.qdw + C - 1) / C;
  jb.gpx = (jb.ngroups + 7) / 8;
  jb.gc = std::max(1, plan.opt.job_groups);
  jb.chunks = std::max(1, (jb.gpx + jb.gc / 2) / jb.gc);  // about job_groups groups per job, all jobs of an XCD equally long
  jb.ns = (plan.up.max_block + 63) & ~63;
  jb.stage_bytes = (1 + (lz_xm ? 1 : 0)) * C * jb.ns * 16;
  jb.wt_bytes = jb.ns * std::max(wc, 1) * 16;
  const int tab = (((2 * plan.ncoef_up + 1) * 16 + 255) & ~255) + 256;  // coefficients, loader bookkeeping words
  jb.nst = std::min(std::min(plan.opt.job_stages, JOB_MAX_STAGES), (160 * 1024 - tab - 2 * jb.wt_bytes) / jb.stage_bytes);
  lds_bytes = jb.nst * jb.stage_bytes + 2 * jb.wt_bytes + tab;
  jb.kin_rows = std::min(plan.up.k_in, (plan.up.k_in_real + 3) & ~3);
  jb.max_outer = plan.up.max_outer;
  jb.debug = plan.opt.job_debug;
  jb.order = plan.up.d_order;
  nwg = (int64_t)8 * jb.chunks * plan.up.nblocks;
}

int64_t job_up_workgroups(const DevSector& s, const TilePlan& plan) {
  JobUp jb{};
  int lds = 0;
  int64_t nwg = 0;
  job_up_geometry(s, plan, false, 0, jb, lds, nwg);
  return nwg;
}

bool job_up_fits(const DevSector& s, const TilePlan& plan, bool lz, int wc) {
  // (the Lanczos epilogue streams the previous vector's tile as well: decided for the worst case so that the choice of
  //  kernel -- and the number of per-workgroup partial sums -- does not change from one iteration to the next)
  if (wc > 0 && wc % plan.opt.job_cols != 0) return false;
  JobUp jb{};
  int lds = 0;
  int64_t nwg = 0;
  job_up_geometry(s, plan, lz, wc, jb, lds, nwg);
  return jb.nst >= 2;
}

hipError_t launch_up_job(const DevSector& s, const TilePlan& plan, const DevTiles& tu, int wc, const double2* v, const double2* wt, double2* hv,
                         const LzEpilogue* lz, hipStream_t st) {
  JobUp jb{};
  int lds_bytes = 0;
  int64_t nwg = 0;
  job_up_geometry(s, plan, lz != nullptr, wc, jb, lds_bytes, nwg);
  if (!job_up_fits(s, plan, lz != nullptr, wc)) return hipErrorInvalidValue;
  jb.v = v;
  jb.wt = wt;
  jb.hv = hv;
  jb.wc = wc;
  // (two columns per tile were measured slower, and every such kernel spills vector and scalar registers: not built any more)
  if (plan.opt.job_cols != 1) return hipErrorInvalidValue;
  if (lz && lz->pair) return launch_up_job_c<1, 2>(s, tu, jb, lds_bytes, nwg, *lz, st);
  return lz ? launch_up_job_c<1, 1>(s, tu, jb, lds_bytes, nwg, *lz, st) : launch_up_job_c<1, 0>(s, tu, jb, lds_bytes, nwg, LzEpilogue(), st);
}

}  // namespace hxv
