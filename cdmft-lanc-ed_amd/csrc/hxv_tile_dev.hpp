// Device-side pieces shared by the tiled kernels (hxv_tiled.hip) and the pipelined job kernels (hxv_jobs.hip).
#pragma once
#include "hxv_device.hpp"
#include "hxv_tiles.hpp"

namespace hxv {

// Pass A's dw part in PIECES (exchange mode 2): after the second transpose the assembled dw part of a slab lies in one block per rank of
// origin -- rows [row0,row1) of all local columns, column stride `stride` elements -- in the receive buffer (the block this rank kept: in
// its panel output).  Pass A reads its accumulator init straight from there: element (row, local column c) = base[c*stride + row - row0].
struct WtRange {
  int32_t row0, row1;
  int64_t stride;
  const void* base;
};

struct DevTiles {
  const uint32_t* start;   // [nblocks+1]
  const uint32_t* tstart;  // [nblocks] start of the block whose in-block tables (perm, ell_in; gstart is aliased likewise) this block shares
  const uint32_t* perm;    // [dim]   sorted position -> index
  const uint32_t* gstart;  // [nblocks+1] first 64-position group of each block
  const uint32_t* gmax;    // [groups] longest in-block list of each 64-position group (low 16 bits)
  const uint32_t* ell_in;  // [k_in][dim]
  const uint32_t* ell16;   // [k_in/2][dim] the same table, two hops per word ((coefficient index << p16_bits) | offset each), or null
  const double2* scoef;    // [nscoef] signed coefficients, last = 0
  const uint32_t* bh_ptr;  // block hops / row slots of the out-of-block part (hxv_tiles.hpp)
  const uint32_t* bh;
  const uint32_t* rs_ptr;
  const uint32_t* rs_off;
  const uint32_t* rs_tab;
  const uint32_t* rs_base; // [nslots] first gather slot of the slot's source block (table words are relative to it; 0: absolute words)
  const uint32_t* rs_neg;  // [nslots] 1: the (shared) table holds this slot's coefficients with the opposite sign
  int nblocks, nscoef;
  int debug;  // timing experiments only: 1 skip out-of-block hops, 2 skip in-block hops, 8 plain instead of streaming hv stores (pass A),
              // 32 natural block order, 64 no packed row-slot words, 256 skip the block hops only, 512 skip the row slots only (round 6's
              // phase budget: profiles/r06_phase_ablate_c3.log)
  int pair_rows;  // pass B: the two row groups that share 128-byte lines run back to back, block by block (large sectors)
  const uint32_t* order;  // [nblocks] blocks by decreasing size (= grouped by table class), or null: pass B visits a row group's blocks in this order
  int p16_bits;           // ell16 words: (coefficient index << p16_bits) | offset, two per 32-bit word
  const uint32_t* rs16;     // half-size row-slot tables: slots sl, sl+1 of a block in one word (same split), or null
  const uint32_t* rs16_off; // [nslots] offset of the packed table of the pair that STARTS at this slot
  const WtRange* wtr;       // pass A, natural-layout dw part (wc == 0) given in row ranges instead of one array, or null
  int nwtr;
};

constexpr int HOP_CHUNK = 8;

// Non-temporal (streaming) accesses.  Measured: SLOWER than plain ones for loads (pass A's wt read: +7 %) and for short
// strided store segments (R*16-byte column segments of a natural-layout vector: they defeat L2 write combining, see
// scripts/strided_bench.py); FASTER for long runs that are not read again soon: pass A's hv (-3 %) and pass B's blocked wt.
typedef double dbl2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 load_stream(const double2* p) {
  dbl2_t x = __builtin_nontemporal_load(reinterpret_cast<const dbl2_t*>(p));
  return make_double2(x.x, x.y);
}
__device__ __forceinline__ double load_stream(const double* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void store_stream(double2* p, double2 a) {
  dbl2_t x;
  x.x = a.x;
  x.y = a.y;
  __builtin_nontemporal_store(x, reinterpret_cast<dbl2_t*>(p));
}

__device__ __forceinline__ void store_stream(double* p, double a) { __builtin_nontemporal_store(a, p); }

constexpr uint32_t TILE_OFF_MASK = (1u << TILE_COEF_SHIFT) - 1u;

// LDS by byte offset (a kernel's dynamic LDS starts at 0 when it declares no static LDS): addressing through the
// extern __shared__ symbol costs one vector add per access that the compiler cannot fold.
// (the HOST pass of hipcc sees 64-bit LDS pointers and warns about the 32-bit offsets; the device pass, the only one that runs them, has 32-bit ones)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
template <typename X>
__device__ __forceinline__ X lds_ld(uint32_t off) { return *(__attribute__((address_space(3))) const X*)(off); }
template <typename X>
__device__ __forceinline__ void lds_st(uint32_t off, X x) { *(__attribute__((address_space(3))) X*)(off) = x; }
template <>
__device__ __forceinline__ double2 lds_ld<double2>(uint32_t off) {
  const dbl2_t x = *(__attribute__((address_space(3))) const dbl2_t*)(off);
  return make_double2(x.x, x.y);
}
template <>
__device__ __forceinline__ void lds_st<double2>(uint32_t off, double2 a) {
  dbl2_t x;
  x.x = a.x;
  x.y = a.y;
  *(__attribute__((address_space(3))) dbl2_t*)(off) = x;
}
#pragma clang diagnostic pop
// (a << SH) + b in one instruction (the compiler turns (x & m) << SH into shift, mask, add)
template <int SH>
__device__ __forceinline__ uint32_t shl_add(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(SH), "s"(b));
  return r;
}

// Vector element type VT: double2 (complex vectors, the reference's complex(8)) or double (REAL vectors: when H is real,
// a real start vector keeps every Lanczos vector real -- half the bytes of every pass; device Lanczos only).
template <typename VT>
__device__ __forceinline__ VT vzero();
template <>
__device__ __forceinline__ double2 vzero<double2>() { return make_double2(0.0, 0.0); }
template <>
__device__ __forceinline__ double vzero<double>() { return 0.0; }
__device__ __forceinline__ void vadd(double2& a, double2 b) { a.x += b.x; a.y += b.y; }
__device__ __forceinline__ void vadd(double& a, double b) { a += b; }
__device__ __forceinline__ void vscale(double2& a, double c) { a.x *= c; a.y *= c; }
__device__ __forceinline__ void vscale(double& a, double c) { a *= c; }
// (written out: a.x*b.x rounded, then one fused multiply-add -- with a.y == 0 this is exactly the paired epilogue's a.x*b.x)
__device__ __forceinline__ double mul_rounded(double a, double b) {
#pragma clang fp contract(off)  // (never fused into a neighbouring add; the _rn intrinsics of HIP are plain operators)
  const double t = a * b;
  return t;
}
__device__ __forceinline__ double vdot(double2 a, double2 b) { return ::fma(a.y, b.y, mul_rounded(a.x, b.x)); }
__device__ __forceinline__ double vdot(double a, double b) { return mul_rounded(a, b); }

// paired Lanczos epilogue: the two components of a complex element are two independent real vectors
__device__ __forceinline__ void pair_scale(double2& a, double c, double c2) { a.x *= c; a.y *= c2; }
__device__ __forceinline__ void pair_scale(double&, double, double) {}
__device__ __forceinline__ void pair_fma(double2& acc, double c, double c2, double2 x) {
  acc.x = ::fma(c, x.x, acc.x);
  acc.y = ::fma(c2, x.y, acc.y);
}
__device__ __forceinline__ void pair_fma(double&, double, double, double) {}
__device__ __forceinline__ double pair_dot_re(double2 a, double2 b) { return mul_rounded(a.x, b.x); }
__device__ __forceinline__ double pair_dot_im(double2 a, double2 b) { return mul_rounded(a.y, b.y); }
__device__ __forceinline__ double pair_dot_re(double a, double b) { return mul_rounded(a, b); }
__device__ __forceinline__ double pair_dot_im(double, double) { return 0.0; }

template <bool REAL>
struct Coef;
template <>
struct Coef<true> {
  using type = double;
  static __device__ __forceinline__ void fma(double2& acc, double c, double2 x) {
    acc.x = ::fma(c, x.x, acc.x);
    acc.y = ::fma(c, x.y, acc.y);
  }
  static __device__ __forceinline__ void fma(double& acc, double c, double x) { acc = ::fma(c, x, acc); }
  static __device__ __forceinline__ double from(double2 c) { return c.x; }
  static __device__ __forceinline__ double neg(double c) { return -c; }
};
template <>
struct Coef<false> {
  using type = double2;
  static __device__ __forceinline__ void fma(double2& acc, double2 c, double2 x) { cfma(acc, c, x); }
  static __device__ __forceinline__ double2 from(double2 c) { return c; }
  static __device__ __forceinline__ double2 neg(double2 c) { return make_double2(-c.x, -c.y); }
};

template <bool NORB1>
__device__ __forceinline__ double diag_value(const DevDiag& dg, double au, uint32_t mu, int c) {
  const uint32_t md = dg.map_dw[c];
  if (NORB1) return au + dg.a_dw[c] + dg.cross.uloc[0] * (double)__popc(mu & md & dg.cross.orbmask[0]);
  return au + dg.a_dw[c] + diag_cross(dg.cross, mu, md);
}

}  // namespace hxv
