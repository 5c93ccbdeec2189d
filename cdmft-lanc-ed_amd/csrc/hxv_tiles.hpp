// Tiling plan for the two-pass kernels.
#pragma once
#include <functional>

#include "hxv_internal.hpp"

namespace hxv {

// Prefix-block decomposition of one spin sector.  States that share their high (ns-lowbits)
// bits are contiguous in the sorted basis (ED_SETUP.f90:748-773 orders by integer value) and
// closed under every hop among the low orbitals.  The ELL table is split accordingly:
//   inner: source inside the row's own block, stored RELATIVE to the block start (LDS offset)
//   outer: source in another block, stored as the absolute index (global-memory gather)
struct SpinTiles {
  int lowbits = 0;
  int nblocks = 0;
  int max_block = 0;
  int k_in = 0, k_out = 0;
  int64_t n_in = 0, n_out = 0;       // entry counts (statistics)
  std::vector<uint32_t> start;       // [nblocks+1]
  uint32_t* d_start = nullptr;
  uint32_t* d_ell_in = nullptr;      // [k_in][dim]
  uint32_t* d_ell_out = nullptr;     // [k_out][dim]
};

struct TileOptions {
  int cols_per_tile = 4;    // pass A (up hops): columns per workgroup tile
  int rows_per_tile = 8;    // pass B (dw hops): rows per workgroup tile
  int lds_budget_kb_up = 64;  // LDS per workgroup tile, pass A
  int lds_budget_kb_dw = 16;  // LDS per workgroup tile, pass B (small tiles keep one row group per XCD in flight)
  int force_bits_up = -1, force_bits_dw = -1;
  int threads_up = 512, threads_dw = 256;
  int passes = 3;  // bit 0: pass A (diag + up hops), bit 1: pass B (dw hops); timing experiments only
};

struct TilePlan {
  TileOptions opt;
  SpinTiles up, dw;
  int ncoef_up = 1, ncoef_dw = 1;
};

using UploadU32 = std::function<hipError_t(const std::vector<uint32_t>&, uint32_t**)>;
std::string make_tile_plan(const SectorHost& s, TilePlan& plan, const UploadU32& upload);
hipError_t launch_hxv_tiled(const DevSector& s, const TilePlan& plan, const double2* v_full, double2* hv_local, hipStream_t st);

}  // namespace hxv
