// Tiling plan for the two-pass kernels.
#pragma once
#include <functional>

#include "hxv_internal.hpp"

namespace hxv {

// Tiled ELL word (differs from the plain table of hxv_internal.hpp):
//   bits  0..19  inner: source offset inside the block (LDS element offset)
//                outer: absolute source index (pass A: row; pass B: column slot of the gather layout)
//   bits 20..30  signed-coefficient index 2*id+sign into an LDS table of 2*ncoef+1 entries;
//                the last entry is 0, and an empty slot is (offset 0, that entry): it gathers in
//                bounds and adds nothing, so the hop loops carry no per-lane branch.
constexpr int TILE_COEF_SHIFT = 20;
constexpr int TILE_MAX_COEF = 255;  // 2*255+1 signed entries fit the LDS table comfortably

// Prefix-block decomposition of one spin sector.  States that share their high (ns-lowbits)
// bits are contiguous in the sorted basis (ED_SETUP.f90:748-773 orders by integer value) and
// closed under every hop among the low orbitals.  Inside a block the rows are visited in an
// order sorted by their entry counts so the 64 lanes of a wave carry equally long hop lists.
struct SpinTiles {
  int lowbits = 0;
  int nblocks = 0;
  int max_block = 0;
  int k_in = 0, k_out = 0;
  int k_in_real = 0;                 // longest in-block list of any row (k_in is this rounded up to the fetch chunk)
  int64_t n_in = 0, n_out = 0;       // entry counts (statistics)
  double slots_in = 0, slots_out = 0;  // processed slots per row after sorting (statistics)
  std::vector<uint32_t> start;       // [nblocks+1]
  uint32_t* d_start = nullptr;
  uint32_t* d_tstart = nullptr;      // [nblocks] start of the first block with identical in-block tables (they are read from there)
  int table_classes = 0;             // distinct in-block tables among the blocks
  uint32_t* d_perm = nullptr;        // [dim] sorted position -> index
  uint32_t* d_gstart = nullptr;      // [nblocks+1] first 64-lane group of each block
  uint32_t* d_gmax = nullptr;        // [ngroups] k_in max | k_out max << 16 of each 64-position group
  uint32_t* d_ell_in = nullptr;      // [k_in][dim], indexed by sorted position
  uint32_t* d_ell16 = nullptr;       // [k_in/2][dim] half-size copy (two hops per word), null when the fields do not fit 16 bits
  int p16_bits = 0;                  // offset bits of a half-size word (the coefficient index sits above them)
  // Out-of-block hops, grouped by (block, source block).  A pair whose hop maps the whole source block onto
  // the block with the identity on the low orbitals and one signed coefficient is a BLOCK hop: no per-row
  // data at all, the partner block is read as one contiguous, coalesced run.  Everything else is a ROW slot:
  // one table entry per row of the block (absolute source index | signed-coefficient index).
  uint32_t* d_bh_ptr = nullptr;      // [nblocks+1] -> d_bh
  uint32_t* d_bh = nullptr;          // [2*nbh] (source start, signed-coefficient index)
  uint32_t* d_rs_ptr = nullptr;      // [nblocks+1] -> d_rs_off
  uint32_t* d_rs_off = nullptr;      // [nslots] offset of each slot's table in d_rs_tab
  uint32_t* d_rs_tab = nullptr;      // flat tables, |block| words per slot (source relative to the slot's base)
  uint32_t* d_rs_base = nullptr;     // [nslots] first gather slot of the source block (0: the words are absolute)
  uint32_t* d_rs_neg = nullptr;      // [nslots] 1: the shared table holds the opposite overall sign
  int rs_tables = 0;                 // distinct row-slot tables after sharing
  uint32_t* d_rs16 = nullptr;        // half-size row-slot tables, two slots per word (pairs of a block's slots), shared like the single ones
  uint32_t* d_rs16_off = nullptr;    // [nslots] packed table of the pair starting at a slot
  bool rs16_on = false;
  double bh_per_row = 0, rs_per_row = 0;  // statistics: block hops / row slots visited per row
  int max_outer = 0;                 // most (row slots + block hops) of any block (register tables of the job kernels)
  uint32_t* d_order = nullptr;       // [nblocks] block indices, largest block first (job order inside a chunk)
  uint32_t* d_order_pc = nullptr;    // [nblocks] block indices by the particle number of the high orbitals (TileOptions::block_order)
};

struct TileOptions {
  int cols_per_tile = 4;      // pass A (up hops): columns per workgroup tile
  int rows_per_tile = 0;      // pass B (dw hops): rows per workgroup tile; 0 = by sector size (4; 8 = whole 128-byte lines when the
                              // panels of two neighbouring row groups no longer fit an XCD's L2), resolved when the plan is made
  int lds_budget_kb_up = 64;  // LDS per workgroup tile, pass A
  int lds_budget_kb_dw = 64;  // LDS per workgroup tile, pass B (two 1024-thread workgroups per CU)
  int force_bits_up = -1, force_bits_dw = -1;
  int threads_up = 1024, threads_dw = 1024;
  int sort_mode = 0;  // pass A visiting order: 0 natural (keeps global accesses coalesced), 1 by inner count, 2 by (outer, inner)
  int sort_mode_dw = 1;  // pass B inner phase (LDS only): 0 natural, 1 by inner count
  int lds_min_kb_up = 0, lds_min_kb_dw = 0;  // request at least this much LDS per workgroup (limits workgroups per CU)
  int spread_banks = 1;  // in-block hop lists re-dealt per wave so that the LDS gathers of a slot spread over the bank quads (host only)
  int block_order = -1;  // dispatch order of a group's blocks in the tile kernels: 0 largest first (= by table class), 1 by the particle
                         // number of the high orbitals (coupled blocks close together: L2 hits of the out-of-block gathers), 2 natural,
                         // -1 automatic
  int wt_colmajor = 1;    // blocked dw-hop scratch with column-major patches (tile kernels; the job kernel keeps row-major ones)
  int real_dw_pairs = 1;  // REAL vectors: pass B runs the complex kernel on pairs of rows (one decode and one 16-byte gather per two elements)
  int pair_rows = -1;  // pass B order: -1 automatic (paired row groups when two panels exceed the XCD's L2), 0 off, 1 on
  int job_max_blocks = 32;  // pass A runs as jobs only up to this many blocks per spin (beyond, one chunk's jobs no longer fit an XCD's CUs)
  int wt_cols = 4;  // columns per group of the blocked dw-hop scratch (>= cols_per_tile): R*wt_cols*16-byte write runs in pass B
  // pipelined job kernels (hxv_jobs.hip)
  int job_up = 2;      // pass A as jobs (block x run of column groups) with an LDS-DMA tile ring: 2 = for the fused Lanczos product only
                       // [default], 1 = always, 0 = never (one tile per workgroup)
  int job_cols = 1;    // columns per tile of a pass-A job: 1 (with 2 pass A does not run as jobs: those kernels were slower and spilled)
  int job_groups = 100; // column groups per job (about: an XCD's groups are cut into equal runs)
  int job_stages = 4;  // depth of the LDS tile ring (clamped to what fits 160 KB)
  int job_debug = 0;   // timing experiments only (JobUp::debug); results are wrong when non-zero
  int debug = 0;   // timing experiments only (see DevTiles::debug); results are wrong when non-zero
  int passes = 3;     // bit 0: pass A (diag + up hops), bit 1: pass B (dw hops); timing experiments only
};

struct TilePlan {
  TileOptions opt;
  SpinTiles up, dw;
  int ncoef_up = 1, ncoef_dw = 1;
  double2* d_scoef_up = nullptr;  // [2*ncoef+1] signed coefficient tables
  double2* d_scoef_dw = nullptr;
  bool usable = true;             // false: too many distinct amplitudes -> the engine uses kernel 0
};

// Optional Lanczos epilogue of pass A (device Lanczos, hxv_lanczos.hip): with x the input vector of the product,
//   w = s*(H x) - c*xm   is stored instead of H x, and   sum Re(conj(s*x) * w)   is reduced per workgroup,
// s = scal[i_s], c = scal[i_c] read from device memory.  Saves two full vector passes per iteration.
struct LzEpilogue {
  const void* xm = nullptr;     // previous (unnormalised) Lanczos vector (same element type as the product's vectors), may be null when c == 0
  const double* scal = nullptr;
  int i_s = 0, i_c = 0;
  double* partial = nullptr;    // one partial sum per workgroup of pass A
  // PAIRED epilogue (real H, complex vectors): Re and Im are two independent real Lanczos vectors; the imaginary part has
  // its own s = scal[i_s2], c = scal[i_c2] and its own partial sums
  int pair = 0;
  int i_s2 = 0, i_c2 = 0;
  double* partial2 = nullptr;
};
// workgroups of pass A (= partial sums of the Lanczos epilogue) for the product launch_hxv_tiled would run with an epilogue
// pieces: the dw part arrives in row ranges (exchange mode 2: always the tile kernel)
int64_t tiled_pass_up_workgroups(const DevSector& s, const TilePlan& plan, bool real_vec = false, bool pieces = false);
int64_t tiled_wt_elems(const DevSector& s, const TilePlan& plan);

struct PlanUploader {
  std::function<hipError_t(const std::vector<uint32_t>&, uint32_t**)> u32;
  std::function<hipError_t(const std::vector<double2>&, double2**)> d2;
};
std::string make_tile_plan(const SectorHost& s, TilePlan& plan, const PlanUploader& up);
// job kernels (hxv_jobs.hip)
struct DevTiles;
struct WtRange;
bool job_up_usable(const DevSector& s, const TilePlan& plan);
int64_t job_up_workgroups(const DevSector& s, const TilePlan& plan);
bool job_up_fits(const DevSector& s, const TilePlan& plan, bool lz, int wc);
hipError_t launch_up_job(const DevSector& s, const TilePlan& plan, const DevTiles& tu, int wc, const double2* v, const double2* wt, double2* hv,
                         const LzEpilogue* lz, hipStream_t st);
// wtr / nwtr (with only_pass = 1, wt_natural): the dw part handed over in row ranges (WtRange, hxv_tile_dev.hpp) instead of one array;
// wt_scratch must still be non-null (it marks "there is a dw part")
hipError_t launch_hxv_tiled(const DevSector& s, const TilePlan& plan, const double2* v_full, double2* wt_scratch, double2* hv_local,
                            hipStream_t st, const LzEpilogue* lz = nullptr, int only_pass = 0, bool wt_natural = false,
                            const WtRange* wtr = nullptr, int nwtr = 0);
// Same product on REAL vectors (double elements; H must be real, nranks == 1): s.pitch = real pitch (multiple of 16).
hipError_t launch_hxv_tiled_real(const DevSector& s, const TilePlan& plan, const double* v, double* wt_scratch, double* hv, hipStream_t st,
                                 const LzEpilogue* lz = nullptr, int only_pass = 0, bool wt_natural = false, const WtRange* wtr = nullptr,
                                 int nwtr = 0);

}  // namespace hxv
