// Internal declarations shared by the host-side sector builder, the HIP kernels and the C-ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <complex>
#include <cstdint>
#include <exception>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hxv.h"

namespace hxv {

using cplx = std::complex<double>;

// ---------------------------------------------------------------------------------------
// ELL entry of a one-spin hopping matrix: 32 bits per stored element instead of the
// reference's 16 B value + 4 B column (ED_SPARSE_MATRIX.f90:13-30).
//   bits  0..19  source index (column of H_sigma), 0-based        (dim <= 2^20)
//   bits 20..29  coefficient id into coef[] (|amplitude| table)   (<= 1023 ids)
//   bit  31      sign: element = sign ? -coef[id] : coef[id]
// 0xFFFFFFFF = empty slot.  Layout [k][row] so consecutive rows are consecutive words.
// ---------------------------------------------------------------------------------------
constexpr uint32_t ELL_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t ND_INVALID = 0xFFFFFFFFu;
constexpr int ELL_SRC_BITS = 20;
constexpr uint32_t ELL_SRC_MASK = (1u << ELL_SRC_BITS) - 1u;
constexpr uint32_t ELL_COEF_MASK = 0x3FFu;
constexpr int MAX_COEF = 1023;

struct SpinOp {                 // H_up or H_dw on one spin sector
  int dim = 0;
  std::vector<int64_t> rowptr;  // CSR in the reference's row-list order
  std::vector<int32_t> cols;    // 0-based
  std::vector<cplx> vals;
  int K = 0;                    // ELL width
  std::vector<uint32_t> ell;    // [K][dim]
  std::vector<cplx> coef;       // coefficient table
  bool real_vals = true;
};

struct CrossParams {            // non-separable part of the diagonal (H_local.f90:35-50)
  int32_t norb = 1, nlat = 0;
  double uloc[5] = {0, 0, 0, 0, 0};
  double ust = 0;
  uint32_t orbmask[5] = {0, 0, 0, 0, 0};
  uint32_t sitemask[16] = {0};
};

// Spin-exchange / pair-hopping block spH0nd (sparse/H_non_local.f90:4-100), applied on the fly.
struct NonLocalParams {
  int32_t active = 0;  // Jhflag: Norb>1 and (Jx or Jp) != 0, ED_SETUP.f90:200-201
  int32_t nlat = 0, norb = 0;
  double jx = 0, jp = 0;
  int32_t fold = 1;    // added inside pass A of the tiled product (option "fold_nd"; 0: a separate pass over hv afterwards)
};

struct SectorHost {
  int ns = 0, nup = 0, ndw = 0;
  int dimup = 0, dimdw = 0;
  int pitch = 0;                    // device column pitch: DimUp rounded up to 8 elements (128-byte lines)
  int64_t dim = 0;
  int panel_rows = 0;               // > 0: a dw-only row panel (all-to-all exchange): dimup = panel_rows, no H_up
  int rank = 0, nranks = 1, qdw = 0, dw0 = 0;
  int cmax = 0;                     // columns per rank in the padded all-gather layout
  std::vector<uint32_t> vcol;       // [dimdw] column -> column slot in the padded layout
  // HALO exchange (exchange == 1): the gathered vector holds this rank's qdw columns (slots 0..qdw-1) followed by only
  // those columns of other ranks that H_dw couples to its rows, ascending (= grouped by owner rank)
  int exchange = 0;                 // 0 all-gather layout, 1 halo layout, 2 all-gather layout with the two-transposes exchange
  std::vector<int32_t> halo_cols;   // global columns received, in slot order
  std::vector<int32_t> halo_ptr;    // [nranks+1] offsets into halo_cols by owner rank
  std::vector<int32_t> send_cols;   // LOCAL column indices to send, grouped by destination rank
  std::vector<int32_t> send_ptr;    // [nranks+1]
  int64_t ishift = 0;
  std::vector<uint32_t> map_up, map_dw;
  SpinOp up, dw;
  bool separable_diag = true;
  std::vector<double> a_up, a_dw;  // separable diagonal tables
  CrossParams cross;
  NonLocalParams nd;
  std::vector<double> diag_stored; // from_csr path: explicit local diagonal
  // spH0nd (Jx/Jp) as Kronecker products of one-spin moves on a site: table [nlat*norb*norb][dim] per spin, entry
  // (il,x,y) of state i = c^+_{il,y} c_{il,x}|i>: target index | sign << 31, ND_INVALID if not applicable
  // (dw targets are column SLOTS of the gather layout).  Empty: the kernel falls back to searching the basis.
  std::vector<uint32_t> nd_up, nd_dw;

  // ---- DEVICE ROW ORDER (round 6).  The rows of every device vector -- the up index -- are NOT stored in the reference's order (orbital p
  // <-> bit p-1, ascending integers, ED_SETUP.f90:720-775) when up_perm is set, but in the order of the up configurations written with orbital
  // o at bit up_pos[o], ascending.  The orbitals that hop into the HIGH orbitals of pass A's prefix blocks take the highest low bits, so a row
  // slot's live rows and their sources are long contiguous runs: pass A's out-of-block gathers touch a third fewer cache lines at C3
  // (scripts/bitorder_sim.py; LABNOTES round 6).  The basis vectors change sign with the order of their creation operators:
  //     v_dev[up_perm[i]] = (up_sign: -1 : +1) * v_ref[i],      H_dev = S P H P^T S  = H written with the relabelled orbitals.
  // Everything the reference sees stays in ITS order (map_up, up, a_up, nd_up above: hxv_get_maps / _csr / _diag; host vectors are converted
  // at the boundary, hxv_capi.hip); the *_dev members below are what is uploaded and tiled.  Empty up_perm = the reference's order.
  std::vector<int32_t> up_pos;        // [ns] orbital -> bit
  std::vector<int32_t> up_perm;       // [dimup] reference row -> device row
  std::vector<int32_t> up_iperm;      // [dimup] device row -> reference row
  std::vector<uint8_t> up_sign;       // [dimup] by DEVICE row: 1 = that basis state carries a minus sign
  std::vector<uint32_t> key_up;       // [dimup] the configurations in up_pos numbering, ascending: the prefix blocks of pass A
  SpinOp up_dev;                      // H_up between device rows
  std::vector<uint32_t> map_up_dev;   // reference bit strings by device row (diagonal, ladder operators)
  std::vector<double> a_up_dev;
  std::vector<uint32_t> nd_up_dev;
  bool row_order() const { return !up_perm.empty(); }
  const SpinOp& dev_up() const { return row_order() ? up_dev : up; }
  const std::vector<uint32_t>& dev_map_up() const { return row_order() ? map_up_dev : map_up; }
  const std::vector<uint32_t>& dev_key_up() const { return row_order() ? key_up : map_up; }
  const std::vector<double>& dev_a_up() const { return row_order() ? a_up_dev : a_up; }
  const std::vector<uint32_t>& dev_nd_up() const { return row_order() ? nd_up_dev : nd_up; }
};

// A host worker thread whose body cannot take the process down: an exception inside the thread -- bad_alloc from the
// multi-hundred-MB tables of an Ns=18 sector -- would end in std::terminate, and an exception on the CALLING side while the thread is
// still joinable would do the same from std::thread's destructor.  run() catches into `err` (the builders' error-string channel),
// the destructor joins.
struct GuardedThread {
  std::thread th;
  std::string err;
  template <typename F>
  void run(F f) {
    th = std::thread([this, f]() mutable {
      try {
        f();
      } catch (const std::exception& e) {
        err = std::string("host worker thread: ") + e.what();
      } catch (...) {
        err = "host worker thread: unknown exception";
      }
    });
  }
  void join() {
    if (th.joinable()) th.join();
  }
  ~GuardedThread() { join(); }
};

// host builders (hxv_sector.cpp); return "" on success, else an error message
std::string build_sector_from_model(const hxv_model& m, int nup, int ndw, int rank, int nranks, SectorHost& out, int panel_rows = 0);
std::string build_sector_from_csr(int dimup, int dimdw, const int64_t* up_rp, const int32_t* up_cols, const double* up_vals,
                                  const int64_t* dw_rp, const int32_t* dw_cols, const double* dw_vals, const double* diag, int rank,
                                  int nranks, SectorHost& out);
std::string build_ell(SpinOp& op);
void dw_split(int dimdw, int rank, int nranks, int& qdw, int& dw0);
void make_vcol(SectorHost& s);
std::string make_panel_host(const SectorHost& main, int nrows, SectorHost& panel);  // the row panel of the all-to-all exchange
// needs s.dw (CSR); replaces the all-gather layout by the halo layout (more_*: further referenced columns per column, CSR-like)
void make_halo(SectorHost& s, const std::vector<int64_t>* more_ptr = nullptr, const std::vector<int32_t>* more_cols = nullptr);
int default_lowbits_up(int ns, int npart, int ncoef);  // the block bits the DEFAULT plan gives pass A (hxv_tiled.hip): decides the device row order
bool row_order_enabled();         // HXV_ROW_ORDER=0 keeps the reference's row order on the device
std::string row_order_env_key();  // (for the sector-image cache key)
int default_exchange();            // hxv_set_exchange_default / HXV_EXCHANGE=halo
void set_default_exchange(int mode);
std::vector<uint32_t> translate_ell_src(const std::vector<uint32_t>& ell, const std::vector<uint32_t>& vcol);
double host_diag_element(const SectorHost& s, int iup, int idw);

// ---------------------------------------------------------------------------------------
// device-side views (plain structs passed by value to kernels)
// ---------------------------------------------------------------------------------------
struct DevSpin {
  const uint32_t* ell;   // [K][dim]
  const double2* coef;   // [ncoef]
  int K;
  int dim;
};

struct DevDiag {
  int mode;               // 0 separable (on the fly), 1 stored
  const double* a_up;
  const double* a_dw;
  const uint32_t* map_up;
  const uint32_t* map_dw;
  const double* stored;   // local rows
  CrossParams cross;
};

struct DevSector {
  DevSpin up, dw;
  DevDiag diag;
  int dimup, dimdw;
  int pitch;              // elements between column starts of v / hv on the device (>= dimup, multiple of 8)
  int qdw, dw0;           // local columns [dw0, dw0+qdw)
  int slab0;              // column slot of local column 0 in the padded all-gather layout (= rank*cmax)
  const uint32_t* vcol;   // [dimdw] column -> column slot (identity when nranks==1)
  int vcol_identity;
  NonLocalParams nd;
  // device row order (SectorHost::up_perm): null = the reference's order.  map_up_ref: the reference's sorted up configurations; up_perm:
  // reference row -> device row; up_iperm: device row -> reference row; up_sign: by device row, 1 = the basis state carries a minus sign
  const uint32_t* map_up_ref;
  const int32_t* up_perm;
  const int32_t* up_iperm;
  const uint8_t* up_sign;
  const uint32_t* nd_up;  // move tables of the spH0nd block (SectorHost::nd_up / nd_dw), null: search the basis instead
  const uint32_t* nd_dw;
  // spH0nd as STORED by the caller (hxv_set_nonlocal_csr): local rows, global 1-based columns; null: none
  const int64_t* ndcsr_rowptr;
  const int32_t* ndcsr_cols;
  const double2* ndcsr_vals;
  int real_h;
};

// kernel launchers (hxv_kernels.hip)
struct TilePlan;  // opaque tiling data for the two-pass kernels
hipError_t launch_hxv_naive(const DevSector& s, const double2* v_full, double2* hv_local, hipStream_t st);
hipError_t launch_ladder(const uint32_t* map_from, int dim_from, const uint32_t* map_to, int dim_to, int pitch_from, int dimup_to,
                         int pitch_to, int dimdw_to, int orbital, int spin, int create, const double2* psi, double2* out, hipStream_t st,
                         double2 coef = double2{1.0, 0.0}, int accumulate = 0, const int32_t* perm_from = nullptr, const uint8_t* sign_from = nullptr,
                         const uint8_t* sign_to = nullptr);
// d_out[k*pitch + i] = d_in[cols[k]*pitch + i]: the columns a peer needs, packed for the halo exchange
struct WtRange;
// hv(row, c) += the dw part handed over in row ranges (exchange mode 2, overlapped form); hv: [ncols][pitch] elements of double2 or double
hipError_t launch_add_pieces(void* hv, const WtRange* wtr, int nwtr, int dimup, int pitch, int ncols, bool real, hipStream_t st);
hipError_t launch_pack_columns(const double2* d_in, double2* d_out, const int32_t* d_cols, int ncols, int pitch, hipStream_t st);
// true: the tiled product adds the spH0nd block itself (pass A, hxv_tiled.hip); false: launch_hxv_nonlocal after the product
inline bool nd_folds(const DevSector& s) { return s.nd.active && s.nd_up && s.nd_dw && s.nd.fold; }
hipError_t launch_hxv_nonlocal(const DevSector& s, const double2* v_full, double2* hv_local, hipStream_t st);

}  // namespace hxv
