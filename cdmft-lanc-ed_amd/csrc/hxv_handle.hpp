// The opaque handle behind include/hxv.h and the error helpers shared by the translation units
// that implement the C-ABI (hxv_capi.hip: handles + products; hxv_lanczos.hip: Lanczos recurrences; hxv_eigh.hip: thick-restart eigensolver).
#pragma once
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "hxv_internal.hpp"
#include "hxv_tiles.hpp"

struct hxv_handle;
namespace hxv {
int fail(int code, const std::string& msg);  // records the message hxv_last_error() returns; returns code
constexpr int RED_BLOCKS = 1024;             // workgroups of the grid-stride reduction kernels
// vector-sized device buffers go through the engine's cache (hxv_pool.cpp): a fresh hipMalloc costs ~25 ms per GB here
hipError_t pool_alloc(int device, size_t bytes, void** out);
void pool_free(int device, void* ptr);
int ensure_wt(hxv_handle* h);                // (re)allocates the dw-hop scratch of the tiled kernels (hxv_capi.hip)
// slab exchange of a split sector (hxv_comm.cpp)
bool comm_ready(const hxv_handle* h);
bool comm_in_gather(const hxv_handle* h, const void* p);  // p lies in one of the handle's gather buffers (hxv_slab_home)
int comm_allreduce_sum(hxv_handle* h, double* d_buf, size_t count, hipStream_t st);  // no-op without a communicator
int comm_sendrecv_cols(hxv_handle* h, const void* send, const int64_t* send_ptr, void* recv, const int64_t* recv_ptr, size_t col_bytes, hipStream_t st);
int comm_agree(hxv_handle* h, int rc_local);  // collective: non-zero on every rank if any rank passes non-zero (no-op without a communicator)
// (H v)|slab from this rank's slab: exchange + product; `ep`: optional Lanczos epilogue of pass A (its partial sums are this rank's share)
int apply_slab(hxv_handle* h, const double2* d_v_local, double2* d_hv_local, hipStream_t st, const LzEpilogue* ep = nullptr);
int apply_slab_real(hxv_handle* h, const double* d_v_local, double* d_hv_local, hipStream_t st, const LzEpilogue* ep = nullptr);
void comm_release(hxv_handle* h);
// REAL-vector mode helpers shared by the Lanczos drivers (hxv_capi.hip / hxv_lanczos.hip)
const char* real_mode_blocker(const hxv_handle* h);  // nullptr when real vectors can be used with this handle
int pitch_real_of(const hxv_handle* h);
// layout conversions between complex [DimDw][pitch] and real [DimDw][pitch_real] device vectors (pads written as zero)
void launch_to_real(const hxv_handle* h, const double2* src, double* dst, hipStream_t st);
void launch_to_complex(const hxv_handle* h, const double* src, double2* dst, hipStream_t st);
// deterministic start vector, real part of the complex one (imaginary part dropped)
void launch_init_real(const hxv_handle* h, double* q, uint64_t seed, hipStream_t st);
// one Lanczos step on normalised vectors through the fused product (hxv_lanczos.hip): w = H q - beta*qm, alpha = <q,w>, w -= alpha*q, |w|
int comm_lz_homes(hxv_handle* h, bool real, double2* out[3]);
int finish_create(hxv_handle* h, int device, hxv_handle** out);  // uploads h->host, builds the tile plan (hxv_capi.hip); deletes h on failure
bool lanczos_local_step_available(const hxv_handle* h);
int lanczos_local_step(hxv_handle* h, bool real, const double2* q, double sq, const double2* qm, double sqm, double beta, double2* w,
                       bool sub_alpha, double* alpha, double* nrm_w);
// exchange mode 2 on a split sector: pass A reads the dw part in pieces (one block per rank of origin) and runs on the tile kernel
inline bool dw_part_in_pieces(const hxv_handle* h);
}  // namespace hxv

#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess)                                                                              \
      return hxv::fail(HXV_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                \
  } while (0)

struct hxv_handle {
  hxv::SectorHost host;
  int device = 0;
  hipStream_t stream = nullptr;
  std::vector<void*> allocs;
  hxv::DevSector dev{};
  hxv::TilePlan plan;
  // staging for hxv_apply_host
  double2* d_stage_v = nullptr;
  double2* d_stage_hv = nullptr;
  double2* d_wt = nullptr;  // dw-hop scratch of the tiled kernels (column-group-blocked, tiled_wt_elems())
  int64_t wt_elems = 0;
  // lanczos scratch
  double* d_partials = nullptr;  // [2][RED_BLOCKS]
  double* d_scalars = nullptr;   // [8]
  double2* d_lz[3] = {nullptr, nullptr, nullptr};
  double2* lz_vec[3] = {nullptr, nullptr, nullptr};  // what the drivers use: d_lz[], or (split sector) the slab's home in three gather buffers
  int lz_inplace = 1;              // option "lanczos_inplace": on a split sector the Lanczos vectors live in their slot of a gather buffer (no slab copy per product)
  double* d_lz_partial = nullptr;  // per-workgroup partial sums of the fused Lanczos epilogue
  int64_t lz_partial_n = 0;
  int lz_fused = 1;                // option "lanczos_fused"
  int lz_graph = 1;                // option "lanczos_graph": fixed-length tridiagonalisations run device-only, three iterations per hipGraph
  int eigh_measure_all = 0;        // option "eigh_measure_all": hxv_eigh_lowest measures every projection at every step (round-1 behaviour)
  int64_t eigh_last_full = 0, eigh_last_local = 0;  // Gram-Schmidt passes of the last hxv_eigh_lowest: whole basis / local only
  int eigh_keep_pct = 20;          // option "eigh_keep_pct": share of the basis beyond the wanted pairs that a thick restart keeps
  int eigh_degenerate = 1;         // option "eigh_degenerate": hxv_eigh_lowest looks for further copies of degenerate levels (locking rounds)
  int real_vectors = 1;            // option "real_vectors": device Lanczos drivers use real vectors when H and the start vector are real
  int lz_buf_mode = 0;             // layout the d_lz work vectors were last used in (0 complex, 1 real): the pad rows differ
  int last_real = 0;               // did the last device Lanczos run use real vectors (get_option "lanczos_real_last")
  int kernel = 1;
  // split sector: RCCL communicator over the nranks handles (hxv_comm_init) and the gathered vector
  void* comm = nullptr;          // ncclComm_t
  void* comm_api = nullptr;      // the RCCL entry points that communicator was created with (hxv_comm.cpp: the system's librccl, or HXV_RCCL_LIB)
  void* lgroup = nullptr;        // thread ranks of one process (hxv_comm_init_local): the group object, see hxv_comm.cpp
  const char* xfer_send = nullptr;         // thread ranks: what this rank offers in the column exchange under way (comm_sendrecv_cols)
  const int64_t* xfer_send_ptr = nullptr;  //               and its per-destination offsets
  double2* d_gather = nullptr;   // nranks * cmax * pitch elements (all-gather layout) / (qdw + halo) * pitch (halo layout)
  double2* d_gather_x[2] = {nullptr, nullptr};  // two more of the same for the device Lanczos on a split sector (three vectors rotate)
  double2* gather_cur = nullptr; // the gather buffer the exchange under way / last done runs on (peers of a thread group read it)
  int a2a_overlap = 0;           // option "exchange_overlap": exchange mode 2 runs diagonal + up hops on a second stream while the transposes are under way
  hipStream_t stream2 = nullptr; // that second stream and its two events (created on first use)
  hipEvent_t ov_ev[2] = {nullptr, nullptr};
  void* a2a = nullptr;           // exchange 2 (two all-to-all transposes): panel handle, staging buffers, per-peer offsets (hxv_comm.cpp)
  int64_t n_slab_copy = 0;       // slab copies into a gather buffer (exchange with a vector that is not at home)
  double2* d_send = nullptr;     // halo exchange: packed columns, grouped by destination rank
  int32_t* d_send_cols = nullptr;
  int64_t n_exchange = 0;
  int64_t n_apply = 0;
  std::vector<void*> owned_vectors;      // hxv_vector_alloc'ed and not yet freed (hxv_destroy returns what is left to the buffer cache)
  int64_t h2d_bytes = 0, d2h_bytes = 0;  // vector-sized PCIe traffic of the host-array entry points and hxv_vector_from/to_host (hxv_get_stats)
  int64_t device_bytes = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t kt_ev[4] = {nullptr, nullptr, nullptr, nullptr};  // hxv_time_apply_slab: events around the kernels of a slab product (two regions in exchange mode 2)
  int kt_on = 0;

  template <typename T>
  hipError_t alloc(T** p, size_t n) {
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    hipError_t e = hipMalloc((void**)p, bytes);
    if (e == hipSuccess) {
      allocs.push_back((void*)*p);
      device_bytes += (int64_t)bytes;
    }
    return e;
  }
  template <typename T>
  hipError_t upload(T** p, const std::vector<T>& src) {
    hipError_t e = alloc(p, src.size());
    if (e != hipSuccess) return e;
    if (!src.empty()) e = hipMemcpy(*p, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
    return e;
  }
};

inline bool hxv::dw_part_in_pieces(const hxv_handle* h) { return h->host.exchange == 2 && h->host.nranks > 1; }
