// The opaque handle behind include/hxv.h and the error helpers shared by the translation units
// that implement the C-ABI (hxv_capi.hip: handles + products; hxv_lanczos.hip: Lanczos recurrences; hxv_eigh.hip: thick-restart eigensolver).
#pragma once
#include <algorithm>
#include <atomic>
#include <cstring>
#include <memory>
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
// this rank's slab between a HOST array in the reference's layout (contiguous columns of DimUp, the reference's row order) and a DEVICE vector
// (columns padded to pitch, rows in the device row order with the basis signs of SectorHost::up_perm).  Synchronous: returns when the copy is
// done.  The reference's order on the device: one 2-D copy; a device row order: one contiguous copy + one permuting pass through a buffer from
// the engine's cache.
int slab_from_host(hxv_handle* h, const void* v_host, double2* d_vec);
int slab_to_host(hxv_handle* h, const double2* d_vec, void* v_host);
int finish_create(hxv_handle* h, int device, hxv_handle** out);  // builds the tile plan, uploads the tables of h->img (hxv_capi.hip); deletes h on failure

// Device tables of a sector travel in ONE allocation and ONE host-to-device copy: a sector has ~50 of them (maps, ELL tables, tile
// tables of both spins) and a hipMalloc + synchronous hipMemcpy each cost more than building them.  add() copies the table into a
// host staging image at once (the caller's vector may die) and remembers where the device pointer is to be stored; commit() allocates,
// copies and patches the pointers, which must still be alive then.
struct TableArena {
  struct Item {
    size_t off;
    void** dst;
  };
  std::vector<char> stage;
  std::vector<Item> items;
  template <typename T>
  hipError_t add(const std::vector<T>& v, T** p) {
    const size_t off = (stage.size() + 255) & ~(size_t)255, bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    stage.resize(off + bytes, 0);
    if (!v.empty()) std::memcpy(stage.data() + off, v.data(), v.size() * sizeof(T));
    items.push_back({off, (void**)p});
    *p = nullptr;
    return hipSuccess;
  }
  // -> the base of the allocation (the caller owns it: hipFree), its size in *bytes
  hipError_t commit(void** base, int64_t* bytes) {
    *base = nullptr;
    *bytes = (int64_t)std::max<size_t>(stage.size(), 256);
    hipError_t e = hipMalloc(base, (size_t)*bytes);
    if (e != hipSuccess) return e;
    if (!stage.empty()) e = hipMemcpy(*base, stage.data(), stage.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    for (const Item& it : items) *it.dst = (char*)*base + it.off;
    items.clear();
    std::vector<char>().swap(stage);
    return hipSuccess;
  }
};

// What opening a sector builds and never changes afterwards: the host description (basis maps, one-spin matrices, diagonal tables),
// the tile plan of the default options and every device table of both.  Handles SHARE it (shared_ptr): the reference re-opens the
// same sector for every Green's-function channel (build_Hv_sector / delete_Hv_sector around each sp_lanc_tridiag, ED_GF_NORMAL.f90:208-222),
// and the images of closed sectors stay in a per-process cache keyed by everything that determines them (hxv_capi.hip: sector_cache).
struct SectorImage {
  SectorHost host;
  DevSector dev{};
  TilePlan plan;
  int kernel = 1;                 // 0: too many distinct amplitudes for the tile kernels
  int device = -1;
  std::vector<void*> allocs;      // device allocations (the table arena)
  int64_t device_bytes = 0, host_bytes = 0;
  std::string key;                // full cache key (compared byte for byte); empty: not cacheable (from_csr, panels)
  double us_host = 0, us_plan = 0, us_upload = 0;  // what building it cost (microseconds)
  bool uploaded = false;
  // between the host half of an open (tile plan built, every table staged) and its device half (one allocation, one copy): the staged
  // tables and the device pointers they will be patched into
  struct Pending;
  std::shared_ptr<Pending> pending;
  ~SectorImage();
};
// the cache of closed sectors' images (hxv_cache.cpp); an empty key means "do not cache"
std::string sector_cache_key(const hxv_model& m, int nup, int ndw, int rank, int nranks, int device, int exchange);
std::shared_ptr<SectorImage> sector_cache_find(const std::string& key);
void sector_cache_insert(const std::shared_ptr<SectorImage>& im);
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

namespace hxv {
inline std::atomic<int64_t> g_live_handles{0};  // hxv_live_handles(): handles created and not yet destroyed (leak checks of host programs)
}
struct hxv_handle {
  std::shared_ptr<hxv::SectorImage> img;  // shared, immutable (see SectorImage); `host` below is img->host
  hxv::SectorHost& host;
  explicit hxv_handle(std::shared_ptr<hxv::SectorImage> i = std::make_shared<hxv::SectorImage>()) : img(std::move(i)), host(img->host) { ++hxv::g_live_handles; }
  ~hxv_handle() { --hxv::g_live_handles; }
  hxv_handle(const hxv_handle&) = delete;
  hxv_handle& operator=(const hxv_handle&) = delete;
  double open_us[4] = {0, 0, 0, 0};       // this open: host build, tile plan, upload, whole call (get_option "open_us_*"); 0 0 0 on a cache hit
  int open_cache_hit = 0;
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
  int64_t eigh_last_search = 0, eigh_last_check = 0;  // products of the last hxv_eigh_lowest: the search / the check rounds for hidden copies
  int eigh_fuse_restart = 1;       // option "eigh_fuse_restart": the restart rotation measures the residual vector, the first step of a cycle removes the arrow and measures in one pass
  int eigh_keep_pct = 20;          // option "eigh_keep_pct": share of the basis beyond the wanted pairs that a thick restart keeps
  int eigh_degenerate = 0;         // option "eigh_degenerate": 1 = hxv_eigh_lowest looks for further copies of degenerate levels (locking rounds; C3: +60
                                   // products on 380); 0 [default] = one Krylov space, what ARPACK (the call this replaces) does
  int real_vectors = 1;            // option "real_vectors": device Lanczos drivers use real vectors when H and the start vector are real
  int lz_buf_mode = 0;             // layout the d_lz work vectors were last used in (0 complex, 1 real): the pad rows differ
  int last_real = 0;               // did the last device Lanczos run use real vectors (get_option "lanczos_real_last")
  int kernel = 1;
  // split sector: RCCL communicator over the nranks handles (hxv_comm_init) and the gathered vector
  void* comm = nullptr;          // ncclComm_t
  void* comm_shared = nullptr;   // hxv_comm.cpp ProcComm: the PROCESS-level communicator `comm` belongs to (one ncclCommInitRank per process and
                                 // (nranks, rank, device, library), shared by every sector the process opens; ED_VARS_GLOBAL.f90:365-380 sets MpiComm once
                                 // per solve); null: thread ranks / none
  std::atomic<int> comm_aborted{0};  // hxv_comm_abort has run (from ANOTHER host thread while this rank's own sits in a collective) on `comm` (freed by ncclCommAbort: never destroyed again, never used again)
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
  hipEvent_t kt_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // hxv_time_apply_slab: events around the kernels of a slab product (two regions in exchange mode 2; [4], [5]: pass A on the second stream of its overlapped form)
  int kt_on = 0;
  int64_t last_overlapped_us = 0;  // last hxv_time_apply_slab: mean time of the kernels that ran on the second stream beside the exchange (overlapped mode 2), microseconds

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
