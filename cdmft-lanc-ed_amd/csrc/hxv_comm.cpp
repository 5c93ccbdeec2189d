// The slab exchange behind the C-ABI: one communicator per open (split) sector.
//
// The reference splits the vector along DimDw (ED_HAMILTONIAN.f90:93-105) and re-assembles what a rank needs with
// MPI collectives inside spMatVec_MPI_main (ED_HAMILTONIAN_SPARSE_HxV.f90:272-296, ED_HAMILTONIAN_COMMON.f90:30-94).
// Here the re-assembly is ONE equal-count all-gather of the padded slabs (or the halo exchange: only the columns H_dw
// couples across ranks), on the stream the product runs on, so a Fortran rank needs nothing but this library.
// Two transports serve the same code path:
//   * RCCL over xGMI, one process per GPU: hxv_comm_unique_id (rank 0) -> broadcast the 128 bytes with the host program's
//     own MPI_Bcast -> hxv_comm_init on every rank.  RCCL is loaded with dlopen at the first of these calls: the library has
//     no link-time dependency on it, and a process that already holds a copy (PyTorch ships its own) reuses it.
//   * THREAD RANKS inside one process: hxv_comm_local_create(nranks) -> hxv_comm_init_local(handle, group) from one host
//     thread per rank.  Slabs travel by device-to-device copies ordered with HIP events, scalars through host memory.  The
//     ranks may share one GPU (how the multi-rank code paths -- uneven slabs, halo lists, the drivers' collectives --
//     are executed and tested on a one-GPU box, where RCCL refuses two ranks on one device) or sit on different GPUs of
//     a node driven by one process.
// The dot products of the device Lanczos drivers are all-reduced on the same stream (comm_allreduce_sum), and a rank that
// fails before a collective tells the others first (comm_agree), so that nobody waits for a peer that has already left.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>

#include "hxv_handle.hpp"
#include "hxv_tile_dev.hpp"

using namespace hxv;

namespace {
struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;  // optional: hxv_comm_abort
  std::string path;                              // file the symbols came from (dladdr)
  std::string err;
};
// One symbol table per library.  HXV_RCCL_LIB (read at every hxv_comm_unique_id / hxv_comm_init) names the library to use instead of the
// system's librccl: the test suite points it at tests/rccl_double (thread ranks of ONE process behind RCCL's ten entry points), which is
// how the RCCL branches below execute with several ranks on a one-GPU box.  A communicator keeps the table it was created with
// (hxv_handle::comm_api), so handles of both kinds can live in one process.
Rccl* rccl() {
  static std::mutex mu;
  static std::map<std::string, std::unique_ptr<Rccl>> tables;
  const char* over = getenv("HXV_RCCL_LIB");
  const std::string key = over && over[0] ? over : "";
  std::lock_guard<std::mutex> lk(mu);
  auto it = tables.find(key);
  if (it != tables.end()) return it->second.get();
  std::unique_ptr<Rccl> t(new Rccl());
  Rccl& r = *t;
  if (!key.empty()) {
    r.lib = dlopen(key.c_str(), RTLD_NOW | RTLD_LOCAL);
  } else {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
  }
  if (!r.lib) {
    const char* de = dlerror();
    r.err = std::string("cannot load ") + (key.empty() ? "librccl" : key) + ": " + (de ? de : "?");
  } else {
#define SYM(field, sym)                                           \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, #sym)); \
  if (!r.field) r.err = "the RCCL library lacks " #sym;
    SYM(GetUniqueId, ncclGetUniqueId)
    SYM(CommInitRank, ncclCommInitRank)
    SYM(CommDestroy, ncclCommDestroy)
    SYM(AllGather, ncclAllGather)
    SYM(AllReduce, ncclAllReduce)
    SYM(GetErrorString, ncclGetErrorString)
    SYM(Send, ncclSend)
    SYM(Recv, ncclRecv)
    SYM(GroupStart, ncclGroupStart)
    SYM(GroupEnd, ncclGroupEnd)
#undef SYM
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.lib, "ncclCommAbort"));  // (not required of a library)
    {
      // where the library was found: the first real N>1 run must be able to say WHICH librccl it ran (bench.py: config.rccl_lib)
      Dl_info di;
      if (r.GetUniqueId && dladdr(reinterpret_cast<void*>(r.GetUniqueId), &di) && di.dli_fname) r.path = di.dli_fname;
    }
  }
  return tables.emplace(key, std::move(t)).first->second.get();
}
Rccl* api(const hxv_handle* h) { return static_cast<Rccl*>(h->comm_api); }
int nccl_fail(const Rccl* r, const char* what, ncclResult_t e) {
  return fail(HXV_ERR_HIP, std::string(what) + ": " + (r && r->GetErrorString ? r->GetErrorString(e) : "RCCL error"));
}

// ---- ONE communicator per process, not per sector ---------------------------------------------------------
// The reference sets MpiComm once per solve (ED_VARS_GLOBAL.f90:365-380, ed_set_MpiComm) and derives a sub-communicator only for sectors with
// DimDw < MpiSize (ED_HAMILTONIAN.f90:63-89); its callers open 289 sectors per ED_DIAG sweep (ED_DIAG.f90:142-190) and 56 per Green's-function
// stage (ED_GF_NORMAL.f90:208-222).  ncclCommInitRank is a collective that builds rings and channels over xGMI -- hundreds of milliseconds on a
// node -- so the engine builds it ONCE per (library, nranks, rank, device) and every later hxv_comm_init with the same key binds the handle to
// that communicator (the id it is given is then not consumed: all ranks take the same branch, because all ranks have made the same calls).
// A sector with DimDw < nranks is opened by the first DimDw ranks with nranks' = DimDw: another key, built once as well.
// HXV_COMM_CACHE=0: one communicator per handle, as before round 6.  An aborted communicator leaves the cache.
struct ProcComm {
  ncclComm_t c = nullptr;
  Rccl* api = nullptr;
  int nranks = 0, rank = 0, device = 0;
  bool cached = false;            // lives in g_comms until hxv_comm_cache_clear (else: owned by the one handle that made it)
  std::atomic<int> users{0};      // handles bound to it
  std::atomic<int> aborted{0};
  std::atomic<int> failed{0};     // a collective on it returned an error: torn down with ncclCommAbort (ncclCommDestroy may wait for the lost peer)
};
struct CommCache {
  std::mutex mu;
  std::vector<ProcComm*> all;     // cached communicators of this process
  int64_t inits = 0, reuses = 0;  // ncclCommInitRank calls made / hxv_comm_init calls served by a cached communicator
};
CommCache& comm_cache() {
  static CommCache c;
  return c;
}
bool comm_cache_on() {
  const char* e = getenv("HXV_COMM_CACHE");
  return !(e && e[0] == '0');
}
ProcComm* pc(const hxv_handle* h) { return static_cast<ProcComm*>(h->comm_shared); }
void proc_comm_destroy(ProcComm* p) {
  if (!p) return;
  if (p->c && !p->aborted) {  // (ncclCommAbort has freed an aborted communicator)
    if (p->failed && p->api->CommAbort)
      (void)p->api->CommAbort(p->c);
    else
      (void)p->api->CommDestroy(p->c);
  }
  delete p;
}
// A collective on the handle's communicator has FAILED (a peer aborted, a link went down): the communicator must not serve the next sector.
// It leaves the cache (the next hxv_comm_init builds a new one from its id); the handles bound to it keep it until they are closed.
int comm_failed(hxv_handle* h, const char* what, ncclResult_t e) {
  if (ProcComm* p = pc(h)) {
    p->failed = 1;
    std::lock_guard<std::mutex> lk(comm_cache().mu);
    auto& all = comm_cache().all;
    all.erase(std::remove(all.begin(), all.end(), p), all.end());
  }
  return nccl_fail(api(h), what, e);
}

// ---- thread ranks: the ranks of a sector are host threads of this process -----------------------------------------
struct LocalGroup {
  int n = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t gen = 0;
  std::vector<hxv_handle*> member;       // by rank
  std::vector<hipEvent_t> ready, done;   // by rank: "what I send is in place" / "I have read what the others sent"
  std::vector<std::vector<double>> red;  // by rank: contribution to the running all-reduce
  std::vector<int> flag;                 // by rank: comm_agree
  bool broken = false;                   // a rank dropped out (hxv_comm_local_abort) or a barrier timed out: every wait returns an error
  double timeout_s = 300.0;              // HXV_LOCAL_TIMEOUT_S
  // Every rank calls it; returns 0 when all have (the ranks issue their collectives in the same order, like MPI ranks), non-zero when the
  // group is broken: a peer was aborted, or did not arrive in time (then this rank breaks the group for everybody else).
  int barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (broken) return 1;
    const uint64_t g = gen;
    if (++arrived == n) {
      arrived = 0;
      ++gen;
      cv.notify_all();
      return 0;
    }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s);
    while (gen == g && !broken)
      if (cv.wait_until(lk, deadline) == std::cv_status::timeout && gen == g) {
        broken = true;
        cv.notify_all();
      }
    return gen == g ? 1 : 0;
  }
};
int broken_group() { return fail(HXV_ERR_STATE, "thread-rank group: a peer rank dropped out (aborted, failed or did not arrive in time); the group is unusable"); }
// a HIP call inside a collective of the thread-rank transport: the error is remembered, the rank still takes part in the collective's
// barriers (its peers must not be left waiting) and reports after the last one
#define HIPREC(expr)                                                                                        \
  do {                                                                                                      \
    hipError_t _e = (expr);                                                                                 \
    if (_e != hipSuccess && rc == HXV_OK) rc = fail(HXV_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)
LocalGroup* lg(const hxv_handle* h) { return reinterpret_cast<LocalGroup*>(h->lgroup); }

// bytes of one column of a Lanczos / product vector: complex(8) columns of `pitch` elements or real ones of pitch_real
size_t col_bytes(const hxv_handle* h, bool real) { return real ? (size_t)pitch_real_of(h) * sizeof(double) : (size_t)h->host.pitch * sizeof(double2); }

int ensure_gather(hxv_handle* h, hipStream_t st) {
  const SectorHost& s = h->host;
  if (h->d_gather) return HXV_OK;
  // sized for complex vectors; real ones (half the bytes per column) use the front of the same buffer
  const size_t cb = col_bytes(h, false);
  if (s.exchange == 1) {
    const size_t nfull = (size_t)s.qdw + s.halo_cols.size();
    HIPCHK(pool_alloc(h->device, std::max<size_t>(nfull, 1) * cb, (void**)&h->d_gather));
    HIPCHK(pool_alloc(h->device, std::max<size_t>(s.send_cols.size(), 1) * cb, (void**)&h->d_send));
    HIPCHK(hipMalloc((void**)&h->d_send_cols, std::max<size_t>(s.send_cols.size(), 1) * sizeof(int32_t)));
    HIPCHK(hipMemcpy(h->d_send_cols, s.send_cols.data(), s.send_cols.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    h->device_bytes += (int64_t)((nfull + s.send_cols.size()) * cb);
  } else {
    const size_t bytes = (size_t)s.cmax * s.nranks * cb;
    HIPCHK(pool_alloc(h->device, std::max<size_t>(bytes, 1), (void**)&h->d_gather));
    HIPCHK(hipMemsetAsync(h->d_gather, 0, bytes, st));
    h->device_bytes += (int64_t)bytes;
  }
  return HXV_OK;
}

}  // namespace

namespace hxv {
// Columns of `cb` bytes between the ranks of h's communicator: rank r sends send[send_ptr[p] .. send_ptr[p+1]) to every peer p and
// receives recv[recv_ptr[p] .. recv_ptr[p+1]) from it (offsets in columns; both buffers on the device; asynchronous on st).  The
// halo exchange of the product and the column moves of the spin-dw ladder operators are this one step.
int comm_sendrecv_cols(hxv_handle* h, const void* send_v, const int64_t* send_ptr, void* recv_v, const int64_t* recv_ptr, size_t cb, hipStream_t st) {
  const SectorHost& s = h->host;
  const char* send = reinterpret_cast<const char*>(send_v);
  char* recv = reinterpret_cast<char*>(recv_v);
  if (LocalGroup* G = lg(h)) {
    int rc = HXV_OK;
    h->xfer_send = send;
    h->xfer_send_ptr = send_ptr;
    HIPREC(hipEventRecord(G->ready[s.rank], st));
    if (G->barrier()) return broken_group();
    int bad = 0;
    for (int p = 0; p < s.nranks; ++p) {
      if (p == s.rank) continue;
      const hxv_handle* o = G->member[p];
      const size_t nr = (size_t)(recv_ptr[p + 1] - recv_ptr[p]);
      if ((size_t)(o->xfer_send_ptr[s.rank + 1] - o->xfer_send_ptr[s.rank]) != nr) bad = 1;  // the two ranks' plans disagree
      if (!nr || bad || rc) continue;
      HIPREC(hipStreamWaitEvent(st, G->ready[p], 0));
      HIPREC(hipMemcpyAsync(recv + (size_t)recv_ptr[p] * cb, o->xfer_send + (size_t)o->xfer_send_ptr[s.rank] * cb, nr * cb, hipMemcpyDefault, st));
    }
    HIPREC(hipEventRecord(G->done[s.rank], st));
    if (G->barrier()) return broken_group();
    for (int p = 0; p < s.nranks; ++p)
      if (p != s.rank) HIPREC(hipStreamWaitEvent(st, G->done[p], 0));  // my send buffer is free again once they have read it
    if (rc) return rc;
    if (bad) return fail(HXV_ERR_STATE, "column exchange: a peer's send list does not match this rank's receive list");
    return HXV_OK;
  }
  if (!h->comm) return fail(HXV_ERR_STATE, "column exchange without a communicator");
  Rccl* r = api(h);
  ncclResult_t e = r->GroupStart();
  for (int p = 0; p < s.nranks && e == ncclSuccess; ++p) {
    if (p == s.rank) continue;
    const size_t ns = (size_t)(send_ptr[p + 1] - send_ptr[p]) * cb / sizeof(double), nr = (size_t)(recv_ptr[p + 1] - recv_ptr[p]) * cb / sizeof(double);
    if (ns) e = r->Send(send + (size_t)send_ptr[p] * cb, ns, ncclFloat64, p, (ncclComm_t)h->comm, st);
    if (nr && e == ncclSuccess) e = r->Recv(recv + (size_t)recv_ptr[p] * cb, nr, ncclFloat64, p, (ncclComm_t)h->comm, st);
  }
  ncclResult_t e2 = r->GroupEnd();
  if (e != ncclSuccess || e2 != ncclSuccess) return comm_failed(h, "grouped send/recv", e != ncclSuccess ? e : e2);
  return HXV_OK;
}
}  // namespace hxv

namespace {
// Exchange: this rank's slab d_v_local -> the gathered vector h->d_gather in the layout the kernels expect
// (all-gather layout or halo layout, hxv.h); asynchronous on st.
int exchange(hxv_handle* h, const void* d_v_local, bool real, hipStream_t st) {
  const SectorHost& s = h->host;
  HIPCHK(hipSetDevice(h->device));
  int rc = ensure_gather(h, st);
  if (rc) return rc;
  const size_t cb = col_bytes(h, real);
  // A vector that already sits at the slab's place of one of the handle's gather buffers (hxv_slab_home; the Lanczos vectors of a split
  // sector) is exchanged where it is; anything else is copied into the first buffer.
  const size_t home_off = s.exchange == 1 ? 0 : (size_t)s.rank * s.cmax * cb;
  char* gather = reinterpret_cast<char*>(h->d_gather);
  bool at_home = false;
  for (double2* cand : {h->d_gather, h->d_gather_x[0], h->d_gather_x[1]})
    if (cand && reinterpret_cast<const char*>(cand) + home_off == reinterpret_cast<const char*>(d_v_local)) {
      gather = reinterpret_cast<char*>(cand);
      at_home = true;
    }
  h->gather_cur = reinterpret_cast<double2*>(gather);
  if (!at_home) h->n_slab_copy++;
  LocalGroup* G = lg(h);
  if (s.exchange == 1) {
    // HALO exchange: only the columns H_dw couples to another rank's rows travel -- packed per destination; they land
    // behind the local slab, where the column -> slot table of this layout expects them
    // (a caller that keeps its slab where the exchange wants it -- hxv_slab_home -- saves this copy)
    if (!at_home) HIPCHK(hipMemcpyAsync(gather, d_v_local, (size_t)s.qdw * cb, hipMemcpyDeviceToDevice, st));
    hipError_t pe = launch_pack_columns((const double2*)d_v_local, h->d_send, h->d_send_cols, (int)s.send_cols.size(), (int)(cb / sizeof(double2)), st);
    if (pe != hipSuccess) return fail(HXV_ERR_HIP, std::string("pack kernel: ") + hipGetErrorString(pe));
    const std::vector<int64_t> sp(s.send_ptr.begin(), s.send_ptr.end()), hp(s.halo_ptr.begin(), s.halo_ptr.end());
    rc = comm_sendrecv_cols(h, h->d_send, sp.data(), gather + (size_t)s.qdw * cb, hp.data(), cb, st);
    if (rc) return rc;
    h->n_exchange++;
    return HXV_OK;
  }
  // ALL-GATHER: copy the slab into its slot of the gather buffer, all-gather in place (equal counts: ranks that own one
  // column less leave their last column unused)
  const size_t slot = (size_t)s.cmax * cb;
  char* mine = gather + (size_t)s.rank * slot;
  if (!at_home) HIPCHK(hipMemcpyAsync(mine, d_v_local, (size_t)s.qdw * cb, hipMemcpyDeviceToDevice, st));
  if (G) {
    rc = HXV_OK;
    HIPREC(hipEventRecord(G->ready[s.rank], st));
    if (G->barrier()) return broken_group();
    for (int p = 0; p < s.nranks; ++p) {
      if (p == s.rank || rc) continue;
      HIPREC(hipStreamWaitEvent(st, G->ready[p], 0));
      HIPREC(hipMemcpyAsync(gather + (size_t)p * slot, reinterpret_cast<const char*>(G->member[p]->gather_cur) + (size_t)p * slot, slot, hipMemcpyDefault, st));
    }
    HIPREC(hipEventRecord(G->done[s.rank], st));
    if (G->barrier()) return broken_group();
    for (int p = 0; p < s.nranks; ++p)
      if (p != s.rank) HIPREC(hipStreamWaitEvent(st, G->done[p], 0));
    if (rc) return rc;
  } else {
    ncclResult_t e = api(h)->AllGather(mine, gather, slot / sizeof(double), ncclFloat64, (ncclComm_t)h->comm, st);
    if (e != ncclSuccess) return comm_failed(h, "ncclAllGather", e);
  }
  h->n_exchange++;
  return HXV_OK;
}
}  // namespace

namespace hxv {

bool comm_ready(const hxv_handle* h) {
  const bool gone = h->comm_aborted || (h->comm_shared && static_cast<const ProcComm*>(h->comm_shared)->aborted);
  return (h->comm != nullptr && !gone) || h->lgroup != nullptr;
}

// Does p point into one of the handle's gather buffers (hxv_slab_home hands out a slot of the first)?  The device Lanczos drivers zero
// the slab's place in all three before they read their start vector: a start vector that lives there is staged first.
bool comm_in_gather(const hxv_handle* h, const void* p) {
  const SectorHost& s = h->host;
  const size_t cbc = col_bytes(h, false);
  const size_t bytes = s.exchange == 1 ? std::max<size_t>((size_t)s.qdw + s.halo_cols.size(), 1) * cbc : std::max<size_t>((size_t)s.cmax * s.nranks * cbc, 1);
  for (const double2* b : {h->d_gather, h->d_gather_x[0], h->d_gather_x[1]})
    if (b && reinterpret_cast<const char*>(p) >= reinterpret_cast<const char*>(b) && reinterpret_cast<const char*>(p) < reinterpret_cast<const char*>(b) + bytes) return true;
  return false;
}

// Three places for the slab of a split sector's Lanczos vectors: its slot in the handle's gather buffer and in two more of the same
// size, so that every vector of the three-term recurrence is exchanged where it lies (no slab copy per product).  The slab regions
// are zeroed (pad rows must be zero; `real`: layout of the coming run).  HXV_OK with out[] set, or an error (the caller falls back to
// slab buffers of its own on HXV_ERR_HIP from the allocation).
int comm_lz_homes(hxv_handle* h, bool real, double2* out[3]) {
  const SectorHost& s = h->host;
  HIPCHK(hipSetDevice(h->device));
  int rc = ensure_gather(h, h->stream);
  if (rc) return rc;
  const size_t cbc = col_bytes(h, false);
  const size_t bytes = s.exchange == 1 ? std::max<size_t>((size_t)s.qdw + s.halo_cols.size(), 1) * cbc : std::max<size_t>((size_t)s.cmax * s.nranks * cbc, 1);
  {
    // Three whole gather buffers are the right trade only where they fit comfortably (in all-gather layout each is a whole-Dim vector,
    // and the sectors that get split are the large ones): with less than the missing buffers + 1/16 of the device free (the engine's
    // cache counted as free) the caller's fall-back runs -- three SLAB buffers, one slab copy per product.
    size_t need = 0;
    for (const double2* p : h->d_gather_x)
      if (!p) need += bytes;
    if (need) {
      size_t fr = 0, tot = 0;
      int64_t cached = 0;
      HIPCHK(hipMemGetInfo(&fr, &tot));
      (void)hxv_pool_stats(h->device, &cached, nullptr, nullptr);
      if (fr + (size_t)cached < need + tot / 16) return fail(HXV_ERR_HIP, "gather buffers of the device Lanczos: not enough free device memory for the in-place layout");
    }
  }
  for (auto& p : h->d_gather_x)
    if (!p) {
      hipError_t e = pool_alloc(h->device, bytes, (void**)&p);
      if (e != hipSuccess) {
        // not enough memory for three whole gather buffers (the sectors that get split are the large ones): give back what this call
        // obtained, so that the caller's fall-back -- three SLAB buffers and one slab copy per product -- starts from a clean slate
        p = nullptr;
        (void)hipGetLastError();
        for (auto& q : h->d_gather_x)
          if (q) {
            pool_free(h->device, q);
            q = nullptr;
            h->device_bytes -= (int64_t)bytes;
          }
        return fail(HXV_ERR_HIP, std::string("gather buffers of the device Lanczos: ") + hipGetErrorString(e));
      }
      h->device_bytes += (int64_t)bytes;
    }
  const size_t cb = col_bytes(h, real);
  const size_t home_off = s.exchange == 1 ? 0 : (size_t)s.rank * s.cmax * cb;
  double2* base[3] = {h->d_gather, h->d_gather_x[0], h->d_gather_x[1]};
  for (int i = 0; i < 3; ++i) {
    out[i] = reinterpret_cast<double2*>(reinterpret_cast<char*>(base[i]) + home_off);
    HIPCHK(hipMemsetAsync(out[i], 0, (size_t)std::max(s.qdw, 1) * col_bytes(h, false), h->stream));  // (the complex slab's extent covers the real one's)
  }
  return HXV_OK;
}

int comm_allreduce_sum(hxv_handle* h, double* d_buf, size_t count, hipStream_t st) {
  if (LocalGroup* G = lg(h)) {
    // through host memory, summed in rank order on every rank: the same bits everywhere
    int rc = HXV_OK;
    const int r = h->host.rank;
    std::vector<double> mine(count, 0.0);
    HIPREC(hipMemcpyAsync(mine.data(), d_buf, count * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPREC(hipStreamSynchronize(st));
    G->red[r] = mine;
    if (G->barrier()) return broken_group();
    std::vector<double> tot(count, 0.0);
    for (int p = 0; p < G->n; ++p)
      for (size_t i = 0; i < count && i < G->red[p].size(); ++i) tot[i] += G->red[p][i];
    if (G->barrier()) return broken_group();  // (everybody has read every contribution before anybody overwrites its own)
    HIPREC(hipMemcpyAsync(d_buf, tot.data(), count * sizeof(double), hipMemcpyHostToDevice, st));
    HIPREC(hipStreamSynchronize(st));
    return rc;
  }
  if (!h->comm) return HXV_OK;  // serial: nothing to add
  ncclResult_t e = api(h)->AllReduce(d_buf, d_buf, count, ncclFloat64, ncclSum, (ncclComm_t)h->comm, st);
  if (e != ncclSuccess) return comm_failed(h, "ncclAllReduce", e);
  return HXV_OK;
}

// Collective error agreement: every rank passes its local status; all return non-zero if any rank failed.  Called by the
// drivers after their rank-local preparations (allocations, argument checks) and BEFORE their first collective, so that a
// rank that cannot go on does not leave its peers waiting inside an all-reduce.
int comm_agree(hxv_handle* h, int rc_local) {
  if (!comm_ready(h)) return rc_local;
  int worst = rc_local;
  if (LocalGroup* G = lg(h)) {
    G->flag[h->host.rank] = rc_local;
    if (G->barrier()) return rc_local ? rc_local : broken_group();
    for (int p = 0; p < G->n; ++p)
      if (G->flag[p] != 0 && worst == 0) worst = G->flag[p];
    if (G->barrier()) return rc_local ? rc_local : broken_group();
  } else {
    double v = rc_local ? 1.0 : 0.0;
    hipError_t e = hipMemcpyAsync(h->d_scalars + 7, &v, sizeof(double), hipMemcpyHostToDevice, h->stream);
    ncclResult_t ne = ncclSuccess;
    if (e == hipSuccess) ne = api(h)->AllReduce(h->d_scalars + 7, h->d_scalars + 7, 1, ncclFloat64, ncclMax, (ncclComm_t)h->comm, h->stream);
    if (e == hipSuccess && ne == ncclSuccess) e = hipMemcpyAsync(&v, h->d_scalars + 7, sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && ne == ncclSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess || ne != ncclSuccess) return fail(HXV_ERR_HIP, "comm_agree: the status all-reduce failed");
    if (v != 0.0 && worst == 0) worst = HXV_ERR_STATE;
  }
  if (worst != 0 && rc_local == 0) return fail(worst, "a peer rank of this sector reported an error before the collective step: all ranks stop");
  return worst;
}

// d_hv_local = (H v)|slab with v given as this rank's slab: exchange, then the product on the gathered vector.
// `ep`: optional Lanczos epilogue of pass A (fused recurrence on a split sector: the partial sums are this rank's share).
// ---------------------------------------------------------------------------------------------------------------------------------
// Exchange 2: the reference's own scheme (spMatVec_mpi_main, ED_HAMILTONIAN_SPARSE_HxV.f90:272-296; vector_transpose_MPI,
// ED_HAMILTONIAN_COMMON.f90:30-94) -- transpose the slab to row panels, dw hops on the panel, transpose back, then diagonal + up hops +
// the assembled dw part on the slab.  Each transpose moves (P-1)/P of ONE slab per rank instead of (P-1) slabs: the lowest-traffic
// exchange (C3 at 8 ranks: 0.58 GB into a GPU per product against 2.32 GB for the all-gather).  Rows are split like the columns
// (mpiQup rule, :274-275).  Blocks are packed / unpacked with strided device copies; the block a rank keeps never leaves its GPU.
// hxv_time_apply_slab: events around the kernels of a slab product
static inline void kt_mark(hxv_handle* h, int i, hipStream_t st) {
  if (h->kt_on && h->kt_ev[i]) (void)hipEventRecord(h->kt_ev[i], st);
}

struct A2A {
  hxv_handle* panel = nullptr;
  std::vector<int> rn, ru0, cq, cc0;         // per rank: its rows (count, first) and columns (count, first)
  std::vector<int64_t> sp1, rp1, sp2, rp2;   // [P+1] element offsets of the per-peer blocks in the send / receive buffers, both transposes
  std::vector<int64_t> pan;                  // [P+1] element offsets of the ranks' column ranges in an UNPADDED panel (direct receive / send)
  double2 *d_send = nullptr, *d_recv = nullptr, *d_x = nullptr, *d_y = nullptr;
  WtRange* d_wtr[2] = {nullptr, nullptr};    // [P] where pass A finds the dw part after the second transpose, per rank of origin (complex / real layout)
  int pp = 0;                                // panel pitch (complex layout)
  int mode = 0;                              // layout the panel / dw-part buffers were last used in (0 complex, 1 real): the pad rows differ
};

static void a2a_release(hxv_handle* h) {
  A2A* a = static_cast<A2A*>(h->a2a);
  if (!a) return;
  for (double2* p : {a->d_send, a->d_recv, a->d_x, a->d_y})
    if (p) pool_free(h->device, p);
  for (WtRange* p : a->d_wtr)
    if (p) (void)hipFree(p);
  if (a->panel) (void)hxv_destroy(a->panel);
  delete a;
  h->a2a = nullptr;
}

static int ensure_a2a(hxv_handle* h, hipStream_t st) {
  if (h->a2a) return HXV_OK;
  const SectorHost& s = h->host;
  const int P = s.nranks;
  std::unique_ptr<A2A> a(new A2A());
  a->rn.resize(P);
  a->ru0.resize(P);
  a->cq.resize(P);
  a->cc0.resize(P);
  // Rows are dealt in units of 16 where every rank still gets some: a rank whose row count is a multiple of the panel's pitch granularity
  // (8 complex / 16 real elements) has an UNPADDED panel, receives its blocks of the first transpose straight into it and sends the
  // blocks of the second straight out of it -- two of the four strided copies per product go away.
  const int units = (s.dimup + 15) / 16;
  for (int p = 0; p < P; ++p) {
    if (units >= P) {
      int nu, u0;
      dw_split(units, p, P, nu, u0);
      a->ru0[p] = std::min(16 * u0, s.dimup);
      a->rn[p] = std::min(16 * (u0 + nu), s.dimup) - a->ru0[p];
    } else {
      dw_split(s.dimup, p, P, a->rn[p], a->ru0[p]);
    }
    dw_split(s.dimdw, p, P, a->cq[p], a->cc0[p]);
  }
  const int nme = a->rn[s.rank], qme = s.qdw;
  if (nme < 1) return fail(HXV_ERR_UNSUPPORTED, "all-to-all exchange: a rank without rows (nranks > DimUp)");
  a->sp1.assign(P + 1, 0);
  a->rp1.assign(P + 1, 0);
  a->sp2.assign(P + 1, 0);
  a->rp2.assign(P + 1, 0);
  for (int p = 0; p < P; ++p) {
    const bool self = p == s.rank;
    a->sp1[p + 1] = a->sp1[p] + (self ? 0 : (int64_t)qme * a->rn[p]);   // my columns, p's rows
    a->rp1[p + 1] = a->rp1[p] + (self ? 0 : (int64_t)a->cq[p] * nme);   // p's columns, my rows
    a->sp2[p + 1] = a->sp2[p] + (self ? 0 : (int64_t)a->cq[p] * nme);   // (the way back: the same blocks, roles swapped)
    a->rp2[p + 1] = a->rp2[p] + (self ? 0 : (int64_t)qme * a->rn[p]);
  }
  a->pan.assign(P + 1, 0);
  for (int p = 0; p < P; ++p) a->pan[p + 1] = a->pan[p] + (int64_t)a->cq[p] * nme;
  HIPCHK(hipSetDevice(h->device));
  hxv_handle* ph = new hxv_handle();
  std::string e = make_panel_host(s, nme, ph->host);
  if (!e.empty()) {
    delete ph;
    return fail(HXV_ERR_STATE, "all-to-all exchange: " + e);
  }
  hxv_handle* out = nullptr;
  int rc = finish_create(ph, h->device, &out);  // (deletes ph on failure)
  if (rc) return rc;
  a->panel = out;
  a->pp = out->host.pitch;
  const size_t nsend = (size_t)std::max<int64_t>(std::max(a->sp1[P], a->sp2[P]), 1), nrecv = (size_t)std::max<int64_t>(std::max(a->rp1[P], a->rp2[P]), 1);
  const size_t npanel = (size_t)s.dimdw * a->pp;
  struct { double2** p; size_t n; } bufs[4] = {{&a->d_send, nsend}, {&a->d_recv, nrecv}, {&a->d_x, npanel}, {&a->d_y, npanel}};
  for (auto& b : bufs) {
    hipError_t ea = pool_alloc(h->device, b.n * sizeof(double2), (void**)b.p);
    if (ea != hipSuccess) {
      h->a2a = a.release();
      a2a_release(h);
      return fail(HXV_ERR_HIP, std::string("all-to-all exchange buffers: ") + hipGetErrorString(ea));
    }
    h->device_bytes += (int64_t)(b.n * sizeof(double2));
  }
  // pad rows of the panel are never written by the unpack copies: zero them once
  HIPCHK(hipMemsetAsync(a->d_x, 0, npanel * sizeof(double2), st));
  // Where the dw part of this rank's slab lies after the second transpose, per rank of origin: the blocks the peers sent stay in the
  // receive buffer ([q columns][rn[p] rows] each), the block this rank kept stays in its panel output (column range cc0[me].., pitch =
  // panel pitch).  Pass A reads them there (WtRange): no unpack copies, no assembled copy of the dw part.
  for (int real = 0; real < 2; ++real) {
    const size_t esz = real ? sizeof(double) : sizeof(double2);
    const int64_t ppr = real ? (int64_t)pitch_real_of(out) : (int64_t)a->pp;
    std::vector<WtRange> tab(P);
    for (int p = 0; p < P; ++p) {
      tab[p].row0 = a->ru0[p];
      tab[p].row1 = a->ru0[p] + a->rn[p];
      if (p == s.rank) {
        tab[p].stride = ppr;
        tab[p].base = reinterpret_cast<const char*>(a->d_y) + (size_t)a->cc0[p] * ppr * esz;
      } else {
        tab[p].stride = a->rn[p];
        tab[p].base = reinterpret_cast<const char*>(a->d_recv) + (size_t)a->rp2[p] * esz;
      }
    }
    hipError_t ea = hipMalloc((void**)&a->d_wtr[real], P * sizeof(WtRange));
    if (ea == hipSuccess) ea = hipMemcpy(a->d_wtr[real], tab.data(), P * sizeof(WtRange), hipMemcpyHostToDevice);
    if (ea != hipSuccess) {
      h->a2a = a.release();
      a2a_release(h);
      return fail(HXV_ERR_HIP, std::string("all-to-all exchange tables: ") + hipGetErrorString(ea));
    }
  }
  h->a2a = a.release();
  return HXV_OK;
}

// (rows [u0,u0+n) of `ncols` columns of pitch `pitch`) <-> contiguous [ncols][n]; esz = bytes per element (16 complex, 8 real)
static hipError_t copy_block(void* dst, size_t dpitch, const void* src, size_t spitch, int n, int ncols, size_t esz, hipStream_t st) {
  if (n <= 0 || ncols <= 0) return hipSuccess;
  return hipMemcpy2DAsync(dst, dpitch * esz, src, spitch * esz, (size_t)n * esz, (size_t)ncols, hipMemcpyDeviceToDevice, st);
}

// `real`: the slabs are double[qdw][pitch_real] (REAL-vector mode of the drivers: half the bytes in every step below)
static int apply_slab_a2a(hxv_handle* h, const void* v_, void* hv_, bool real, hipStream_t st, const LzEpilogue* ep) {
  const SectorHost& s = h->host;
  if (!comm_ready(h)) return fail(HXV_ERR_STATE, "split sector without a communicator: call hxv_comm_init after opening the sector");
  HIPCHK(hipSetDevice(h->device));
  int rc = ensure_a2a(h, st);
  if (rc) return rc;
  A2A& a = *static_cast<A2A*>(h->a2a);
  const int P = s.nranks, me = s.rank, nme = a.rn[me], q = s.qdw;
  const size_t esz = real ? sizeof(double) : sizeof(double2);
  const size_t pit = real ? (size_t)pitch_real_of(h) : (size_t)s.pitch;          // slab pitch
  const size_t pp = real ? (size_t)pitch_real_of(a.panel) : (size_t)a.pp;         // panel pitch
  if (a.mode != (real ? 1 : 0)) {
    // the pad rows of the two layouts sit at different places and the unpack copies never write pads
    HIPCHK(hipMemsetAsync(a.d_x, 0, (size_t)s.dimdw * a.pp * sizeof(double2), st));
    a.mode = real ? 1 : 0;
  }
  const char* v = static_cast<const char*>(v_);
  char *send = reinterpret_cast<char*>(a.d_send), *recv = reinterpret_cast<char*>(a.d_recv), *x = reinterpret_cast<char*>(a.d_x),
       *y = reinterpret_cast<char*>(a.d_y);
  // OVERLAPPED form (option "exchange_overlap", plain products): the reference computes the diagonal and the up hops before its first
  // transpose (ED_HAMILTONIAN_SPARSE_HxV.f90:250-270, then :272-296); here they run on a second stream WHILE pack, transpose, panel
  // product and transpose back are under way, and the dw part is added at the end (hv += pieces: 32 B per local state more than the
  // fused form, which reads the pieces as pass A's accumulator init).  Pays where a transpose takes longer than pass A on the slab.
  const bool overlap = h->a2a_overlap && !ep && h->plan.usable;
  //  every error return below the launch on the second stream waits for it first: the caller may free or reuse v / hv at once
  struct JoinSecondStream {
    hipStream_t s2 = nullptr;
    ~JoinSecondStream() {
      if (s2) (void)hipStreamSynchronize(s2);
    }
  } join2;
  if (overlap) {
    if (!h->stream2) HIPCHK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
    for (auto& e : h->ov_ev)
      if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(hipEventRecord(h->ov_ev[0], st));                 // v is ready (and the previous user of hv is done) on st
    HIPCHK(hipStreamWaitEvent(h->stream2, h->ov_ev[0], 0));
    kt_mark(h, 4, h->stream2);
    hipError_t eo;
    if (real) {
      DevSector d = h->dev;
      d.pitch = (int)pit;
      eo = launch_hxv_tiled_real(d, h->plan, reinterpret_cast<const double*>(v) - (int64_t)d.slab0 * d.pitch, nullptr, static_cast<double*>(hv_), h->stream2, nullptr, 1, true);
    } else {
      eo = launch_hxv_tiled(h->dev, h->plan, reinterpret_cast<const double2*>(v) - (int64_t)h->dev.slab0 * h->dev.pitch, nullptr, static_cast<double2*>(hv_), h->stream2,
                            nullptr, 1, true);
    }
    if (eo != hipSuccess) return fail(HXV_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(eo));
    kt_mark(h, 5, h->stream2);
    join2.s2 = h->stream2;  // from here on pass A is writing hv on the second stream: no return may leave it unjoined
    HIPCHK(hipEventRecord(h->ov_ev[1], h->stream2));
  }
  // 1. my slab cut by the receivers' row ranges; my own block goes straight into the panel
  for (int p = 0; p < P; ++p) {
    if (p == me)
      HIPCHK(copy_block(x + (size_t)a.cc0[me] * pp * esz, pp, v + (size_t)a.ru0[me] * esz, pit, nme, q, esz, st));
    else
      HIPCHK(copy_block(send + (size_t)a.sp1[p] * esz, a.rn[p], v + (size_t)a.ru0[p] * esz, pit, a.rn[p], q, esz, st));
  }
  const bool direct = pp == (size_t)nme;  // unpadded panel: a peer's block IS its column range of the panel
  rc = direct ? comm_sendrecv_cols(h, send, a.sp1.data(), x, a.pan.data(), esz, st) : comm_sendrecv_cols(h, send, a.sp1.data(), recv, a.rp1.data(), esz, st);
  if (rc) return rc;
  if (!direct)
    for (int p = 0; p < P; ++p)
      if (p != me) HIPCHK(copy_block(x + (size_t)a.cc0[p] * pp * esz, pp, recv + (size_t)a.rp1[p] * esz, nme, nme, a.cq[p], esz, st));
  // 2. dw hops on the row panel [my rows] x [all columns]
  kt_mark(h, 0, st);
  if (real) {
    if (!a.panel->plan.usable) return fail(HXV_ERR_UNSUPPORTED, "all-to-all exchange: tiled kernels unavailable on the panel");
    DevSector d = a.panel->dev;
    d.pitch = (int)pp;
    hipError_t e = launch_hxv_tiled_real(d, a.panel->plan, reinterpret_cast<const double*>(x), reinterpret_cast<double*>(y), nullptr, st, nullptr, 2, true);
    if (e != hipSuccess) return fail(HXV_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  } else {
    rc = hxv_apply_dw_panel(a.panel, a.d_x, a.d_y, st);
    if (rc) return rc;
  }
  kt_mark(h, 1, st);
  // 3. back to the column owners: peers' blocks go out of the panel output (direct) or through the send buffer; what arrives STAYS in the
  //    receive buffer and the block this rank keeps stays in the panel output -- pass A reads the pieces where they are (a.d_wtr)
  if (!direct)
    for (int p = 0; p < P; ++p)
      if (p != me) HIPCHK(copy_block(send + (size_t)a.sp2[p] * esz, nme, y + (size_t)a.cc0[p] * pp * esz, pp, nme, a.cq[p], esz, st));
  rc = direct ? comm_sendrecv_cols(h, y, a.pan.data(), recv, a.rp2.data(), esz, st) : comm_sendrecv_cols(h, send, a.sp2.data(), recv, a.rp2.data(), esz, st);
  if (rc) return rc;
  if (overlap) {
    // 4'. the dw part joins what the second stream has computed meanwhile
    HIPCHK(hipStreamWaitEvent(st, h->ov_ev[1], 0));
    join2.s2 = nullptr;  // joined in stream order: st now waits for pass A
    kt_mark(h, 2, st);
    hipError_t ea = launch_add_pieces(hv_, a.d_wtr[real ? 1 : 0], P, s.dimup, (int)pit, q, real, st);
    if (ea != hipSuccess) return fail(HXV_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(ea));
    kt_mark(h, 3, st);
    h->n_apply++;
    h->n_exchange += 2;
    return HXV_OK;
  }
  // 4. diagonal + up hops + the assembled dw part on the slab (pass A alone, with the Lanczos epilogue when asked for)
  if (!h->plan.usable) return fail(HXV_ERR_UNSUPPORTED, "all-to-all exchange: tiled kernels unavailable (too many distinct amplitudes)");
  hipError_t e;
  kt_mark(h, 2, st);
  if (real) {
    DevSector d = h->dev;
    d.pitch = (int)pit;
    const double* vbase = reinterpret_cast<const double*>(v) - (int64_t)d.slab0 * d.pitch;  // (pass A addresses its slab as column slots slab0..)
    e = launch_hxv_tiled_real(d, h->plan, vbase, reinterpret_cast<double*>(recv), static_cast<double*>(hv_), st, ep, 1, true, a.d_wtr[1], P);
  } else {
    const double2* vbase = reinterpret_cast<const double2*>(v) - (int64_t)h->dev.slab0 * h->dev.pitch;
    e = launch_hxv_tiled(h->dev, h->plan, vbase, a.d_recv, static_cast<double2*>(hv_), st, ep, 1, true, a.d_wtr[0], P);
  }
  if (e != hipSuccess) return fail(HXV_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  kt_mark(h, 3, st);
  h->n_apply++;
  h->n_exchange += 2;
  return HXV_OK;
}

int apply_slab(hxv_handle* h, const double2* d_v_local, double2* d_hv_local, hipStream_t st, const LzEpilogue* ep) {
  const SectorHost& s = h->host;
  if (s.exchange == 2 && s.nranks > 1) return apply_slab_a2a(h, d_v_local, d_hv_local, false, st, ep);
  const double2* vfull = d_v_local;
  if (s.nranks != 1 || comm_ready(h)) {
    if (!comm_ready(h)) return fail(HXV_ERR_STATE, "split sector without a communicator: call hxv_comm_init after opening the sector");
    int rc = exchange(h, d_v_local, false, st);
    if (rc) return rc;
    vfull = h->gather_cur;
  }
  kt_mark(h, 0, st);
  if (!ep) {
    const int rcp = hxv_apply_device(h, vfull, d_hv_local, st);
    kt_mark(h, 1, st);
    return rcp;
  }
  int rcw = ensure_wt(h);
  if (rcw) return rcw;
  hipError_t e = launch_hxv_tiled(h->dev, h->plan, vfull, h->d_wt, d_hv_local, st, ep);
  if (e != hipSuccess) return fail(HXV_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  kt_mark(h, 1, st);
  h->n_apply++;
  return HXV_OK;
}

// the same on REAL vectors (double[qdw][pitch_real] slabs; half the bytes on the links)
int apply_slab_real(hxv_handle* h, const double* d_v_local, double* d_hv_local, hipStream_t st, const LzEpilogue* ep) {
  const SectorHost& s = h->host;
  if (s.exchange == 2 && s.nranks > 1) return apply_slab_a2a(h, d_v_local, d_hv_local, true, st, ep);
  const double* vfull = d_v_local;
  if (s.nranks != 1 || comm_ready(h)) {
    if (!comm_ready(h)) return fail(HXV_ERR_STATE, "split sector without a communicator: call hxv_comm_init after opening the sector");
    int rc = exchange(h, d_v_local, true, st);
    if (rc) return rc;
    vfull = reinterpret_cast<const double*>(h->gather_cur);
  }
  int rcw = ensure_wt(h);
  if (rcw) return rcw;
  DevSector d = h->dev;
  d.pitch = pitch_real_of(h);
  hipError_t e = launch_hxv_tiled_real(d, h->plan, vfull, (double*)h->d_wt, d_hv_local, st, ep);
  if (e != hipSuccess) return fail(HXV_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  h->n_apply++;
  return HXV_OK;
}

void comm_release(hxv_handle* h) {
  if (h->comm) {
    // the communicator belongs to the process (cache) or to this handle alone (HXV_COMM_CACHE=0 / taken out of the cache by an abort)
    ProcComm* p = pc(h);
    if (p) {
      const int left = --p->users;
      bool in_cache = false;
      if (p->cached) {
        std::lock_guard<std::mutex> lk(comm_cache().mu);
        for (ProcComm* q : comm_cache().all) in_cache = in_cache || q == p;
      }
      if (!in_cache && left == 0) proc_comm_destroy(p);
    }
    h->comm = nullptr;
    h->comm_shared = nullptr;
    h->comm_api = nullptr;
    h->comm_aborted = 0;
  }
  if (LocalGroup* G = lg(h)) {
    // (the group object itself belongs to whoever created it: hxv_comm_local_destroy)
    std::lock_guard<std::mutex> lk(G->mu);
    if (h->host.rank < (int)G->member.size() && G->member[h->host.rank] == h) G->member[h->host.rank] = nullptr;
    h->lgroup = nullptr;
  }
  if (h->d_gather) {
    pool_free(h->device, h->d_gather);
    h->d_gather = nullptr;
  }
  for (auto& p : h->d_gather_x)
    if (p) {
      pool_free(h->device, p);
      p = nullptr;
    }
  h->gather_cur = nullptr;
  for (auto& p : h->lz_vec) p = nullptr;
  a2a_release(h);
  if (h->d_send) {
    pool_free(h->device, h->d_send);
    h->d_send = nullptr;
  }
  if (h->d_send_cols) {
    (void)hipFree(h->d_send_cols);
    h->d_send_cols = nullptr;
  }
}

}  // namespace hxv

extern "C" {

int hxv_comm_unique_id(void* id128) {
  if (!id128) return fail(HXV_ERR_ARG, "hxv_comm_unique_id: NULL");
  Rccl* r = rccl();
  if (!r->err.empty()) return fail(HXV_ERR_UNSUPPORTED, r->err);
  ncclUniqueId id;
  ncclResult_t e = r->GetUniqueId(&id);
  if (e != ncclSuccess) return nccl_fail(r, "ncclGetUniqueId", e);
  static_assert(sizeof(id) == HXV_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  std::memcpy(id128, &id, sizeof(id));
  return HXV_OK;
}

int hxv_comm_init(hxv_handle* h, const void* id128) {
  if (!h || !id128) return fail(HXV_ERR_ARG, "hxv_comm_init: NULL");
  if (comm_ready(h)) return fail(HXV_ERR_STATE, "hxv_comm_init: the handle already has a communicator");
  if (h->host.panel_rows > 0) return fail(HXV_ERR_STATE, "hxv_comm_init: panel handles take no communicator");
  Rccl* r = rccl();
  if (!r->err.empty()) return fail(HXV_ERR_UNSUPPORTED, r->err);
  HIPCHK(hipSetDevice(h->device));
  const bool use_cache = comm_cache_on();
  CommCache& cc = comm_cache();
  if (use_cache) {
    std::lock_guard<std::mutex> lk(cc.mu);
    for (ProcComm* p : cc.all)
      if (p->api == r && p->nranks == h->host.nranks && p->rank == h->host.rank && p->device == h->device && !p->aborted) {
        ++p->users;
        ++cc.reuses;
        h->comm = p->c;
        h->comm_shared = p;
        h->comm_api = r;
        return HXV_OK;
      }
  }
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  ncclResult_t e = r->CommInitRank(&c, h->host.nranks, id, h->host.rank);
  if (e != ncclSuccess) return nccl_fail(r, "ncclCommInitRank", e);
  ProcComm* p = new ProcComm();
  p->c = c;
  p->api = r;
  p->nranks = h->host.nranks;
  p->rank = h->host.rank;
  p->device = h->device;
  p->cached = use_cache;
  p->users = 1;
  {
    std::lock_guard<std::mutex> lk(cc.mu);
    ++cc.inits;
    if (use_cache) cc.all.push_back(p);
  }
  h->comm = c;
  h->comm_shared = p;
  h->comm_api = r;
  return HXV_OK;
}

// The cached communicators of this process (see ProcComm): *entries alive, *inits = ncclCommInitRank calls made since the process started,
// *reuses = hxv_comm_init calls that bound a handle to a communicator that existed already.  Any out may be NULL.
int hxv_comm_cache_stats(int64_t* entries, int64_t* inits, int64_t* reuses) {
  CommCache& cc = comm_cache();
  std::lock_guard<std::mutex> lk(cc.mu);
  if (entries) *entries = (int64_t)cc.all.size();
  if (inits) *inits = cc.inits;
  if (reuses) *reuses = cc.reuses;
  return HXV_OK;
}

// ncclCommDestroy of every cached communicator no handle is bound to (collective in the RCCL sense: every rank of a communicator calls it at the
// same point of the program -- the end of a solve); communicators still in use stay.  Returns the number destroyed through *destroyed.
int hxv_comm_cache_clear(int64_t* destroyed) {
  CommCache& cc = comm_cache();
  std::vector<ProcComm*> gone;
  {
    std::lock_guard<std::mutex> lk(cc.mu);
    std::vector<ProcComm*> keep;
    for (ProcComm* p : cc.all) (p->users == 0 ? gone : keep).push_back(p);
    cc.all.swap(keep);
  }
  for (ProcComm* p : gone) {
    (void)hipSetDevice(p->device);
    proc_comm_destroy(p);
  }
  if (destroyed) *destroyed = (int64_t)gone.size();
  return HXV_OK;
}

int hxv_comm_local_create(int32_t nranks, void** group) {
  if (nranks < 1 || !group) return fail(HXV_ERR_ARG, "hxv_comm_local_create: bad argument");
  LocalGroup* G = new LocalGroup();
  G->n = nranks;
  G->member.assign(nranks, nullptr);
  G->ready.assign(nranks, nullptr);
  G->done.assign(nranks, nullptr);
  G->red.resize(nranks);
  G->flag.assign(nranks, 0);
  if (const char* t = getenv("HXV_LOCAL_TIMEOUT_S")) {
    const double v = atof(t);
    if (v > 0.0) G->timeout_s = v;
  }
  *group = G;
  return HXV_OK;
}

// A rank of the group cannot go on (its host thread failed outside the library): wake every peer that waits in a collective and make
// every later collective of the group return HXV_ERR_STATE.  Callable from any thread, any number of times.
int hxv_comm_local_abort(void* group) {
  LocalGroup* G = reinterpret_cast<LocalGroup*>(group);
  if (!G) return fail(HXV_ERR_ARG, "hxv_comm_local_abort: NULL");
  std::lock_guard<std::mutex> lk(G->mu);
  G->broken = true;
  G->cv.notify_all();
  return HXV_OK;
}

// A rank of the communicator has failed OUTSIDE the library: wake this handle's collectives instead of leaving them waiting for it.
// RCCL: ncclCommAbort (callable from another host thread while this rank's thread sits in a collective; the communicator is gone
// afterwards -- later calls on the handle report a missing communicator, hxv_comm_free / hxv_destroy still clean up).  Thread ranks:
// the whole group is marked broken, like hxv_comm_local_abort.
int hxv_comm_abort(hxv_handle* h) {
  if (!h) return HXV_OK;
  if (LocalGroup* G = lg(h)) {
    std::lock_guard<std::mutex> lk(G->mu);
    G->broken = true;
    G->cv.notify_all();
    return HXV_OK;
  }
  // (not to be raced with hxv_destroy / hxv_comm_free of the SAME handle: the caller joins or serialises -- hxv/engine.py takes a per-sector
  //  lock around both; the handle's fields are read once, up front)
  void* const comm = h->comm;
  ProcComm* const p = pc(h);
  Rccl* const r = api(h);
  if (!comm || h->comm_aborted || (p && p->aborted)) return HXV_OK;
  if (!r || !r->CommAbort) return fail(HXV_ERR_UNSUPPORTED, "hxv_comm_abort: the RCCL library in use exports no ncclCommAbort");
  h->comm_aborted = 1;
  if (p) {
    // the communicator is shared by every sector of this process: all of them lose it, and the cache forgets it (the next hxv_comm_init
    // builds a new one from the id it is given -- every rank's, because an aborted communicator fails its collectives on every rank)
    if (p->aborted.exchange(1)) return HXV_OK;
    std::lock_guard<std::mutex> lk(comm_cache().mu);
    auto& all = comm_cache().all;
    all.erase(std::remove(all.begin(), all.end(), p), all.end());
  }
  ncclResult_t e = r->CommAbort((ncclComm_t)comm);
  if (e != ncclSuccess) return nccl_fail(r, "ncclCommAbort", e);
  return HXV_OK;
}

// the file the RCCL entry points of this handle's communicator were resolved from ("" without one): what bench.py reports as config.rccl_lib
const char* hxv_comm_library(const hxv_handle* h) {
  if (!h || !h->comm_api) return "";
  return static_cast<const Rccl*>(h->comm_api)->path.c_str();
}

int hxv_comm_local_destroy(void* group) {
  LocalGroup* G = reinterpret_cast<LocalGroup*>(group);
  if (!G) return HXV_OK;
  for (auto* m : G->member)
    if (m) return fail(HXV_ERR_STATE, "hxv_comm_local_destroy: a handle still belongs to the group (hxv_comm_free / hxv_destroy it first)");
  for (auto& e : G->ready)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : G->done)
    if (e) (void)hipEventDestroy(e);
  delete G;
  return HXV_OK;
}

int hxv_comm_init_local(hxv_handle* h, void* group) {
  LocalGroup* G = reinterpret_cast<LocalGroup*>(group);
  if (!h || !G) return fail(HXV_ERR_ARG, "hxv_comm_init_local: NULL");
  if (comm_ready(h)) return fail(HXV_ERR_STATE, "hxv_comm_init_local: the handle already has a communicator");
  if (h->host.panel_rows > 0) return fail(HXV_ERR_STATE, "hxv_comm_init_local: panel handles take no communicator");
  if (h->host.nranks != G->n) return fail(HXV_ERR_ARG, "hxv_comm_init_local: the group's size is not the handle's nranks");
  HIPCHK(hipSetDevice(h->device));
  const int r = h->host.rank;
  {
    std::lock_guard<std::mutex> lk(G->mu);
    if (G->member[r]) return fail(HXV_ERR_STATE, "hxv_comm_init_local: this rank has already joined the group");
    G->member[r] = h;
  }
  if (!G->ready[r]) HIPCHK(hipEventCreateWithFlags(&G->ready[r], hipEventDisableTiming));
  if (!G->done[r]) HIPCHK(hipEventCreateWithFlags(&G->done[r], hipEventDisableTiming));
  h->lgroup = G;
  if (G->barrier()) return broken_group();  // collective: every rank has joined (one host thread per rank)
  return HXV_OK;
}

int hxv_comm_free(hxv_handle* h) {
  if (!h) return HXV_OK;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  comm_release(h);
  return HXV_OK;
}

int hxv_apply_device_slab(hxv_handle* h, const void* d_v_local, void* d_hv_local, void* stream) {
  if (!h || !d_v_local || !d_hv_local) return fail(HXV_ERR_ARG, "hxv_apply_device_slab: NULL argument");
  if (h->host.panel_rows > 0) return fail(HXV_ERR_STATE, "hxv_apply_device_slab: panel handles only do hxv_apply_dw_panel");
  return apply_slab(h, (const double2*)d_v_local, (double2*)d_hv_local, (hipStream_t)stream);
}

// Measurement: nrep slab products (exchange + kernels, COLLECTIVE on a split sector: every rank calls it with the same nrep) timed with HIP
// events on the handle's stream; *ms_step = mean time of a whole product on this rank, *ms_kernels = mean time of its product kernels alone
// (the panel product and pass A in exchange mode 2) -- what bench.py's roofline leg quotes for N > 1.
int hxv_time_apply_slab(hxv_handle* h, const void* d_v_local, void* d_hv_local, int32_t nrep, float* ms_step, float* ms_kernels) {
  if (!h || !d_v_local || !d_hv_local || nrep < 1 || !ms_step || !ms_kernels) return fail(HXV_ERR_ARG, "hxv_time_apply_slab: bad argument");
  if (h->host.panel_rows > 0) return fail(HXV_ERR_STATE, "hxv_time_apply_slab: not on a panel handle");
  HIPCHK(hipSetDevice(h->device));
  for (auto& e : h->kt_ev)
    if (!e) HIPCHK(hipEventCreate(&e));
  const bool two = h->host.exchange == 2 && h->host.nranks > 1;
  const bool ov = two && h->a2a_overlap && h->plan.usable;  // overlapped mode 2: pass A runs on the second stream, between kt_ev[4] and [5]
  double tot = 0.0, ker = 0.0, ovl = 0.0;
  int rc = HXV_OK;
  h->kt_on = 1;
  for (int i = 0; i < nrep && rc == HXV_OK; ++i) {
    hipError_t e = hipEventRecord(h->ev0, h->stream);
    rc = apply_slab(h, (const double2*)d_v_local, (double2*)d_hv_local, h->stream);
    if (rc) break;
    if (e == hipSuccess) e = hipEventRecord(h->ev1, h->stream);
    if (e == hipSuccess) e = hipEventSynchronize(h->ev1);
    float a = 0, b = 0, c = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&a, h->ev0, h->ev1);
    if (e == hipSuccess) e = hipEventElapsedTime(&b, h->kt_ev[0], h->kt_ev[1]);
    if (e == hipSuccess && two) e = hipEventElapsedTime(&c, h->kt_ev[2], h->kt_ev[3]);
    float d = 0;
    if (e == hipSuccess && ov) e = hipEventElapsedTime(&d, h->kt_ev[4], h->kt_ev[5]);
    if (e != hipSuccess) rc = fail(HXV_ERR_HIP, std::string("hxv_time_apply_slab: ") + hipGetErrorString(e));
    tot += a;
    ker += b + c + d;  // (the kernels' own time: with the overlap, more than their share of the step)
    ovl += d;          // (the part that ran on the SECOND stream, beside the exchange: not to be subtracted from the step when the exchange's
                       //  share is computed -- get_option "time_kernels_overlapped_us")
  }
  h->last_overlapped_us = (int64_t)(ovl / nrep * 1e3);
  h->kt_on = 0;
  if (rc) return rc;
  *ms_step = (float)(tot / nrep);
  *ms_kernels = (float)(ker / nrep);
  return HXV_OK;
}

// Where the exchange wants this rank's slab: its slot of the gather buffer (all-gather) / the front of the halo layout.  A caller that
// builds its vector THERE and passes that pointer to hxv_apply_device_slab saves the slab copy of every product (the vector's other
// slots are overwritten by the exchange; the slab itself is only read).  Complex layout: [qdw columns][pitch].
int hxv_slab_home(hxv_handle* h, void** d_slab) {
  if (!h || !d_slab) return fail(HXV_ERR_ARG, "hxv_slab_home: NULL argument");
  if (h->host.panel_rows > 0) return fail(HXV_ERR_STATE, "hxv_slab_home: not on a panel handle");
  if (h->host.exchange == 2) return fail(HXV_ERR_STATE, "hxv_slab_home: the all-to-all exchange has no gathered vector (any slab buffer serves)");
  HIPCHK(hipSetDevice(h->device));
  int rc = ensure_gather(h, h->stream);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));  // (the buffer is zeroed on the handle's stream)
  const size_t slot = (size_t)h->host.cmax * col_bytes(h, false);
  *d_slab = reinterpret_cast<char*>(h->d_gather) + (h->host.exchange == 1 ? 0 : (size_t)h->host.rank * slot);
  return HXV_OK;
}

int64_t hxv_exchange_count(const hxv_handle* h) { return h ? h->n_exchange : -1; }

// The plan, not the transport: what rank `rank` of an `nranks`-way split of the DimDw axis receives from and sends to every
// peer in the halo exchange, computed from the one-spin matrix H_dw alone (CSR in the reference's convention: 1-based columns, as
// hxv_get_csr returns and spH0dws(1) stores) -- no handle, no device, no communicator.  Lets one process check that rank p's send
// list towards q equals rank q's receive list from p for every pair.
// counts: [nranks] columns per peer; cols: GLOBAL 0-based column indices (recv: slot order = ascending; send: grouped by destination).
int hxv_halo_plan_from_csr(int32_t dimdw, const int64_t* dw_rowptr, const int32_t* dw_cols, int32_t rank, int32_t nranks, int32_t* recv_counts,
                           int32_t* send_counts, int32_t* recv_cols, int32_t* send_cols, int32_t* n_recv, int32_t* n_send) {
  if (dimdw < 1 || !dw_rowptr || !dw_cols || rank < 0 || nranks < 1 || rank >= nranks) return fail(HXV_ERR_ARG, "hxv_halo_plan_from_csr: bad argument");
  if (dw_rowptr[0] != 0) return fail(HXV_ERR_ARG, "hxv_halo_plan_from_csr: rowptr[0] must be 0");
  for (int32_t i = 0; i < dimdw; ++i)
    if (dw_rowptr[i + 1] < dw_rowptr[i]) return fail(HXV_ERR_ARG, "hxv_halo_plan_from_csr: rowptr must be non-decreasing");
  SectorHost s;
  s.dimdw = dimdw;
  s.dw.rowptr.assign(dw_rowptr, dw_rowptr + dimdw + 1);
  s.dw.cols.resize((size_t)dw_rowptr[dimdw]);
  for (size_t p = 0; p < s.dw.cols.size(); ++p) {
    if (dw_cols[p] < 1 || dw_cols[p] > dimdw) return fail(HXV_ERR_ARG, "hxv_halo_plan_from_csr: column index out of range (1-based expected)");
    s.dw.cols[p] = dw_cols[p] - 1;
  }
  s.rank = rank;
  s.nranks = nranks;
  dw_split(s.dimdw, rank, nranks, s.qdw, s.dw0);
  make_halo(s);
  for (int r = 0; r < nranks; ++r) {
    if (recv_counts) recv_counts[r] = s.halo_ptr[r + 1] - s.halo_ptr[r];
    if (send_counts) send_counts[r] = s.send_ptr[r + 1] - s.send_ptr[r];
  }
  if (n_recv) *n_recv = (int32_t)s.halo_cols.size();
  if (n_send) *n_send = (int32_t)s.send_cols.size();
  if (recv_cols) std::copy(s.halo_cols.begin(), s.halo_cols.end(), recv_cols);
  if (send_cols)
    for (size_t k = 0; k < s.send_cols.size(); ++k) send_cols[k] = s.send_cols[k] + s.dw0;  // local -> global
  return HXV_OK;
}

}  // extern "C"
