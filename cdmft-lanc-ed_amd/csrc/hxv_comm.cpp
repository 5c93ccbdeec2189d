// The slab exchange behind the C-ABI: one RCCL communicator per open (split) sector.
//
// The reference splits the vector along DimDw (ED_HAMILTONIAN.f90:93-105) and re-assembles what a rank needs with
// MPI collectives inside spMatVec_MPI_main (ED_HAMILTONIAN_SPARSE_HxV.f90:272-296, ED_HAMILTONIAN_COMMON.f90:30-94).
// Here the re-assembly is ONE equal-count ncclAllGather over xGMI of the padded slabs, on the handle's stream, so a
// Fortran rank needs nothing but this library: hxv_comm_unique_id (rank 0) -> broadcast the 128 bytes with the host
// program's own MPI_Bcast -> hxv_comm_init on every rank.  The dot products of the device Lanczos drivers become
// ncclAllReduce on the same stream.
//
// RCCL is loaded with dlopen at the first hxv_comm_* call: the library has no link-time dependency on it (serial runs
// and the CPU-only checks never touch it), and a process that already holds a copy (PyTorch ships its own) reuses it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>

#include "hxv_handle.hpp"

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
  std::string err;
};
Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) {
      r.err = std::string("cannot load librccl: ") + dlerror();
      return;
    }
#define SYM(field, sym)                                           \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, #sym)); \
  if (!r.field) r.err = "librccl lacks " #sym;
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
  });
  return &r;
}
int nccl_fail(const char* what, ncclResult_t e) {
  Rccl* r = rccl();
  return fail(HXV_ERR_HIP, std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(e) : "RCCL error"));
}
}  // namespace

namespace hxv {

bool comm_ready(const hxv_handle* h) { return h->comm != nullptr; }

int comm_allreduce_sum(hxv_handle* h, double* d_buf, size_t count, hipStream_t st) {
  if (!h->comm) return HXV_OK;  // serial: nothing to add
  ncclResult_t e = rccl()->AllReduce(d_buf, d_buf, count, ncclFloat64, ncclSum, (ncclComm_t)h->comm, st);
  if (e != ncclSuccess) return nccl_fail("ncclAllReduce", e);
  return HXV_OK;
}

// d_hv_local = (H v)|slab with v given as this rank's slab: copy the slab into its slot of the gather buffer, all-gather
// in place, run the product on the gathered vector
int apply_slab(hxv_handle* h, const double2* d_v_local, double2* d_hv_local, hipStream_t st) {
  const SectorHost& s = h->host;
  if (s.nranks == 1 && !h->comm) return hxv_apply_device(h, d_v_local, d_hv_local, st);
  if (!h->comm) return fail(HXV_ERR_STATE, "split sector without a communicator: call hxv_comm_init after opening the sector");
  HIPCHK(hipSetDevice(h->device));
  if (s.exchange == 1) {
    // HALO exchange: only the columns H_dw couples to another rank's rows travel -- packed per destination, one grouped
    // send/receive; they land behind the local slab, where the column -> slot table of this layout expects them
    const size_t nfull = (size_t)(s.qdw + s.halo_cols.size()) * s.pitch;
    if (!h->d_gather) {
      HIPCHK(pool_alloc(h->device, std::max<size_t>(nfull, 1) * sizeof(double2), (void**)&h->d_gather));
      HIPCHK(pool_alloc(h->device, std::max<size_t>(s.send_cols.size(), 1) * s.pitch * sizeof(double2), (void**)&h->d_send));
      HIPCHK(hipMalloc((void**)&h->d_send_cols, std::max<size_t>(s.send_cols.size(), 1) * sizeof(int32_t)));
      HIPCHK(hipMemcpy(h->d_send_cols, s.send_cols.data(), s.send_cols.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemcpyAsync(h->d_gather, d_v_local, (size_t)s.qdw * s.pitch * sizeof(double2), hipMemcpyDeviceToDevice, st));
    hipError_t pe = launch_pack_columns(d_v_local, h->d_send, h->d_send_cols, (int)s.send_cols.size(), s.pitch, st);
    if (pe != hipSuccess) return fail(HXV_ERR_HIP, std::string("pack kernel: ") + hipGetErrorString(pe));
    Rccl* r = rccl();
    ncclResult_t e = r->GroupStart();
    for (int p = 0; p < s.nranks && e == ncclSuccess; ++p) {
      if (p == s.rank) continue;
      const size_t ns = (size_t)(s.send_ptr[p + 1] - s.send_ptr[p]) * s.pitch, nr = (size_t)(s.halo_ptr[p + 1] - s.halo_ptr[p]) * s.pitch;
      if (ns) e = r->Send(h->d_send + (size_t)s.send_ptr[p] * s.pitch, ns * 2, ncclFloat64, p, (ncclComm_t)h->comm, st);
      if (nr && e == ncclSuccess) e = r->Recv(h->d_gather + (size_t)(s.qdw + s.halo_ptr[p]) * s.pitch, nr * 2, ncclFloat64, p, (ncclComm_t)h->comm, st);
    }
    ncclResult_t e2 = r->GroupEnd();
    if (e != ncclSuccess || e2 != ncclSuccess) return nccl_fail("halo send/recv", e != ncclSuccess ? e : e2);
    h->n_exchange++;
    return hxv_apply_device(h, h->d_gather, d_hv_local, st);
  }
  const size_t slot = (size_t)s.cmax * s.pitch;
  if (!h->d_gather) {
    HIPCHK(pool_alloc(h->device, slot * s.nranks * sizeof(double2), (void**)&h->d_gather));
    HIPCHK(hipMemsetAsync(h->d_gather, 0, slot * s.nranks * sizeof(double2), st));
    h->device_bytes += (int64_t)(slot * s.nranks * sizeof(double2));
  }
  double2* mine = h->d_gather + (size_t)s.rank * slot;
  HIPCHK(hipMemcpyAsync(mine, d_v_local, (size_t)s.qdw * s.pitch * sizeof(double2), hipMemcpyDeviceToDevice, st));
  ncclResult_t e = rccl()->AllGather(mine, h->d_gather, slot * 2, ncclFloat64, (ncclComm_t)h->comm, st);
  if (e != ncclSuccess) return nccl_fail("ncclAllGather", e);
  h->n_exchange++;
  return hxv_apply_device(h, h->d_gather, d_hv_local, st);
}

void comm_release(hxv_handle* h) {
  if (h->comm) {
    (void)rccl()->CommDestroy((ncclComm_t)h->comm);
    h->comm = nullptr;
  }
  if (h->d_gather) {
    pool_free(h->device, h->d_gather);
    h->d_gather = nullptr;
  }
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
  if (e != ncclSuccess) return nccl_fail("ncclGetUniqueId", e);
  static_assert(sizeof(id) == HXV_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  std::memcpy(id128, &id, sizeof(id));
  return HXV_OK;
}

int hxv_comm_init(hxv_handle* h, const void* id128) {
  if (!h || !id128) return fail(HXV_ERR_ARG, "hxv_comm_init: NULL");
  if (h->comm) return fail(HXV_ERR_STATE, "hxv_comm_init: the handle already has a communicator");
  if (h->host.panel_rows > 0) return fail(HXV_ERR_STATE, "hxv_comm_init: panel handles take no communicator");
  Rccl* r = rccl();
  if (!r->err.empty()) return fail(HXV_ERR_UNSUPPORTED, r->err);
  HIPCHK(hipSetDevice(h->device));
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  ncclResult_t e = r->CommInitRank(&c, h->host.nranks, id, h->host.rank);
  if (e != ncclSuccess) return nccl_fail("ncclCommInitRank", e);
  h->comm = c;
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

int64_t hxv_exchange_count(const hxv_handle* h) { return h ? h->n_exchange : -1; }

}  // extern "C"
