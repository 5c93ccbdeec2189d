// C-ABI of the engine (include/hxv.h): handle management, host/device products, options and introspection
// (the Lanczos drivers live in hxv_lanczos.hip / hxv_eigh.hip).  Everything runs on one HIP device and one stream per handle
// (SURVEY.md 8b "Threading / re-entrancy": one open sector at a time per rank).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>

#include "hxv_internal.hpp"
#include "hxv_tiles.hpp"

using namespace hxv;

#include "hxv_handle.hpp"

namespace hxv {
namespace {
thread_local std::string g_err;
}
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
}  // namespace hxv

namespace hxv {
int ensure_wt(hxv_handle* h) {
  const int64_t need = std::max<int64_t>(tiled_wt_elems(h->dev, h->plan), 1);
  if (h->d_wt && h->wt_elems >= need) return HXV_OK;
  HIPCHK(hipSetDevice(h->device));
  if (h->d_wt) {
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipDeviceSynchronize());
    pool_free(h->device, h->d_wt);
    h->device_bytes -= h->wt_elems * (int64_t)sizeof(double2);
    h->d_wt = nullptr;
  }
  HIPCHK(pool_alloc(h->device, (size_t)need * sizeof(double2), (void**)&h->d_wt));
  h->wt_elems = need;
  h->device_bytes += need * (int64_t)sizeof(double2);
  return HXV_OK;
}
}  // namespace hxv

namespace hxv {
namespace {
double us_since(const std::chrono::steady_clock::time_point& t0) {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}
int64_t host_bytes_of(const SectorHost& s) {
  auto vb = [](const auto& v) { return (int64_t)(v.capacity() * sizeof(v[0])); };
  auto sp = [&](const SpinOp& o) { return vb(o.rowptr) + vb(o.cols) + vb(o.vals) + vb(o.ell) + vb(o.coef); };
  return sp(s.up) + sp(s.dw) + vb(s.vcol) + vb(s.map_up) + vb(s.map_dw) + vb(s.a_up) + vb(s.a_dw) + vb(s.diag_stored) + vb(s.nd_up) + vb(s.nd_dw) +
         vb(s.halo_cols) + vb(s.send_cols) + sp(s.up_dev) + vb(s.up_perm) + vb(s.up_iperm) + vb(s.up_sign) + vb(s.key_up) + vb(s.map_up_dev) + vb(s.a_up_dev) +
         vb(s.nd_up_dev);
}
}  // namespace
// the staged tables of an image whose host half is done (SectorImage::pending)
struct SectorImage::Pending {
  TableArena ar;
  uint32_t *ell_up = nullptr, *ell_dw = nullptr, *vcol = nullptr, *map_up = nullptr, *map_dw = nullptr, *ndu = nullptr, *ndd = nullptr;
  uint32_t* map_up_ref = nullptr;   // device row order only: the reference's sorted up configurations (ladder operators look sources up in it)
  int32_t *up_perm = nullptr, *up_iperm = nullptr;
  uint8_t* up_sign = nullptr;
  double2 *coef_up = nullptr, *coef_dw = nullptr;
  double *a_up = nullptr, *a_dw = nullptr, *stored = nullptr;
};
namespace {
// Host half of an open: the tile plan, and every device table of the sector staged in one image (TableArena).  No HIP call.
int prepare_image(SectorImage& im) {
  const SectorHost& s = im.host;
  const auto t0 = std::chrono::steady_clock::now();
  auto pd = std::make_shared<SectorImage::Pending>();
  TableArena& ar = pd->ar;
  const SpinOp& sup = s.dev_up();   // (H_up between DEVICE rows: SectorHost::up_perm)
  (void)ar.add(sup.ell, &pd->ell_up);
  (void)ar.add(translate_ell_src(s.dw.ell, s.vcol), &pd->ell_dw);
  (void)ar.add(s.vcol, &pd->vcol);
  std::vector<double2> cu(sup.coef.size()), cd(s.dw.coef.size());
  for (size_t i = 0; i < cu.size(); ++i) cu[i] = make_double2(sup.coef[i].real(), sup.coef[i].imag());
  for (size_t i = 0; i < cd.size(); ++i) cd[i] = make_double2(s.dw.coef[i].real(), s.dw.coef[i].imag());
  (void)ar.add(cu, &pd->coef_up);
  (void)ar.add(cd, &pd->coef_dw);
  if (s.separable_diag) {
    (void)ar.add(s.dev_map_up(), &pd->map_up);
    (void)ar.add(s.map_dw, &pd->map_dw);
    (void)ar.add(s.dev_a_up(), &pd->a_up);
    (void)ar.add(s.a_dw, &pd->a_dw);
  } else {
    (void)ar.add(s.diag_stored, &pd->stored);
  }
  if (!s.nd_up.empty()) {
    (void)ar.add(s.dev_nd_up(), &pd->ndu);
    (void)ar.add(s.nd_dw, &pd->ndd);
  }
  if (s.row_order()) {
    (void)ar.add(s.map_up, &pd->map_up_ref);
    (void)ar.add(s.up_perm, &pd->up_perm);
    (void)ar.add(s.up_iperm, &pd->up_iperm);
    (void)ar.add(s.up_sign, &pd->up_sign);
  }
  PlanUploader pu{[&ar](const std::vector<uint32_t>& v, uint32_t** p) { return ar.add(v, p); },
                  [&ar](const std::vector<double2>& v, double2** p) { return ar.add(v, p); }};
  std::string perr = make_tile_plan(s, im.plan, pu);
  if (!perr.empty()) return fail(HXV_ERR_HIP, perr);
  if (!im.plan.usable) im.kernel = 0;  // too many distinct amplitudes for the LDS coefficient table
  im.host_bytes = host_bytes_of(s);
  im.pending = pd;
  im.us_plan = us_since(t0);
  return HXV_OK;
}
// Device half: one allocation, one copy, pointers patched.
int upload_image(SectorImage& im, int device) {
  if (!im.pending) {
    int rc = prepare_image(im);
    if (rc) return rc;
  }
  const SectorHost& s = im.host;
  const auto t0 = std::chrono::steady_clock::now();
  SectorImage::Pending& pd = *im.pending;
  void* base = nullptr;
  int64_t bytes = 0;
  hipError_t e = pd.ar.commit(&base, &bytes);
  if (base) {
    im.allocs.push_back(base);
    im.device_bytes += bytes;
  }
  im.device = device;
  if (e != hipSuccess) return fail(HXV_ERR_HIP, std::string("upload of the sector tables: ") + hipGetErrorString(e));
  DevSector& d = im.dev;
  d.up = DevSpin{pd.ell_up, pd.coef_up, s.dev_up().K, s.dev_up().dim};
  d.map_up_ref = pd.map_up_ref;
  d.up_perm = pd.up_perm;
  d.up_iperm = pd.up_iperm;
  d.up_sign = pd.up_sign;
  d.dw = DevSpin{pd.ell_dw, pd.coef_dw, s.dw.K, s.dw.dim};
  d.diag.mode = s.separable_diag ? 0 : 1;
  d.diag.a_up = pd.a_up;
  d.diag.a_dw = pd.a_dw;
  d.diag.map_up = pd.map_up;
  d.diag.map_dw = pd.map_dw;
  d.diag.stored = pd.stored;
  d.diag.cross = s.cross;
  d.dimup = s.dimup;
  d.dimdw = s.dimdw;
  d.pitch = s.pitch;
  d.qdw = s.qdw;
  d.dw0 = s.dw0;
  d.slab0 = s.exchange == 1 ? 0 : s.rank * s.cmax;
  d.vcol = pd.vcol;
  d.vcol_identity = (s.nranks == 1) ? 1 : 0;
  d.nd = s.nd;
  d.nd_up = pd.ndu;
  d.nd_dw = pd.ndd;
  d.ndcsr_rowptr = nullptr;
  d.ndcsr_cols = nullptr;
  d.ndcsr_vals = nullptr;
  d.real_h = (s.dev_up().real_vals && s.dw.real_vals) ? 1 : 0;
  im.pending.reset();
  im.us_upload = us_since(t0);
  im.uploaded = true;
  return HXV_OK;
}
}  // namespace
int finish_create(hxv_handle* h, int device, hxv_handle** out);
}  // namespace hxv
int hxv::finish_create(hxv_handle* h, int device, hxv_handle** out) {
  int ndev = 0;
  hipError_t e0 = hipGetDeviceCount(&ndev);
  if (e0 != hipSuccess || ndev == 0) {
    delete h;
    return fail(HXV_ERR_HIP, std::string("no HIP device available (hipGetDeviceCount: ") + hipGetErrorString(e0) + ", count " +
                                 std::to_string(ndev) + "): the HxV engine has no CPU fallback");
  }
  if (device < 0 || device >= ndev) {
    delete h;
    return fail(HXV_ERR_ARG, "device index out of range");
  }
  h->device = device;
  auto cleanup = [&](int code, const std::string& m) {
    hxv_destroy(h);
    return fail(code, m);
  };
#define HC(expr)                                                                                       \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess) return cleanup(HXV_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)
  HC(hipSetDevice(device));
  HC(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  HC(hipEventCreateWithFlags(&h->ev0, hipEventDefault));
  HC(hipEventCreateWithFlags(&h->ev1, hipEventDefault));
  SectorImage& im = *h->img;
  if (!im.uploaded) {
    int rc = upload_image(im, device);
    if (rc) {
      const std::string msg = hxv_last_error();
      return cleanup(rc, msg);
    }
    h->open_us[1] = im.us_plan;
    h->open_us[2] = im.us_upload;
    sector_cache_insert(h->img);
  }
  h->dev = im.dev;
  h->plan = im.plan;
  h->kernel = im.kernel;
  // the handle's own scratch: partial sums [2][RED_BLOCKS] and the scalars [8] of the Lanczos drivers, one allocation
  HC(h->alloc(&h->d_partials, 2 * RED_BLOCKS + 8));
  h->d_scalars = h->d_partials + 2 * RED_BLOCKS;
#undef HC
  *out = h;
  return HXV_OK;
}

namespace hxv {
// REAL-vector mode: available when every amplitude of H is real, the tiled kernels run and there is
// no spH0nd block.  Vectors are double[qdw local columns][pitch_real], pitch_real = roundup16(DimUp); on a split sector the
// exchange moves real slabs -- half the bytes on the links.
const char* real_mode_blocker(const hxv_handle* h) {
  if (!h->dev.real_h) return "H has complex amplitudes";
  if (h->kernel != 1 || !h->plan.usable) return "the tiled kernels are not in use";
  if (h->dev.nd.active) return "the spH0nd block (Jx/Jp) is active";
  if (h->host.panel_rows > 0) return "panel handle";
  if (h->plan.opt.passes != 3 || h->plan.opt.debug != 0) return "debug options are set";
  return nullptr;
}
int pitch_real_of(const hxv_handle* h) { return (h->host.dimup + 15) & ~15; }
}  // namespace hxv

extern "C" {

const char* hxv_last_error(void) { return g_err.c_str(); }
const char* hxv_version(void) { return "hxv-mi355x 0.1 (gfx950)"; }

int hxv_create_from_model(const hxv_model* model, int32_t nup, int32_t ndw, int32_t rank, int32_t nranks, int32_t device,
                          hxv_handle** out) {
  if (!model || !out) return fail(HXV_ERR_ARG, "NULL model/out");
  *out = nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  // a sector that was open before (same model bytes, sector, split, exchange, device) shares the image of that open (hxv_cache.cpp)
  const std::string key = sector_cache_key(*model, nup, ndw, rank, nranks, device, nranks > 1 ? default_exchange() : 0);
  std::shared_ptr<SectorImage> img = sector_cache_find(key);
  const bool hit = img != nullptr;
  if (!hit) {
    img = std::make_shared<SectorImage>();
    std::string e = build_sector_from_model(*model, nup, ndw, rank, nranks, img->host);
    if (!e.empty()) {
      bool unsup = e.find("not implemented") != std::string::npos;
      return fail(unsup ? HXV_ERR_UNSUPPORTED : HXV_ERR_ARG, "hxv_create_from_model: " + e);
    }
    img->key = key;
    img->us_host = us_since(t0);
    int rcp = prepare_image(*img);  // (the host half of the open: the plan and the staged tables, before any device is touched)
    if (rcp) return rcp;
  }
  hxv_handle* h = new hxv_handle(img);
  h->open_cache_hit = hit ? 1 : 0;
  if (!hit) h->open_us[0] = img->us_host;
  int rc = finish_create(h, device, out);
  if (rc == HXV_OK) {
    h->open_us[3] = us_since(t0);
    if (const char* tr = getenv("HXV_TRACE_OPEN"))
      if (tr[0] == '1')
        fprintf(stderr, "[hxv open] (%d,%d) rank %d/%d dim %lld: %s host %.0f us, plan %.0f us, upload %.0f us, total %.0f us\n", nup, ndw, rank, nranks,
                (long long)h->host.dim, hit ? "CACHED" : "built", h->open_us[0], h->open_us[1], h->open_us[2], h->open_us[3]);
  } else if (const char* tr = getenv("HXV_TRACE_OPEN")) {
    if (tr[0] == '1' && !hit) fprintf(stderr, "[hxv open] (%d,%d) dim %lld: host %.0f us, plan %.0f us (create failed with status %d)\n", nup, ndw, (long long)img->host.dim, img->us_host, img->us_plan, rc);
  }
  return rc;
}

int hxv_create_dw_panel(const hxv_model* model, int32_t nup, int32_t ndw, int32_t nrows, int32_t device, hxv_handle** out) {
  if (!model || !out) return fail(HXV_ERR_ARG, "NULL model/out");
  *out = nullptr;
  if (nrows < 1) return fail(HXV_ERR_ARG, "hxv_create_dw_panel: nrows < 1");
  hxv_handle* h = new hxv_handle();  // (a private image: panels are not cached)
  std::string e = build_sector_from_model(*model, nup, ndw, 0, 1, h->host, nrows);
  if (!e.empty()) {
    delete h;
    return fail(HXV_ERR_ARG, "hxv_create_dw_panel: " + e);
  }
  return finish_create(h, device, out);
}

int hxv_create_from_csr(int32_t dimup, int32_t dimdw, const int64_t* up_rowptr, const int32_t* up_cols, const double* up_vals,
                        const int64_t* dw_rowptr, const int32_t* dw_cols, const double* dw_vals, const double* diag, int32_t rank,
                        int32_t nranks, int32_t device, hxv_handle** out) {
  if (!out) return fail(HXV_ERR_ARG, "NULL out");
  *out = nullptr;
  hxv_handle* h = new hxv_handle();
  std::string e = build_sector_from_csr(dimup, dimdw, up_rowptr, up_cols, up_vals, dw_rowptr, dw_cols, dw_vals, diag, rank, nranks,
                                        h->host);
  if (!e.empty()) {
    delete h;
    return fail(HXV_ERR_ARG, "hxv_create_from_csr: " + e);
  }
  return finish_create(h, device, out);
}

int hxv_set_nonlocal_csr(hxv_handle* h, const int64_t* rowptr, const int32_t* cols, const double* vals) {
  if (!h || !rowptr) return fail(HXV_ERR_ARG, "hxv_set_nonlocal_csr: NULL argument");
  if (h->host.panel_rows > 0) return fail(HXV_ERR_STATE, "hxv_set_nonlocal_csr: not on a panel handle");
  if (h->host.nd.active) return fail(HXV_ERR_STATE, "hxv_set_nonlocal_csr: the handle already has an spH0nd block");
  if (h->host.exchange != 0) return fail(HXV_ERR_UNSUPPORTED, "hxv_set_nonlocal_csr: the spH0nd block needs the whole gathered vector (all-gather exchange)");
  const int64_t nloc = (int64_t)h->host.qdw * h->host.dimup;
  if (rowptr[0] != 0) return fail(HXV_ERR_ARG, "hxv_set_nonlocal_csr: rowptr[0] must be 0");
  for (int64_t i = 0; i < nloc; ++i)
    if (rowptr[i + 1] < rowptr[i]) return fail(HXV_ERR_ARG, "hxv_set_nonlocal_csr: rowptr must not decrease");
  const int64_t nnz = rowptr[nloc];
  if (nnz > 0 && (!cols || !vals)) return fail(HXV_ERR_ARG, "hxv_set_nonlocal_csr: NULL cols / vals");
  bool real = true;
  for (int64_t k = 0; k < nnz; ++k) {
    if (cols[k] < 1 || (int64_t)cols[k] > h->host.dim) return fail(HXV_ERR_ARG, "hxv_set_nonlocal_csr: column index outside 1..Dim");
    real = real && vals[2 * k + 1] == 0.0;
  }
  if (nnz == 0) return HXV_OK;  // (an all-zero block, stored: nothing to add)
  HIPCHK(hipSetDevice(h->device));
  int64_t* d_rp = nullptr;
  int32_t* d_c = nullptr;
  double2* d_v = nullptr;
  HIPCHK(h->alloc(&d_rp, (size_t)nloc + 1));
  HIPCHK(h->alloc(&d_c, (size_t)nnz));
  HIPCHK(h->alloc(&d_v, (size_t)nnz));
  HIPCHK(hipMemcpy(d_rp, rowptr, ((size_t)nloc + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_c, cols, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_v, vals, (size_t)nnz * sizeof(double2), hipMemcpyHostToDevice));
  h->dev.ndcsr_rowptr = d_rp;
  h->dev.ndcsr_cols = d_c;
  h->dev.ndcsr_vals = d_v;
  // from here on the handle behaves like one with Jx / Jp: the block is its own pass after the product (never folded, no REAL-vector
  // mode, plain Lanczos recurrence)
  h->dev.nd.active = h->host.nd.active = 1;  // (from_csr handles own their image: nothing shares this host description)
  h->dev.nd.fold = h->host.nd.fold = 0;
  (void)real;
  return HXV_OK;
}

int hxv_destroy(hxv_handle* h) {
  if (!h) return HXV_OK;
  (void)hipSetDevice(h->device);
  // kernels of this handle may still run on a CALLER's stream (hxv_apply_device & co. take one) and use the buffers
  // below; the cache would hand them to the next handle at once (hipFree used to synchronise implicitly)
  (void)hipDeviceSynchronize();
  comm_release(h);
  for (void* p : h->allocs) (void)hipFree(p);
  pool_free(h->device, h->d_stage_v);
  pool_free(h->device, h->d_stage_hv);
  pool_free(h->device, h->d_wt);
  for (auto& p : h->d_lz) pool_free(h->device, p);
  if (h->d_lz_partial) (void)hipFree(h->d_lz_partial);
  for (void* v : h->owned_vectors) pool_free(h->device, v);  // (vectors the caller never freed)
  h->owned_vectors.clear();
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  for (auto e : h->kt_ev)
    if (e) (void)hipEventDestroy(e);
  for (auto e : h->ov_ev)
    if (e) (void)hipEventDestroy(e);
  if (h->stream2) (void)hipStreamDestroy(h->stream2);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return HXV_OK;
}

int64_t hxv_vecdim(const hxv_handle* h) { return h ? (int64_t)h->host.qdw * h->host.dimup : -1; }
int64_t hxv_fullvec_elems(const hxv_handle* h) {
  if (!h) return -1;
  if (h->host.exchange == 1) return (int64_t)(h->host.qdw + (int64_t)h->host.halo_cols.size()) * h->host.pitch;
  return (int64_t)h->host.nranks * h->host.cmax * h->host.pitch;
}
int32_t hxv_exchange_mode(const hxv_handle* h) { return h ? h->host.exchange : -1; }
int hxv_halo_counts(const hxv_handle* h, int32_t* recv_counts, int32_t* send_counts) {
  if (!h || h->host.exchange != 1) return fail(HXV_ERR_STATE, "hxv_halo_counts: the handle does not use the halo exchange");
  for (int r = 0; r < h->host.nranks; ++r) {
    if (recv_counts) recv_counts[r] = h->host.halo_ptr[r + 1] - h->host.halo_ptr[r];
    if (send_counts) send_counts[r] = h->host.send_ptr[r + 1] - h->host.send_ptr[r];
  }
  return HXV_OK;
}
int hxv_halo_lists(const hxv_handle* h, int32_t* recv_cols, int32_t* send_cols) {
  if (!h || h->host.exchange != 1) return fail(HXV_ERR_STATE, "hxv_halo_lists: the handle does not use the halo exchange");
  if (recv_cols) std::copy(h->host.halo_cols.begin(), h->host.halo_cols.end(), recv_cols);
  if (send_cols) std::copy(h->host.send_cols.begin(), h->host.send_cols.end(), send_cols);
  return HXV_OK;
}
int hxv_set_exchange_default(int32_t mode) {
  if (mode < 0 || mode > 2) return fail(HXV_ERR_ARG, "exchange mode must be 0 (all-gather), 1 (halo) or 2 (two all-to-all transposes)");
  set_default_exchange(mode);
  return HXV_OK;
}
int64_t hxv_localvec_elems(const hxv_handle* h) { return h ? (int64_t)h->host.qdw * h->host.pitch : -1; }
int32_t hxv_pitch(const hxv_handle* h) { return h ? h->host.pitch : -1; }

int hxv_dims(const hxv_handle* h, int32_t* dimup, int32_t* dimdw, int64_t* dim, int32_t* qdw, int64_t* ishift) {
  if (!h) return fail(HXV_ERR_ARG, "NULL handle");
  if (dimup) *dimup = h->host.dimup;
  if (dimdw) *dimdw = h->host.dimdw;
  if (dim) *dim = h->host.dim;
  if (qdw) *qdw = h->host.qdw;
  if (ishift) *ishift = h->host.ishift;
  return HXV_OK;
}

int hxv_apply_device(hxv_handle* h, const void* d_v_full, void* d_hv_local, void* stream) {
  if (!h || !d_v_full || !d_hv_local) return fail(HXV_ERR_ARG, "hxv_apply_device: NULL argument");
  if (h->host.panel_rows > 0) return fail(HXV_ERR_STATE, "hxv_apply_device: panel handles only do hxv_apply_dw_panel");
  hipStream_t st = (hipStream_t)stream;  // NULL = the legacy default stream, as everywhere in HIP
  hipError_t e;
  if (h->kernel == 0 || !h->plan.usable)
    e = launch_hxv_naive(h->dev, (const double2*)d_v_full, (double2*)d_hv_local, st);
  else {
    int rcw = ensure_wt(h);
    if (rcw) return rcw;
    e = launch_hxv_tiled(h->dev, h->plan, (const double2*)d_v_full, h->d_wt, (double2*)d_hv_local, st);
  }
  // (the tiled product adds the spH0nd block in its pass A when the move tables exist)
  const bool nd_done = h->kernel != 0 && h->plan.usable && nd_folds(h->dev) && (h->plan.opt.passes & 1);
  if (e == hipSuccess && h->dev.nd.active && !nd_done) e = launch_hxv_nonlocal(h->dev, (const double2*)d_v_full, (double2*)d_hv_local, st);
  if (e != hipSuccess) return fail(HXV_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  h->n_apply++;
  return HXV_OK;
}

int32_t hxv_pitch_real(const hxv_handle* h) { return h ? pitch_real_of(h) : -1; }
int64_t hxv_realvec_elems(const hxv_handle* h) { return h ? (int64_t)h->host.dimdw * pitch_real_of(h) : -1; }  // (the FULL vector)
int32_t hxv_real_vectors_available(const hxv_handle* h) { return h && !real_mode_blocker(h) ? 1 : 0; }

int hxv_apply_device_real(hxv_handle* h, const void* d_v_real, void* d_hv_real, void* stream) {
  if (!h || !d_v_real || !d_hv_real) return fail(HXV_ERR_ARG, "hxv_apply_device_real: NULL argument");
  if (const char* why = real_mode_blocker(h)) return fail(HXV_ERR_UNSUPPORTED, std::string("hxv_apply_device_real: real vectors unavailable: ") + why);
  if (h->host.nranks != 1) return fail(HXV_ERR_UNSUPPORTED, "hxv_apply_device_real takes the whole vector of an unsplit sector (split sectors: the device Lanczos drivers exchange real slabs themselves)");
  int rcw = ensure_wt(h);
  if (rcw) return rcw;
  DevSector d = h->dev;
  d.pitch = pitch_real_of(h);
  hipError_t e = launch_hxv_tiled_real(d, h->plan, (const double*)d_v_real, (double*)h->d_wt, (double*)d_hv_real, (hipStream_t)stream);
  if (e != hipSuccess) return fail(HXV_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  h->n_apply++;
  return HXV_OK;
}

int hxv_apply_dw_panel(hxv_handle* h, const void* d_x, void* d_y, void* stream) {
  if (!h || !d_x || !d_y) return fail(HXV_ERR_ARG, "hxv_apply_dw_panel: NULL argument");
  if (h->host.panel_rows <= 0) return fail(HXV_ERR_STATE, "hxv_apply_dw_panel needs a handle from hxv_create_dw_panel");
  if (!h->plan.usable) return fail(HXV_ERR_UNSUPPORTED, "hxv_apply_dw_panel: tiled kernels unavailable (too many distinct amplitudes)");
  hipError_t e = launch_hxv_tiled(h->dev, h->plan, (const double2*)d_x, (double2*)d_y, nullptr, (hipStream_t)stream, nullptr, 2, true);
  if (e != hipSuccess) return fail(HXV_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  h->n_apply++;
  return HXV_OK;
}

int hxv_apply_up_add(hxv_handle* h, const void* d_v_local, const void* d_w, void* d_hv_local, void* stream) {
  if (!h || !d_v_local || !d_w || !d_hv_local) return fail(HXV_ERR_ARG, "hxv_apply_up_add: NULL argument");
  if (h->host.panel_rows > 0) return fail(HXV_ERR_STATE, "hxv_apply_up_add: panel handles have no up part");
  if (h->dev.nd.active) return fail(HXV_ERR_UNSUPPORTED, "hxv_apply_up_add: the spH0nd block needs the gathered vector (use hxv_apply_device)");
  if (!h->plan.usable) return fail(HXV_ERR_UNSUPPORTED, "hxv_apply_up_add: tiled kernels unavailable (too many distinct amplitudes)");
  // pass A addresses its slab as column slots slab0.. of a gathered vector: shift the base so the local slab lands there
  const double2* vbase = (const double2*)d_v_local - (int64_t)h->dev.slab0 * h->dev.pitch;
  hipError_t e = launch_hxv_tiled(h->dev, h->plan, vbase, const_cast<double2*>((const double2*)d_w), (double2*)d_hv_local,
                                  (hipStream_t)stream, nullptr, 1, true);
  if (e != hipSuccess) return fail(HXV_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  h->n_apply++;
  return HXV_OK;
}

}  // extern "C"

namespace hxv {
namespace {
// contiguous reference layout [ncols][dimup] -> device layout [ncols][pitch]: device row d holds reference row iperm[d], times the basis sign
__global__ void __launch_bounds__(256) rows_ref_to_dev(const double2* __restrict__ src, double2* __restrict__ dst, const int32_t* __restrict__ iperm,
                                                       const uint8_t* __restrict__ sign, int dimup, int pitch, int64_t n) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
    const int64_t c = t / pitch;
    const int d = (int)(t - c * pitch);
    double2 x = make_double2(0.0, 0.0);   // (pad rows are zero in every device vector)
    if (d < dimup) {
      x = src[c * dimup + iperm[d]];
      if (sign[d]) x = make_double2(-x.x, -x.y);
    }
    dst[t] = x;
  }
}
// device layout -> contiguous reference layout: reference row i sits at device row perm[i]
__global__ void __launch_bounds__(256) rows_dev_to_ref(const double2* __restrict__ src, double2* __restrict__ dst, const int32_t* __restrict__ perm,
                                                       const uint8_t* __restrict__ sign, int dimup, int pitch, int64_t n) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
    const int64_t c = t / dimup;
    const int i = (int)(t - c * dimup);
    const int d = perm[i];
    double2 x = src[c * pitch + d];
    if (sign[d]) x = make_double2(-x.x, -x.y);
    dst[t] = x;
  }
}
}  // namespace

int slab_from_host(hxv_handle* h, const void* v_host, double2* d_vec) {
  const SectorHost& s = h->host;
  const size_t col = (size_t)s.dimup * sizeof(double2), pit = (size_t)s.pitch * sizeof(double2);
  if (s.qdw <= 0) return HXV_OK;
  if (!s.row_order()) {
    HIPCHK(hipMemcpy2DAsync(d_vec, pit, v_host, col, col, (size_t)s.qdw, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  } else {
    double2* tmp = nullptr;
    HIPCHK(pool_alloc(h->device, col * (size_t)s.qdw, (void**)&tmp));
    hipError_t e = hipMemcpyAsync(tmp, v_host, col * (size_t)s.qdw, hipMemcpyHostToDevice, h->stream);
    const int64_t n = (int64_t)s.pitch * s.qdw;
    if (e == hipSuccess) {
      hipLaunchKernelGGL(rows_ref_to_dev, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 65536)), dim3(256), 0, h->stream, tmp, d_vec, h->dev.up_iperm,
                         h->dev.up_sign, s.dimup, s.pitch, n);
      e = hipGetLastError();
    }
    const hipError_t e2 = hipStreamSynchronize(h->stream);
    pool_free(h->device, tmp);
    if (e != hipSuccess || e2 != hipSuccess) return fail(HXV_ERR_HIP, std::string("host -> device vector: ") + hipGetErrorString(e != hipSuccess ? e : e2));
  }
  h->h2d_bytes += (int64_t)(col * s.qdw);
  return HXV_OK;
}

int slab_to_host(hxv_handle* h, const double2* d_vec, void* v_host) {
  const SectorHost& s = h->host;
  const size_t col = (size_t)s.dimup * sizeof(double2), pit = (size_t)s.pitch * sizeof(double2);
  if (s.qdw <= 0) return HXV_OK;
  if (!s.row_order()) {
    HIPCHK(hipMemcpy2DAsync(v_host, col, d_vec, pit, col, (size_t)s.qdw, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  } else {
    double2* tmp = nullptr;
    HIPCHK(pool_alloc(h->device, col * (size_t)s.qdw, (void**)&tmp));
    const int64_t n = (int64_t)s.dimup * s.qdw;
    hipLaunchKernelGGL(rows_dev_to_ref, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 65536)), dim3(256), 0, h->stream, d_vec, tmp, h->dev.up_perm,
                       h->dev.up_sign, s.dimup, s.pitch, n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(v_host, tmp, col * (size_t)s.qdw, hipMemcpyDeviceToHost, h->stream);
    const hipError_t e2 = hipStreamSynchronize(h->stream);
    pool_free(h->device, tmp);
    if (e != hipSuccess || e2 != hipSuccess) return fail(HXV_ERR_HIP, std::string("device -> host vector: ") + hipGetErrorString(e != hipSuccess ? e : e2));
  }
  h->d2h_bytes += (int64_t)(col * s.qdw);
  return HXV_OK;
}
}  // namespace hxv

extern "C" {

int hxv_apply_host(hxv_handle* h, int64_t nloc, const void* v, void* hv) {
  if (!h || !v || !hv) return fail(HXV_ERR_ARG, "hxv_apply_host: NULL argument");
  if (h->host.panel_rows > 0) return fail(HXV_ERR_STATE, "hxv_apply_host: panel handles only do hxv_apply_dw_panel");
  if (h->host.nranks != 1 && !comm_ready(h))
    return fail(HXV_ERR_STATE, "hxv_apply_host on a split sector needs the slab exchange: call hxv_comm_init after opening the sector "
                               "(or gather the vector yourself and use hxv_apply_device)");
  // v, hv: this rank's slab, vecDim_Hv_sector = DimUp*mpiQdw elements (spMatVec_MPI_main, ED_HAMILTONIAN_SPARSE_HxV.f90:230-315)
  if (nloc != (int64_t)h->host.qdw * h->host.dimup) return fail(HXV_ERR_ARG, "hxv_apply_host: Nloc != vecDim of the open sector");
  HIPCHK(hipSetDevice(h->device));
  // host arrays are in the reference's contiguous layout; the device layout pads every column to `pitch`
  const size_t col = (size_t)h->host.dimup * sizeof(double2), pit = (size_t)h->host.pitch * sizeof(double2);
  const size_t bytes = pit * (size_t)std::max(h->host.qdw, 1);
  if (!h->d_stage_v) {
    HIPCHK(pool_alloc(h->device, bytes, (void**)&h->d_stage_v));
    HIPCHK(pool_alloc(h->device, bytes, (void**)&h->d_stage_hv));
    HIPCHK(hipMemsetAsync(h->d_stage_v, 0, bytes, h->stream));  // (on the handle's stream: it does not synchronise with the null stream)
    HIPCHK(hipMemsetAsync(h->d_stage_hv, 0, bytes, h->stream));
    h->device_bytes += 2 * (int64_t)bytes;
  }
  (void)col;
  int rc = slab_from_host(h, v, h->d_stage_v);
  if (rc) return rc;
  rc = apply_slab(h, h->d_stage_v, h->d_stage_hv, h->stream);
  if (rc) return rc;
  return slab_to_host(h, h->d_stage_hv, hv);
}

// Page-lock a host array the host program keeps passing to hxv_apply_host / the *_host drivers (the Lanczos work vectors of ED_DIAG): a
// pinned array is copied by DMA at the link's rate, a pageable one through the runtime's staging buffers.  hipHostRegister costs ~0.1 s per
// GB: once per array, not per product.
int hxv_host_register(void* ptr, int64_t bytes) {
  if (!ptr || bytes <= 0) return fail(HXV_ERR_ARG, "hxv_host_register: bad argument");
  hipError_t e = hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault);
  if (e == hipErrorHostMemoryAlreadyRegistered) {
    (void)hipGetLastError();
    return HXV_OK;
  }
  if (e != hipSuccess) return fail(HXV_ERR_HIP, std::string("hipHostRegister: ") + hipGetErrorString(e));
  return HXV_OK;
}
int hxv_host_unregister(void* ptr) {
  if (!ptr) return HXV_OK;
  hipError_t e = hipHostUnregister(ptr);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(HXV_ERR_HIP, std::string("hipHostUnregister: ") + hipGetErrorString(e));
  }
  return HXV_OK;
}

int hxv_time_apply(hxv_handle* h, const void* d_v_full, void* d_hv_local, int32_t nrep, float* ms_per_apply) {
  if (!h || nrep < 1 || !ms_per_apply) return fail(HXV_ERR_ARG, "hxv_time_apply: bad argument");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipEventRecord(h->ev0, h->stream));
  for (int i = 0; i < nrep; ++i) {
    int rc = hxv_apply_device(h, d_v_full, d_hv_local, h->stream);
    if (rc) return rc;
  }
  HIPCHK(hipEventRecord(h->ev1, h->stream));
  HIPCHK(hipEventSynchronize(h->ev1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *ms_per_apply = ms / (float)nrep;
  return HXV_OK;
}

// ---- device vectors owned by the library: what a host program without a HIP binding of its own (the Fortran glue) keeps between calls ----
int hxv_vector_alloc_many(hxv_handle* h, int32_t count, void** d_vec);

int hxv_vector_alloc(hxv_handle* h, void** d_vec) { return hxv_vector_alloc_many(h, 1, d_vec); }

// `count` local vectors in ONE allocation, consecutive (hxv_localvec_elems() elements apart): what hxv_eigh_lowest writes its eigenvectors to
int hxv_vector_alloc_many(hxv_handle* h, int32_t count, void** d_vec) {
  if (!h || !d_vec || count < 1) return fail(HXV_ERR_ARG, "hxv_vector_alloc: bad argument");
  HIPCHK(hipSetDevice(h->device));
  const size_t bytes = (size_t)count * (size_t)h->host.pitch * std::max(h->host.qdw, 1) * sizeof(double2);
  void* p = nullptr;
  HIPCHK(pool_alloc(h->device, bytes, &p));
  HIPCHK(hipMemsetAsync(p, 0, bytes, h->stream));   // (pad rows must be zero; on the handle's stream, like every other fill)
  HIPCHK(hipStreamSynchronize(h->stream));
  h->owned_vectors.push_back(p);
  *d_vec = p;
  return HXV_OK;
}

int hxv_vector_free(hxv_handle* h, void* d_vec) {
  if (!h) return fail(HXV_ERR_ARG, "hxv_vector_free: NULL handle");
  if (!d_vec) return HXV_OK;
  auto it = std::find(h->owned_vectors.begin(), h->owned_vectors.end(), d_vec);
  if (it == h->owned_vectors.end()) return fail(HXV_ERR_ARG, "hxv_vector_free: not a vector of this handle (or freed already)");
  h->owned_vectors.erase(it);
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  pool_free(h->device, d_vec);
  return HXV_OK;
}

int hxv_vector_from_host(hxv_handle* h, const void* v_host, void* d_vec) {
  if (!h || !v_host || !d_vec) return fail(HXV_ERR_ARG, "hxv_vector_from_host: NULL argument");
  HIPCHK(hipSetDevice(h->device));
  return slab_from_host(h, v_host, (double2*)d_vec);
}

int hxv_vector_to_host(hxv_handle* h, const void* d_vec, void* v_host) {
  if (!h || !v_host || !d_vec) return fail(HXV_ERR_ARG, "hxv_vector_to_host: NULL argument");
  HIPCHK(hipSetDevice(h->device));
  return slab_to_host(h, (const double2*)d_vec, v_host);
}

int64_t hxv_live_handles(void) { return hxv::g_live_handles.load(); }

int hxv_row_order(const hxv_handle* h, int32_t* perm, int8_t* sign) {
  if (!h) return -1;
  const SectorHost& s = h->host;
  if (perm)
    for (int i = 0; i < s.dimup; ++i) perm[i] = s.row_order() ? s.up_perm[i] : i;
  if (sign)
    for (int i = 0; i < s.dimup; ++i) sign[i] = (s.row_order() && s.up_sign[s.up_perm[i]]) ? (int8_t)-1 : (int8_t)1;
  return s.row_order() ? 1 : 0;
}

int hxv_get_maps(const hxv_handle* h, int32_t* map_up, int32_t* map_dw) {
  if (!h) return fail(HXV_ERR_ARG, "NULL handle");
  if (h->host.map_up.empty()) return fail(HXV_ERR_STATE, "handle built from CSR has no basis maps");
  if (map_up) std::copy(h->host.map_up.begin(), h->host.map_up.end(), map_up);
  if (map_dw) std::copy(h->host.map_dw.begin(), h->host.map_dw.end(), map_dw);
  return HXV_OK;
}

int64_t hxv_nnz(const hxv_handle* h, int32_t which) {
  if (!h || which < 0 || which > 1) return -1;
  const SpinOp& op = which == 0 ? h->host.up : h->host.dw;
  return op.rowptr.empty() ? 0 : op.rowptr.back();
}

int hxv_get_csr(const hxv_handle* h, int32_t which, int64_t* rowptr, int32_t* cols, double* vals) {
  if (!h || which < 0 || which > 1 || !rowptr || !cols || !vals) return fail(HXV_ERR_ARG, "hxv_get_csr: bad argument");
  const SpinOp& op = which == 0 ? h->host.up : h->host.dw;
  std::copy(op.rowptr.begin(), op.rowptr.end(), rowptr);
  for (size_t p = 0; p < op.cols.size(); ++p) {
    cols[p] = op.cols[p] + 1;
    vals[2 * p] = op.vals[p].real();
    vals[2 * p + 1] = op.vals[p].imag();
  }
  return HXV_OK;
}

int hxv_get_diag(const hxv_handle* h, double* diag) {
  if (!h || !diag) return fail(HXV_ERR_ARG, "hxv_get_diag: bad argument");
  const SectorHost& s = h->host;
  for (int c = 0; c < s.qdw; ++c)
    for (int i = 0; i < s.dimup; ++i) diag[(size_t)i + (size_t)c * s.dimup] = host_diag_element(s, i, c + s.dw0);
  return HXV_OK;
}

int hxv_set_option(hxv_handle* h, const char* name, int64_t value) {
  if (!h || !name) return fail(HXV_ERR_ARG, "hxv_set_option: NULL");
  // timing experiments (results are wrong or partial when set): only with HXV_EXPERIMENTS=1 in the environment
  if (!strcmp(name, "debug") || !strcmp(name, "passes") || !strcmp(name, "job_debug")) {
    const char* ex = getenv("HXV_EXPERIMENTS");
    if (!(ex && ex[0] == '1') && value != (!strcmp(name, "passes") ? 3 : 0))
      return fail(HXV_ERR_ARG, std::string("option ") + name + " is a timing experiment that gives wrong results; set HXV_EXPERIMENTS=1 to enable it");
  }
  if (!strcmp(name, "kernel")) {
    if (value < 0 || value > 1) return fail(HXV_ERR_ARG, "kernel must be 0 or 1");
    h->kernel = (int)value;
    return HXV_OK;
  }
  if (!strcmp(name, "lds_min_kb_up") || !strcmp(name, "lds_min_kb_dw")) {
    if (value < 0 || value > 160) return fail(HXV_ERR_ARG, "lds_min_kb must be in [0,160]");
    (name[11] == 'u' ? h->plan.opt.lds_min_kb_up : h->plan.opt.lds_min_kb_dw) = (int)value;
    return HXV_OK;
  }
  if (!strcmp(name, "lanczos_fused")) {
    h->lz_fused = value ? 1 : 0;
    return HXV_OK;
  }
  if (!strcmp(name, "real_vectors")) {
    h->real_vectors = value ? 1 : 0;
    return HXV_OK;
  }
  if (!strcmp(name, "lanczos_graph")) {
    h->lz_graph = value ? 1 : 0;
    return HXV_OK;
  }
  if (!strcmp(name, "eigh_keep_pct")) {
    if (value < 5 || value > 80) return fail(HXV_ERR_ARG, "eigh_keep_pct must be in [5,80]");
    h->eigh_keep_pct = (int)value;
    return HXV_OK;
  }
  if (!strcmp(name, "eigh_degenerate")) {
    h->eigh_degenerate = value ? 1 : 0;
    return HXV_OK;
  }
  if (!strcmp(name, "eigh_fuse_restart")) {
    h->eigh_fuse_restart = value ? 1 : 0;
    return HXV_OK;
  }
  if (!strcmp(name, "eigh_measure_all")) {
    h->eigh_measure_all = value ? 1 : 0;
    return HXV_OK;
  }
  if (!strcmp(name, "debug")) {
    h->plan.opt.debug = (int)value;
    return HXV_OK;
  }
  if (!strcmp(name, "job_debug")) {
    h->plan.opt.job_debug = (int)value;
    return HXV_OK;
  }
  if (!strcmp(name, "pair_rows")) {
    if (value < -1 || value > 1) return fail(HXV_ERR_ARG, "pair_rows must be -1, 0 or 1");
    h->plan.opt.pair_rows = (int)value;
    return HXV_OK;
  }
  if (!strcmp(name, "real_dw_pairs")) {
    h->plan.opt.real_dw_pairs = value ? 1 : 0;
    return HXV_OK;
  }
  if (!strcmp(name, "wt_colmajor")) {
    h->plan.opt.wt_colmajor = value ? 1 : 0;
    return HXV_OK;
  }
  if (!strcmp(name, "block_order")) {
    if (value < -1 || value > 2) return fail(HXV_ERR_ARG, "block_order must be -1, 0, 1 or 2");
    h->plan.opt.block_order = (int)value;
    return HXV_OK;
  }
  if (!strcmp(name, "job_max_blocks")) {
    h->plan.opt.job_max_blocks = (int)value;
    return HXV_OK;
  }
  if (!strcmp(name, "passes")) {
    if (value < 1 || value > 3) return fail(HXV_ERR_ARG, "passes must be 1, 2 or 3");
    h->plan.opt.passes = (int)value;
    return HXV_OK;
  }
  if (!strcmp(name, "lanczos_inplace")) {  // split sectors: Lanczos vectors in their slot of a gather buffer (no slab copy per product)
    h->lz_inplace = value ? 1 : 0;
    return HXV_OK;
  }
  if (!strcmp(name, "exchange_overlap")) {  // exchange mode 2: diagonal + up hops on a second stream during the transposes (plain products)
    h->a2a_overlap = value ? 1 : 0;
    return HXV_OK;
  }
  if (!strcmp(name, "fold_nd")) {  // spH0nd inside pass A (default) or as its own pass over hv
    h->dev.nd.fold = value ? 1 : 0;  // (the handle's copy: the host description may be shared with other handles)
    return HXV_OK;
  }
  // tiling knobs: rebuild the plan (old tables stay allocated until hxv_destroy)
  TileOptions o = h->plan.opt;
  if (!strcmp(name, "cols_per_tile")) o.cols_per_tile = (int)value;
  else if (!strcmp(name, "rows_per_tile")) o.rows_per_tile = (int)value;
  else if (!strcmp(name, "lds_budget_kb")) o.lds_budget_kb_up = o.lds_budget_kb_dw = (int)value;
  else if (!strcmp(name, "lds_budget_kb_up")) o.lds_budget_kb_up = (int)value;
  else if (!strcmp(name, "lds_budget_kb_dw")) o.lds_budget_kb_dw = (int)value;
  else if (!strcmp(name, "tile_bits_up")) o.force_bits_up = (int)value;
  else if (!strcmp(name, "tile_bits_dw")) o.force_bits_dw = (int)value;
  else if (!strcmp(name, "threads_up")) o.threads_up = (int)value;
  else if (!strcmp(name, "threads_dw")) o.threads_dw = (int)value;

  else if (!strcmp(name, "sort_mode")) o.sort_mode = (int)value;
  else if (!strcmp(name, "wt_cols")) o.wt_cols = (int)value;
  else if (!strcmp(name, "sort_mode_dw")) o.sort_mode_dw = (int)value;
  else if (!strcmp(name, "spread_banks")) o.spread_banks = value ? 1 : 0;
  else if (!strcmp(name, "job_up")) o.job_up = value < 0 ? 0 : (value > 2 ? 2 : (int)value);
  else if (!strcmp(name, "job_cols")) o.job_cols = (int)value;
  else if (!strcmp(name, "job_groups")) o.job_groups = (int)value;
  else if (!strcmp(name, "job_stages")) o.job_stages = (int)value;
  else return fail(HXV_ERR_ARG, std::string("unknown option ") + name);
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  TilePlan np;
  np.opt = o;
  TableArena ar;  // (the handle's own tables from here on: the shared image keeps the default plan)
  PlanUploader pu{[&ar](const std::vector<uint32_t>& v, uint32_t** p) { return ar.add(v, p); },
                  [&ar](const std::vector<double2>& v, double2** p) { return ar.add(v, p); }};
  std::string perr = make_tile_plan(h->host, np, pu);
  if (!perr.empty()) return fail(HXV_ERR_ARG, perr);
  if (np.usable) {
    void* base = nullptr;
    int64_t bytes = 0;
    hipError_t ea = ar.commit(&base, &bytes);
    if (base) {
      h->allocs.push_back(base);
      h->device_bytes += bytes;
    }
    if (ea != hipSuccess) return fail(HXV_ERR_HIP, std::string("upload of the tile tables: ") + hipGetErrorString(ea));
  }
  h->plan = np;
  return HXV_OK;
}

int64_t hxv_get_option(const hxv_handle* h, const char* name) {
  if (!h || !name) return -1;
  if (!strcmp(name, "real_vectors")) return h->real_vectors;
  // what THIS open cost, microseconds (0 for the parts a cached image saved): host description, tile plan, table upload, whole create call
  if (!strcmp(name, "open_us_host")) return (int64_t)h->open_us[0];
  if (!strcmp(name, "open_us_plan")) return (int64_t)h->open_us[1];
  if (!strcmp(name, "open_us_upload")) return (int64_t)h->open_us[2];
  if (!strcmp(name, "open_us_total")) return (int64_t)h->open_us[3];
  if (!strcmp(name, "open_cache_hit")) return h->open_cache_hit;
  if (!strcmp(name, "lanczos_graph")) return h->lz_graph;
  if (!strcmp(name, "eigh_degenerate")) return h->eigh_degenerate;
  if (!strcmp(name, "eigh_fuse_restart")) return h->eigh_fuse_restart;
  if (!strcmp(name, "eigh_keep_pct")) return h->eigh_keep_pct;
  if (!strcmp(name, "eigh_measure_all")) return h->eigh_measure_all;
  if (!strcmp(name, "eigh_last_full_passes")) return h->eigh_last_full;
  if (!strcmp(name, "eigh_last_local_passes")) return h->eigh_last_local;
  if (!strcmp(name, "eigh_last_search_products")) return h->eigh_last_search;  // *nmatvec of the last hxv_eigh_lowest = search + check
  if (!strcmp(name, "eigh_last_check_products")) return h->eigh_last_check;
  if (!strcmp(name, "lanczos_real_last")) return h->last_real;
  if (!strcmp(name, "slab_copies")) return h->n_slab_copy;  // exchanges whose vector was not at home in a gather buffer
  if (!strcmp(name, "lanczos_inplace")) return h->lz_inplace;
  if (!strcmp(name, "exchange_overlap")) return h->a2a_overlap;
  if (!strcmp(name, "kernel")) return h->kernel;
  if (!strcmp(name, "time_kernels_overlapped_us")) return h->last_overlapped_us;
  if (!strcmp(name, "tile_bits_up")) return h->plan.up.lowbits;
  if (!strcmp(name, "tile_bits_dw")) return h->plan.dw.lowbits;
  if (!strcmp(name, "real_dw_pairs")) return h->plan.opt.real_dw_pairs;
  if (!strcmp(name, "wt_colmajor")) return h->plan.opt.wt_colmajor;
  if (!strcmp(name, "cols_per_tile")) return h->plan.opt.cols_per_tile;
  if (!strcmp(name, "rows_per_tile")) return h->plan.opt.rows_per_tile;
  if (!strcmp(name, "p16_bits_up")) return h->plan.up.p16_bits;
  if (!strcmp(name, "p16_bits_dw")) return h->plan.dw.p16_bits;
  if (!strcmp(name, "lds_budget_kb_up")) return h->plan.opt.lds_budget_kb_up;
  if (!strcmp(name, "lds_budget_kb_dw")) return h->plan.opt.lds_budget_kb_dw;
  if (!strcmp(name, "k_in_up")) return h->plan.up.k_in;
  if (!strcmp(name, "k_out_up")) return h->plan.up.k_out;
  if (!strcmp(name, "k_in_dw")) return h->plan.dw.k_in;
  if (!strcmp(name, "k_out_dw")) return h->plan.dw.k_out;
  if (!strcmp(name, "n_in_up")) return h->plan.up.n_in;
  if (!strcmp(name, "n_out_up")) return h->plan.up.n_out;
  if (!strcmp(name, "n_in_dw")) return h->plan.dw.n_in;
  if (!strcmp(name, "n_out_dw")) return h->plan.dw.n_out;
  if (!strcmp(name, "slots_in_up_x100")) return (int64_t)(100 * h->plan.up.slots_in);
  if (!strcmp(name, "slots_out_up_x100")) return (int64_t)(100 * h->plan.up.slots_out);
  if (!strcmp(name, "slots_in_dw_x100")) return (int64_t)(100 * h->plan.dw.slots_in);
  if (!strcmp(name, "bh_up_x100")) return (int64_t)(100 * h->plan.up.bh_per_row);
  if (!strcmp(name, "rs_up_x100")) return (int64_t)(100 * h->plan.up.rs_per_row);
  if (!strcmp(name, "bh_dw_x100")) return (int64_t)(100 * h->plan.dw.bh_per_row);
  if (!strcmp(name, "rs_dw_x100")) return (int64_t)(100 * h->plan.dw.rs_per_row);
  if (!strcmp(name, "max_block_up")) return h->plan.up.max_block;
  if (!strcmp(name, "max_block_dw")) return h->plan.dw.max_block;
  if (!strcmp(name, "nblocks_up")) return h->plan.up.nblocks;
  if (!strcmp(name, "table_classes_up")) return h->plan.up.table_classes;
  if (!strcmp(name, "table_classes_dw")) return h->plan.dw.table_classes;
  if (!strcmp(name, "rs_tables_up")) return h->plan.up.rs_tables;
  if (!strcmp(name, "rs_tables_dw")) return h->plan.dw.rs_tables;
  if (!strcmp(name, "nblocks_dw")) return h->plan.dw.nblocks;
  if (!strcmp(name, "job_up")) return h->plan.opt.job_up;
  if (!strcmp(name, "job_up_active"))
    return (h->plan.opt.job_up == 1 && h->plan.opt.sort_mode == 0 && job_up_usable(h->dev, h->plan) &&
            job_up_fits(h->dev, h->plan, false, std::max(h->plan.opt.job_cols, h->plan.opt.wt_cols)))
               ? 1
               : 0;
  if (!strcmp(name, "max_outer_up")) return h->plan.up.max_outer;
  if (!strcmp(name, "max_outer_dw")) return h->plan.dw.max_outer;
  return -1;
}

int hxv_get_stats(const hxv_handle* h, hxv_stats* out) {
  if (!h || !out) return fail(HXV_ERR_ARG, "hxv_get_stats: NULL");
  out->n_apply = h->n_apply;
  out->algorithmic_bytes = 32 * (int64_t)h->host.qdw * h->host.dimup;
  out->device_bytes = h->device_bytes + h->img->device_bytes;  // (the sector's tables may be shared with other handles of the same sector)
  out->kernel = h->kernel;
  out->real_h = h->dev.real_h;
  out->k_up = h->host.up.K;
  out->k_dw = h->host.dw.K;
  out->n_hops_up = (int)h->host.up.coef.size();
  out->n_hops_dw = (int)h->host.dw.coef.size();
  out->h2d_bytes = h->h2d_bytes;
  out->d2h_bytes = h->d2h_bytes;
  return HXV_OK;
}

}  // extern "C"

