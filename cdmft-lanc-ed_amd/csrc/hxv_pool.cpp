// Device-buffer cache of the engine.  Opening a sector needs a few vector-sized buffers (the dw-hop scratch, three
// Lanczos vectors, for hxv_eigh_lowest a Krylov basis of tens of GB), and on this platform a fresh hipMalloc costs
// ~25 ms per GB (page mapping): 0.7 s for a 28 GB basis, 4 s for 150 GB -- comparable to the solve itself.  An ED run
// opens sectors one after another (ED_DIAG.f90:78-184; 56 Green's-function channels per solve), so freed buffers are
// kept per device and handed to the next handle instead of going back to the driver.
//   HXV_POOL=0            disables the cache (every free is a hipFree)
//   HXV_POOL_MAX_GB=<n>   cap on cached bytes per device (default: 40 % of the device memory)
// Cached memory is NOT zeroed: callers that need zeros memset (they did so after hipMalloc too).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <map>
#include <mutex>

#include "../../include/hxv.h"

namespace hxv {
namespace {
struct DevicePool {
  std::multimap<size_t, void*> cached;  // size -> block
  std::map<void*, size_t> live;         // blocks handed out (size needed at free time)
  size_t cached_bytes = 0;
  int64_t hits = 0, misses = 0;
};
std::mutex g_mu;
std::map<int, DevicePool> g_pools;

bool enabled() {
  static const bool on = [] {
    const char* e = std::getenv("HXV_POOL");
    return !(e && e[0] == '0');
  }();
  return on;
}

size_t cap_bytes() {
  static const size_t cap = [] {
    if (const char* e = std::getenv("HXV_POOL_MAX_GB")) return (size_t)std::max(0.0, std::atof(e)) << 30;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return (size_t)0;
    return tot / 10 * 4;
  }();
  return cap;
}

void trim_locked(DevicePool& p) {
  for (auto& kv : p.cached) (void)hipFree(kv.second);
  p.cached.clear();
  p.cached_bytes = 0;
}
}  // namespace

hipError_t pool_alloc(int device, size_t bytes, void** out) {
  if (bytes == 0) bytes = 1;
  std::lock_guard<std::mutex> lk(g_mu);
  DevicePool& p = g_pools[device];
  if (enabled()) {
    // smallest cached block that fits without wasting more than a quarter of it
    auto it = p.cached.lower_bound(bytes);
    if (it != p.cached.end() && it->first <= bytes + bytes / 4 + (1u << 20)) {
      *out = it->second;
      p.live[*out] = it->first;
      p.cached_bytes -= it->first;
      p.cached.erase(it);
      ++p.hits;
      return hipSuccess;
    }
  }
  ++p.misses;
  hipError_t e = hipMalloc(out, bytes);
  if (e == hipErrorOutOfMemory && !p.cached.empty()) {  // our own cache may be what is in the way
    (void)hipGetLastError();
    trim_locked(p);
    e = hipMalloc(out, bytes);
  }
  if (e == hipSuccess) p.live[*out] = bytes;
  return e;
}

void pool_free(int device, void* ptr) {
  if (!ptr) return;
  std::lock_guard<std::mutex> lk(g_mu);
  DevicePool& p = g_pools[device];
  auto it = p.live.find(ptr);
  if (it == p.live.end()) {  // not ours: plain free
    (void)hipFree(ptr);
    return;
  }
  const size_t sz = it->second;
  p.live.erase(it);
  if (enabled() && p.cached_bytes + sz <= cap_bytes()) {
    p.cached.emplace(sz, ptr);
    p.cached_bytes += sz;
  } else {
    (void)hipFree(ptr);
  }
}
}  // namespace hxv

extern "C" {

int hxv_pool_trim(int32_t device) {
  std::lock_guard<std::mutex> lk(hxv::g_mu);
  for (auto& kv : hxv::g_pools)
    if (device < 0 || kv.first == device) {
      (void)hipSetDevice(kv.first);
      hxv::trim_locked(kv.second);
    }
  return HXV_OK;
}

int hxv_pool_stats(int32_t device, int64_t* cached_bytes, int64_t* hits, int64_t* misses) {
  std::lock_guard<std::mutex> lk(hxv::g_mu);
  const hxv::DevicePool& p = hxv::g_pools[device];
  if (cached_bytes) *cached_bytes = (int64_t)p.cached_bytes;
  if (hits) *hits = p.hits;
  if (misses) *misses = p.misses;
  return HXV_OK;
}

}  // extern "C"
