// Sector-image cache.  The reference opens and closes a sector around every Lanczos run: ED_DIAG.f90:142-186 once per sector,
// ED_GF_NORMAL.f90:208-222 (and its seven siblings) once per Green's-function CHANNEL -- 56 channels of a 2x2 cluster re-open the same
// four sectors N+-1 fourteen times each.  What an open builds (basis maps, one-spin matrices, the tile plan, ~50 device tables) depends
// only on the model, the sector, the split and the device, so the engine keeps the images of closed sectors and a re-open shares
// them (SectorImage, hxv_handle.hpp).  The key holds every byte that enters the construction and is compared byte for byte, so a new
// bath (the next DMFT iteration) is a different key, never a stale hit.
//   HXV_SECTOR_CACHE=0        disables it (every open builds and uploads)
//   HXV_SECTOR_CACHE_MB=<n>   cap on the cached images' host + device bytes (default 2048), least recently used first out
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <list>
#include <mutex>
#include <unordered_map>

#include "hxv_handle.hpp"

namespace hxv {

SectorImage::~SectorImage() {
  if (allocs.empty()) return;
  int cur = -1;
  (void)hipGetDevice(&cur);
  if (device >= 0) (void)hipSetDevice(device);
  for (void* p : allocs) (void)hipFree(p);
  if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
}

namespace {
struct Cache {
  std::mutex mu;
  std::list<std::shared_ptr<SectorImage>> lru;  // most recently used first
  std::unordered_map<std::string, std::list<std::shared_ptr<SectorImage>>::iterator> by_key;
  int64_t bytes = 0, hits = 0, misses = 0;
};
Cache& cache() {
  static Cache* c = new Cache();  // never destroyed: its images hold device memory, and HIP may be gone when statics die
  return *c;
}
bool enabled() {
  static const bool on = [] {
    const char* e = std::getenv("HXV_SECTOR_CACHE");
    return !(e && e[0] == '0');
  }();
  return on;
}
int64_t cap_bytes() {
  static const int64_t cap = [] {
    if (const char* e = std::getenv("HXV_SECTOR_CACHE_MB")) return (int64_t)std::max(0.0, std::atof(e)) << 20;
    return (int64_t)2048 << 20;
  }();
  return cap;
}
int64_t image_bytes(const SectorImage& im) { return im.host_bytes + im.device_bytes; }
}  // namespace

template <typename T>
static void put(std::string& k, const T& x) {
  k.append(reinterpret_cast<const char*>(&x), sizeof(T));
}

// Everything build_sector_from_model and make_tile_plan read: the model (scalars and the three arrays, byte for byte), the sector, the
// split, the exchange the split will use, the device.
std::string sector_cache_key(const hxv_model& m, int nup, int ndw, int rank, int nranks, int device, int exchange) {
  if (!enabled()) return std::string();
  if (m.nlat < 1 || m.norb < 1 || m.nspin < 1 || m.nspin > 2 || m.nbath < 0 || m.nlat > 16 || m.norb > 5 || !m.imphloc) return std::string();
  if (m.nbath > 0 && (!m.hbath || !m.vbath)) return std::string();
  std::string k;
  const size_t nloc = (size_t)m.nlat * m.nlat * m.nspin * m.nspin * m.norb * m.norb;
  k.reserve(256 + 16 * nloc * (size_t)(1 + m.nbath));
  k.append("hxv1");
  put(k, m.nlat); put(k, m.norb); put(k, m.nspin); put(k, m.nbath); put(k, m.hfmode);
  for (double u : m.uloc) put(k, u);
  put(k, m.ust); put(k, m.jh); put(k, m.jx); put(k, m.jp); put(k, m.xmu);
  put(k, nup); put(k, ndw); put(k, rank); put(k, nranks); put(k, device); put(k, exchange);
  k.append(row_order_env_key());  // (HXV_ROW_ORDER...: the image holds the tables of ONE device row order)
  k.append(reinterpret_cast<const char*>(m.imphloc), 16 * nloc);
  if (m.nbath > 0) {
    k.append(reinterpret_cast<const char*>(m.hbath), 16 * nloc * (size_t)m.nbath);
    k.append(reinterpret_cast<const char*>(m.vbath), 8 * (size_t)m.nlat * m.nspin * m.norb * m.nbath);
  }
  return k;
}

std::shared_ptr<SectorImage> sector_cache_find(const std::string& key) {
  if (key.empty()) return nullptr;
  Cache& c = cache();
  std::lock_guard<std::mutex> lk(c.mu);
  auto it = c.by_key.find(key);
  if (it == c.by_key.end()) {
    ++c.misses;
    return nullptr;
  }
  c.lru.splice(c.lru.begin(), c.lru, it->second);
  ++c.hits;
  return *it->second;
}

void sector_cache_insert(const std::shared_ptr<SectorImage>& im) {
  if (!im || im->key.empty() || !im->uploaded) return;
  Cache& c = cache();
  std::vector<std::shared_ptr<SectorImage>> dropped;  // destroyed (hipFree) outside the lock
  {
    std::lock_guard<std::mutex> lk(c.mu);
    if (c.by_key.count(im->key)) return;  // (another thread of this process built the same image meanwhile: both stay valid)
    if (image_bytes(*im) > cap_bytes()) return;
    c.lru.push_front(im);
    c.by_key[im->key] = c.lru.begin();
    c.bytes += image_bytes(*im);
    while (c.bytes > cap_bytes() && c.lru.size() > 1) {
      auto last = std::prev(c.lru.end());
      c.bytes -= image_bytes(**last);
      c.by_key.erase((*last)->key);
      dropped.push_back(*last);
      c.lru.erase(last);
    }
  }
}

}  // namespace hxv

extern "C" {

int hxv_sector_cache_clear(void) {
  hxv::Cache& c = hxv::cache();
  std::list<std::shared_ptr<hxv::SectorImage>> gone;
  {
    std::lock_guard<std::mutex> lk(c.mu);
    gone.swap(c.lru);
    c.by_key.clear();
    c.bytes = 0;
  }
  // (images that open handles still share live on until those handles are destroyed; hxv_destroy has synchronised the device for the rest)
  gone.clear();
  return HXV_OK;
}

int hxv_sector_cache_stats(int64_t* entries, int64_t* bytes, int64_t* hits, int64_t* misses) {
  hxv::Cache& c = hxv::cache();
  std::lock_guard<std::mutex> lk(c.mu);
  if (entries) *entries = (int64_t)c.lru.size();
  if (bytes) *bytes = c.bytes;
  if (hits) *hits = c.hits;
  if (misses) *misses = c.misses;
  return HXV_OK;
}

}  // extern "C"
