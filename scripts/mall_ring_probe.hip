// Memory-system probe (NOT part of libhxv.so): what does the 256 MiB Infinity Cache give the product's STREAMS if pass B and pass A run
// unit by unit instead of one whole-vector launch each?
// The product moves 5 vector transits through HBM: pass B reads v and writes the scratch wt; pass A reads v and wt and writes hv.  Run as
// B(chunk 0), A(chunk 0), B(chunk 1), ... the second read of v and the read of wt come a few tens of MB after the first touch of the same lines.
// The skeleton has the streams only (no hops): B: w = 2 x, A: y = x + w, tiles of 64 KB per workgroup of 1024 threads.
//   plan 0  two launches over the whole vectors (today)
//   plan 1  per chunk of c tiles: B(chunk), A(chunk); full-size w
//   plan 2  the same with w in a ring of 2 c tiles (the scratch never needs to reach HBM if the cache keeps dirty lines)
//   plan 3  merged launches, full-size w: launch k = A(chunk k-1) and B(chunk k) in one grid (no dependent launch boundary between them)
//   plan 4  merged launches, ring of 3 c tiles
// usage: mall_ring_probe [GB of one vector = 2.65] ; prints ms per "product" and 13.25 GB / t for chunk sizes of 16 MB ... 256 MB
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#define CHK(x)                                                                 \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

constexpr int T = 1024, NP = 4, TILE = T * NP;  // elements of 16 B per tile (64 KB)

__device__ inline void tile_b(const double2* __restrict__ x, double2* __restrict__ w, int tid) {
  double2 a[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) a[k] = x[k * T + tid];
#pragma unroll
  for (int k = 0; k < NP; ++k) w[k * T + tid] = make_double2(2.0 * a[k].x, 2.0 * a[k].y);
}
__device__ inline void tile_a(const double2* __restrict__ x, const double2* __restrict__ w, double2* __restrict__ y, int tid) {
  double2 a[NP], b[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) a[k] = x[k * T + tid];
#pragma unroll
  for (int k = 0; k < NP; ++k) b[k] = w[k * T + tid];
#pragma unroll
  for (int k = 0; k < NP; ++k) y[k * T + tid] = make_double2(a[k].x + b[k].x, a[k].y + b[k].y);
}

// tiles [t0, t0 + gridDim.x); w tile index = ring ? tile % ring : tile
__global__ __launch_bounds__(T) void kern_b(const double2* x, double2* w, uint32_t t0, uint32_t ring) {
  const uint32_t t = t0 + blockIdx.x;
  tile_b(x + (uint64_t)t * TILE, w + (uint64_t)(ring ? t % ring : t) * TILE, threadIdx.x);
}
__global__ __launch_bounds__(T) void kern_a(const double2* x, const double2* w, double2* y, uint32_t t0, uint32_t ring) {
  const uint32_t t = t0 + blockIdx.x;
  tile_a(x + (uint64_t)t * TILE, w + (uint64_t)(ring ? t % ring : t) * TILE, y + (uint64_t)t * TILE, threadIdx.x);
}
// one grid: the first na workgroups run A on tiles [ta0, ta0 + na), the others B on tiles [tb0, ...)
__global__ __launch_bounds__(T) void kern_ab(const double2* x, double2* w, double2* y, uint32_t ta0, uint32_t na, uint32_t tb0, uint32_t ring) {
  if (blockIdx.x < na) {
    const uint32_t t = ta0 + blockIdx.x;
    tile_a(x + (uint64_t)t * TILE, w + (uint64_t)(ring ? t % ring : t) * TILE, y + (uint64_t)t * TILE, threadIdx.x);
  } else {
    const uint32_t t = tb0 + (blockIdx.x - na);
    tile_b(x + (uint64_t)t * TILE, w + (uint64_t)(ring ? t % ring : t) * TILE, threadIdx.x);
  }
}

int main(int argc, char** argv) {
  const double gb = argc > 1 ? atof(argv[1]) : 2.65;
  const uint32_t ntiles = (uint32_t)(gb * 1e9 / (TILE * 16.0));
  const size_t bytes = (size_t)ntiles * TILE * 16;
  double2 *x, *w, *y;
  CHK(hipMalloc(&x, bytes));
  CHK(hipMalloc(&w, bytes));
  CHK(hipMalloc(&y, bytes));
  CHK(hipMemset(x, 0, bytes));
  CHK(hipMemset(w, 0, bytes));
  CHK(hipMemset(y, 0, bytes));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  printf("vector %.3f GB, %u tiles of 64 KB; 5 transits = %.2f GB\n", bytes / 1e9, ntiles, 5 * bytes / 1e9);
  auto run = [&](int plan, uint32_t c) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      CHK(hipEventRecord(e0));
      if (plan == 0) {
        kern_b<<<ntiles, T>>>(x, w, 0, 0);
        kern_a<<<ntiles, T>>>(x, w, y, 0, 0);
      } else if (plan == 1 || plan == 2) {
        const uint32_t ring = plan == 2 ? 2 * c : 0;
        for (uint32_t t0 = 0; t0 < ntiles; t0 += c) {
          const uint32_t n = std::min(c, ntiles - t0);
          kern_b<<<n, T>>>(x, w, t0, ring);
          kern_a<<<n, T>>>(x, w, y, t0, ring);
        }
      } else {
        const uint32_t ring = plan == 4 ? 3 * c : 0;
        const uint32_t nch = (ntiles + c - 1) / c;
        for (uint32_t k = 0; k <= nch; ++k) {  // launch k: A(chunk k-1) with B(chunk k)
          const uint32_t na = k >= 1 ? std::min(c, ntiles - (k - 1) * c) : 0;
          const uint32_t nb = k < nch ? std::min(c, ntiles - k * c) : 0;
          kern_ab<<<na + nb, T>>>(x, w, y, k >= 1 ? (k - 1) * c : 0, na, k * c, ring);
        }
      }
      CHK(hipEventRecord(e1));
      CHK(hipEventSynchronize(e1));
      float ms;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) best = std::min(best, ms);
    }
    printf("plan %d chunk %5u tiles (%6.1f MB per stream): %7.3f ms = %6.2f TB/s of the 5 transits\n", plan, c, c * TILE * 16.0 / 1e6, best,
           5 * bytes / 1e9 / best);
    fflush(stdout);
  };
  run(0, 0);
  for (int plan = 1; plan <= 4; ++plan)
    for (uint32_t c : {256u, 512u, 1024u, 2048u, 4096u}) run(plan, c);
  run(0, 0);
  // correctness of the last plan's data flow (y = 3 x with x = 0 tells nothing: fill x with 1 and check a ring plan)
  CHK(hipMemset(y, 0, bytes));
  std::vector<double2> hx(TILE, make_double2(1.0, -1.0));
  for (uint32_t t = 0; t < ntiles; t += std::max(1u, ntiles / 64)) CHK(hipMemcpy(x + (uint64_t)t * TILE, hx.data(), TILE * 16, hipMemcpyHostToDevice));
  {
    const uint32_t c = 1024, ring = 3 * c, nch = (ntiles + c - 1) / c;
    for (uint32_t k = 0; k <= nch; ++k) {
      const uint32_t na = k >= 1 ? std::min(c, ntiles - (k - 1) * c) : 0;
      const uint32_t nb = k < nch ? std::min(c, ntiles - k * c) : 0;
      kern_ab<<<na + nb, T>>>(x, w, y, k >= 1 ? (k - 1) * c : 0, na, k * c, ring);
    }
    CHK(hipDeviceSynchronize());
    int bad = 0;
    std::vector<double2> hy(TILE);
    for (uint32_t t = 0; t < ntiles; t += std::max(1u, ntiles / 64)) {
      CHK(hipMemcpy(hy.data(), y + (uint64_t)t * TILE, TILE * 16, hipMemcpyDeviceToHost));
      for (int i = 0; i < TILE; ++i) bad += !(hy[i].x == 3.0 && hy[i].y == -3.0);
    }
    printf("ring plan data flow: %s\n", bad ? "WRONG" : "ok");
  }
  return 0;
}
