// Memory-system probe (NOT part of libhxv.so): do HBM-streaming loads and L2-served 64-byte gathers of one CU overlap?
// Pass B of the product is, per tile: stream a tile in (HBM), gather ~4.2 out-of-block 64-byte segments per element (L2 of the XCD), store.
// Measured inside the kernel (round 6, debug bits): stream alone 0.96 ms, stream + gathers 1.78 ms -- the sum, with two workgroups per CU.
// This probe isolates the two access streams:
//   mode 0  stream only        every thread loads NP x 16 B of a 59 KB tile (coalesced), writes it to LDS, barrier
//   mode 1  gathers only       every thread does NG gathers of 16 B; 4 lanes share a 64-byte segment, segments pseudo-random inside a region
//                              of `small` bytes owned by blockIdx % 8 (one XCD's L2); HB loads in flight per thread, then a dependent sum
//   mode 2  both, one after the other in every workgroup (the structure of pass B)
//   mode 3  both, SPLIT inside the workgroup: waves 0-7 stream (2 x the loads each), waves 8-15 gather (2 x each); one barrier per tile
//   mode 4  split by workgroup parity (even workgroups stream 2 tiles per step, odd ones gather for 2)
// `missp` per cent of the gathers go to the big buffer instead (L2 misses).   usage: l1_mix_probe [small_KB] [missp] [ng] [hb]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
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

__device__ inline uint32_t hash32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

template <int HB, int SL>
__device__ inline double2 gather_part(const double2* __restrict__ big, const double2* __restrict__ small, uint32_t small_segs, uint64_t big_segs,
                                      int ng, int missp, uint32_t seed, int tid) {
  double2 acc = make_double2(0.0, 0.0);
  for (int g0 = 0; g0 < ng; g0 += HB) {
    double2 x[HB];
#pragma unroll
    for (int k = 0; k < HB; ++k) {
      const uint32_t h = hash32(seed + (uint32_t)(g0 + k) * 0x9E3779B9u + (uint32_t)(tid / SL) * 0x85EBCA6Bu);
      const bool miss = (int)(h % 100u) < missp;
      const double2* p = miss ? big + ((uint64_t)(hash32(h) % (big_segs * 4 / SL)) * SL + (tid % SL)) : small + ((uint64_t)(h % (small_segs * 4 / SL)) * SL + (tid % SL));
      x[k] = *p;
    }
#pragma unroll
    for (int k = 0; k < HB; ++k) {
      acc.x += x[k].x;
      acc.y += x[k].y;
    }
  }
  return acc;
}

template <int MODE, int HB, int SL>
__global__ void __launch_bounds__(1024, 8) mix(const double2* __restrict__ big, const double2* __restrict__ small, double2* __restrict__ out,
                                               int tiles_per_wg, uint64_t big_tiles, uint32_t small_segs, int ng, int missp) {
  extern __shared__ double2 lds[];
  const int tid = threadIdx.x, b = blockIdx.x;
  const double2* __restrict__ sm = small + (uint64_t)(b & 7) * small_segs * 4;
  const uint64_t big_segs = big_tiles * (TILE / 4);
  double2 acc = make_double2(0.0, 0.0);
  for (int t = 0; t < tiles_per_wg; ++t) {
    const uint64_t tile = ((uint64_t)b * tiles_per_wg + t) % big_tiles;
    const uint32_t seed = (uint32_t)tile * 2654435761u;
    const bool do_stream = MODE == 0 || MODE == 2 || (MODE == 3 && tid < T / 2) || (MODE == 4 && !(b & 8));
    const bool do_gather = MODE == 1 || MODE == 2 || (MODE == 3 && tid >= T / 2) || (MODE == 4 && (b & 8));
    if (do_stream) {
      if (MODE == 3) {
        double2 x[2 * NP];
#pragma unroll
        for (int it = 0; it < 2 * NP; ++it) x[it] = big[tile * TILE + (uint64_t)it * (T / 2) + tid];
#pragma unroll
        for (int it = 0; it < 2 * NP; ++it) lds[it * (T / 2) + tid] = x[it];
      } else {
        for (int rep = 0; rep < (MODE == 4 ? 2 : 1); ++rep) {
          double2 x[NP];
          const uint64_t tl = MODE == 4 ? (tile * 2 + rep) % big_tiles : tile;
#pragma unroll
          for (int it = 0; it < NP; ++it) x[it] = big[tl * TILE + (uint64_t)it * T + tid];
#pragma unroll
          for (int it = 0; it < NP; ++it) lds[it * T + tid] = x[it];
        }
      }
    }
    if (MODE == 2) __syncthreads();
    if (do_gather) {
      const int mult = (MODE == 3 || MODE == 4) ? 2 : 1;
      const double2 a = gather_part<HB, SL>(big, sm, small_segs, big_segs, ng * mult, missp, seed, MODE == 3 ? tid - T / 2 + (t & 1) * 4096 : tid);
      acc.x += a.x;
      acc.y += a.y;
    }
    __syncthreads();
    acc.x += lds[(tid * 7 + t) & (TILE - 1)].x;
    __syncthreads();
  }
  if (acc.x == 1.2345e300) out[b] = acc;
}

// gathers only, by per-lane LDS-DMA (global_load_lds_dwordx4: the data lands in LDS at base + 16 x lane, no staging registers)
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
template <int HB, int SL>
__global__ void __launch_bounds__(1024, 8) gather_dma(const double2* __restrict__ small, double2* __restrict__ out, int tiles_per_wg, uint64_t big_tiles,
                                                      uint32_t small_segs, int ng) {
  extern __shared__ double2 lds[];
  const int tid = threadIdx.x, b = blockIdx.x, wave = tid >> 6, lane = tid & 63;
  const double2* __restrict__ sm = small + (uint64_t)(b & 7) * small_segs * 4;
  double2 acc = make_double2(0.0, 0.0);
  const uint32_t base = (uint32_t)(wave * HB * 1024);   // this wave's HB KB of the staging slab
  for (int t = 0; t < tiles_per_wg; ++t) {
    const uint64_t tile = ((uint64_t)b * tiles_per_wg + t) % big_tiles;
    const uint32_t seed = (uint32_t)tile * 2654435761u;
    for (int g0 = 0; g0 < ng; g0 += HB) {
#pragma unroll
      for (int k = 0; k < HB; ++k) {
        const uint32_t h = hash32(seed + (uint32_t)(g0 + k) * 0x9E3779B9u + (uint32_t)(tid / SL) * 0x85EBCA6Bu);
        const double2* p = sm + ((uint64_t)(h % (small_segs * 4 / SL)) * SL + (tid % SL));
        __builtin_amdgcn_global_load_lds((glb_void_t*)p, (lds_void_t*)(base + (uint32_t)k * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < HB; ++k) {
        const double2 x = lds[(base >> 4) + k * 64 + lane];
        acc.x += x.x;
        acc.y += x.y;
      }
    }
  }
  if (acc.x == 1.2345e300) out[b] = acc;
}

// gathers only, 8 bytes per lane (the real-vector pass A: one double per column and row), same bytes moved: 2 x the loads
template <int HB, int SL>
__global__ void __launch_bounds__(1024, 8) gather_b64(const double* __restrict__ small, double2* __restrict__ out, int tiles_per_wg, uint64_t big_tiles,
                                                      uint32_t small_dsegs, int ng) {
  const int tid = threadIdx.x, b = blockIdx.x;
  const double* __restrict__ sm = small + (uint64_t)(b & 7) * small_dsegs * 8;   // (small_dsegs segments of 64 B = 8 doubles)
  double acc = 0.0;
  for (int t = 0; t < tiles_per_wg; ++t) {
    const uint64_t tile = ((uint64_t)b * tiles_per_wg + t) % big_tiles;
    const uint32_t seed = (uint32_t)tile * 2654435761u;
    for (int g0 = 0; g0 < 2 * ng; g0 += HB) {
      double x[HB];
#pragma unroll
      for (int k = 0; k < HB; ++k) {
        const uint32_t h = hash32(seed + (uint32_t)(g0 + k) * 0x9E3779B9u + (uint32_t)(tid / SL) * 0x85EBCA6Bu);
        x[k] = sm[(uint64_t)(h % (small_dsegs * 8 / SL)) * SL + (tid % SL)];
      }
#pragma unroll
      for (int k = 0; k < HB; ++k) acc += x[k];
    }
  }
  if (acc == 1.2345e300) out[b] = make_double2(acc, 0.0);
}

template <int HB, int SL>
void sweep_forms(const double2* small, double2* out, int tiles_per_wg, uint64_t big_tiles, uint32_t small_segs, int ng) {
  const double gb = 512.0 * tiles_per_wg * TILE * 16.0 * ng / NP;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  float ms;
  for (int form = 0; form < 2; ++form) {
    for (int rep = 0; rep < 2; ++rep) {
      CHK(hipEventRecord(e0, 0));
      for (int i = 0; i < (rep ? 5 : 1); ++i) {
        if (form == 0)
          hipLaunchKernelGGL((gather_dma<HB, SL>), dim3(512), dim3(T), 16 * HB * 1024, 0, small, out, tiles_per_wg, big_tiles, small_segs, ng);
        else
          hipLaunchKernelGGL((gather_b64<HB, SL>), dim3(512), dim3(T), 0, 0, (const double*)small, out, tiles_per_wg, big_tiles, small_segs, ng);
      }
      CHK(hipEventRecord(e1, 0));
      CHK(hipEventSynchronize(e1));
      CHK(hipEventElapsedTime(&ms, e0, e1));
    }
    CHK(hipGetLastError());
    ms /= 5;
    printf("  seg %4d B HB=%d %-40s %8.3f ms   %7.1f GB/s chip  %6.1f GB/s per CU\n", SL * (form ? 8 : 16), HB,
           form ? "gathers only, 8 B per lane (dwordx2)" : "gathers only, per-lane LDS-DMA", ms, gb / ms * 1e-6, gb / ms * 1e-6 / 256);
  }
}

template <int MODE, int HB, int SL>
float run(const double2* big, const double2* small, double2* out, int tiles_per_wg, uint64_t big_tiles, uint32_t small_segs, int ng, int missp, int nrep) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  const int grid = 512;
  hipLaunchKernelGGL((mix<MODE, HB, SL>), dim3(grid), dim3(T), TILE * 16, 0, big, small, out, tiles_per_wg, big_tiles, small_segs, ng, missp);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0, 0));
  for (int i = 0; i < nrep; ++i)
    hipLaunchKernelGGL((mix<MODE, HB, SL>), dim3(grid), dim3(T), TILE * 16, 0, big, small, out, tiles_per_wg, big_tiles, small_segs, ng, missp);
  CHK(hipEventRecord(e1, 0));
  CHK(hipEventSynchronize(e1));
  float ms;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  CHK(hipGetLastError());
  return ms / nrep;
}

template <int HB, int SL>
void sweep(const double2* big, const double2* small, double2* out, int tiles_per_wg, uint64_t big_tiles, uint32_t small_segs, int ng, int missp) {
  const double tiles = 512.0 * tiles_per_wg;
  const double sb = tiles * TILE * 16, gb = tiles * TILE * 16.0 * ng / NP;
  const char* names[5] = {"stream only", "gathers only", "both, sequential per workgroup", "both, split by waves of a workgroup", "both, split by workgroup parity"};
  float ms[5];
  ms[0] = run<0, HB, SL>(big, small, out, tiles_per_wg, big_tiles, small_segs, ng, missp, 5);
  ms[1] = run<1, HB, SL>(big, small, out, tiles_per_wg, big_tiles, small_segs, ng, missp, 5);
  ms[2] = run<2, HB, SL>(big, small, out, tiles_per_wg, big_tiles, small_segs, ng, missp, 5);
  ms[3] = run<3, HB, SL>(big, small, out, tiles_per_wg, big_tiles, small_segs, ng, missp, 5);
  ms[4] = run<4, HB, SL>(big, small, out, tiles_per_wg, big_tiles, small_segs, ng, missp, 5);
  for (int m = 0; m < 5; ++m) {
    const double bytes = (m == 0 ? sb : m == 1 ? gb : sb + gb);
    printf("  seg %4d B HB=%d %-40s %8.3f ms   %7.1f GB/s chip  %6.1f GB/s per CU\n", SL * 16, HB, names[m], ms[m], bytes / ms[m] * 1e-6, bytes / ms[m] * 1e-6 / 256);
  }
  printf("  seg %4d B HB=%d sum of the two alone %.3f ms, max %.3f ms\n", SL * 16, HB, ms[0] + ms[1], ms[0] > ms[1] ? ms[0] : ms[1]);
}

int main(int argc, char** argv) {
  const int small_kb = argc > 1 ? atoi(argv[1]) : 2048;
  const int missp = argc > 2 ? atoi(argv[2]) : 0;
  const int ng = argc > 3 ? atoi(argv[3]) : 16;  // gathers per thread and tile (pass B at C3: 4 pairs x 4.2 live = ~17, 6.1 issued)
  const uint64_t big_tiles = 40000;               // 2.6 GB
  const int tiles_per_wg = 80;                    // 512 x 80 tiles = 2.7 GB streamed per launch
  double2 *big, *small, *out;
  CHK(hipMalloc(&big, big_tiles * TILE * 16));
  CHK(hipMalloc(&small, (size_t)8 * small_kb * 1024));
  CHK(hipMalloc(&out, 4096 * 16));
  CHK(hipMemset(big, 0, big_tiles * TILE * 16));
  CHK(hipMemset(small, 0, (size_t)8 * small_kb * 1024));
  const uint32_t small_segs = (uint32_t)small_kb * 1024 / 64;
  printf("l1_mix_probe: tile 64 KB, 512 workgroups x %d tiles, gathers per thread and tile %d (64-byte segments, region %d KB per XCD, %d %% to HBM)\n",
         tiles_per_wg, ng, small_kb, missp);
  sweep<4, 4>(big, small, out, tiles_per_wg, big_tiles, small_segs, ng, missp);
  sweep<8, 4>(big, small, out, tiles_per_wg, big_tiles, small_segs, ng, missp);
  sweep<4, 8>(big, small, out, tiles_per_wg, big_tiles, small_segs, ng, missp);
  sweep<8, 8>(big, small, out, tiles_per_wg, big_tiles, small_segs, ng, missp);
  sweep<4, 16>(big, small, out, tiles_per_wg, big_tiles, small_segs, ng, missp);
  sweep<4, 64>(big, small, out, tiles_per_wg, big_tiles, small_segs, ng, missp);
  sweep<8, 64>(big, small, out, tiles_per_wg, big_tiles, small_segs, ng, missp);
  if (missp == 0) {
    printf("other forms of the gather (all L2 hits):\n");
    sweep_forms<4, 4>(small, out, tiles_per_wg, big_tiles, small_segs, ng);
    sweep_forms<8, 4>(small, out, tiles_per_wg, big_tiles, small_segs, ng);
    sweep_forms<4, 8>(small, out, tiles_per_wg, big_tiles, small_segs, ng);
    sweep_forms<4, 64>(small, out, tiles_per_wg, big_tiles, small_segs, ng);
    sweep_forms<8, 64>(small, out, tiles_per_wg, big_tiles, small_segs, ng);
  }
  return 0;
}
