// Micro-benchmark kernels used by scripts/strided_bench.py (NOT part of libhxv.so): what HBM gives for R*16-byte
// column segments at a column stride of pitch*16 bytes, read / staged through LDS / written back in several ways.
// Built on demand: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o gpurun_out/libstrided_bench.so scripts/strided_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>

typedef double dbl2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 load_stream(const double2* p) {
  dbl2_t x = __builtin_nontemporal_load(reinterpret_cast<const dbl2_t*>(p));
  return make_double2(x.x, x.y);
}
__device__ __forceinline__ void store_stream(double2* p, double2 a) {
  dbl2_t x;
  x.x = a.x;
  x.y = a.y;
  __builtin_nontemporal_store(x, reinterpret_cast<dbl2_t*>(p));
}
// read every element of a DimUp x ncols matrix once,
// in pass-B tile order: workgroup = [R consecutive rows] x [n consecutive columns].  Tells what HBM gives
// for R*16-byte segments at a column stride of DimUp*16 bytes.
// ---------------------------------------------------------------------------------------
template <int R>
__global__ void __launch_bounds__(1024) strided_read_kernel(const double2* __restrict__ v, double2* __restrict__ out, int dimup, int ncols,
                                                           int n, int ngroups, int groups_per_xcd, int nblocks, int mode) {
  extern __shared__ double2 lds[];
  const int b = blockIdx.x;
  const int xcd = b & 7, j = b >> 3;
  const int gl = j / nblocks, kb = j - gl * nblocks;
  const int rg = xcd * groups_per_xcd + gl;
  if (gl >= groups_per_xcd || rg >= ngroups) return;
  const int cb0 = kb * n, nn = min(n, ncols - cb0), i0 = rg * R;
  double2 acc = make_double2(0.0, 0.0);
  if (mode == 0) {  // read only
    for (int q = threadIdx.x; q < nn * R; q += blockDim.x) {
      const double2 x = v[(int64_t)(cb0 + q / R) * dimup + min(i0 + q % R, dimup - 1)];
      acc.x += x.x;
      acc.y += x.y;
    }
    if (acc.x == 1.2345e300) out[b] = acc;  // keep the loads alive
    return;
  }
  // mode >= 1: stage through LDS transposed (lds[r*nn + col]) like pass B
  for (int q = threadIdx.x; q < nn * R; q += blockDim.x)
    lds[(q % R) * nn + q / R] = v[(int64_t)(cb0 + q / R) * dimup + min(i0 + q % R, dimup - 1)];
  __syncthreads();
  if (mode == 1) {
    for (int q = threadIdx.x; q < nn * R; q += blockDim.x) {
      const double2 x = lds[(q % R) * nn + q / R];
      acc.x += x.x;
      acc.y += x.y;
    }
    if (acc.x == 1.2345e300) out[b] = acc;
    return;
  }
  for (int q = threadIdx.x; q < nn * R; q += blockDim.x) {
    if (i0 + q % R >= dimup) continue;
    const int64_t o = (int64_t)(cb0 + q / R) * dimup + i0 + q % R;
    double2 x = lds[(q % R) * nn + q / R];
    if (mode == 4 || mode == 5) {  // read-modify-write
      const double2 h = (mode == 5) ? load_stream(&out[o]) : out[o];
      x.x += h.x;
      x.y += h.y;
    }
    if (mode == 3 || mode == 5)
      store_stream(&out[o], x);
    else
      out[o] = x;
  }
}

hipError_t launch_strided_read(const double2* v, double2* out, int dimup, int ncols, int R, int n, int mode, hipStream_t st) {
  const int ngroups = (dimup + R - 1) / R, gpx = (ngroups + 7) / 8, nblocks = (ncols + n - 1) / n;
  const int64_t nwg = (int64_t)gpx * 8 * nblocks;
  const size_t lds = mode ? (size_t)n * R * 16 : 0;
#define SR(RR)                                                                                                         \
  case RR: {                                                                                                           \
    (void)hipFuncSetAttribute((const void*)strided_read_kernel<RR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(strided_read_kernel<RR>, dim3((unsigned)nwg), dim3(1024), lds, st, v, out, dimup, ncols, n, ngroups, gpx, nblocks, mode); \
  } break;
  switch (R) {
    SR(4) SR(8) SR(16) SR(32) SR(64)
    default: return hipErrorInvalidValue;
  }
#undef SR
  return hipGetLastError();
}

extern "C" int strided_bench(const void* d_v, void* d_out, int pitch, int ncols, int R, int n, int mode, int nrep, float* ms) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, nullptr);
  for (int i = 0; i < nrep; ++i)
    if (launch_strided_read((const double2*)d_v, (double2*)d_out, pitch, ncols, R, n, mode, nullptr) != hipSuccess) return 2;
  (void)hipEventRecord(e1, nullptr);
  (void)hipEventSynchronize(e1);
  float t = 0;
  (void)hipEventElapsedTime(&t, e0, e1);
  *ms = t / nrep;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return 0;
}
