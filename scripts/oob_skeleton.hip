// Skeleton of the experiment VERDICT r3 item 1(c) asks to MEASURE FIRST (NOT part of libhxv.so): pass B's out-of-block half as an
// LDS-free kernel on whole cache lines.  One wavefront = 16 rows x 4 columns (256-byte contiguous runs per column); for every column the
// list of out-of-block source columns of H_dw (uniform over rows) is walked, the partner elements are loaded and summed (loads + stores
// only, no coefficients), and the result is written ONCE into the blocked scratch wt[group][row][4].  XCD-aware: blockIdx % 8 owns a
// contiguous range of 16-row chunks, the column groups of a chunk are swept by consecutive workgroups, so a chunk's panel
// (16 rows x DimDw x 16 B = 3.3 MB at Ns=16) is what the XCD's L2 sees.  Kill criterion: <= 0.6 ms at C3 (the phase costs 0.88 ms today).
#include <hip/hip_runtime.h>
#include <cstdint>

typedef double dbl2_t __attribute__((ext_vector_type(2)));

template <int ROWS>  // rows per wave footprint: 16 (4 columns) or 8 (8 columns)
__global__ void __launch_bounds__(1024) oob_skeleton(const double2* __restrict__ v, double2* __restrict__ wt, const int32_t* __restrict__ src,
                                                    int nsrc, int dimup, int dimdw, int pitch, int chunks_per_xcd, int groups_per_wg, int mode) {
  constexpr int COLS = 64 / ROWS;
  const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
  const int wgs_per_chunk = (dimdw / COLS + groups_per_wg - 1) / groups_per_wg;
  const int chunk = xcd * chunks_per_xcd + j / wgs_per_chunk;
  const int wg_in_chunk = j % wgs_per_chunk;
  if (j / wgs_per_chunk >= chunks_per_xcd || chunk * ROWS >= dimup) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = wg_in_chunk * groups_per_wg + wave;  // column group of this wave
  if (g * COLS >= dimdw) return;
  const int r = lane % ROWS, cc = lane / ROWS;
  const int row = min(chunk * ROWS + r, dimup - 1), c = min(g * COLS + cc, dimdw - 1);
  double2 acc = make_double2(0.0, 0.0);
  if (mode & 1) {
    for (int s = 0; s < nsrc; ++s) {
      const int sc = src[(int64_t)s * dimdw + c];
      if (sc >= 0) {
        const double2 x = v[(int64_t)sc * pitch + row];
        acc.x += x.x;
        acc.y += x.y;
      }
    }
  }
  if (mode & 2) {
    // blocked scratch: [group][row][COLS]
    dbl2_t o;
    o.x = acc.x;
    o.y = acc.y;
    __builtin_nontemporal_store(o, reinterpret_cast<dbl2_t*>(wt + ((int64_t)g * dimup + row) * COLS + cc));
  } else if (acc.x == 1.2345e300) {
    wt[0] = acc;
  }
}

extern "C" int oob_skeleton_run(const void* v, void* wt, const void* src, int nsrc, int dimup, int dimdw, int pitch, int rows, int mode, int nrep, float* ms) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int nchunks = (dimup + rows - 1) / rows, cpx = (nchunks + 7) / 8;
  const int cols = 64 / rows, groups = (dimdw + cols - 1) / cols, gpw = 16;
  const int wgs_per_chunk = (groups + gpw - 1) / gpw;
  const unsigned grid = (unsigned)(cpx * 8 * wgs_per_chunk);
  hipEventRecord(e0, 0);
  for (int i = 0; i < nrep; ++i) {
    if (rows == 16)
      hipLaunchKernelGGL(oob_skeleton<16>, dim3(grid), dim3(1024), 0, 0, (const double2*)v, (double2*)wt, (const int32_t*)src, nsrc, dimup, dimdw, pitch, cpx, gpw, mode);
    else
      hipLaunchKernelGGL(oob_skeleton<8>, dim3(grid), dim3(1024), 0, 0, (const double2*)v, (double2*)wt, (const int32_t*)src, nsrc, dimup, dimdw, pitch, cpx, gpw, mode);
  }
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  hipEventElapsedTime(ms, e0, e1);
  *ms /= nrep;
  return (int)hipGetLastError();
}
