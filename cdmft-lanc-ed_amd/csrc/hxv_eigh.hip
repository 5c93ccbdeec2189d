// hxv_eigh_lowest: the `neigen` lowest eigenpairs of the open sector, entirely on the device.
//
// It stands where the reference calls SciFortran's sp_eigh (P-ARPACK: implicitly restarted Lanczos with a
// Krylov basis of Nblock = ncv vectors) at ED_DIAG.f90:152-160.  Thick-restart Lanczos (Wu & Simon 2000) is
// the explicit-restart form of the same method for Hermitian operators: build the basis up to ncv vectors
// with measured re-orthogonalisation (every step measures all projections, subtracts the ones above rounding noise;
// one refinement pass when the norm drops 10x, the DGKS scheme ARPACK uses with a looser trigger), diagonalise the small projected matrix on the host, keep the lowest Ritz vectors by a
// tall-skinny rotation of the basis in place, continue.  Convergence test = ARPACK's:
// |beta_m * s_mi| <= tol * max(eps^(2/3), |theta_i|).
//
// Everything Dim-sized stays in HBM: ncv+1 basis vectors (C3, ncv=20: 56 GB of the 288 GB).  The vector
// kernels are plain streaming kernels (HBM-bound); per Lanczos step they read the j+1 basis vectors twice
// (multi-dot, multi-axpy), which dominates the HxV itself -- the same trade ARPACK makes on the host.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "hxv_handle.hpp"

using namespace hxv;

namespace {

constexpr int JB = 8;       // basis vectors per multi-dot pass (w is re-read once per JB vectors)
constexpr int TR_BLOCKS = 4096;  // workgroups of the streaming kernels (16 per CU)
constexpr int MAXCV = 64;
constexpr double DGKS_ETA2 = 0.01;  // refine when |w_after|^2 < eta^2 |w_before|^2
constexpr double GS_TAU = 1e-13;     // projections below tau*|w| are measured but not subtracted (see gs_pass)   // largest Krylov basis (the rotation keeps one element of every vector in registers)

__device__ inline double wave_sum(double x) {
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;
}

// partial[blockIdx][2*j..2*j+1] = sum_i conj(V_j[i]) * w[i]   for j < nb (nb <= JB)
__global__ void __launch_bounds__(256) tr_mdot(int64_t n, const double2* __restrict__ V, int64_t stride, int nb,
                                               const double2* __restrict__ w, double* __restrict__ partial) {
  double re[JB], im[JB];
#pragma unroll
  for (int j = 0; j < JB; ++j) re[j] = im[j] = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double2 x = w[i];
#pragma unroll
    for (int j = 0; j < JB; ++j)
      if (j < nb) {
        const double2 y = V[(int64_t)j * stride + i];
        re[j] += y.x * x.x + y.y * x.y;
        im[j] += y.x * x.y - y.y * x.x;
      }
  }
  __shared__ double red[4][2 * JB];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < JB; ++j) {
    const double a = wave_sum(re[j]), b = wave_sum(im[j]);
    if (lane == 0) {
      red[wave][2 * j] = a;
      red[wave][2 * j + 1] = b;
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * JB)
    partial[(int64_t)blockIdx.x * 2 * JB + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// out[t] = sum_b partial[b][t]  (t < nval), partial rows of `ld` values
__global__ void __launch_bounds__(256) tr_colsum(const double* __restrict__ partial, int nblocks, int ld, int nval, double* __restrict__ out,
                                                 int zero_odd) {
  __shared__ double red[256];
  for (int t = 0; t < nval; ++t) {
    if (zero_odd && (t & 1)) {  // REAL-vector mode: the "imaginary parts" of the reinterpreted dot products are not coefficients
      if (threadIdx.x == 0) out[t] = 0.0;
      continue;
    }
    double acc = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) acc += partial[(int64_t)b * ld + t];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[t] = red[0];
    __syncthreads();
  }
}

// w -= sum_{j<nj} c_j V_{idx[j]} ; partial[blockIdx] = sum |w|^2
__global__ void __launch_bounds__(256) tr_maxpy(int64_t n, const double2* __restrict__ V, int64_t stride, int nj, const int* __restrict__ idx,
                                                const double* __restrict__ coef, double2* __restrict__ w, double* __restrict__ partial) {
  __shared__ double sc[2 * (MAXCV + 1)];
  __shared__ int sidx[MAXCV + 1];
  for (int t = threadIdx.x; t < 2 * nj; t += 256) sc[t] = coef[t];
  for (int t = threadIdx.x; t < nj; t += 256) sidx[t] = idx[t];
  __syncthreads();
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    double2 x = w[i];
#pragma unroll 4
    for (int j = 0; j < nj; ++j) {
      const double2 y = V[(int64_t)sidx[j] * stride + i];
      const double cr = sc[2 * j], ci = sc[2 * j + 1];
      x.x -= cr * y.x - ci * y.y;
      x.y -= cr * y.y + ci * y.x;
    }
    w[i] = x;
    acc += x.x * x.x + x.y * x.y;
  }
  __shared__ double red[4];
  const double a = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// x *= r ; also usable for the norm alone (r == 1 skips the store)
__global__ void __launch_bounds__(256) tr_scale_nrm(int64_t n, double2* __restrict__ x, double r, double* __restrict__ partial) {
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    double2 a = x[i];
    acc += a.x * a.x + a.y * a.y;
    if (r != 1.0) x[i] = make_double2(a.x * r, a.y * r);
  }
  __shared__ double red[4];
  const double a = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0 && partial) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// deterministic start vector (same hash as the single-vector Lanczos in hxv_lanczos.hip): pad rows stay zero
__global__ void __launch_bounds__(256) tr_init(int64_t n, double2* __restrict__ q, uint64_t seed, int dimup, int pitch, int col0,
                                               const int32_t* __restrict__ iperm, const uint8_t* __restrict__ sign) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t lcol = i / pitch;
    const int row = (int)(i - lcol * pitch);
    const int64_t col = lcol + col0;  // global column: a split sector starts from the same vector as the unsplit one
    if (row >= dimup) {
      q[i] = make_double2(0.0, 0.0);
      continue;
    }
    // (device row order: the vector is defined on the reference index, see lz_init)
    const int rrow = iperm ? iperm[row] : row;
    const double sgn = (sign && sign[row]) ? -1.0 : 1.0;
    const uint64_t z = (uint64_t)(col * dimup + rrow) * 2 + seed;
    double r[2];
    for (int k = 0; k < 2; ++k) {
      uint64_t x = z + (uint64_t)k + 0x9E3779B97F4A7C15ull;
      x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
      x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
      x = x ^ (x >> 31);
      r[k] = (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    }
    q[i] = make_double2(sgn * r[0], sgn * r[1]);
  }
}

// In-place basis rotation  V[:, 0..k) <- V[:, 0..m) * S  (S real m x k, column-major S[l + j*m]).
// Each thread keeps one element of all m vectors in registers, so every vector is read once and the
// first k are written once: (m + k) vector passes instead of m*k.
template <int MAXM>
__global__ void __launch_bounds__(256) tr_rotate(int64_t n, double2* __restrict__ V, int64_t stride, int m, int k,
                                                 const double* __restrict__ S) {
  __shared__ double sS[MAXM * MAXM];
  for (int t = threadIdx.x; t < m * k; t += 256) sS[t] = S[t];
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    double2 x[MAXM];
#pragma unroll
    for (int l = 0; l < MAXM; ++l)
      if (l < m) x[l] = V[(int64_t)l * stride + i];
    for (int j = 0; j < k; ++j) {
      double yr = 0.0, yi = 0.0;
#pragma unroll
      for (int l = 0; l < MAXM; ++l)
        if (l < m) {
          const double s = sS[l + j * m];
          yr += s * x[l].x;
          yi += s * x[l].y;
        }
      V[(int64_t)j * stride + i] = make_double2(yr, yi);
    }
  }
}

void launch_rotate(int grid, hipStream_t st, int64_t n, double2* V, int64_t stride, int m, int k, const double* S) {
  if (m <= 8)
    hipLaunchKernelGGL(tr_rotate<8>, dim3(grid), dim3(256), 0, st, n, V, stride, m, k, S);
  else if (m <= 16)
    hipLaunchKernelGGL(tr_rotate<16>, dim3(grid), dim3(256), 0, st, n, V, stride, m, k, S);
  else if (m <= 32)
    hipLaunchKernelGGL(tr_rotate<32>, dim3(grid), dim3(256), 0, st, n, V, stride, m, k, S);
  else
    hipLaunchKernelGGL(tr_rotate<MAXCV>, dim3(grid), dim3(256), 0, st, n, V, stride, m, k, S);
}

// First step of a restart cycle, one pass instead of two: x = a*w - sum_{l<nv} c_l V_l (the arrow of T taken out with its KNOWN coefficients;
// c_l = 0 for vectors that are only measured), stored back to w, and partial[blockIdx][2l..2l+1] = sum_i conj(V_l[i]) x[i] for all l < nv --
// the measurement the following Gram-Schmidt pass needs.  Every V_l[i] is loaded once for both.  nv <= NJ.
template <int NJ>
__global__ void __launch_bounds__(256) tr_axpy_mdot(int64_t n, const double2* __restrict__ V, int64_t stride, int nv, const double* __restrict__ coef,
                                                    double a, double2* __restrict__ w, double* __restrict__ partial) {
  __shared__ double sc[NJ];
  if ((int)threadIdx.x < NJ) sc[threadIdx.x] = (int)threadIdx.x < nv ? coef[threadIdx.x] : 0.0;
  __syncthreads();
  double re[NJ], im[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) re[j] = im[j] = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    double2 y[NJ];
    double2 x = w[i];
    x.x *= a;
    x.y *= a;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (j < nv) {
        y[j] = V[(int64_t)j * stride + i];
        x.x -= sc[j] * y[j].x;
        x.y -= sc[j] * y[j].y;
      }
    w[i] = x;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (j < nv) {
        re[j] += y[j].x * x.x + y[j].y * x.y;
        im[j] += y[j].x * x.y - y[j].y * x.x;
      }
  }
  __shared__ double red[4][2 * NJ];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const double p = wave_sum(re[j]), q = wave_sum(im[j]);
    if (lane == 0) {
      red[wave][2 * j] = p;
      red[wave][2 * j + 1] = q;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < 2 * NJ)
    partial[(int64_t)blockIdx.x * 2 * NJ + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// The restart rotation (tr_rotate) that also brings the residual vector r = V[m] to its new place V[k] and measures it against the k new
// Ritz vectors as they are formed: partial[blockIdx][2j..2j+1] = sum_i conj(y_j[i]) r[i].  Replaces rotation + copy + one multi-dot pass.  k <= KD.
template <int MAXM, int KD>
__global__ void __launch_bounds__(256) tr_rotate_dots(int64_t n, double2* __restrict__ V, int64_t stride, int m, int k, const double* __restrict__ S,
                                                      double* __restrict__ partial) {
  __shared__ double sS[MAXM * KD];
  for (int t = threadIdx.x; t < m * k; t += 256) sS[t] = S[t];
  __syncthreads();
  double re[KD], im[KD];
#pragma unroll
  for (int j = 0; j < KD; ++j) re[j] = im[j] = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    double2 x[MAXM];
#pragma unroll
    for (int l = 0; l < MAXM; ++l)
      if (l < m) x[l] = V[(int64_t)l * stride + i];
    const double2 r = V[(int64_t)m * stride + i];
#pragma unroll
    for (int j = 0; j < KD; ++j)
      if (j < k) {
        double yr = 0.0, yi = 0.0;
#pragma unroll
        for (int l = 0; l < MAXM; ++l)
          if (l < m) {
            const double sv = sS[l + j * m];
            yr += sv * x[l].x;
            yi += sv * x[l].y;
          }
        V[(int64_t)j * stride + i] = make_double2(yr, yi);
        re[j] += yr * r.x + yi * r.y;
        im[j] += yr * r.y - yi * r.x;
      }
    V[(int64_t)k * stride + i] = r;
  }
  __shared__ double red[4][2 * KD];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < KD; ++j) {
    const double p = wave_sum(re[j]), q = wave_sum(im[j]);
    if (lane == 0) {
      red[wave][2 * j] = p;
      red[wave][2 * j + 1] = q;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < 2 * KD)
    partial[(int64_t)blockIdx.x * 2 * KD + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

constexpr int FUSE_NJ = 16;  // most vectors the two fused restart kernels handle (larger kept sets take the separate passes)

bool launch_rotate_dots(int grid, hipStream_t st, int64_t n, double2* V, int64_t stride, int m, int k, const double* S, double* partial) {
  if (k > FUSE_NJ) return false;
  if (m <= 8)
    hipLaunchKernelGGL((tr_rotate_dots<8, FUSE_NJ>), dim3(grid), dim3(256), 0, st, n, V, stride, m, k, S, partial);
  else if (m <= 16)
    hipLaunchKernelGGL((tr_rotate_dots<16, FUSE_NJ>), dim3(grid), dim3(256), 0, st, n, V, stride, m, k, S, partial);
  else if (m <= 32)
    hipLaunchKernelGGL((tr_rotate_dots<32, FUSE_NJ>), dim3(grid), dim3(256), 0, st, n, V, stride, m, k, S, partial);
  else
    hipLaunchKernelGGL((tr_rotate_dots<MAXCV, FUSE_NJ>), dim3(grid), dim3(256), 0, st, n, V, stride, m, k, S, partial);
  return true;
}

// Cyclic Jacobi for a small dense real symmetric matrix (column-major n x n in A, destroyed).
// On exit w = eigenvalues ascending, Z[:, i] = eigenvector i.
bool jacobi_eigh(int n, std::vector<double>& A, std::vector<double>& w, std::vector<double>& Z) {
  Z.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) Z[i + (size_t)i * n] = 1.0;
  auto a = [&](int i, int j) -> double& { return A[i + (size_t)j * n]; };
  double scale = 0.0;
  for (double x : A) scale = std::max(scale, std::fabs(x));
  if (scale == 0.0) scale = 1.0;
  bool done = false;
  for (int sweep = 0; sweep < 100 && !done; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) off += a(p, q) * a(p, q);
    if (std::sqrt(off) <= 1e-15 * scale) {
      done = true;
      break;
    }
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = a(p, q);
        if (std::fabs(apq) <= 1e-300) continue;
        const double tau = (a(q, q) - a(p, p)) / (2.0 * apq);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c;
        for (int r = 0; r < n; ++r) {  // columns p, q
          const double arp = a(r, p), arq = a(r, q);
          a(r, p) = c * arp - s * arq;
          a(r, q) = s * arp + c * arq;
        }
        for (int r = 0; r < n; ++r) {  // rows p, q
          const double apr = a(p, r), aqr = a(q, r);
          a(p, r) = c * apr - s * aqr;
          a(q, r) = s * apr + c * aqr;
        }
        a(p, q) = a(q, p) = 0.0;
        for (int r = 0; r < n; ++r) {
          const double zrp = Z[r + (size_t)p * n], zrq = Z[r + (size_t)q * n];
          Z[r + (size_t)p * n] = c * zrp - s * zrq;
          Z[r + (size_t)q * n] = s * zrp + c * zrq;
        }
      }
  }
  std::vector<int> idx(n);
  for (int i = 0; i < n; ++i) idx[i] = i;
  std::sort(idx.begin(), idx.end(), [&](int x, int y) { return a(x, x) < a(y, y); });
  w.resize(n);
  std::vector<double> Zs((size_t)n * n);
  for (int i = 0; i < n; ++i) {
    w[i] = a(idx[i], idx[i]);
    std::memcpy(&Zs[(size_t)i * n], &Z[(size_t)idx[i] * n], (size_t)n * sizeof(double));
  }
  Z.swap(Zs);
  return done;
}

// Ritz pairs kept at a restart: the wanted ones, those already converged, and a share of the rest of the basis (option "eigh_keep_pct",
// per cent of m - neigen; the restart cycle then has m - k new products)
int keep_count(int m, int neigen, int nconv, int pct) {
  int k = neigen + std::min(nconv, (m - neigen) / 2) + std::max(1, (m - neigen) * pct / 100);
  return std::max(1, std::min(k, m - 1));
}

struct DevFree {  // small buffers: hipFree; `pooled`: vector-sized ones, back to the engine's cache (hxv_pool.cpp)
  std::vector<void*> p, pooled;
  int device = 0;
  ~DevFree() {
    if (!pooled.empty()) (void)hipDeviceSynchronize();  // nothing may still be using a block that the next handle gets
    for (void* q : pooled) pool_free(device, q);
    for (void* q : p)
      if (q) (void)hipFree(q);
  }
};

}  // namespace

extern "C" {

int hxv_eigh_lowest(hxv_handle* h, int32_t neigen, int32_t ncv, int32_t maxrestart, double tol, double* evals, void* d_evecs,
                    int32_t* nconv_out, int32_t* nmatvec_out) {
  if (!h || !evals || neigen < 1 || maxrestart < 0) return fail(HXV_ERR_ARG, "hxv_eigh_lowest: bad argument");
  if (h->host.nranks != 1 && !comm_ready(h))
    return fail(HXV_ERR_STATE, "hxv_eigh_lowest on a split sector needs the communicator: call hxv_comm_init after opening the sector");
  HIPCHK(hipSetDevice(h->device));  // (before the first collective: thread ranks on several GPUs each have their own current device)
  const int64_t dim = h->host.dim;
  if (ncv <= 0) ncv = 10 * neigen;  // the reference's default: lanc_ncv_factor=10, lanc_ncv_add=0 (ED_INPUT_VARS.f90:174-175)
  const int m = (int)std::min<int64_t>(std::max(ncv, neigen + 1), dim);
  {
    // argument errors are agreed on by the ranks of a split sector as well (a rank whose caller passed something else must not
    // leave the others waiting in the first all-reduce)
    int arg_rc = HXV_OK;
    if (neigen > dim) arg_rc = fail(HXV_ERR_ARG, "hxv_eigh_lowest: neigen > Dim");
    else if (m > MAXCV) arg_rc = fail(HXV_ERR_ARG, "hxv_eigh_lowest: ncv > 64 is not supported");
    arg_rc = comm_agree(h, arg_rc);
    if (arg_rc) return arg_rc;
  }
  HIPCHK(hipSetDevice(h->device));
  // REAL-vector mode (H real, our own real start vector): the basis holds double[DimDw][pitch_real]; every kernel below
  // is elementwise with real coefficients, so it runs unchanged on the vectors viewed as n double2 elements.
  const bool real = h->real_vectors && !real_mode_blocker(h);
  h->last_real = real ? 1 : 0;
  const int64_t nc = (int64_t)h->host.pitch * h->host.qdw;  // this rank's padded complex slab (pads are zero and stay zero)
  const int64_t n = real ? (int64_t)pitch_real_of(h) * h->host.qdw / 2 : nc;
  const int g = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, TR_BLOCKS));  // (never an empty grid: a rank may own no column)
  const double eps = 2.220446049250313e-16, eps23 = std::pow(eps, 2.0 / 3.0);
  tol = std::max(tol, eps);

  size_t free_b = 0, total_b = 0;
  HIPCHK(hipMemGetInfo(&free_b, &total_b));
  {  // blocks held by the engine's own device-buffer cache are available to pool_alloc (reused or trimmed on OOM)
    int64_t cached = 0;
    (void)hxv_pool_stats(h->device, &cached, nullptr, nullptr);
    free_b += (size_t)std::max<int64_t>(cached, 0);
  }
  const size_t need = (size_t)(m + 1) * (size_t)std::max<int64_t>(n, 1) * sizeof(double2);
  // (the shortfall is rank-local: the ranks of a split sector agree on it before anybody enters a collective)
  int short_rc = HXV_OK;
  if (need + ((size_t)n * sizeof(double2)) > free_b)
    short_rc = fail(HXV_ERR_HIP, "hxv_eigh_lowest: the Krylov basis needs " + std::to_string(need >> 20) + " MiB of HBM for ncv=" + std::to_string(m) +
                                     " but only " + std::to_string(free_b >> 20) + " MiB are free: lower ncv or use hxv_lanczos_eigh (3 vectors)");
  short_rc = comm_agree(h, short_rc);
  if (short_rc) return short_rc;
  DevFree mem;
  double2* V = nullptr;
  double *d_part = nullptr, *d_coef = nullptr, *d_S = nullptr;
  mem.device = h->device;
  {
    hipError_t ea = pool_alloc(h->device, need, (void**)&V);
    int rca = comm_agree(h, ea == hipSuccess ? HXV_OK : fail(HXV_ERR_HIP, std::string("hxv_eigh_lowest: Krylov basis allocation: ") + hipGetErrorString(ea)));
    if (rca) {
      if (ea == hipSuccess) pool_free(h->device, V);
      return rca;
    }
  }
  mem.pooled.push_back(V);
  HIPCHK(hipMemsetAsync(V, 0, need, h->stream));  // pad rows must be zero: the products never write them, the dots read them
  HIPCHK(hipMalloc((void**)&d_part, (size_t)TR_BLOCKS * (2 * FUSE_NJ + 1) * sizeof(double)));  // ([blocks][2*JB] of tr_mdot or [blocks][2*FUSE_NJ] of the fused kernels, + [blocks] norms)
  mem.p.push_back(d_part);
  HIPCHK(hipMalloc((void**)&d_coef, (size_t)(2 * (MAXCV + 1) + 2) * sizeof(double)));
  mem.p.push_back(d_coef);
  HIPCHK(hipMalloc((void**)&d_S, (size_t)MAXCV * MAXCV * sizeof(double)));
  mem.p.push_back(d_S);
  double* d_csel = nullptr;
  int* d_isel = nullptr;
  HIPCHK(hipMalloc((void**)&d_csel, (size_t)2 * (MAXCV + 1) * sizeof(double)));
  mem.p.push_back(d_csel);
  HIPCHK(hipMalloc((void**)&d_isel, (size_t)(MAXCV + 1) * sizeof(int)));
  mem.p.push_back(d_isel);
  double* d_npart = d_part + (size_t)TR_BLOCKS * 2 * FUSE_NJ;
  double* d_nrm = d_coef + 2 * (MAXCV + 1);
  hipStream_t st = h->stream;
  auto vec = [&](int j) { return V + (int64_t)j * n; };

  // One Gram-Schmidt pass of w = V[jt+1] against V[0..jt] (jt = absolute index; the first `nlock` vectors are LOCKED
  // eigenvectors, see below).  ALL jt+1 projections are measured every step (tr_mdot), so the orthogonality of the basis
  // is known, not assumed; but only those that matter are subtracted (tr_maxpy): the locked vectors, the two local ones
  // (alpha_j v_j, beta_j v_{j-1}), everything right after a restart (the arrow) or in a refinement pass, and whatever
  // exceeds GS_TAU*|w| -- in practice the kept Ritz vectors that are close to convergence, which is where a Lanczos basis
  // loses orthogonality (Paige).  The rest are rounding noise (<= 1e-13 relative): skipping them leaves the basis
  // orthogonal to ~1e-12 and saves about two thirds of the update traffic.
  // The basis is stored UNNORMALISED where that saves a pass: slot i holds nv[i] * (the unit vector v_i).  A Lanczos step leaves its new
  // vector as it comes out of the recurrence (nv = its norm = beta) and the next product divides in its epilogue; the projections below
  // fold the factors into their coefficients (<v_i, w> = <V_i, w> / nv[i]; w -= <v_i, w> v_i = (<V_i, w> / nv[i]^2) V_i), the restart
  // rotation into the rows of S.  c[] always holds the coefficients on the UNIT vectors.
  std::vector<double> nv(MAXCV + 2, 1.0);
  std::vector<double> c(2 * (MAXCV + 1)), csel(2 * (MAXCV + 1));
  std::vector<int> isel(MAXCV + 1);
  // (dots_ready: d_coef already holds the raw sums <V_i, w> of all jt+1 vectors -- a fused kernel measured them on its way)
  auto gs_pass = [&](int jt, int nlock, bool all, double* nrm2_after, bool dots_ready = false) -> int {
    const int nj = jt + 1;
    for (int g0 = 0; g0 < nj && !dots_ready; g0 += JB) {
      const int nb = std::min(JB, nj - g0);
      hipLaunchKernelGGL(tr_mdot, dim3(g), dim3(256), 0, st, n, vec(g0), n, nb, vec(jt + 1), d_part);
      hipLaunchKernelGGL(tr_colsum, dim3(1), dim3(256), 0, st, d_part, g, 2 * JB, 2 * nb, d_coef + 2 * g0, real ? 1 : 0);
    }
    if (int rca = comm_allreduce_sum(h, d_coef, (size_t)2 * nj, st)) return rca;
    HIPCHK(hipMemcpyAsync(c.data(), d_coef, (size_t)2 * nj * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < nj; ++i)
      if (nv[i] != 1.0) {
        c[2 * i] /= nv[i];
        c[2 * i + 1] /= nv[i];
      }
    double ssum = 0.0;
    for (int t = 0; t < 2 * nj; ++t) ssum += c[t] * c[t];
    const double thr2 = GS_TAU * GS_TAU * ssum;
    int nsel = 0;
    for (int i = 0; i < nj; ++i)
      if (all || i < nlock || i + 1 >= jt || c[2 * i] * c[2 * i] + c[2 * i + 1] * c[2 * i + 1] > thr2) {
        isel[nsel] = i;
        csel[2 * nsel] = c[2 * i] / nv[i];
        csel[2 * nsel + 1] = c[2 * i + 1] / nv[i];
        ++nsel;
      }
    HIPCHK(hipMemcpyAsync(d_csel, csel.data(), (size_t)2 * nsel * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_isel, isel.data(), (size_t)nsel * sizeof(int), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(tr_maxpy, dim3(g), dim3(256), 0, st, n, V, n, nsel, d_isel, d_csel, vec(jt + 1), d_npart);
    hipLaunchKernelGGL(tr_colsum, dim3(1), dim3(256), 0, st, d_npart, g, 1, 1, d_nrm, 0);
    if (int rca = comm_allreduce_sum(h, d_nrm, 1, st)) return rca;
    HIPCHK(hipMemcpyAsync(nrm2_after, d_nrm, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return HXV_OK;
  };
  // (HXV_EIGH_TRACE only) the true projections of w = V[jt+1] on V[0..jt-2], measured and NOT removed: max |<v_b, w>| and where -- printed next
  // to the estimate of the omega recurrence at every local step (how round 4 found the single-pass flaw of the cleaning steps)
  auto measure_only = [&](int jt, double* mx, int* where) -> int {
    const int nj = jt - 1;
    *mx = 0.0;
    *where = -1;
    if (nj <= 0) return HXV_OK;
    std::vector<double> cc(2 * (MAXCV + 1));
    for (int g0 = 0; g0 < nj; g0 += JB) {
      const int nb = std::min(JB, nj - g0);
      hipLaunchKernelGGL(tr_mdot, dim3(g), dim3(256), 0, st, n, vec(g0), n, nb, vec(jt + 1), d_part);
      hipLaunchKernelGGL(tr_colsum, dim3(1), dim3(256), 0, st, d_part, g, 2 * JB, 2 * nb, d_coef + 2 * g0, real ? 1 : 0);
    }
    HIPCHK(hipMemcpyAsync(cc.data(), d_coef, (size_t)2 * nj * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < nj; ++i) {
      const double a = std::sqrt(cc[2 * i] * cc[2 * i] + cc[2 * i + 1] * cc[2 * i + 1]) / nv[i];
      if (a > *mx) *mx = a, *where = i;
    }
    return HXV_OK;
  };
  auto norm2_of = [&](double2* x, double* out) -> int {
    hipLaunchKernelGGL(tr_scale_nrm, dim3(g), dim3(256), 0, st, n, x, 1.0, d_npart);
    hipLaunchKernelGGL(tr_colsum, dim3(1), dim3(256), 0, st, d_npart, g, 1, 1, d_nrm, 0);
    if (int rca = comm_allreduce_sum(h, d_nrm, 1, st)) return rca;  // split sector: projections and norms are sums over the ranks
    HIPCHK(hipMemcpyAsync(out, d_nrm, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return HXV_OK;
  };

  // The two LOCAL projections (alpha_j on V[jt], beta_j on V[jt-1]) -- what a Lanczos step needs when the basis is known (by the
  // omega recurrence below) to be semi-orthogonal -- PLUS the projections on the LOCKED eigenvectors V[0..nlock): a locked pair
  // is converged only to the caller's tolerance, so every product re-injects a component along it of the order of its residual
  // (tol*|theta|, far above rounding when the caller asks for a loose tol); left in, the complement round could converge back onto
  // a locked state and report it as a second copy.  nlock <= neigen vectors: cheap.  c[2*i] is set for every vector touched.
  auto gs_local = [&](int jt, int nlock, bool has_prev, double* nrm2_after) -> int {
    const int b0 = has_prev ? jt - 1 : jt, nb = has_prev ? 2 : 1;
    for (int g0 = 0; g0 < nlock; g0 += JB) {
      const int nl = std::min(JB, nlock - g0);
      hipLaunchKernelGGL(tr_mdot, dim3(g), dim3(256), 0, st, n, vec(g0), n, nl, vec(jt + 1), d_part);
      hipLaunchKernelGGL(tr_colsum, dim3(1), dim3(256), 0, st, d_part, g, 2 * JB, 2 * nl, d_coef + 2 * g0, real ? 1 : 0);
    }
    hipLaunchKernelGGL(tr_mdot, dim3(g), dim3(256), 0, st, n, vec(b0), n, nb, vec(jt + 1), d_part);
    hipLaunchKernelGGL(tr_colsum, dim3(1), dim3(256), 0, st, d_part, g, 2 * JB, 2 * nb, d_coef + 2 * nlock, real ? 1 : 0);
    const int nsel = nlock + nb;
    if (int rca = comm_allreduce_sum(h, d_coef, (size_t)2 * nsel, st)) return rca;
    HIPCHK(hipMemcpyAsync(csel.data(), d_coef, (size_t)2 * nsel * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < nsel; ++i) {
      const int iv = i < nlock ? i : b0 + (i - nlock);
      isel[i] = iv;
      c[2 * iv] = csel[2 * i] / nv[iv];
      c[2 * iv + 1] = csel[2 * i + 1] / nv[iv];
      csel[2 * i] = c[2 * iv] / nv[iv];
      csel[2 * i + 1] = c[2 * iv + 1] / nv[iv];
    }
    HIPCHK(hipMemcpyAsync(d_csel, csel.data(), (size_t)2 * nsel * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_isel, isel.data(), (size_t)nsel * sizeof(int), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(tr_maxpy, dim3(g), dim3(256), 0, st, n, V, n, nsel, d_isel, d_csel, vec(jt + 1), d_npart);
    hipLaunchKernelGGL(tr_colsum, dim3(1), dim3(256), 0, st, d_npart, g, 1, 1, d_nrm, 0);
    if (int rca = comm_allreduce_sum(h, d_nrm, 1, st)) return rca;
    HIPCHK(hipMemcpyAsync(nrm2_after, d_nrm, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return HXV_OK;
  };

  // A fused step of a LOCKING round: the product's epilogue has already removed beta_j v_{j-1} from w = V[jt+1] and measured alpha_j;
  // here the projections on the locked eigenvectors are measured and removed together with alpha_j v_j in ONE update pass.
  auto gs_locked_alpha = [&](int jt, int nlock, double alpha, double* nrm2_after) -> int {
    for (int g0 = 0; g0 < nlock; g0 += JB) {
      const int nl = std::min(JB, nlock - g0);
      hipLaunchKernelGGL(tr_mdot, dim3(g), dim3(256), 0, st, n, vec(g0), n, nl, vec(jt + 1), d_part);
      hipLaunchKernelGGL(tr_colsum, dim3(1), dim3(256), 0, st, d_part, g, 2 * JB, 2 * nl, d_coef + 2 * g0, real ? 1 : 0);
    }
    if (int rca = comm_allreduce_sum(h, d_coef, (size_t)2 * nlock, st)) return rca;
    HIPCHK(hipMemcpyAsync(csel.data(), d_coef, (size_t)2 * nlock * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < nlock; ++i) {
      isel[i] = i;
      c[2 * i] = csel[2 * i] / nv[i];
      c[2 * i + 1] = csel[2 * i + 1] / nv[i];
      csel[2 * i] = c[2 * i] / nv[i];
      csel[2 * i + 1] = c[2 * i + 1] / nv[i];
    }
    isel[nlock] = jt;
    csel[2 * nlock] = alpha / nv[jt];
    csel[2 * nlock + 1] = 0.0;
    const int nsel = nlock + 1;
    HIPCHK(hipMemcpyAsync(d_csel, csel.data(), (size_t)2 * nsel * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_isel, isel.data(), (size_t)nsel * sizeof(int), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(tr_maxpy, dim3(g), dim3(256), 0, st, n, V, n, nsel, d_isel, d_csel, vec(jt + 1), d_npart);
    hipLaunchKernelGGL(tr_colsum, dim3(1), dim3(256), 0, st, d_npart, g, 1, 1, d_nrm, 0);
    if (int rca = comm_allreduce_sum(h, d_nrm, 1, st)) return rca;
    HIPCHK(hipMemcpyAsync(nrm2_after, d_nrm, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return HXV_OK;
  };
  // bring a stored vector to unit length where a kernel needs it so (the plain product, the start of a round)
  auto normalise_slot = [&](int i) {
    if (nv[i] != 1.0) {
      hipLaunchKernelGGL(tr_scale_nrm, dim3(g), dim3(256), 0, st, n, vec(i), 1.0 / nv[i], (double*)nullptr);
      nv[i] = 1.0;
    }
  };

  // Thick-restart Lanczos for the `nwant` lowest pairs of H restricted to the orthogonal complement of the `nlock`
  // LOCKED eigenvectors V[0..nlock) (nlock = 0: H itself).  The active basis is V[nlock..nlock+ma]; on return its first
  // `ne` vectors are the Ritz vectors of theta[0..ne).
  std::vector<double> T, A, theta, S, Ssc;
  std::vector<double> lockval;  // eigenvalues of the locked vectors V[0..nlock)
  int nmv = 0, nconv = 0, ne = neigen;
  int n_full = 0, n_local = 0;  // Gram-Schmidt passes against the whole basis / against the two local vectors only
  bool closed = false;  // the Krylov space closed (invariant subspace): every returned pair is exact
  bool above = false;   // a check round stopped early: the lowest Ritz value minus its residual bound is already above `stop_above`
  const bool local_fused = lanczos_local_step_available(h);
  const bool trace = getenv("HXV_EIGH_TRACE") != nullptr;
  auto trl = [&](int nlock, int nwant, uint64_t seed, double stop_above) -> int {
    above = false;
    const int ma = m - nlock;  // active basis size
    double2* Va = vec(nlock);
    auto av = [&](int j) { return vec(nlock + j); };
    T.assign((size_t)ma * ma, 0.0);
    auto t_at = [&](int i, int j) -> double& { return T[i + (size_t)j * ma]; };
    for (int i = nlock; i <= m; ++i) nv[i] = 1.0;  // (the locked vectors are unit vectors: every round ends with a rotation)
    // start vector (deterministic hash of the global index; a different seed per round), made orthogonal to the locked set
    if (real)
      launch_init_real(h, (double*)av(0), seed, st);
    else
      hipLaunchKernelGGL(tr_init, dim3(g), dim3(256), 0, st, n, av(0), seed, h->host.dimup, h->host.pitch, h->host.dw0, h->dev.up_iperm, h->dev.up_sign);
    double nrm2 = 0.0;
    if (nlock > 0) {
      for (int pass = 0; pass < 2; ++pass) {  // V[nlock] plays w: project the locked vectors out, twice
        int rc = gs_pass(nlock - 1, nlock, true, &nrm2);
        if (rc) return rc;
      }
    } else {
      int rc = norm2_of(av(0), &nrm2);
      if (rc) return rc;
    }
    if (!(nrm2 > 0.0)) return fail(HXV_ERR_STATE, "hxv_eigh_lowest: start vector vanished");
    hipLaunchKernelGGL(tr_scale_nrm, dim3(g), dim3(256), 0, st, n, av(0), 1.0 / std::sqrt(nrm2), (double*)nullptr);
    int k = 0, meff = ma;
    double beta_last = 0.0;
    std::vector<double> om_prev(m + 2, eps), om_cur(m + 2, eps), om_next(m + 2, eps), theta_keep, s_keep;
    bool force_full = false;
    double res0_prev = 1e300;
    for (int it = 0;; ++it) {
      meff = ma;
      beta_last = 0.0;
      for (int j = k; j < ma; ++j) {
        const int jt = nlock + j;
        // a step that is known in advance to need only the two local projections runs through the fused product: pass A
        // subtracts beta_j q_{j-1} and reduces alpha_j in its epilogue, one more pass subtracts alpha_j q_j and measures |w|
        // (not in the locking rounds: there every step also removes the locked eigenvectors, see gs_local)
        // (a FORCED step -- the second vector in a row to be cleaned against the whole basis -- takes its local step first, like any other:
        //  one classical Gram-Schmidt pass over "H q_j" would measure the projection on an old vector v_b BEFORE beta_j q_{j-1} is taken
        //  out, and q_{j-1} carries the very overlap with v_b that the step is there to stop -- the new vector would inherit it, the
        //  estimates would say "clean", and a basis of more than ~30 vectors would lose a converged Ritz vector within a few cycles)
        const bool forced = force_full && j > k && !h->eigh_measure_all;
        const bool fused_local = local_fused && !h->eigh_measure_all && j > k;
        int rc;
        double w2 = 0.0, arrow2 = 0.0;  // arrow2: squared length of what the first step of a cycle takes out with the known coefficients
        bool first_dots_ready = false;  // the fused first step has measured the projections already
        if (fused_local) {
          double al = 0.0, nw = 0.0;
          rc = lanczos_local_step(h, real, av(j), nv[jt], av(j - 1), nv[jt - 1], t_at(j, j - 1), av(j + 1), nlock == 0, &al, &nw);
          if (rc) return rc;
          c[2 * jt] = al;
          c[2 * jt + 1] = 0.0;
          c[2 * (jt - 1)] = t_at(j, j - 1);
          c[2 * (jt - 1) + 1] = 0.0;
          if (nlock == 0)
            w2 = nw * nw;
          else if ((rc = gs_locked_alpha(jt, nlock, al, &w2)))
            return rc;
        } else {
          // (the first step of a cycle can take its input as stored: the fused kernel below divides by its length)
          const bool fuse_first = j == k && k > 0 && !h->eigh_measure_all && jt + 1 <= FUSE_NJ && h->eigh_fuse_restart;
          if (!fuse_first) normalise_slot(jt);
          rc = real ? apply_slab_real(h, (const double*)av(j), (double*)av(j + 1), st) : apply_slab(h, av(j), av(j + 1), st);
          if (rc) return rc;
          if (j == k && k > 0 && !h->eigh_measure_all) {
            // First step of a cycle: H q_k couples to every kept Ritz vector with the KNOWN coefficient s_l (the arrow of T).  They are
            // taken out first, with those coefficients, and the measured pass below then only finds what is left (rounding-size
            // projections).  One classical pass over "H q_k" would instead measure a projection BEFORE the others are subtracted, and the
            // kept Ritz vectors are orthogonal to each other only as far as the old basis was: the new vector would inherit
            // sum_l s_l <y_b, y_l> against each of them -- above the estimates' starting level after a long cycle.
            for (int l = 0; l < k; ++l) arrow2 += s_keep[l] * s_keep[l];
            if (fuse_first) {
              // ONE pass: w = H(stored q_k) / |stored q_k| - arrow, and the projections of that w on all jt + 1 vectors (round 5; was an
              // update pass plus a multi-dot pass)
              std::vector<double> cf(FUSE_NJ, 0.0);
              for (int l = 0; l < k; ++l) cf[nlock + l] = s_keep[l] / (nv[nlock + l] * nv[nlock + l]);
              HIPCHK(hipMemcpyAsync(d_csel, cf.data(), (size_t)FUSE_NJ * sizeof(double), hipMemcpyHostToDevice, st));
              hipLaunchKernelGGL((tr_axpy_mdot<FUSE_NJ>), dim3(g), dim3(256), 0, st, n, V, n, jt + 1, d_csel, 1.0 / nv[jt], av(j + 1), d_part);
              hipLaunchKernelGGL(tr_colsum, dim3(1), dim3(256), 0, st, d_part, g, 2 * FUSE_NJ, 2 * (jt + 1), d_coef, real ? 1 : 0);
              HIPCHK(hipStreamSynchronize(st));  // (cf is a host buffer)
              first_dots_ready = true;
            } else {
              for (int l = 0; l < k; ++l) {
                isel[l] = nlock + l;
                csel[2 * l] = s_keep[l] / (nv[nlock + l] * nv[nlock + l]);
                csel[2 * l + 1] = 0.0;
              }
              HIPCHK(hipMemcpyAsync(d_csel, csel.data(), (size_t)2 * k * sizeof(double), hipMemcpyHostToDevice, st));
              HIPCHK(hipMemcpyAsync(d_isel, isel.data(), (size_t)k * sizeof(int), hipMemcpyHostToDevice, st));
              hipLaunchKernelGGL(tr_maxpy, dim3(g), dim3(256), 0, st, n, V, n, k, d_isel, d_csel, av(j + 1), d_npart);
              HIPCHK(hipStreamSynchronize(st));  // (csel / isel are host buffers the pass below reuses)
            }
          }
        }
        ++nmv;
        // Partial re-orthogonalisation (Simon 1984) in its thick-restart form: the loss of orthogonality of the next
        // vector against every earlier basis vector is ESTIMATED from the recurrence the exact quantities obey
        // (om_*: <q_j, v_b>), and the whole basis is only touched when an estimate passes sqrt(eps) -- then for two
        // consecutive vectors.  Otherwise a step costs the two local projections.  (Option "eigh_measure_all" = 1
        // restores the round-1 behaviour: every projection measured at every step.)
        const bool first_after_restart = j == k;
        bool full = h->eigh_measure_all || first_after_restart;
        if (full) {
          rc = gs_pass(jt, nlock, first_after_restart, &w2, first_dots_ready);
          ++n_full;
        } else if (fused_local) {
          ++n_local;
        } else {
          rc = gs_local(jt, nlock, j > k, &w2);
          ++n_local;
        }
        if (rc) return rc;
        t_at(j, j) = c[2 * jt];
        double c2sum = 0.0;
        if (full) {
          for (int t = 0; t < 2 * (jt + 1); ++t) c2sum += c[t] * c[t];
          c2sum += arrow2;
        } else
          c2sum = c[2 * jt] * c[2 * jt] + c[2 * jt + 1] * c[2 * jt + 1] + (j > k ? c[2 * (jt - 1)] * c[2 * (jt - 1)] + c[2 * (jt - 1) + 1] * c[2 * (jt - 1) + 1] : 0.0);
        if (!full)
          for (int b = 0; b < nlock; ++b) c2sum += c[2 * b] * c[2 * b] + c[2 * b + 1] * c[2 * b + 1];
        double nrm = std::sqrt(std::max(w2, 0.0));
        bool was_forced = force_full;
        force_full = false;
        if (!h->eigh_measure_all) {
          double tsc = 1.0;
          for (int a = 0; a <= j; ++a) tsc = std::max(tsc, std::fabs(t_at(a, a)));
          const double noise = 2.0 * eps * tsc / std::max(nrm, 1e-300);
          if (full) {
            for (int b = 0; b <= jt; ++b) om_next[b] = noise;  // measured and removed: orthogonal to rounding
          } else {
            auto omc = [&](int x) -> double { return x == jt ? 1.0 : (x == jt - 1 ? noise : om_cur[x]); };
            const double alpha_j = t_at(j, j), beta_j = j > k ? t_at(j, j - 1) : 0.0;
            double maxom = 0.0;
            for (int b = 0; b < jt - 1; ++b) {
              double at;
              if (b < nlock) {
                om_next[b] = noise;  // measured and removed at every step (gs_local)
                continue;
              } else {
                const int l = b - nlock;
                if (l < k) {
                  at = theta_keep[l] * om_cur[b] + s_keep[l] * omc(nlock + k);
                } else {
                  at = t_at(l, l) * om_cur[b] + t_at(l + 1, l) * omc(b + 1);
                  if (l > k)
                    at += t_at(l, l - 1) * omc(b - 1);
                  else
                    for (int i2 = 0; i2 < k; ++i2) at += s_keep[i2] * omc(nlock + i2);
                }
              }
              double val = (at - alpha_j * om_cur[b] - beta_j * om_prev[b]) / std::max(nrm, 1e-300);
              val += std::copysign(noise, val);
              om_next[b] = val;
              maxom = std::max(maxom, std::fabs(val));
            }
            om_next[jt - 1 < 0 ? 0 : jt - 1] = noise;
            om_next[jt] = noise;
            // sqrt(eps) would keep the EIGENVALUES at rounding level (Simon); the eigenvectors are handed to the Green's-function
            // stage, so the basis is kept orthogonal to 1e-12 instead and the returned vectors are orthonormalised once more at the end
            if (trace) {
              double mx = 0.0;
              int wh = -1;
              (void)measure_only(jt, &mx, &wh);
              int we = -1;
              double me = 0.0;
              for (int b = 0; b < jt - 1; ++b)
                if (std::fabs(om_next[b]) > me) me = std::fabs(om_next[b]), we = b;
              fprintf(stderr, "[eigh]   j %d estimate max %.3e at %d | measured max %.3e at %d (relative to |w| %.3e) est there %.3e\n", j, me, we, mx / std::max(nrm, 1e-300), wh, nrm,
                      wh >= 0 ? std::fabs(om_next[wh]) : 0.0);
            }
            if (maxom > 1e-12 || forced) {  // about to be lost: remove everything now, and again at the next step
              double w3 = 0.0;
              rc = gs_pass(jt, nlock, true, &w3);
              ++n_full;
              if (rc) return rc;
              t_at(j, j) += c[2 * jt];
              nrm = std::sqrt(std::max(w3, 0.0));
              for (int b = 0; b <= jt; ++b) om_next[b] = noise;
              force_full = !was_forced;
              full = true;
              c2sum = 0.0;
              for (int t = 0; t < 2 * (jt + 1); ++t) c2sum += c[t] * c[t];
              w2 = w3;
            }
          }
          om_prev.swap(om_cur);
          om_cur.swap(om_next);
        }
        // One classical Gram-Schmidt pass leaves an orthogonality error ~ eps*|w_before|/|w_after|.  H v_j always carries
        // alpha_j v_j + beta_j v_{j-1}, so the textbook DGKS bound 1/sqrt(2) would refine nearly every step; Lanczos only
        // needs semi-orthogonality (sqrt(eps)), so refine when the norm dropped by more than 10x (error <= ~1e-14 otherwise).
        if (w2 < DGKS_ETA2 * (c2sum + w2)) {
          double w3 = 0.0;
          rc = gs_pass(jt, nlock, true, &w3);
          ++n_full;
          if (rc) return rc;
          t_at(j, j) += c[2 * jt];
          double nrm_b = std::sqrt(std::max(w3, 0.0));
          if (nrm_b < 0.5 * nrm) nrm_b = 0.0;  // w lies in span(V): invariant subspace
          nrm = nrm_b;
        }
        double tscale = 1.0;
        for (int a = 0; a <= j; ++a)
          for (int b = 0; b <= j; ++b) tscale = std::max(tscale, std::fabs(t_at(a, b)));
        if (nrm <= 1e-13 * tscale) {
          meff = j + 1;
          beta_last = 0.0;
          break;
        }
        if (trace) fprintf(stderr, "[eigh] it %d j %d jt %d %s%s alpha % .6e beta %.6e c2sum %.3e\n", it, j, jt, fused_local ? "fused " : "plain ", full ? "FULL" : "local", t_at(j, j), nrm, c2sum);
        if (j + 1 < ma) t_at(j + 1, j) = t_at(j, j + 1) = nrm;
        beta_last = nrm;
        nv[jt + 1] = nrm;  // left unnormalised: the next product (or whoever needs a unit vector) divides
      }
      A.assign((size_t)meff * meff, 0.0);
      for (int a = 0; a < meff; ++a)
        for (int b = 0; b < meff; ++b) A[a + (size_t)b * meff] = t_at(a, b);
      if (!jacobi_eigh(meff, A, theta, S)) return fail(HXV_ERR_STATE, "hxv_eigh_lowest: projected eigenproblem did not converge");
      ne = std::min(nwant, meff);
      nconv = 0;
      if (trace) fprintf(stderr, "[eigh] it %d restart: theta0 % .10e theta1 % .10e beta_last %.3e\n", it, theta[0], meff > 1 ? theta[1] : 0.0, beta_last);
      for (int i = 0; i < ne; ++i) {
        const double res = std::fabs(beta_last * S[(meff - 1) + (size_t)i * meff]);
        if (res <= tol * std::max(eps23, std::fabs(theta[i]))) ++nconv;
      }
      closed = meff < ma;
      // check rounds only ask "is there a state below stop_above?": Ritz values come down monotonically and the residual
      // bounds how far the lowest one can still move, so the answer "no" does not need a converged pair
      // (the residual only says that SOME eigenvalue lies within it of the Ritz value -- a copy with a tiny overlap with the start
      //  vector can still hide below -- so the answer is trusted only from the second restart cycle on and while the residual of
      //  the lowest pair keeps falling, or once that pair has converged; a heuristic, documented as such in hxv.h)
      const double res0 = ne >= 1 ? std::fabs(beta_last * S[(meff - 1)]) : 0.0;
      const bool settled = nconv >= 1 || closed || (it >= 1 && res0 <= res0_prev);
      res0_prev = res0;
      if (ne >= 1 && settled && theta[0] - res0 > stop_above) {
        above = true;
        break;
      }
      if (nconv == ne || closed || it >= maxrestart) break;
      k = keep_count(ma, nwant, nconv, h->eigh_keep_pct);
      // Ritz vectors = (stored vectors) * diag(1/nv) * S
      Ssc.assign(S.begin(), S.begin() + (size_t)ma * k);
      for (int i = 0; i < k; ++i)
        for (int l = 0; l < ma; ++l) Ssc[l + (size_t)i * ma] /= nv[nlock + l];
      HIPCHK(hipMemcpyAsync(d_S, Ssc.data(), (size_t)ma * k * sizeof(double), hipMemcpyHostToDevice, st));
      // (round 5) with nothing locked the rotation kernel also moves the residual vector to its new place and measures it against the Ritz
      // vectors it forms: rotation + copy + one multi-dot pass in one pass over the basis
      const bool fuse_rot = nlock == 0 && !h->eigh_measure_all && h->eigh_fuse_restart && launch_rotate_dots(g, st, n, Va, n, ma, k, d_S, d_part);
      if (fuse_rot) {
        hipLaunchKernelGGL(tr_colsum, dim3(1), dim3(256), 0, st, d_part, g, 2 * FUSE_NJ, 2 * k, d_coef, real ? 1 : 0);
      } else {
        launch_rotate(g, st, n, Va, n, ma, k, d_S);
        HIPCHK(hipMemcpyAsync(av(k), av(ma), (size_t)n * sizeof(double2), hipMemcpyDeviceToDevice, st));
      }
      HIPCHK(hipStreamSynchronize(st));  // Ssc (host) is reused at the next restart
      for (int i = 0; i < k; ++i) nv[nlock + i] = 1.0;
      nv[nlock + k] = nv[nlock + ma];
      if (!h->eigh_measure_all) {
        // The residual vector inherits whatever orthogonality the old basis had lost against the directions that are now
        // the kept Ritz vectors: clean it once per restart, so that the estimates of the new cycle start from rounding
        // level for every pair they track
        double r2 = 0.0;
        int rcr = gs_pass(nlock + k - 1, nlock, true, &r2, fuse_rot);
        if (rcr) return rcr;
        ++n_full;
        if (r2 > 0.0) nv[nlock + k] = std::sqrt(r2);  // (its length after the clean-up; it stays unnormalised)
      }
      std::fill(T.begin(), T.end(), 0.0);
      theta_keep.assign(k, 0.0);
      s_keep.assign(k, 0.0);
      for (int i = 0; i < k; ++i) {
        t_at(i, i) = theta[i];
        t_at(k, i) = t_at(i, k) = beta_last * S[(ma - 1) + (size_t)i * ma];
        theta_keep[i] = theta[i];
        s_keep[i] = t_at(k, i);
      }
      std::fill(om_prev.begin(), om_prev.end(), eps);
      std::fill(om_cur.begin(), om_cur.end(), eps);
      force_full = false;
    }
    if (above) return HXV_OK;
    // Ritz vectors of the wanted pairs to the front of the active basis
    Ssc.assign(S.begin(), S.begin() + (size_t)meff * ne);
    for (int i = 0; i < ne; ++i)
      for (int l = 0; l < meff; ++l) Ssc[l + (size_t)i * meff] /= nv[nlock + l];
    HIPCHK(hipMemcpyAsync(d_S, Ssc.data(), (size_t)meff * ne * sizeof(double), hipMemcpyHostToDevice, st));
    launch_rotate(g, st, n, Va, n, meff, ne, d_S);
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < ne; ++i) nv[nlock + i] = 1.0;
    return HXV_OK;
  };

  int rc0 = trl(0, neigen, (uint64_t)0x5EED5EEDull, 1e300);
  if (rc0) return rc0;
  const int nmv_search = nmv;  // products of the search itself; what follows are the optional check rounds for hidden copies
  // values and (absolute) basis slots of the pairs found so far
  std::vector<double> fval(theta.begin(), theta.begin() + ne);
  lockval = fval;
  int nfound = ne;
  const int nconv0 = nconv, ne0 = ne;
  const bool closed0 = closed;
  // A single-vector Krylov method sees ONE vector of an exactly degenerate level (ARPACK included; the reference keeps
  // every state within gs_threshold of the minimum, ED_DIAG.f90:234-244, and relies on the copies showing up).  So:
  // lock what was found and look, in its orthogonal complement, for a state BELOW the current neigen-th lowest value --
  // a second copy of a degenerate level that displaces a higher one; repeat until there is none.
  if (h->eigh_degenerate && (nconv0 == ne0 || closed0)) {
    for (int round = 1; round <= 2 * neigen && nfound + std::max(neigen, 1) + 2 <= m && nfound < dim; ++round) {
      double kth = 1e300;  // fewer pairs than wanted so far (the first Krylov space closed early): take whatever comes
      if (nfound >= neigen) {
        std::vector<double> sorted(fval);
        std::sort(sorted.begin(), sorted.end());
        kth = sorted[neigen - 1] - 1e-9 * std::max(1.0, std::fabs(sorted[neigen - 1]));
      }
      int rc = trl(nfound, 1, (uint64_t)0x5EED5EEDull + 0x9E3779B97F4A7C15ull * (uint64_t)round, kth);
      if (rc) return rc;
      if (above) break;                  // nothing below the wanted set
      if (nconv < 1 && !closed) break;   // could not settle the question within maxrestart: keep what is certain
      if (!(theta[0] < kth)) break;
      fval.push_back(theta[0]);  // its vector sits at V[nfound]: locked from now on
      lockval = fval;
      ++nfound;
    }
  }
  // the found vectors (slots 0..nfound-1) come from bases that were orthogonal to ~1e-10: one Gram-Schmidt sweep among them
  if (!h->eigh_measure_all && d_evecs)
    for (int b = 1; b < nfound; ++b) {
      double r2 = 0.0;
      int rc = gs_pass(b - 1, 0, true, &r2);
      if (rc) return rc;
      if (r2 > 0.0) hipLaunchKernelGGL(tr_scale_nrm, dim3(g), dim3(256), 0, st, n, vec(b), 1.0 / std::sqrt(r2), (double*)nullptr);
    }
  // the neigen lowest of everything found, ascending, vectors gathered to the end of the basis and copied out
  std::vector<int> order(nfound);
  for (int i = 0; i < nfound; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return fval[a] < fval[b]; });
  ne = std::min(neigen, nfound);
  theta.assign(ne, 0.0);
  for (int i = 0; i < ne; ++i) theta[i] = fval[order[i]];
  nconv = (nconv0 == ne0 || closed0) ? ne : std::min(nconv0, ne);  // (pairs added by the locking rounds had converged)
  for (int i = 0; i < neigen; ++i) evals[i] = i < ne ? theta[i] : 0.0;
  if (nconv_out) *nconv_out = nconv;
  if (nmatvec_out) *nmatvec_out = nmv;
  h->eigh_last_full = n_full;
  h->eigh_last_local = n_local;
  h->eigh_last_search = nmv_search;
  h->eigh_last_check = nmv - nmv_search;
  if (d_evecs) {
    for (int i = 0; i < ne; ++i) {
      if (real)
        launch_to_complex(h, (const double*)vec(order[i]), (double2*)d_evecs + (int64_t)i * nc, st);
      else
        HIPCHK(hipMemcpyAsync((double2*)d_evecs + (int64_t)i * nc, vec(order[i]), (size_t)nc * sizeof(double2), hipMemcpyDeviceToDevice, st));
    }
    if (ne < neigen) HIPCHK(hipMemsetAsync((double2*)d_evecs + (int64_t)ne * nc, 0, (size_t)(neigen - ne) * nc * sizeof(double2), st));
  }
  HIPCHK(hipStreamSynchronize(st));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(HXV_ERR_HIP, std::string("hxv_eigh_lowest kernels: ") + hipGetErrorString(e));
  return HXV_OK;
}

int hxv_eigh_lowest_host(hxv_handle* h, int32_t neigen, int32_t ncv, int32_t maxrestart, double tol, double* evals, void* evecs_host,
                         int32_t* nconv_out, int32_t* nmatvec_out) {
  if (!h || neigen < 1) return fail(HXV_ERR_ARG, "hxv_eigh_lowest_host: bad argument");
  if (!evecs_host) return hxv_eigh_lowest(h, neigen, ncv, maxrestart, tol, evals, nullptr, nconv_out, nmatvec_out);
  HIPCHK(hipSetDevice(h->device));
  const int64_t n = (int64_t)h->host.pitch * h->host.qdw;
  const int64_t vecdim = (int64_t)h->host.dimup * h->host.qdw;  // eig_basis(vecDim, Neigen): this rank's slab of every vector
  DevFree mem;
  double2* d = nullptr;
  mem.device = h->device;
  HIPCHK(pool_alloc(h->device, (size_t)neigen * n * sizeof(double2), (void**)&d));
  mem.pooled.push_back(d);
  int rc = hxv_eigh_lowest(h, neigen, ncv, maxrestart, tol, evals, d, nconv_out, nmatvec_out);
  if (rc) return rc;
  // eig_basis(vecDim, Neigen) in the reference's layout: columns unpadded, eigenvectors consecutive (ED_DIAG.f90:145)
  for (int i = 0; i < neigen; ++i) {
    rc = slab_to_host(h, d + (int64_t)i * n, (char*)evecs_host + (size_t)i * vecdim * sizeof(double2));
    if (rc) return rc;
  }
  return HXV_OK;
}

}  // extern "C"
