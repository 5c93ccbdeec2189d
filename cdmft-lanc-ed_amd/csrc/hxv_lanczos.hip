// Device Lanczos behind the C-ABI (include/hxv.h): the single-vector recurrences with the SciFortran call shapes
// (sp_lanc_tridiag, sp_lanc_eigh; ED_GF_NORMAL.f90:215-220, ED_DIAG.f90:176-184), the Green's-function start vectors
// (c / c^dagger on the ground state, ED_GF_NORMAL.f90:180-199) and the iteration timer of bench.py.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "hxv_handle.hpp"

using namespace hxv;

// ===========================================================================================
// Device Lanczos
// ===========================================================================================
namespace {

// w -= b*qm ; partial sums of Re<q,w>
__global__ void __launch_bounds__(256) lz_sub_dot(int64_t n, double2* __restrict__ w, const double2* __restrict__ qm,
                                                  const double2* __restrict__ q, const double* __restrict__ scal, int ib,
                                                  double* __restrict__ partial) {
  const double b = ib >= 0 ? scal[ib] : 0.0;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double2 x = w[i];
    if (ib >= 0) {
      double2 p = qm[i];
      x.x -= b * p.x;
      x.y -= b * p.y;
      w[i] = x;
    }
    double2 y = q[i];
    acc += y.x * x.x + y.y * x.y;
  }
  __shared__ double red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// w -= a*q ; partial sums of |w|^2
__global__ void __launch_bounds__(256) lz_sub_nrm(int64_t n, double2* __restrict__ w, const double2* __restrict__ q,
                                                  const double* __restrict__ scal, int ia, double* __restrict__ partial) {
  const double a = scal[ia];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double2 x = w[i], y = q[i];
    // (written out -- one fused multiply-add per component, the squares rounded before they are added -- so that the paired
    //  kernel below reproduces two runs of this one bit for bit)
    x.x = fma(-a, y.x, x.x);
    x.y = fma(-a, y.y, x.y);
    w[i] = x;
    {
#pragma clang fp contract(off)  // (the _rn intrinsics of HIP are plain operators, open to contraction)
      const double t = x.x * x.x;
      acc = acc + fma(x.y, x.y, t);
    }
  }
  __shared__ double red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// PAIRED vectors (real part = vector a, imaginary part = vector b): w.re -= a_a*q.re, w.im -= a_b*q.im; partial sums of
// |w.re|^2 and |w.im|^2 separately
__global__ void __launch_bounds__(256) lz_sub_nrm_pair(int64_t n, double2* __restrict__ w, const double2* __restrict__ q,
                                                       const double* __restrict__ scal, int ia, int ib, double* __restrict__ partial_a,
                                                       double* __restrict__ partial_b) {
  const double a = scal[ia], b = scal[ib];
  double acc = 0.0, acc2 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double2 x = w[i], y = q[i];
    x.x = fma(-a, y.x, x.x);
    x.y = fma(-b, y.y, x.y);
    w[i] = x;
    {
#pragma clang fp contract(off)
      const double t = x.x * x.x, t2 = x.y * x.y;
      acc = acc + t;
      acc2 = acc2 + t2;
    }
  }
  __shared__ double red[256], red2[256];
  red[threadIdx.x] = acc;
  red2[threadIdx.x] = acc2;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[threadIdx.x] += red[threadIdx.x + s];
      red2[threadIdx.x] += red2[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial_a[blockIdx.x] = red[0];
    partial_b[blockIdx.x] = red2[0];
  }
}

// q = (Re a, Re b) from two complex vectors; partial sums of Im(a)^2 + Im(b)^2 (must be zero), |Re a|^2, |Re b|^2
__global__ void __launch_bounds__(256) lz_pack_pair(int64_t n, const double2* __restrict__ a, const double2* __restrict__ b,
                                                    double2* __restrict__ q, double* __restrict__ p_im, double* __restrict__ p_a,
                                                    double* __restrict__ p_b) {
  double im = 0.0, na = 0.0, nb = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double2 x = a[i], y = b[i];
    q[i] = make_double2(x.x, y.x);
    im += x.y * x.y + y.y * y.y;
    {
#pragma clang fp contract(off)
      const double t = x.x * x.x, t2 = y.x * y.x;
      na = na + t;
      nb = nb + t2;
    }
  }
  __shared__ double red[3][256];
  red[0][threadIdx.x] = im;
  red[1][threadIdx.x] = na;
  red[2][threadIdx.x] = nb;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
      for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    p_im[blockIdx.x] = red[0][0];
    p_a[blockIdx.x] = red[1][0];
    p_b[blockIdx.x] = red[2][0];
  }
}

// partial sums of |x|^2
__global__ void __launch_bounds__(256) lz_nrm(int64_t n, const double2* __restrict__ x, double* __restrict__ partial) {
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double2 a = x[i];
    {
#pragma clang fp contract(off)  // (written out: see lz_sub_nrm)
      const double t = a.x * a.x;
      acc = acc + fma(a.y, a.y, t);
    }
  }
  __shared__ double red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// scal[io] = op(sum partial[0..np))   op: 0 identity, 1 sqrt
__global__ void __launch_bounds__(256) lz_final(const double* __restrict__ partial, int np, double* __restrict__ scal, int io, int op) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < np; i += 256) acc += partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) scal[io] = op ? sqrt(red[0]) : red[0];
}

// q = w / scal[ib]
__global__ void __launch_bounds__(256) lz_scale(int64_t n, double2* q, const double2* w,
                                                const double* __restrict__ scal, int ib) {
  const double r = 1.0 / scal[ib];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double2 x = w[i];
    q[i] = make_double2(x.x * r, x.y * r);
  }
}

// y += c * q   (c real, host scalar)
__global__ void __launch_bounds__(256) lz_axpy(int64_t n, double2* __restrict__ y, const double2* __restrict__ q, double c) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double2 x = q[i], z = y[i];
    y[i] = make_double2(z.x + c * x.x, z.y + c * x.y);
  }
}

// deterministic start vector: splitmix64 hash of the global index -> uniform(-0.5,0.5) re and im
// (col0 = first global column of this rank's slab: a split sector starts from the same global vector as the unsplit one)
// (device row order, SectorHost::up_perm: the vector is defined on the REFERENCE index -- device row `row` holds reference row iperm[row],
//  times its basis sign -- so a sector starts from the same vector whatever order its rows are stored in)
__global__ void __launch_bounds__(256) lz_init(int64_t n, double2* __restrict__ q, uint64_t seed, int dimup, int pitch, int col0,
                                               const int32_t* __restrict__ iperm, const uint8_t* __restrict__ sign) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t lcol = i / pitch;
    const int row = (int)(i - lcol * pitch);
    const int64_t col = lcol + col0;
    if (row >= dimup) {  // pad rows stay zero: they must not enter the dot products
      q[i] = make_double2(0.0, 0.0);
      continue;
    }
    const int rrow = iperm ? iperm[row] : row;
    const double sgn = (sign && sign[row]) ? -1.0 : 1.0;
    uint64_t z = (uint64_t)(col * dimup + rrow) * 2 + seed;
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

int grid_for(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, RED_BLOCKS)); }  // (never an empty grid: a rank may own no column)

// ---- REAL-vector mode: layout conversions and the real start vector ----------------------------------------------
// real [DimDw][pr] <- Re(complex [DimDw][pc]); pads zero.  partial = per-block sum of Im^2 (may be null)
__global__ void __launch_bounds__(256) lz_to_real(int dimup, int dimdw, int pc, int pr, const double2* __restrict__ src,
                                                  double* __restrict__ dst, double* __restrict__ partial) {
  const int64_t n = (int64_t)dimdw * pr;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t col = i / pr;
    const int row = (int)(i - col * pr);
    double x = 0.0;
    if (row < dimup) {
      const double2 z = src[col * pc + row];
      x = z.x;
      acc += z.y * z.y;
    }
    if (dst) dst[i] = x;
  }
  __shared__ double red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0 && partial) partial[blockIdx.x] = red[0];
}

// complex [DimDw][pc] <- real [DimDw][pr]; pads zero
__global__ void __launch_bounds__(256) lz_to_complex(int dimup, int dimdw, int pc, int pr, const double* __restrict__ src,
                                                     double2* __restrict__ dst) {
  const int64_t n = (int64_t)dimdw * pc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t col = i / pc;
    const int row = (int)(i - col * pc);
    dst[i] = make_double2(row < dimup ? src[col * pr + row] : 0.0, 0.0);
  }
}

// real start vector: the real part of lz_init's vector
__global__ void __launch_bounds__(256) lz_init_real(int64_t n, double* __restrict__ q, uint64_t seed, int dimup, int pitch, int col0,
                                                    const int32_t* __restrict__ iperm, const uint8_t* __restrict__ sign) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t lcol = i / pitch;
    const int row = (int)(i - lcol * pitch);
    const int64_t col = lcol + col0;  // global column: a split sector starts from the same global vector as the unsplit one
    if (row >= dimup) {
      q[i] = 0.0;
      continue;
    }
    const int rrow = iperm ? iperm[row] : row;
    uint64_t x = (uint64_t)(col * dimup + rrow) * 2 + seed + 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x = x ^ (x >> 31);
    const double r = (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    q[i] = (sign && sign[row]) ? -r : r;
  }
}

// Symmetric tridiagonal eigen-solver (implicit QL with Wilkinson shifts): d[n] diagonal,
// e[n] sub-diagonal in e[1..n-1] (e[0] unused).  On exit d = eigenvalues (unsorted) and, if z,
// z (n x n, column-major, initialised to identity by the caller) = eigenvectors.
bool tridiag_ql(int n, std::vector<double>& d, std::vector<double>& e, std::vector<double>* z) {
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  for (int l = 0; l < n; ++l) {
    int iter = 0, m;
    do {
      for (m = l; m < n - 1; ++m) {
        double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
        if (std::fabs(e[m]) <= 2.3e-16 * dd) break;
      }
      if (m != l) {
        if (iter++ == 200) return false;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = std::hypot(g, 1.0);
        g = d[m] - d[l] + e[l] / (g + (g >= 0 ? std::fabs(r) : -std::fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = m - 1; i >= l; --i) {
          double f = s * e[i], b = c * e[i];
          r = std::hypot(f, g);
          e[i + 1] = r;
          if (r == 0.0) {
            d[i + 1] -= p;
            e[m] = 0.0;
            break;
          }
          s = f / r;
          c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          p = s * r;
          d[i + 1] = g + p;
          g = c * r - b;
          if (z)
            for (int k = 0; k < n; ++k) {
              double* zz = z->data();
              f = zz[k + (size_t)(i + 1) * n];
              zz[k + (size_t)(i + 1) * n] = s * zz[k + (size_t)i * n] + c * f;
              zz[k + (size_t)i * n] = c * zz[k + (size_t)i * n] - s * f;
            }
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p;
        e[l] = g;
        e[m] = 0.0;
      }
    } while (m != l);
  }
  return true;
}

struct LzBuf {
  double2 *q, *qm, *w;
};

// scal[i] = scal[a] * scal[b]
__global__ void lz_mul(double* scal, int i, int a, int b) { scal[i] = scal[a] * scal[b]; }
// scal[i] = sqrt(scal[i])   (split sector: the square root comes after the all-reduce of the partial sums of squares)
__global__ void lz_sqrt(double* scal, int i) { scal[i] = sqrt(scal[i]); }

// sum (op 0) or 2-norm (op 1) of the per-workgroup partials into scal[io]; on a split sector the sums of all ranks are added
int reduce_scalar(hxv_handle* h, const double* partial, int np, int io, int op) {
  const bool dist = comm_ready(h);
  hipLaunchKernelGGL(lz_final, dim3(1), dim3(256), 0, h->stream, partial, np, h->d_scalars, io, dist ? 0 : op);
  if (dist) {
    int rc = comm_allreduce_sum(h, h->d_scalars + io, 1, h->stream);
    if (rc) return rc;
    if (op) hipLaunchKernelGGL(lz_sqrt, dim3(1), dim3(1), 0, h->stream, h->d_scalars, io);
  }
  return HXV_OK;
}

// End of one fused iteration without the host: record alpha_k = scal[0], beta_{k+1} = scal[1] at the device-side step
// counter scal[6], and prepare the next iteration's scalars  s = 1/beta_{k+1} (scal[2]),  c = s_old/s = beta_{k+1}/beta_k (scal[3]).
__global__ void lz_next(double* scal, double* alpha_out, double* beta_out, int nmax) {
  const int k = (int)scal[6];
  if (k < nmax) {
    alpha_out[k] = scal[0];
    if (k + 1 < nmax) beta_out[k + 1] = scal[1];
  }
  // (the same operations, in the same order, as LzRunner::advance() + step() on the host: bit-identical recurrences)
  const double beta_prev = 1.0 / scal[2], s_new = 1.0 / scal[1];
  scal[2] = s_new;
  scal[3] = 1.0 / (s_new * beta_prev);
  scal[6] = (double)(k + 1);
}

// Lanczos recurrence on device.  Two implementations of one step:
//  * plain : w = H q (any kernel), then lz_sub_dot / lz_sub_nrm / lz_scale on normalised vectors;
//  * fused : vectors are kept UNNORMALISED (q_k = s*X, s = 1/beta_k); pass A's epilogue produces
//            w = s*H X - c*Xm and the partial sums of alpha, one more pass subtracts alpha*q and reduces beta.
//            144 B/state per iteration instead of 224.
struct LzRunner {
  hxv_handle* h;
  LzBuf b;
  bool fused;
  bool real;            // REAL-vector mode: the buffers hold double[DimDw][pitch_real]; every streaming kernel below is
                        // elementwise with real scalars, so it runs unchanged on the buffer viewed as n2 double2 elements
  int64_t n2;           // double2 elements of one vector
  bool first = true;
  double s_cur = 1.0;   // q = s_cur * b.q (fused) ; 1 (plain)
  double beta_prev = 1.0;

  LzRunner(hxv_handle* hh, double2* x, double2* xm, double2* w, bool real_vec = false) : h(hh), b{x, xm, w}, real(real_vec) {
    // (split sectors included: the epilogue's partial sums are this rank's share, alpha and beta are all-reduced)
    fused = hh->kernel == 1 && hh->plan.usable && (!hh->dev.nd.active || nd_folds(hh->dev)) && hh->plan.opt.passes == 3 && hh->plan.opt.debug == 0 && hh->lz_fused;
    // (local slab: qdw == DimDw on an unsplit sector)
    n2 = real ? (int64_t)pitch_real_of(hh) * hh->host.qdw / 2 : (int64_t)hh->host.pitch * hh->host.qdw;
    hh->last_real = real ? 1 : 0;
  }

  // b.q holds a vector of norm `nrm` (pass 1.0 if already normalised)
  int begin(double nrm) {
    first = true;
    s_cur = 1.0 / nrm;
    beta_prev = nrm;
    if (!fused && nrm != 1.0) return fail(HXV_ERR_STATE, "plain Lanczos expects a normalised start vector");
    return HXV_OK;
  }

  int step(double* alpha, double* beta) {
    const int64_t n = n2;
    const int g = grid_for(n);
    if (!fused) {
      // (apply_slab = exchange + product on a split sector, the plain product otherwise)
      int rc = real ? apply_slab_real(h, (const double*)b.q, (double*)b.w, h->stream) : apply_slab(h, b.q, b.w, h->stream);
      if (rc) return rc;
      hipLaunchKernelGGL(lz_sub_dot, dim3(g), dim3(256), 0, h->stream, n, b.w, b.qm, b.q, h->d_scalars, first ? -1 : 2, h->d_partials);
      rc = reduce_scalar(h, h->d_partials, g, 0, 0);
      if (rc) return rc;
      hipLaunchKernelGGL(lz_sub_nrm, dim3(g), dim3(256), 0, h->stream, n, b.w, b.q, h->d_scalars, 0, h->d_partials + RED_BLOCKS);
      rc = reduce_scalar(h, h->d_partials + RED_BLOCKS, g, 1, 1);
      if (rc) return rc;
    } else {
      const int64_t nwg = std::max<int64_t>(1, tiled_pass_up_workgroups(h->dev, h->plan, real, dw_part_in_pieces(h)));
      if (nwg > h->lz_partial_n) {
        if (h->d_lz_partial) (void)hipFree(h->d_lz_partial);
        h->d_lz_partial = nullptr;
        h->lz_partial_n = 0;
        HIPCHK(hipMalloc((void**)&h->d_lz_partial, (size_t)nwg * sizeof(double)));
        HIPCHK(hipMemsetAsync(h->d_lz_partial, 0, (size_t)nwg * sizeof(double), h->stream));  // (a rank without columns launches nothing)
        h->lz_partial_n = nwg;
      }
      // scal[2] = s, scal[3] = c = beta_k / beta_{k-1}
      const double sc[2] = {s_cur, first ? 0.0 : 1.0 / (s_cur * beta_prev)};
      HIPCHK(hipMemcpyAsync(h->d_scalars + 2, sc, 2 * sizeof(double), hipMemcpyHostToDevice, h->stream));
      LzEpilogue ep;
      ep.xm = first ? nullptr : b.qm;
      ep.scal = h->d_scalars;
      ep.i_s = 2;
      ep.i_c = 3;
      ep.partial = h->d_lz_partial;
      int rc = real ? apply_slab_real(h, (const double*)b.q, (double*)b.w, h->stream, &ep) : apply_slab(h, b.q, b.w, h->stream, &ep);
      if (rc) return rc;
      rc = reduce_scalar(h, h->d_lz_partial, (int)nwg, 0, 0);  // alpha: summed over the ranks of a split sector
      if (rc) return rc;
      hipLaunchKernelGGL(lz_mul, dim3(1), dim3(1), 0, h->stream, h->d_scalars, 4, 0, 2);
      hipLaunchKernelGGL(lz_sub_nrm, dim3(g), dim3(256), 0, h->stream, n, b.w, b.q, h->d_scalars, 4, h->d_partials + RED_BLOCKS);
      rc = reduce_scalar(h, h->d_partials + RED_BLOCKS, g, 1, 1);
      if (rc) return rc;
    }
    double host[2];
    HIPCHK(hipMemcpyAsync(host, h->d_scalars, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *alpha = host[0];
    *beta = host[1];
    last_beta = host[1];
    return HXV_OK;
  }

  // One fused iteration enqueued with NO host involvement: s and c are read from scal[2], scal[3] (left there by the
  // previous lz_next), alpha/beta go to device arrays.  q, qm, w are explicit so that a rotation cycle can be captured.
  int enqueue_device_iteration(double2* q, double2* qm, double2* w, double* d_alpha, double* d_beta, int nmax) {
    const int g = grid_for(n2);
    const int64_t nwg = tiled_pass_up_workgroups(h->dev, h->plan, real, dw_part_in_pieces(h));
    LzEpilogue ep;
    ep.xm = qm;
    ep.scal = h->d_scalars;
    ep.i_s = 2;
    ep.i_c = 3;
    ep.partial = h->d_lz_partial;
    hipError_t e;
    if (real) {
      DevSector d = h->dev;
      d.pitch = pitch_real_of(h);
      e = launch_hxv_tiled_real(d, h->plan, (const double*)q, (double*)h->d_wt, (double*)w, h->stream, &ep);
    } else {
      e = launch_hxv_tiled(h->dev, h->plan, q, h->d_wt, w, h->stream, &ep);
    }
    if (e != hipSuccess) return fail(HXV_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    h->n_apply++;
    hipLaunchKernelGGL(lz_final, dim3(1), dim3(256), 0, h->stream, h->d_lz_partial, (int)nwg, h->d_scalars, 0, 0);
    hipLaunchKernelGGL(lz_mul, dim3(1), dim3(1), 0, h->stream, h->d_scalars, 4, 0, 2);
    hipLaunchKernelGGL(lz_sub_nrm, dim3(g), dim3(256), 0, h->stream, n2, w, q, h->d_scalars, 4, h->d_partials + RED_BLOCKS);
    hipLaunchKernelGGL(lz_final, dim3(1), dim3(256), 0, h->stream, h->d_partials + RED_BLOCKS, g, h->d_scalars, 1, 1);
    hipLaunchKernelGGL(lz_next, dim3(1), dim3(1), 0, h->stream, h->d_scalars, d_alpha, d_beta, nmax);
    return HXV_OK;
  }

  // Iterations 1..nmax-1 of a fixed-length run on the device alone (iteration 0 was done by step(); scal[2], scal[3],
  // scal[6] are set): the three-iteration pointer-rotation cycle is captured ONCE into a hipGraph and replayed, so a
  // small sector pays one graph launch per three iterations instead of ~9 kernel launches, two copies and a
  // synchronisation per iteration.  Leaves b.q/b.qm/b.w as after the last iteration.
  int run_device_iterations(int nmax, double* d_alpha, double* d_beta, bool use_graph) {
    double2* buf[3] = {b.q, b.w, b.qm};  // iteration k uses q = buf[k%3] (k counted from 1: q = former w), w = buf[(k+1)%3], qm = buf[(k+2)%3]
    // after step() of iteration 0 and before any rotation: X_1 = b.w, X_0 = b.q, free = b.qm
    int k = 1;
    auto one = [&](int kk) -> int {
      double2* q = buf[kk % 3];
      double2* w = buf[(kk + 1) % 3];
      double2* qm = buf[(kk + 2) % 3];
      return enqueue_device_iteration(q, qm, w, d_alpha, d_beta, nmax);
    };
    const int cycles = use_graph ? (nmax - 1) / 3 : 0;
    if (cycles >= 2) {
      hipGraph_t graph = nullptr;
      hipGraphExec_t exec = nullptr;
      HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
      int rc = HXV_OK;
      for (int u = 0; u < 3 && rc == HXV_OK; ++u) rc = one(k + u);
      hipError_t ec = hipStreamEndCapture(h->stream, &graph);
      if (rc) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
      }
      if (ec != hipSuccess) return fail(HXV_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ec));
      ec = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      if (ec != hipSuccess) {
        (void)hipGraphDestroy(graph);
        return fail(HXV_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ec));
      }
      for (int cy = 0; cy < cycles; ++cy) {
        ec = hipGraphLaunch(exec, h->stream);
        if (ec != hipSuccess) break;
      }
      k += 3 * cycles;  // (k-1) % 3 is unchanged: the remainder below continues the same rotation
      (void)hipGraphExecDestroy(exec);
      (void)hipGraphDestroy(graph);
      if (ec != hipSuccess) return fail(HXV_ERR_HIP, std::string("hipGraphLaunch: ") + hipGetErrorString(ec));
    }
    for (; k < nmax; ++k) {
      int rc = one(k);
      if (rc) return rc;
    }
    b.q = buf[(nmax - 1) % 3];
    b.w = buf[nmax % 3];
    b.qm = buf[(nmax + 1) % 3];
    first = false;
    return HXV_OK;
  }

  // rotate to the next Lanczos vector (needs the beta returned by step())
  int advance() {
    const int64_t n = n2;
    if (!fused) {
      HIPCHK(hipMemcpyAsync(h->d_scalars + 2, h->d_scalars + 1, sizeof(double), hipMemcpyDeviceToDevice, h->stream));
      std::swap(b.q, b.qm);
      hipLaunchKernelGGL(lz_scale, dim3(grid_for(n)), dim3(256), 0, h->stream, n, b.q, b.w, h->d_scalars, 2);
    } else {
      double2* old_m = b.qm;
      b.qm = b.q;      // X_{k-1}
      b.q = b.w;       // X_k = unnormalised residual
      b.w = old_m;
      beta_prev = 1.0 / s_cur;   // beta_k
      s_cur = 1.0 / last_beta;   // 1/beta_{k+1}
    }
    first = false;
    return HXV_OK;
  }

  double last_beta = 0.0;
  // current normalised Lanczos vector = scale() * vec()
  const double2* vec() const { return b.q; }
  double scale() const { return fused ? s_cur : 1.0; }
};

// may this Lanczos run use real vectors?  (d_vin: optional complex start vector that must then be purely real)
bool want_real(hxv_handle* h) { return h->real_vectors && !real_mode_blocker(h); }

// the three work vectors of the single-vector Lanczos; `real` = layout of the coming run.  The pad rows of the two
// layouts sit at different places and must be zero (the reductions run over the padded arrays, the products never
// write pads), so the buffers are cleared whenever the layout changes.
int ensure_lz(hxv_handle* h, bool real) {
  if (h->host.nranks != 1 && !comm_ready(h))
    return fail(HXV_ERR_STATE, "device Lanczos on a split sector needs the communicator: call hxv_comm_init after opening the sector");
  HIPCHK(hipSetDevice(h->device));
  const size_t bytes = (size_t)h->host.pitch * std::max(h->host.qdw, 1) * sizeof(double2);
  if (comm_ready(h) && h->host.nranks > 1 && h->lz_inplace && h->host.exchange != 2) {
    // split sector: the three vectors live where the exchange wants the slab, in three gather buffers (hxv_comm.cpp); when the memory for
    // them is not there the slab buffers below serve, with one slab copy per product
    int rc = comm_lz_homes(h, real, h->lz_vec);
    if (rc == HXV_OK) {
      h->lz_buf_mode = real ? 1 : 0;
      return HXV_OK;
    }
    if (rc != HXV_ERR_HIP) return rc;
  }
  for (auto& p : h->d_lz)
    if (!p) {
      HIPCHK(pool_alloc(h->device, bytes, (void**)&p));
      // on the handle's (non-blocking) stream: a null-stream memset is not ordered with the kernels that follow and could
      // land on a recycled pool block after the start vector had been written into it
      HIPCHK(hipMemsetAsync(p, 0, bytes, h->stream));
      h->device_bytes += (int64_t)bytes;
    }
  if (h->lz_buf_mode != (real ? 1 : 0)) {
    for (auto& p : h->d_lz) HIPCHK(hipMemsetAsync(p, 0, bytes, h->stream));
    h->lz_buf_mode = real ? 1 : 0;
  }
  for (int i = 0; i < 3; ++i) h->lz_vec[i] = h->d_lz[i];
  return HXV_OK;
}

// A start vector that lives in one of the handle's gather buffers (built at hxv_slab_home) would be zeroed by ensure_lz, which clears the
// slab's place in all three before the driver reads its input: such a vector is copied to a staging slab first (one slab copy per RUN).
// *tmp is the staging buffer to pool_free after the run, or null.
int stage_start_vector(hxv_handle* h, const void*& d_vin, double2** tmp) {
  *tmp = nullptr;
  if (!comm_ready(h) || !comm_in_gather(h, d_vin)) return HXV_OK;
  const size_t bytes = (size_t)h->host.pitch * std::max(h->host.qdw, 1) * sizeof(double2);
  HIPCHK(pool_alloc(h->device, bytes, (void**)tmp));
  HIPCHK(hipMemcpyAsync(*tmp, d_vin, bytes, hipMemcpyDeviceToDevice, h->stream));
  d_vin = *tmp;
  return HXV_OK;
}
struct StageFree {
  hxv_handle* h;
  double2* p[2];
  ~StageFree() {
    for (double2* q : p)
      if (q) {
        (void)hipStreamSynchronize(h->stream);
        pool_free(h->device, q);
      }
  }
};

}  // namespace

namespace hxv {
void launch_to_real(const hxv_handle* h, const double2* src, double* dst, hipStream_t st) {
  const int pr = pitch_real_of(h);
  hipLaunchKernelGGL(lz_to_real, dim3(grid_for((int64_t)pr * h->host.qdw)), dim3(256), 0, st, h->host.dimup, h->host.qdw, h->host.pitch, pr,
                     src, dst, (double*)nullptr);
}
void launch_to_complex(const hxv_handle* h, const double* src, double2* dst, hipStream_t st) {
  hipLaunchKernelGGL(lz_to_complex, dim3(grid_for((int64_t)h->host.pitch * h->host.qdw)), dim3(256), 0, st, h->host.dimup, h->host.qdw,
                     h->host.pitch, pitch_real_of(h), src, dst);
}
// One Lanczos step on NORMALISED vectors for callers that keep their own basis (hxv_eigh_lowest): w = H q - beta*qm through
// pass A's epilogue (qm may be null), alpha = <q,w> from its partial sums, then w -= alpha*q and |w|.  False if the fused
// product does not apply to this handle (the caller then measures the two projections itself).
bool lanczos_local_step_available(const hxv_handle* h) {
  return h->kernel == 1 && h->plan.usable && (!h->dev.nd.active || nd_folds(h->dev)) && h->plan.opt.passes == 3 && h->plan.opt.debug == 0 && h->lz_fused;
}

// One Lanczos step on an UNNORMALISED pair: q = sq * (unit q_j), qm = sqm * (unit q_{j-1}).  The product's epilogue stores
//   w = H q / sq - (beta / sqm) qm   and reduces   alpha = <q / sq, w>;
// with sub_alpha one more pass subtracts alpha (q / sq) and measures |w|, otherwise w is left as it is and *nrm_w is not set (the caller
// subtracts alpha together with other projections).  Scale factors 1.0 reproduce the normalised step bit for bit.
int lanczos_local_step(hxv_handle* h, bool real, const double2* q, double sq, const double2* qm, double sqm, double beta, double2* w,
                       bool sub_alpha, double* alpha, double* nrm_w) {
  const int64_t n2 = real ? (int64_t)pitch_real_of(h) * h->host.qdw / 2 : (int64_t)h->host.pitch * h->host.qdw;
  const int g = grid_for(n2);
  const int64_t nwg = std::max<int64_t>(1, tiled_pass_up_workgroups(h->dev, h->plan, real, dw_part_in_pieces(h)));
  if (nwg > h->lz_partial_n) {
    if (h->d_lz_partial) (void)hipFree(h->d_lz_partial);
    h->d_lz_partial = nullptr;
    h->lz_partial_n = 0;
    HIPCHK(hipMalloc((void**)&h->d_lz_partial, (size_t)nwg * sizeof(double)));
    HIPCHK(hipMemsetAsync(h->d_lz_partial, 0, (size_t)nwg * sizeof(double), h->stream));
    h->lz_partial_n = nwg;
  }
  const double sc[2] = {1.0 / sq, qm ? beta / sqm : 0.0};
  HIPCHK(hipMemcpyAsync(h->d_scalars + 2, sc, 2 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  LzEpilogue ep;
  ep.xm = qm;
  ep.scal = h->d_scalars;
  ep.i_s = 2;
  ep.i_c = 3;
  ep.partial = h->d_lz_partial;
  int rc = real ? apply_slab_real(h, (const double*)q, (double*)w, h->stream, &ep) : apply_slab(h, q, w, h->stream, &ep);
  if (rc) return rc;
  rc = reduce_scalar(h, h->d_lz_partial, (int)nwg, 0, 0);
  if (rc) return rc;
  if (sub_alpha) {
    hipLaunchKernelGGL(lz_mul, dim3(1), dim3(1), 0, h->stream, h->d_scalars, 4, 0, 2);  // alpha / sq: the coefficient of the stored q
    hipLaunchKernelGGL(lz_sub_nrm, dim3(g), dim3(256), 0, h->stream, n2, w, q, h->d_scalars, 4, h->d_partials + RED_BLOCKS);
    rc = reduce_scalar(h, h->d_partials + RED_BLOCKS, g, 1, 1);
    if (rc) return rc;
  }
  double host[2] = {0.0, 0.0};
  HIPCHK(hipMemcpyAsync(host, h->d_scalars, (sub_alpha ? 2 : 1) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  *alpha = host[0];
  if (sub_alpha && nrm_w) *nrm_w = host[1];
  return HXV_OK;
}

void launch_init_real(const hxv_handle* h, double* q, uint64_t seed, hipStream_t st) {
  const int64_t n = (int64_t)pitch_real_of(h) * h->host.qdw;
  hipLaunchKernelGGL(lz_init_real, dim3(grid_for(n)), dim3(256), 0, st, n, q, seed, h->host.dimup, pitch_real_of(h), h->host.dw0, h->dev.up_iperm, h->dev.up_sign);
}
}  // namespace hxv

extern "C" {

int hxv_lanczos_tridiag(hxv_handle* h, const void* d_vin, int32_t nlanc, double* alanc, double* blanc, double threshold,
                        int32_t* nsteps) {
  if (!h) return fail(HXV_ERR_ARG, "hxv_lanczos_tridiag: NULL handle");
  HIPCHK(hipSetDevice(h->device));  // (before the first collective: thread ranks on several GPUs each have their own current device)
  // (split sector: a rank whose arguments are bad tells its peers instead of leaving them in the first all-reduce)
  if (int rca = comm_agree(h, (!d_vin || nlanc < 1 || !alanc || !blanc) ? fail(HXV_ERR_ARG, "hxv_lanczos_tridiag: bad argument") : HXV_OK)) return rca;
  const int64_t n = (int64_t)h->host.pitch * h->host.qdw;  // this rank's slab
  StageFree staged{h, {nullptr, nullptr}};
  // REAL-vector mode: H real and the start vector purely real (c / c^dagger applied to a real ground state is) ->
  // the whole recurrence stays real; alanc/blanc are the same numbers at half the bytes per pass
  bool real = want_real(h);
  int rc_pre = HXV_OK;
  if (h->host.nranks != 1 && !comm_ready(h))
    return fail(HXV_ERR_STATE, "device Lanczos on a split sector needs the communicator: call hxv_comm_init after opening the sector");
  if (real) {  // sum of Im(vin)^2 over ALL ranks (dst = null: reduction only): every rank must take the same path
    const int pr = pitch_real_of(h);
    const int g = grid_for((int64_t)pr * h->host.qdw);
    hipLaunchKernelGGL(lz_to_real, dim3(g), dim3(256), 0, h->stream, h->host.dimup, h->host.qdw, h->host.pitch, pr,
                       (const double2*)d_vin, (double*)nullptr, h->d_partials);
    // (a rank that fails here must still meet its peers in the agreement below, not leave them there: ADVICE r4)
    rc_pre = reduce_scalar(h, h->d_partials, g, 5, 0);
    double im2 = 0.0;
    if (!rc_pre && hipMemcpyAsync(&im2, h->d_scalars + 5, sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess) rc_pre = fail(HXV_ERR_HIP, "hxv_lanczos_tridiag: reading the real-vector check failed");
    if (!rc_pre && hipStreamSynchronize(h->stream) != hipSuccess) rc_pre = fail(HXV_ERR_HIP, "hxv_lanczos_tridiag: the real-vector check failed");
    real = im2 == 0.0;
  }
  int rc = rc_pre ? rc_pre : stage_start_vector(h, d_vin, &staged.p[0]);  // (a start vector at hxv_slab_home: staged before its home is cleared)
  rc = comm_agree(h, rc ? rc : ensure_lz(h, real));  // (a rank that could not allocate tells its peers before the first all-reduce)
  if (rc) return rc;
  LzRunner lz(h, h->lz_vec[0], h->lz_vec[1], h->lz_vec[2], real);
  if (real)
    launch_to_real(h, (const double2*)d_vin, (double*)lz.b.q, h->stream);
  else
    HIPCHK(hipMemcpyAsync(lz.b.q, d_vin, (size_t)n * sizeof(double2), hipMemcpyDeviceToDevice, h->stream));
  // SciFortran's sp_lanc_tridiag normalises vin on its first iteration; the reference's call sites pass a normalised
  // vector already (ED_GF_NORMAL.f90:197-199).  Measure the (global) norm and normalise only when it is not 1.
  {
    const int gq = grid_for(lz.n2);
    hipLaunchKernelGGL(lz_nrm, dim3(gq), dim3(256), 0, h->stream, lz.n2, lz.b.q, h->d_partials + RED_BLOCKS);
    rc = reduce_scalar(h, h->d_partials + RED_BLOCKS, gq, 5, 1);
    if (rc) return rc;
    double nrm = 0.0;
    HIPCHK(hipMemcpyAsync(&nrm, h->d_scalars + 5, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (!(nrm > 0.0) || !std::isfinite(nrm)) return fail(HXV_ERR_ARG, "hxv_lanczos_tridiag: the start vector is zero or not finite");
    if (std::fabs(nrm - 1.0) <= 1e-14) {
      rc = lz.begin(1.0);
    } else if (lz.fused) {
      rc = lz.begin(nrm);  // the fused recurrence carries the scale factor
    } else {
      hipLaunchKernelGGL(lz_scale, dim3(gq), dim3(256), 0, h->stream, lz.n2, lz.b.q, lz.b.q, h->d_scalars, 5);
      rc = lz.begin(1.0);
    }
  }
  if (rc) return rc;
  for (int k = 0; k < nlanc; ++k) {
    alanc[k] = 0;
    blanc[k] = 0;
  }
  int k = 0;
  if (lz.fused && h->lz_graph && nlanc >= 8 && !comm_ready(h)) {  // (device-only iterations: serial sectors; the all-reduces of a split one go through the host loop)
    // Fixed-length run (the Green's-function use, ED_GF_NORMAL.f90:204-220): iteration 0 with the host in the loop, the
    // other nlanc-1 on the device alone -- scalars stay in device memory, the pointer-rotation cycle of three
    // iterations is one hipGraph -- and alpha/beta come back once at the end.  A breakdown (beta < threshold) is
    // found afterwards; whatever was computed past it is discarded.
    double a, bt;
    rc = lz.step(&a, &bt);
    if (rc) return rc;
    alanc[0] = a;
    blanc[1] = bt;
    k = 1;
    if (std::fabs(bt) >= threshold) {
      double* d_ab = nullptr;
      HIPCHK(hipMalloc((void**)&d_ab, (size_t)2 * nlanc * sizeof(double)));
      HIPCHK(hipMemsetAsync(d_ab, 0, (size_t)2 * nlanc * sizeof(double), h->stream));
      const double s1 = 1.0 / bt, bprev = 1.0 / lz.s_cur;
      const double init[2] = {s1, 1.0 / (s1 * bprev)};   // s_1 = 1/beta_1 ; c_1 = beta_1/beta_0, rounded like LzRunner::step()
      const double one = 1.0;
      hipError_t e1 = hipMemcpyAsync(h->d_scalars + 2, init, 2 * sizeof(double), hipMemcpyHostToDevice, h->stream);
      hipError_t e2 = hipMemcpyAsync(h->d_scalars + 6, &one, sizeof(double), hipMemcpyHostToDevice, h->stream);
      if (e1 == hipSuccess && e2 == hipSuccess) e1 = hipStreamSynchronize(h->stream);
      rc = (e1 != hipSuccess || e2 != hipSuccess) ? fail(HXV_ERR_HIP, "scalar upload failed") : lz.run_device_iterations(nlanc, d_ab, d_ab + nlanc, true);
      std::vector<double> ab((size_t)2 * nlanc, 0.0);
      if (rc == HXV_OK) {
        hipError_t e3 = hipMemcpyAsync(ab.data(), d_ab, ab.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream);
        if (e3 == hipSuccess) e3 = hipStreamSynchronize(h->stream);
        if (e3 != hipSuccess) rc = fail(HXV_ERR_HIP, std::string("reading alpha/beta back: ") + hipGetErrorString(e3));
      }
      (void)hipStreamSynchronize(h->stream);
      (void)hipFree(d_ab);
      if (rc) return rc;
      for (k = 1; k < nlanc; ++k) {
        alanc[k] = ab[k];
        const double b1 = k + 1 < nlanc ? ab[(size_t)nlanc + k + 1] : 1.0;
        if (k + 1 < nlanc) blanc[k + 1] = b1;
        if (!std::isfinite(alanc[k]) || !std::isfinite(b1) || (k + 1 < nlanc && std::fabs(b1) < threshold)) {
          if (!std::isfinite(alanc[k])) alanc[k] = 0.0;
          if (k + 1 < nlanc && !std::isfinite(b1)) blanc[k + 1] = 0.0;
          ++k;
          break;
        }
      }
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    if (nsteps) *nsteps = k;
    return HXV_OK;
  }
  for (; k < nlanc; ++k) {
    double a, bt;
    rc = lz.step(&a, &bt);
    if (rc) return rc;
    alanc[k] = a;
    if (k + 1 < nlanc) blanc[k + 1] = bt;
    if (std::fabs(bt) < threshold) {
      ++k;
      break;
    }
    if (k + 1 < nlanc) {
      rc = lz.advance();
      if (rc) return rc;
    }
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  if (nsteps) *nsteps = k;
  return HXV_OK;
}

int hxv_lanczos_eigh(hxv_handle* h, int32_t nitermax, double threshold, double* egs, void* d_vect, int32_t* niter) {
  if (!h || nitermax < 1 || !egs) return fail(HXV_ERR_ARG, "hxv_lanczos_eigh: bad argument");
  HIPCHK(hipSetDevice(h->device));  // (before the first collective)
  const bool real = want_real(h);  // the start vector is ours: real when H is (REAL-vector mode)
  int rc = comm_agree(h, ensure_lz(h, real));
  if (rc) return rc;
  const int64_t nc = (int64_t)h->host.pitch * h->host.qdw;                      // this rank's slab
  const int64_t n = real ? (int64_t)pitch_real_of(h) * h->host.qdw / 2 : nc;  // double2 elements per vector
  const int g = grid_for(n);
  const int nmax = (int)std::min<int64_t>(nitermax, h->host.dim);
  const uint64_t seed = 0x5EED5EEDull;
  // deterministic start vector, normalised
  auto start = [&](LzRunner& lz) -> int {
    if (real)
      launch_init_real(h, (double*)lz.b.w, seed, h->stream);
    else
      hipLaunchKernelGGL(lz_init, dim3(g), dim3(256), 0, h->stream, n, lz.b.w, seed, h->host.dimup, h->host.pitch, h->host.dw0, h->dev.up_iperm, h->dev.up_sign);
    HIPCHK(hipMemsetAsync(h->d_scalars, 0, 8 * sizeof(double), h->stream));
    hipLaunchKernelGGL(lz_nrm, dim3(g), dim3(256), 0, h->stream, n, lz.b.w, h->d_partials + RED_BLOCKS);
    int rcn = reduce_scalar(h, h->d_partials + RED_BLOCKS, g, 1, 1);
    if (rcn) return rcn;
    hipLaunchKernelGGL(lz_scale, dim3(g), dim3(256), 0, h->stream, n, lz.b.q, lz.b.w, h->d_scalars, 1);
    return lz.begin(1.0);
  };
  LzRunner lz(h, h->lz_vec[0], h->lz_vec[1], h->lz_vec[2], real);
  rc = start(lz);
  if (rc) return rc;
  std::vector<double> al, be(1, 0.0);
  double e_old = 1e300, e_new = 0;
  int k = 0;
  std::vector<double> d, e;
  for (; k < nmax; ++k) {
    double a, bt;
    rc = lz.step(&a, &bt);
    if (rc) return rc;
    al.push_back(a);
    d = al;
    e = be;
    if (!tridiag_ql((int)d.size(), d, e, nullptr)) return fail(HXV_ERR_STATE, "tridiagonal QL did not converge");
    e_new = *std::min_element(d.begin(), d.end());
    bool conv = std::fabs(e_new - e_old) < threshold;
    e_old = e_new;
    if (conv && d_vect) {
      // the energy converges quadratically faster than the vector: before accepting, require the Ritz
      // residual estimate |beta_{k+1} * y_k| (last component of the tridiagonal eigenvector) to be small too
      const int m = (int)al.size();
      std::vector<double> dd = al, ee = be, zz((size_t)m * m, 0.0);
      ee.resize(m, 0.0);
      for (int i = 0; i < m; ++i) zz[i + (size_t)i * m] = 1.0;
      if (!tridiag_ql(m, dd, ee, &zz)) return fail(HXV_ERR_STATE, "tridiagonal QL did not converge");
      int jm = (int)(std::min_element(dd.begin(), dd.end()) - dd.begin());
      const double resid = std::fabs(bt * zz[(size_t)(m - 1) + (size_t)jm * m]);
      conv = resid < 1e-11 * std::max(1.0, std::fabs(e_new));
    }
    if (conv || std::fabs(bt) < 1e-14 || k + 1 == nmax) {
      ++k;
      break;
    }
    be.push_back(bt);
    rc = lz.advance();
    if (rc) return rc;
  }
  *egs = e_new;
  if (niter) *niter = k;
  if (d_vect) {
    // second pass: re-run the recurrence and accumulate the Ritz vector sum_j y_j q_j
    const int m = (int)al.size();
    d = al;
    e = be;
    e.resize(m, 0.0);
    std::vector<double> z((size_t)m * m, 0.0);
    for (int i = 0; i < m; ++i) z[i + (size_t)i * m] = 1.0;
    if (!tridiag_ql(m, d, e, &z)) return fail(HXV_ERR_STATE, "tridiagonal QL did not converge");
    int jmin = (int)(std::min_element(d.begin(), d.end()) - d.begin());
    const double* y = &z[(size_t)jmin * m];
    double2* out = (double2*)d_vect;
    LzRunner lz2(h, h->lz_vec[0], h->lz_vec[1], h->lz_vec[2], real);
    rc = start(lz2);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(out, 0, (size_t)n * sizeof(double2), h->stream));
    for (int j = 0; j < m; ++j) {
      hipLaunchKernelGGL(lz_axpy, dim3(g), dim3(256), 0, h->stream, n, out, lz2.vec(), y[j] * lz2.scale());
      if (j + 1 == m) break;
      double a, bt;
      rc = lz2.step(&a, &bt);
      if (rc) return rc;
      rc = lz2.advance();
      if (rc) return rc;
    }
    // normalise the Ritz vector
    hipLaunchKernelGGL(lz_nrm, dim3(g), dim3(256), 0, h->stream, n, out, h->d_partials + RED_BLOCKS);
    rc = reduce_scalar(h, h->d_partials + RED_BLOCKS, g, 1, 1);
    if (rc) return rc;
    hipLaunchKernelGGL(lz_scale, dim3(g), dim3(256), 0, h->stream, n, out, out, h->d_scalars, 1);
    if (real) {
      // the Ritz vector was accumulated as a real vector in d_vect's memory: expand it to the complex layout of the API
      HIPCHK(hipMemcpyAsync(h->lz_vec[2], out, (size_t)n * sizeof(double2), hipMemcpyDeviceToDevice, h->stream));
      launch_to_complex(h, (const double*)h->lz_vec[2], out, h->stream);
    }
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return HXV_OK;
}

// Two REAL Lanczos runs on one complex product (real H): the start vectors travel as real and imaginary part of one
// complex vector -- H(x + i y) = H x + i H y -- and every scalar of the recurrence exists once per component.  The channels of
// one Green's-function solve are independent (ED_GF_NORMAL.f90:123-306: one sp_lanc_tridiag per channel), so two of them share
// the product's passes.  The arithmetic of each component is, operation for operation, that of hxv_lanczos_tridiag on (x, 0)
// through the same kernels (options real_vectors = 0, job_up = 0): alanc/blanc are bit-identical to two such runs.
int hxv_lanczos_tridiag_pair(hxv_handle* h, const void* d_vin_a, const void* d_vin_b, int32_t nlanc, double* alanc_a, double* blanc_a,
                             double* alanc_b, double* blanc_b, double threshold, int32_t* nsteps_a, int32_t* nsteps_b) {
  if (!h || !d_vin_a || !d_vin_b || nlanc < 1 || !alanc_a || !blanc_a || !alanc_b || !blanc_b)
    return fail(HXV_ERR_ARG, "hxv_lanczos_tridiag_pair: bad argument");
  if (const char* why = real_mode_blocker(h)) return fail(HXV_ERR_UNSUPPORTED, std::string("hxv_lanczos_tridiag_pair: unavailable: ") + why);
  if (h->host.nranks != 1 && !comm_ready(h))
    return fail(HXV_ERR_STATE, "device Lanczos on a split sector needs the communicator: call hxv_comm_init after opening the sector");
  if (!h->lz_fused) return fail(HXV_ERR_UNSUPPORTED, "hxv_lanczos_tridiag_pair needs the fused recurrence (option lanczos_fused)");
  HIPCHK(hipSetDevice(h->device));
  StageFree staged{h, {nullptr, nullptr}};
  const bool dist = comm_ready(h);
  const int64_t n = (int64_t)h->host.pitch * h->host.qdw;
  const int g = grid_for(n);
  const int64_t nwg = std::max<int64_t>(1, tiled_pass_up_workgroups(h->dev, h->plan, false, dw_part_in_pieces(h)));
  // scalars: [0] alpha_a [1] beta_a [2] s_a [3] c_a [4] alpha_a*s_a ; [8..12] the same for b ; block partials behind them
  double* d_sc = nullptr;
  // every rank-local preparation that can fail comes BEFORE the agreement: a rank that cannot go on makes all ranks return
  auto prepare = [&]() -> int {
    int r = stage_start_vector(h, d_vin_a, &staged.p[0]);  // (start vectors at hxv_slab_home: staged before their home is cleared)
    if (!r) r = stage_start_vector(h, d_vin_b, &staged.p[1]);
    if (!r) r = ensure_lz(h, false);  // (split sector: vectors are this rank's slab, every sum below is all-reduced)
    if (!r) r = ensure_wt(h);
    if (r) return r;
    if (2 * nwg > h->lz_partial_n) {
      if (h->d_lz_partial) (void)hipFree(h->d_lz_partial);
      h->d_lz_partial = nullptr;
      h->lz_partial_n = 0;
      HIPCHK(hipMalloc((void**)&h->d_lz_partial, (size_t)2 * nwg * sizeof(double)));
      HIPCHK(hipMemsetAsync(h->d_lz_partial, 0, (size_t)2 * nwg * sizeof(double), h->stream));  // (a rank without columns launches nothing)
      h->lz_partial_n = 2 * nwg;
    }
    HIPCHK(hipMalloc((void**)&d_sc, (size_t)(16 + 3 * RED_BLOCKS) * sizeof(double)));
    return HXV_OK;
  };
  int rc = comm_agree(h, prepare());
  struct Free {
    double*& p;
    ~Free() {
      if (p) (void)hipFree(p);
    }
  } guard{d_sc};
  if (rc) return rc;
  double* d_p0 = d_sc + 16;
  double* d_p1 = d_p0 + RED_BLOCKS;
  double* d_p2 = d_p1 + RED_BLOCKS;
  HIPCHK(hipMemsetAsync(d_sc, 0, 16 * sizeof(double), h->stream));
  // d_sc[slot] = sum (op 0) or 2-norm (op 1) of `np` partial sums -- over all ranks on a split sector
  auto reduce = [&](const double* part, int np, int slot, int op) -> int {
    hipLaunchKernelGGL(lz_final, dim3(1), dim3(256), 0, h->stream, part, np, d_sc, slot, dist ? 0 : op);
    if (dist) {
      if (int rca = comm_allreduce_sum(h, d_sc + slot, 1, h->stream)) return rca;
      if (op) hipLaunchKernelGGL(lz_sqrt, dim3(1), dim3(1), 0, h->stream, d_sc, slot);
    }
    return HXV_OK;
  };
  double2 *q = h->lz_vec[0], *qm = h->lz_vec[1], *w = h->lz_vec[2];
  hipLaunchKernelGGL(lz_pack_pair, dim3(g), dim3(256), 0, h->stream, n, (const double2*)d_vin_a, (const double2*)d_vin_b, q, d_p0, d_p1, d_p2);
  if ((rc = reduce(d_p0, g, 5, 0)) || (rc = reduce(d_p1, g, 6, 1)) || (rc = reduce(d_p2, g, 7, 1))) return rc;
  double head[3];
  HIPCHK(hipMemcpyAsync(head, d_sc + 5, 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (head[0] != 0.0) return fail(HXV_ERR_ARG, "hxv_lanczos_tridiag_pair: the start vectors must be real (zero imaginary parts)");
  for (int c = 0; c < 2; ++c)
    if (!(head[1 + c] > 0.0) || !std::isfinite(head[1 + c])) return fail(HXV_ERR_ARG, "hxv_lanczos_tridiag_pair: a start vector is zero or not finite");
  for (int k = 0; k < nlanc; ++k) alanc_a[k] = blanc_a[k] = alanc_b[k] = blanc_b[k] = 0.0;
  // per component, exactly LzRunner's fused recurrence: q = s*X, s = 1/beta_k (the start vector's norm carried as beta_0)
  double s_cur[2], beta_prev[2];
  bool alive[2] = {true, true};
  int done[2] = {0, 0};
  for (int c = 0; c < 2; ++c) {
    const double nrm = std::fabs(head[1 + c] - 1.0) <= 1e-14 ? 1.0 : head[1 + c];
    s_cur[c] = 1.0 / nrm;
    beta_prev[c] = nrm;
  }
  double* al[2] = {alanc_a, alanc_b};
  double* bl[2] = {blanc_a, blanc_b};
  bool first = true;
  for (int k = 0; k < nlanc && (alive[0] || alive[1]); ++k) {
    double sc[16] = {0};
    for (int c = 0; c < 2; ++c) {
      sc[8 * c + 2] = alive[c] ? s_cur[c] : 0.0;  // (a finished component is multiplied away: the other one goes on alone)
      sc[8 * c + 3] = (first || !alive[c]) ? 0.0 : 1.0 / (s_cur[c] * beta_prev[c]);
    }
    HIPCHK(hipMemcpyAsync(d_sc + 2, sc + 2, 2 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(d_sc + 10, sc + 10, 2 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    LzEpilogue ep;
    ep.xm = first ? nullptr : qm;
    ep.scal = d_sc;
    ep.i_s = 2;
    ep.i_c = 3;
    ep.pair = 1;
    ep.i_s2 = 10;
    ep.i_c2 = 11;
    ep.partial = h->d_lz_partial;
    ep.partial2 = h->d_lz_partial + nwg;
    if (dist) {  // exchange + product (every rank takes this branch together)
      if ((rc = apply_slab(h, q, w, h->stream, &ep))) return rc;
    } else {
      hipError_t e = launch_hxv_tiled(h->dev, h->plan, q, h->d_wt, w, h->stream, &ep);
      if (e != hipSuccess) return fail(HXV_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
      h->n_apply++;
    }
    if ((rc = reduce(h->d_lz_partial, (int)nwg, 0, 0)) || (rc = reduce(h->d_lz_partial + nwg, (int)nwg, 8, 0))) return rc;
    hipLaunchKernelGGL(lz_mul, dim3(1), dim3(1), 0, h->stream, d_sc, 4, 0, 2);
    hipLaunchKernelGGL(lz_mul, dim3(1), dim3(1), 0, h->stream, d_sc, 12, 8, 10);
    hipLaunchKernelGGL(lz_sub_nrm_pair, dim3(g), dim3(256), 0, h->stream, n, w, q, d_sc, 4, 12, d_p0, d_p1);
    if ((rc = reduce(d_p0, g, 1, 1)) || (rc = reduce(d_p1, g, 9, 1))) return rc;
    double host[16];
    HIPCHK(hipMemcpyAsync(host, d_sc, 16 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int c = 0; c < 2; ++c) {
      if (!alive[c]) continue;
      const double a = host[8 * c], bt = host[8 * c + 1];
      al[c][k] = a;
      if (k + 1 < nlanc) bl[c][k + 1] = bt;
      done[c] = k + 1;
      if (std::fabs(bt) < threshold || !std::isfinite(bt)) {
        alive[c] = false;  // breakdown of this component's Krylov space (the early exit of hxv_lanczos_tridiag)
        continue;
      }
      beta_prev[c] = 1.0 / s_cur[c];
      s_cur[c] = 1.0 / bt;
    }
    // rotate: X_{k+1} = w (unnormalised residual), X_k = q, the old X_{k-1} becomes the next output
    double2* old_m = qm;
    qm = q;
    q = w;
    w = old_m;
    first = false;
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  h->last_real = 2;
  if (nsteps_a) *nsteps_a = done[0];
  if (nsteps_b) *nsteps_b = done[1];
  return HXV_OK;
}

namespace {
int ensure_stage(hxv_handle* h) {
  const size_t bytes = (size_t)h->host.pitch * std::max(h->host.qdw, 1) * sizeof(double2);
  if (!h->d_stage_v) {
    HIPCHK(pool_alloc(h->device, bytes, (void**)&h->d_stage_v));
    HIPCHK(pool_alloc(h->device, bytes, (void**)&h->d_stage_hv));
    HIPCHK(hipMemsetAsync(h->d_stage_v, 0, bytes, h->stream));  // (on the handle's stream: it does not synchronise with the null stream)
    HIPCHK(hipMemsetAsync(h->d_stage_hv, 0, bytes, h->stream));
    h->device_bytes += 2 * (int64_t)bytes;
  }
  return HXV_OK;
}
}  // namespace

int hxv_lanczos_tridiag_host(hxv_handle* h, const void* vin_host, int32_t nlanc, double* alanc, double* blanc, double threshold,
                             int32_t* nsteps) {
  if (!h || !vin_host) return fail(HXV_ERR_ARG, "hxv_lanczos_tridiag_host: NULL argument");
  HIPCHK(hipSetDevice(h->device));
  int rc = ensure_stage(h);
  if (rc) return rc;
  rc = slab_from_host(h, vin_host, h->d_stage_v);
  if (rc) return rc;
  return hxv_lanczos_tridiag(h, h->d_stage_v, nlanc, alanc, blanc, threshold, nsteps);
}

int hxv_lanczos_tridiag_pair_host(hxv_handle* h, const void* vin_a_host, const void* vin_b_host, int32_t nlanc, double* alanc_a,
                                  double* blanc_a, double* alanc_b, double* blanc_b, double threshold, int32_t* nsteps_a, int32_t* nsteps_b) {
  if (!h || !vin_a_host || !vin_b_host) return fail(HXV_ERR_ARG, "hxv_lanczos_tridiag_pair_host: NULL argument");
  HIPCHK(hipSetDevice(h->device));
  int rc = ensure_stage(h);
  if (rc) return rc;
  rc = slab_from_host(h, vin_a_host, h->d_stage_v);
  if (rc) return rc;
  rc = slab_from_host(h, vin_b_host, h->d_stage_hv);
  if (rc) return rc;
  return hxv_lanczos_tridiag_pair(h, h->d_stage_v, h->d_stage_hv, nlanc, alanc_a, blanc_a, alanc_b, blanc_b, threshold, nsteps_a, nsteps_b);
}

int hxv_lanczos_eigh_host(hxv_handle* h, int32_t nitermax, double threshold, double* egs, void* vect_host, int32_t* niter) {
  if (!h) return fail(HXV_ERR_ARG, "hxv_lanczos_eigh_host: NULL handle");
  HIPCHK(hipSetDevice(h->device));
  int rc = ensure_stage(h);
  if (rc) return rc;
  rc = hxv_lanczos_eigh(h, nitermax, threshold, egs, vect_host ? h->d_stage_hv : nullptr, niter);
  if (rc) return rc;
  if (vect_host) return slab_to_host(h, h->d_stage_hv, vect_host);
  return HXV_OK;
}

int hxv_apply_ladder(hxv_handle* from, hxv_handle* to, int32_t orbital, int32_t spin, int32_t create, const void* d_psi, void* d_out,
                     double* norm2) {
  return hxv_apply_ladder_axpy(from, to, orbital, spin, create, 1.0, 0.0, 0, d_psi, d_out, norm2);
}

namespace {
// out[c][:] = (accumulate ? out[c][:] : 0) + coef * sign[c] * src_c[:]  for the local target columns c of a spin-dw ladder operator on a
// split sector: src_c = column slot[c] of the local source slab (slot >= 0), column -1-slot[c] of the received columns (slot < 0),
// or nothing (sign[c] == 0: the target column is not reached).  Rows are untouched by a dw operator; pad rows stay as they are.
__global__ void __launch_bounds__(256) ladder_cols_kernel(int dimup, int ncols, int pitch_from, int pitch_to, const int32_t* __restrict__ slot,
                                                         const int32_t* __restrict__ sign, const double2* __restrict__ psi_local,
                                                         const double2* __restrict__ recv, double2* __restrict__ out, double2 coef, int accumulate) {
  const int c = blockIdx.x;
  if (c >= ncols) return;
  const int sg = sign[c];
  const double2* __restrict__ src = sg == 0 ? nullptr : (slot[c] >= 0 ? psi_local + (int64_t)slot[c] * pitch_from : recv + (int64_t)(-1 - slot[c]) * pitch_from);
  double2* __restrict__ dst = out + (int64_t)c * pitch_to;
  for (int i = threadIdx.x; i < dimup; i += 256) {
    double2 r = make_double2(0.0, 0.0);
    if (sg != 0) {
      const double2 x = src[i];
      r = make_double2(sg * (coef.x * x.x - coef.y * x.y), sg * (coef.x * x.y + coef.y * x.x));
    }
    if (accumulate) {
      const double2 o = dst[i];
      r.x += o.x;
      r.y += o.y;
    }
    dst[i] = r;
  }
}

// position of a bit pattern in a sorted sector map (ED_SETUP.f90:1044-1061 binary_search), -1 if absent
int rank_in_map_host(const std::vector<uint32_t>& map, uint32_t m) {
  auto it = std::lower_bound(map.begin(), map.end(), m);
  return (it != map.end() && *it == m) ? (int)(it - map.begin()) : -1;
}
}  // namespace

int hxv_apply_ladder_axpy(hxv_handle* from, hxv_handle* to, int32_t orbital, int32_t spin, int32_t create, double coef_re, double coef_im,
                          int32_t accumulate, const void* d_psi, void* d_out, double* norm2) {
  if (!from || !to || !d_psi || !d_out) return fail(HXV_ERR_ARG, "hxv_apply_ladder: NULL argument");
  const SectorHost &a = from->host, &b = to->host;
  if (a.map_up.empty() || b.map_up.empty()) return fail(HXV_ERR_STATE, "hxv_apply_ladder needs handles built from a model (basis maps)");
  if (from->device != to->device) return fail(HXV_ERR_ARG, "hxv_apply_ladder: handles on different devices");
  if (a.ns != b.ns || orbital < 0 || orbital >= a.ns || spin < 0 || spin > 1) return fail(HXV_ERR_ARG, "hxv_apply_ladder: bad orbital/spin");
  const int d = create ? 1 : -1;
  if (spin == 0 ? (b.nup != a.nup + d || b.ndw != a.ndw) : (b.ndw != a.ndw + d || b.nup != a.nup))
    return fail(HXV_ERR_ARG, "hxv_apply_ladder: `to` is not the sector reached by this operator");
  if (a.nranks != b.nranks || a.rank != b.rank) return fail(HXV_ERR_ARG, "hxv_apply_ladder: the two sectors must be split over the same ranks");
  const bool split = b.nranks != 1 || comm_ready(to);
  if (split && !comm_ready(to)) return fail(HXV_ERR_STATE, "hxv_apply_ladder on split sectors needs the communicator of `to` (hxv_comm_init after opening it)");
  HIPCHK(hipSetDevice(to->device));
  hipStream_t st = to->stream;
  const double2 coef = make_double2(coef_re, coef_im);
  // The reference applies c / c^dagger on the master and scatters the result (ED_GF_NORMAL.f90:174-214).  Here every rank builds its
  // own slab of the new vector:
  //  * spin up: the operator acts inside a column and both sectors have the same DimDw, hence the same split -- purely local;
  //  * spin dw: target column j of sector B is (a sign times) ONE column of sector A, which may belong to another rank of A's split:
  //    a column permutation.  Every rank derives from the two dw maps what it needs from whom and what everybody needs from it
  //    (deterministic, no negotiation), packs, exchanges once, and assembles.
  if (!accumulate) HIPCHK(hipMemsetAsync(d_out, 0, (size_t)b.pitch * std::max(b.qdw, 1) * sizeof(double2), st));  // pad rows = 0
  if (spin == 0 || !split) {
    // spin up with a device row order (SectorHost::up_perm): the source row is looked up in the source sector's SORTED reference map and sent
    // through its permutation; `to`'s map is by device row already; both basis signs ride along.  A dw operator keeps the row, and the two
    // sectors -- same nup, same model -- have the same row order: nothing to do.
    const uint32_t* mf = spin == 0 ? (a.row_order() ? from->dev.map_up_ref : from->dev.diag.map_up) : from->dev.diag.map_dw;
    const uint32_t* mt = spin == 0 ? to->dev.diag.map_up : to->dev.diag.map_dw + b.dw0;  // (dw: the local target columns)
    if (spin == 0 && a.qdw != b.qdw) return fail(HXV_ERR_STATE, "hxv_apply_ladder: the DimDw splits of the two sectors differ");
    if (spin == 1 && (a.row_order() != b.row_order() || (a.row_order() && a.up_pos != b.up_pos)))
      return fail(HXV_ERR_STATE, "hxv_apply_ladder: the two sectors store their rows in different orders (HXV_ROW_ORDER changed between the opens?)");
    hipError_t e = launch_ladder(mf, spin == 0 ? a.dimup : a.dimdw, mt, spin == 0 ? b.dimup : b.dimdw, a.pitch, b.dimup, b.pitch, b.qdw,
                                 orbital, spin, create ? 1 : 0, (const double2*)d_psi, (double2*)d_out, st, coef, accumulate ? 1 : 0,
                                 spin == 0 ? from->dev.up_perm : nullptr, spin == 0 ? from->dev.up_sign : nullptr, spin == 0 ? to->dev.up_sign : nullptr);
    if (e != hipSuccess) return fail(HXV_ERR_HIP, std::string("ladder kernel: ") + hipGetErrorString(e));
  } else {
    const int P = b.nranks, r = b.rank;
    const uint32_t bit = 1u << orbital;
    // source column (global, in A) and sign of a target column (global, in B); -1: not reached
    auto source_of = [&](int jb, int& sg) -> int {
      const uint32_t mb = b.map_dw[jb];
      if (((mb & bit) != 0u) != (create != 0)) return -1;
      const uint32_t ma = mb ^ bit;
      sg = (__builtin_popcount(ma & (bit - 1u)) & 1) ? -1 : 1;  // (-1)^(occupied orbitals of this spin below `orbital`): c/cdg, ED_SETUP.f90:807-833
      return rank_in_map_host(a.map_dw, ma);
    };
    std::vector<int> first_a(P + 1), first_b(P + 1);
    for (int p = 0; p < P; ++p) {
      int q, c0;
      dw_split(a.dimdw, p, P, q, c0);
      first_a[p] = c0;
      dw_split(b.dimdw, p, P, q, c0);
      first_b[p] = c0;
    }
    first_a[P] = a.dimdw;
    first_b[P] = b.dimdw;
    auto owner_a = [&](int ja) { return (int)(std::upper_bound(first_a.begin(), first_a.end(), ja) - first_a.begin()) - 1; };
    // what I receive: my target columns in order, grouped by the source's owner; what I send: every peer's target columns in ITS order
    std::vector<int32_t> slot(std::max(b.qdw, 1), 0), sgn(std::max(b.qdw, 1), 0), send_cols;
    std::vector<int64_t> recv_ptr(P + 1, 0), send_ptr(P + 1, 0);
    std::vector<std::vector<int>> want(P);  // per owner: my local target columns whose source it holds, ascending
    for (int c = 0; c < b.qdw; ++c) {
      int sg = 0;
      const int ja = source_of(b.dw0 + c, sg);
      if (ja < 0) continue;
      sgn[c] = sg;
      const int o = owner_a(ja);
      if (o == r)
        slot[c] = ja - a.dw0;
      else
        want[o].push_back(c);
    }
    int nrecv = 0;
    for (int p = 0; p < P; ++p) {
      recv_ptr[p] = nrecv;
      for (int c : want[p]) slot[c] = -1 - nrecv++;
    }
    recv_ptr[P] = nrecv;
    for (int p = 0; p < P; ++p) {
      send_ptr[p] = (int64_t)send_cols.size();
      if (p == r) continue;
      for (int jb = first_b[p]; jb < first_b[p + 1]; ++jb) {
        int sg = 0;
        const int ja = source_of(jb, sg);
        if (ja >= 0 && owner_a(ja) == r) send_cols.push_back(ja - a.dw0);
      }
    }
    send_ptr[P] = (int64_t)send_cols.size();
    const size_t cb = (size_t)a.pitch * sizeof(double2);
    double2 *d_sendbuf = nullptr, *d_recvbuf = nullptr;
    int32_t* d_lists = nullptr;
    const size_t nl = send_cols.size() + 2 * (size_t)std::max(b.qdw, 1);
    int rc_local = HXV_OK;
    hipError_t e1 = pool_alloc(to->device, std::max<size_t>(send_cols.size(), 1) * cb, (void**)&d_sendbuf);
    hipError_t e2 = pool_alloc(to->device, std::max<size_t>((size_t)nrecv, 1) * cb, (void**)&d_recvbuf);
    hipError_t e3 = hipMalloc((void**)&d_lists, std::max<size_t>(nl, 1) * sizeof(int32_t));
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) rc_local = fail(HXV_ERR_HIP, "hxv_apply_ladder: staging buffers for the column exchange");
    auto release = [&]() {
      (void)hipStreamSynchronize(st);
      if (d_sendbuf) pool_free(to->device, d_sendbuf);
      if (d_recvbuf) pool_free(to->device, d_recvbuf);
      if (d_lists) (void)hipFree(d_lists);
    };
    int rc = comm_agree(to, rc_local);
    if (rc) {
      release();
      return rc;
    }
    int32_t* d_send_cols = d_lists;
    int32_t* d_slot = d_lists + send_cols.size();
    int32_t* d_sgn = d_slot + std::max(b.qdw, 1);
    hipError_t e = hipSuccess;
    if (!send_cols.empty()) e = hipMemcpyAsync(d_send_cols, send_cols.data(), send_cols.size() * sizeof(int32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_slot, slot.data(), slot.size() * sizeof(int32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_sgn, sgn.data(), sgn.size() * sizeof(int32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = launch_pack_columns((const double2*)d_psi, d_sendbuf, d_send_cols, (int)send_cols.size(), a.pitch, st);
    if (e != hipSuccess) {
      release();
      return fail(HXV_ERR_HIP, std::string("hxv_apply_ladder: ") + hipGetErrorString(e));
    }
    rc = comm_sendrecv_cols(to, d_sendbuf, send_ptr.data(), d_recvbuf, recv_ptr.data(), cb, st);
    if (rc) {
      release();
      return rc;
    }
    if (b.qdw > 0)
      hipLaunchKernelGGL(ladder_cols_kernel, dim3((unsigned)b.qdw), dim3(256), 0, st, b.dimup, b.qdw, a.pitch, b.pitch, d_slot, d_sgn, (const double2*)d_psi,
                         d_recvbuf, (double2*)d_out, coef, accumulate ? 1 : 0);
    release();  // (synchronises the stream: the host lists above are read by asynchronous copies)
  }
  if (norm2) {
    const int64_t n = (int64_t)b.pitch * b.qdw;  // pads of d_out must be zero (hxv.h)
    const int g = grid_for(n);
    hipLaunchKernelGGL(lz_nrm, dim3(g), dim3(256), 0, st, n, (const double2*)d_out, to->d_partials);
    int rcn = reduce_scalar(to, to->d_partials, g, 5, 0);  // <out|out> over ALL ranks of a split sector
    if (rcn) return rcn;
    HIPCHK(hipMemcpyAsync(norm2, to->d_scalars + 5, sizeof(double), hipMemcpyDeviceToHost, st));
  }
  HIPCHK(hipStreamSynchronize(st));
  return HXV_OK;
}

int hxv_time_lanczos(hxv_handle* h, void* d_work3, int32_t nrep, float* ms_per_iter) {
  if (!h || !d_work3 || nrep < 1 || !ms_per_iter) return fail(HXV_ERR_ARG, "hxv_time_lanczos: bad argument");
  if (h->host.nranks != 1) return fail(HXV_ERR_STATE, "hxv_time_lanczos needs nranks==1");
  HIPCHK(hipSetDevice(h->device));
  int64_t n = (int64_t)h->host.pitch * h->host.dimdw;
  HIPCHK(hipMemsetAsync(d_work3, 0, (size_t)3 * n * sizeof(double2), h->stream));
  const bool real = want_real(h);
  LzRunner lz(h, (double2*)d_work3, (double2*)d_work3 + n, (double2*)d_work3 + 2 * n, real);
  const int64_t nfull = n;
  n = lz.n2;
  const int g = grid_for(n);
  if (real)
    launch_init_real(h, (double*)lz.b.w, 0x1234ull, h->stream);
  else
    hipLaunchKernelGGL(lz_init, dim3(g), dim3(256), 0, h->stream, n, lz.b.w, 0x1234ull, h->host.dimup, h->host.pitch, 0, h->dev.up_iperm, h->dev.up_sign);
  (void)nfull;
  HIPCHK(hipMemsetAsync(h->d_scalars, 0, 8 * sizeof(double), h->stream));
  hipLaunchKernelGGL(lz_nrm, dim3(g), dim3(256), 0, h->stream, n, lz.b.w, h->d_partials + RED_BLOCKS);
  hipLaunchKernelGGL(lz_final, dim3(1), dim3(256), 0, h->stream, h->d_partials + RED_BLOCKS, g, h->d_scalars, 1, 1);
  hipLaunchKernelGGL(lz_scale, dim3(g), dim3(256), 0, h->stream, n, lz.b.q, lz.b.w, h->d_scalars, 1);
  HIPCHK(hipMemsetAsync(lz.b.qm, 0, (size_t)n * sizeof(double2), h->stream));
  int rc = lz.begin(1.0);
  if (rc) return rc;
  double a, bt;
  rc = lz.step(&a, &bt);  // untimed first step (lazy allocations)
  if (rc) return rc;
  if (lz.fused && h->lz_graph && nrep >= 8 && !comm_ready(h)) {
    // the fixed-length device-only path of hxv_lanczos_tridiag: iterations 1..nrep, three per hipGraph
    double* d_ab = nullptr;
    HIPCHK(hipMalloc((void**)&d_ab, (size_t)2 * (nrep + 1) * sizeof(double)));
    const double s1 = 1.0 / bt, bprev = 1.0 / lz.s_cur;
    const double init[2] = {s1, 1.0 / (s1 * bprev)};
    const double one = 1.0;
    hipError_t e = hipMemcpyAsync(h->d_scalars + 2, init, 2 * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h->d_scalars + 6, &one, sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = hipEventRecord(h->ev0, h->stream);
    rc = e != hipSuccess ? fail(HXV_ERR_HIP, std::string("hxv_time_lanczos: ") + hipGetErrorString(e))
                         : lz.run_device_iterations(nrep + 1, d_ab, d_ab + nrep + 1, true);
    if (rc == HXV_OK) {
      e = hipEventRecord(h->ev1, h->stream);
      if (e == hipSuccess) e = hipEventSynchronize(h->ev1);
      if (e != hipSuccess) rc = fail(HXV_ERR_HIP, std::string("hxv_time_lanczos: ") + hipGetErrorString(e));
    }
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(d_ab);
    if (rc) return rc;
    float msg = 0;
    HIPCHK(hipEventElapsedTime(&msg, h->ev0, h->ev1));
    *ms_per_iter = msg / (float)nrep;
    return HXV_OK;
  }
  rc = lz.advance();
  if (rc) return rc;
  HIPCHK(hipEventRecord(h->ev0, h->stream));
  for (int k = 0; k < nrep; ++k) {
    rc = lz.step(&a, &bt);
    if (rc) return rc;
    rc = lz.advance();
    if (rc) return rc;
  }
  HIPCHK(hipEventRecord(h->ev1, h->stream));
  HIPCHK(hipEventSynchronize(h->ev1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *ms_per_iter = ms / (float)nrep;
  return HXV_OK;
}

}  // extern "C"
