// Preconditioners beyond the diagonal ones, and the
// sweeps of the PCG form that takes z = M^-1 r as a vector:
//   LSB_PRECOND_CHEBYSHEV    z = p_m(D^-1 S) D^-1 r, the Chebyshev polynomial of
//                            degree m on [lmax/30, lmax] -- the smoother family the
//                            reference's AMG backends configure (src/hypre.c:126-158
//                            relax types / sweeps, src/amgx.c:78-85); m SpMVs and NO
//                            reduction per application: on 8 GPUs an outer iteration
//                            carries m + 1 SpMVs per pair of all-reduces;
//   LSB_PRECOND_BLOCKJACOBI  z = blockdiag(S)^-1 r with dense diagonal blocks of
//                            opts.block_size rows inverted at setup (in place, on the
//                            device, by Gauss-Jordan sweeps -- the blocks of an SPD
//                            matrix are SPD, no pivoting); the block form of the
//                            Jacobi preconditioner of src/ginkgo.cpp:57-58.  One block
//                            as large as the operator is a cached dense inverse: what
//                            the reference's CHOLMOD path keeps (a factor, computed
//                            outside the timed loop, src/cholmod-impl.h:25-26) in the
//                            form a GPU streams best.
// All HBM-bound elementwise / dense-GEMV sweeps; no MFMA (fp64 GEMV at 0.25 flop/B).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lsb_impl.h"

#define WG 256

__device__ __forceinline__ void pwg_sum2(double (&v)[2], double *sred) {
#pragma unroll
  for (int k = 0; k < 2; k++)
    for (int off = 32; off > 0; off >>= 1)
      v[k] += __shfl_xor(v[k], off, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0)
    sred[wave * 2] = v[0], sred[wave * 2 + 1] = v[1];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; k++)
    v[k] = (sred[k] + sred[2 + k]) + (sred[4 + k] + sred[6 + k]);
}

// partials (r.z, r.r), the record k_pcg_update_p consumes
__global__ __launch_bounds__(WG) void k_dot2(unsigned n, const double *__restrict__ r,
                                             const double *__restrict__ z,
                                             double *__restrict__ partials2,
                                             const lsb_pcg_state *__restrict__ st) {
  if (st && st->status)
    return;
  __shared__ double sred[8];
  double acc[2] = {0.0, 0.0};
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
    const double ri = r[i];
    acc[0] += ri * z[i];
    acc[1] += ri * ri;
  }
  pwg_sum2(acc, sred);
  if (threadIdx.x == 0)
    partials2[2 * blockIdx.x] = acc[0], partials2[2 * blockIdx.x + 1] = acc[1];
}

// Chebyshev, first term: d = c0 D^-1 r ; z = d
__global__ __launch_bounds__(WG) void k_cheb_first(unsigned n, const double *__restrict__ r,
                                                   const double *__restrict__ dinv, double dc,
                                                   double c0, double *__restrict__ d,
                                                   double *__restrict__ z,
                                                   const lsb_pcg_state *__restrict__ st) {
  if (st && st->status)
    return;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
    const double v = c0 * ((dinv ? dinv[i] : dc) * r[i]);
    d[i] = v, z[i] = v;
  }
}

// Chebyshev, step k: d = a d + b D^-1 (r - w) ; z += d          (w = S z)
__global__ __launch_bounds__(WG) void k_cheb_step(unsigned n, const double *__restrict__ r,
                                                  const double *__restrict__ w,
                                                  const double *__restrict__ dinv, double dc,
                                                  double a, double b, double *__restrict__ d,
                                                  double *__restrict__ z,
                                                  const lsb_pcg_state *__restrict__ st) {
  if (st && st->status)
    return;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
    // (explicit fma: the same bits as the step fused into the SpMV's epilogue, hip_kernels.hip)
    const double v = fma(a, d[i], b * ((dinv ? dinv[i] : dc) * (r[i] - w[i])));
    d[i] = v;
    z[i] += v;
  }
}

// v = c * D^-1 w   (power iteration of D^-1 S at setup)
__global__ __launch_bounds__(WG) void k_scale_dinv(unsigned n, double c,
                                                   const double *__restrict__ dinv,
                                                   const double *w, double *v) { // w may be v
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG)
    v[i] = c * (dinv[i] * w[i]);
}

__global__ __launch_bounds__(WG) void k_scale_vec(unsigned n, double c, double *__restrict__ v) {
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG)
    v[i] *= c;
}

__global__ __launch_bounds__(WG) void k_power_start(unsigned n, unsigned first,
                                                    double *__restrict__ v) {
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG)
    v[i] = 1.0 + (double)(((unsigned)(i + first) * 7919u) % 1024u) / 1024.0;
}

// ---- block-Jacobi -----------------------------------------------------------
// Storage: block k (rows [k bs, k bs + m_k), m_k = bs except the last) is a dense
// m_k x m_k array at binv + k bs^2 with leading dimension m_k.  The inverse of a
// symmetric block is symmetric, so row i of it can be read as column i: lane i
// of a block reads element (j, i) at j m_k + i -- consecutive lanes, consecutive
// addresses.

// In-place Gauss-Jordan inversion of all blocks at once, pivot p of every block
// per step (no pivoting: the diagonal blocks of an SPD operator are SPD):
//   A'[p][p] = 1/piv          A'[p][j] = A[p][j] / piv
//   A'[i][p] = -A[i][p]/piv   A'[i][j] = A[i][j] - A[i][p] A[p][j] / piv   (i, j != p)
// after the last pivot the array holds A^-1 (symmetric again; in between it is
// not, so a step needs row p AND column p of the state before it).
//   step A (k_gj_rowcol): save row p and column p of every block
//   step B (k_gj_update): one thread per element
__global__ __launch_bounds__(WG) void k_gj_rowcol(unsigned n, unsigned bs, unsigned p,
                                                  const double *__restrict__ binv,
                                                  double *__restrict__ colp,
                                                  double *__restrict__ rowp) {
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
    const unsigned k = (unsigned)(i / bs), il = (unsigned)(i % bs);
    const unsigned m = min(bs, n - k * bs);
    if (p < m) {
      const double *blk = binv + (size_t)k * bs * bs;
      colp[i] = blk[(size_t)il * m + p];
      rowp[i] = blk[(size_t)p * m + il];
    }
  }
}

__global__ __launch_bounds__(WG) void k_gj_update(unsigned n, unsigned bs, unsigned p,
                                                  double *__restrict__ binv,
                                                  const double *__restrict__ colp,
                                                  const double *__restrict__ rowp) {
  const size_t total = (size_t)n * bs;
  for (size_t e = (size_t)blockIdx.x * WG + threadIdx.x; e < total; e += (size_t)gridDim.x * WG) {
    const size_t i = e / bs;
    const unsigned j = (unsigned)(e % bs);
    const unsigned k = (unsigned)(i / bs), il = (unsigned)(i % bs);
    const unsigned m = min(bs, n - k * bs);
    if (p >= m || j >= m)
      continue;
    double *blk = binv + (size_t)k * bs * bs;
    const double piv = colp[(size_t)k * bs + p];
    const double aip = colp[(size_t)k * bs + il], apj = rowp[(size_t)k * bs + j];
    double v;
    if (il == p && j == p)
      v = 1.0 / piv;
    else if (il == p)
      v = apj / piv;
    else if (j == p)
      v = -aip / piv;
    else
      v = blk[(size_t)il * m + j] - aip * apj / piv;
    blk[(size_t)il * m + j] = v;
  }
}

// "The residual already meets the tolerance": the sweep before the preconditioner
// (k_pcg_update_xr) left partial sums of r.r; when they add up to <= thresh2 the
// iteration will end in k_pcg_update_p anyway and z is never used -- a dense block
// need not be streamed for it (one shard only: over shards r.r needs an all-reduce).
__device__ __forceinline__ bool bj_residual_small(const double *__restrict__ parts2, unsigned np2,
                                                  const lsb_pcg_state *__restrict__ st,
                                                  double *sred) {
  double v[2] = {0.0, 0.0};
  for (unsigned i = threadIdx.x; i < np2; i += WG)
    v[1] += parts2[2 * (size_t)i + 1];
  pwg_sum2(v, sred);
  return v[1] <= st->thresh2;
}

// z = Binv r, one thread per row, small blocks (bs <= 64): reads bs entries
__global__ __launch_bounds__(WG) void k_bj_apply(unsigned n, unsigned bs,
                                                 const double *__restrict__ binv,
                                                 const double *__restrict__ r,
                                                 double *__restrict__ z,
                                                 const lsb_pcg_state *__restrict__ st) {
  if (st && st->status)
    return;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
    const unsigned k = (unsigned)(i / bs), il = (unsigned)(i % bs);
    const unsigned m = min(bs, n - k * bs);
    const double *blk = binv + (size_t)k * bs * bs;
    const double *rb = r + (size_t)k * bs;
    double s = 0.0;
    for (unsigned j = 0; j < m; j++)
      s += blk[(size_t)j * m + il] * rb[j];
    z[i] = s;
  }
}

// large blocks: the columns of a block are cut into chunks of BJ_CH; workgroup
// (row tile, chunk) leaves partial sums, k_bj_sum adds the chunks in order
#define BJ_CH 128
__global__ __launch_bounds__(WG) void k_bj_apply_part(unsigned n, unsigned bs, unsigned nch,
                                                      const double *__restrict__ binv,
                                                      const double *__restrict__ r,
                                                      double *__restrict__ part,
                                                      const lsb_pcg_state *__restrict__ st,
                                                      const double *__restrict__ skip2,
                                                      unsigned nskip) {
  __shared__ double sred[8];
  if (st && st->status)
    return;
  if (skip2 && bj_residual_small(skip2, nskip, st, sred))
    return;
  const size_t i = (size_t)blockIdx.x * WG + threadIdx.x;
  const unsigned ch = blockIdx.y;
  if (i >= n)
    return;
  const unsigned k = (unsigned)(i / bs), il = (unsigned)(i % bs);
  const unsigned m = min(bs, n - k * bs);
  const double *blk = binv + (size_t)k * bs * bs;
  const double *rb = r + (size_t)k * bs;
  const unsigned j0 = ch * BJ_CH, j1 = min(j0 + BJ_CH, m);
  double s = 0.0;
  for (unsigned j = j0; j < j1; j++)
    s += blk[(size_t)j * m + il] * rb[j];
  part[(size_t)ch * n + i] = s;
}

__global__ __launch_bounds__(WG) void k_bj_sum(unsigned n, unsigned nch,
                                               const double *__restrict__ part,
                                               double *__restrict__ z,
                                               const lsb_pcg_state *__restrict__ st) {
  if (st && st->status)
    return;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
    double s = 0.0;
    for (unsigned c = 0; c < nch; c++)
      s += part[(size_t)c * n + i];
    z[i] = s;
  }
}

extern "C" {

static unsigned pgrid(size_t n) {
  size_t g = (n + WG * 4 - 1) / (WG * 4);
  return (unsigned)(g < 1 ? 1 : (g > LSB_STREAM_GRID_CAP ? LSB_STREAM_GRID_CAP : g));
}

void lsb_k_dot2(unsigned n, const double *r, const double *z, double *partials2,
                unsigned *npartials, const struct lsb_pcg_state *st, void *stream) {
  unsigned g = pgrid(n);
  if (g > LSB_MAX_PARTIALS)
    g = LSB_MAX_PARTIALS;
  *npartials = g;
  k_dot2<<<g, WG, 0, (hipStream_t)stream>>>(n, r, z, partials2, st);
}

void lsb_k_cheb_first(unsigned n, const double *r, const double *dinv, double dc, double c0,
                      double *d, double *z, const struct lsb_pcg_state *st, void *stream) {
  k_cheb_first<<<pgrid(n), WG, 0, (hipStream_t)stream>>>(n, r, dinv, dc, c0, d, z, st);
}

void lsb_k_cheb_step(unsigned n, const double *r, const double *w, const double *dinv, double dc,
                     double a, double b, double *d, double *z, const struct lsb_pcg_state *st,
                     void *stream) {
  k_cheb_step<<<pgrid(n), WG, 0, (hipStream_t)stream>>>(n, r, w, dinv, dc, a, b, d, z, st);
}

void lsb_k_scale_dinv(unsigned n, double c, const double *dinv, const double *w, double *v,
                      void *stream) {
  k_scale_dinv<<<pgrid(n), WG, 0, (hipStream_t)stream>>>(n, c, dinv, w, v);
}

void lsb_k_scale_vec(unsigned n, double c, double *v, void *stream) {
  k_scale_vec<<<pgrid(n), WG, 0, (hipStream_t)stream>>>(n, c, v);
}

void lsb_k_power_start(unsigned n, unsigned first, double *v, void *stream) {
  k_power_start<<<pgrid(n), WG, 0, (hipStream_t)stream>>>(n, first, v);
}

/* binv: the dense diagonal blocks (as extracted); on return their inverses.
 * scratch: 2 n doubles (row p and column p of every block). */
void lsb_k_bj_invert(unsigned n, unsigned bs, double *binv, double *scratch, void *stream) {
  const size_t total = (size_t)n * bs;
  size_t g = (total + WG * 4 - 1) / (WG * 4);
  if (g > 4096)
    g = 4096;
  for (unsigned p = 0; p < bs && p < n; p++) {
    k_gj_rowcol<<<pgrid(n), WG, 0, (hipStream_t)stream>>>(n, bs, p, binv, scratch, scratch + n);
    k_gj_update<<<(unsigned)(g ? g : 1), WG, 0, (hipStream_t)stream>>>(n, bs, p, binv, scratch,
                                                                        scratch + n);
  }
}

void lsb_k_bj_apply(unsigned n, unsigned bs, const double *binv, const double *r, double *z,
                    double *part, const struct lsb_pcg_state *st, const double *skip2,
                    unsigned nskip, void *stream) {
  if (bs <= 64 || !part) {
    k_bj_apply<<<pgrid(n), WG, 0, (hipStream_t)stream>>>(n, bs, binv, r, z, st);
    return;
  }
  const unsigned nch = (bs + BJ_CH - 1) / BJ_CH;
  dim3 grid((n + WG - 1) / WG, nch);
  k_bj_apply_part<<<grid, WG, 0, (hipStream_t)stream>>>(n, bs, nch, binv, r, part, st,
                                                        st ? skip2 : NULL, nskip);
  k_bj_sum<<<pgrid(n), WG, 0, (hipStream_t)stream>>>(n, nch, part, z, st);
}

unsigned lsb_k_bj_chunks(unsigned bs) { return bs <= 64 ? 0 : (bs + BJ_CH - 1) / BJ_CH; }

} // extern "C"
