// Restarted GMRES(m) with right Jacobi preconditioning -- SURVEY.md section 8
// (a2-6): the Krylov method for operators that are not symmetric (the raw file
// matrix, LSB_OP_RAW; the reference's only in-tree Krylov call site is
// non-symmetric too: BiCGSTAB+Jacobi, src/ginkgo.cpp:55-64).  No reference
// source exists for the arithmetic; the oracle restates the same algorithm
// (oracle/lsb_oracle.c orc_gmres_jacobi).
//
// Arnoldi by classical Gram-Schmidt applied twice (CGS2): the j+1 inner
// products of a step are ONE sweep over the basis (k_gm_multidot), not j+1
// dependent reductions as in modified Gram-Schmidt -- the shape a GPU wants.
// All scalars (Hessenberg column, Givens rotations, residual estimate, stop
// decision) live in lsb_gmres_state on the device; the host enqueues whole
// restart cycles and polls once per cycle.  Kernels turn into no-ops once
// status != 0 or the cycle is closed.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lsb_impl.h"

#define WG 256
#define GM_MAXV (LSB_GMRES_MAX_RESTART + 1)

template <int W>
__device__ __forceinline__ void gm_wg_sum(double (&v)[W], double *sred /*4*W*/) {
#pragma unroll
  for (int k = 0; k < W; k++) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
      v[k] += __shfl_xor(v[k], off, 64);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < W; k++)
      sred[wave * W + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < W; k++)
    v[k] = (sred[0 * W + k] + sred[1 * W + k]) + (sred[2 * W + k] + sred[3 * W + k]);
}

__device__ __forceinline__ bool gm_idle(const lsb_gmres_state *st) {
  return st->status != LSB_STATUS_RUNNING || !st->cycle_open;
}

// r = b - ax (into v0), partial sum of r^2
__global__ __launch_bounds__(WG) void k_gm_resid(unsigned n, const double *__restrict__ b,
                                                 const double *__restrict__ ax,
                                                 double *__restrict__ v0,
                                                 double *__restrict__ partials,
                                                 const lsb_gmres_state *__restrict__ st) {
  if (st->status != LSB_STATUS_RUNNING)
    return;
  __shared__ double sred[4];
  double acc[1] = {0.0};
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
    const double r = b[i] - ax[i];
    v0[i] = r;
    acc[0] += r * r;
  }
  gm_wg_sum<1>(acc, sred);
  if (threadIdx.x == 0)
    partials[blockIdx.x] = acc[0];
}

// one workgroup: beta = ||r||; open a restart cycle (or finish)
__global__ __launch_bounds__(WG) void k_gm_begin(lsb_gmres_state *__restrict__ st,
                                                 const double *__restrict__ partials,
                                                 unsigned nparts, double tol, int maxit,
                                                 int restart, int first) {
  if (!first && st->status != LSB_STATUS_RUNNING)
    return;
  __shared__ double sred[4];
  double v[1] = {0.0};
  for (unsigned i = threadIdx.x; i < nparts; i += WG)
    v[0] += partials[i];
  gm_wg_sum<1>(v, sred);
  if (threadIdx.x == 0) {
    const double beta = sqrt(v[0]);
    if (first) {
      st->bnorm = beta; // x0 = 0 => r0 = b
      st->thresh = tol * beta;
      st->iters = 0, st->maxit = maxit, st->restart = restart;
      st->status = LSB_STATUS_RUNNING;
    }
    st->beta = beta, st->resid = beta;
    st->jlast = 0;
    for (int i = 0; i < GM_MAXV; i++)
      st->g[i] = 0.0;
    st->g[0] = beta;
    st->hnorm = beta;
    if (beta <= st->thresh || beta == 0.0) {
      st->status = LSB_STATUS_CONVERGED;
      st->cycle_open = 0;
    } else if (st->iters >= st->maxit) {
      st->status = LSB_STATUS_MAXIT;
      st->cycle_open = 0;
    } else {
      st->cycle_open = 1;
    }
  }
}

// v = w / st->hnorm (in place when w == v) and z = dinv .* v (into the gather
// vector the SpMV reads)
__global__ __launch_bounds__(WG) void k_gm_scale_prec(unsigned n, const double *__restrict__ w,
                                                      double *__restrict__ v,
                                                      const double *__restrict__ dinv,
                                                      double *__restrict__ z,
                                                      const lsb_gmres_state *__restrict__ st) {
  if (gm_idle(st))
    return;
  const double s = 1.0 / st->hnorm;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
    const double vi = w[i] * s;
    v[i] = vi;
    z[i] = dinv[i] * vi;
  }
}

// partial h_k = sum_i V[k][i] * w[i], k < cnt, one record of GM_MAXV per block.
// Round 4 (first timed: 0.40 of peak over a GMRES(30) step): 16 bytes per lane and operand (the
// basis columns start on 16-byte boundaries: ld is even), the columns taken four at a time with
// their loads issued together, <= 768 workgroups (three per CU: what streams fastest out of HBM,
// DESIGN.md section 4 "How many workgroups stream a vector").  A lane adds its elements in index
// order, pair after pair -- fixed by n and the grid, so bitwise repeatable run to run.
typedef double gm_d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(WG) void k_gm_multidot(unsigned n, const double *__restrict__ V,
                                                    size_t ld, int cnt,
                                                    const double *__restrict__ w,
                                                    double *__restrict__ partials,
                                                    const lsb_gmres_state *__restrict__ st) {
  if (gm_idle(st))
    return;
  __shared__ double sred[4 * GM_MAXV];
  double acc[GM_MAXV];
#pragma unroll
  for (int k = 0; k < GM_MAXV; k++)
    acc[k] = 0.0;
  const size_t npair = n / 2;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < npair; i += (size_t)gridDim.x * WG) {
    const gm_d2 wi = ((const gm_d2 *)w)[i];
#pragma unroll
    for (int k0 = 0; k0 < GM_MAXV; k0 += 4)
      if (k0 < cnt) { // (wave-uniform)
        gm_d2 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) // (columns past cnt: column cnt - 1 again, result dropped)
          v[u] = ((const gm_d2 *)(V + (size_t)(k0 + u < cnt ? k0 + u : cnt - 1) * ld))[i];
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (k0 + u < GM_MAXV && k0 + u < cnt)
            acc[k0 + u] = fma(v[u].y, wi.y, fma(v[u].x, wi.x, acc[k0 + u]));
      }
  }
  if ((n & 1u) && blockIdx.x == 0 && threadIdx.x == 0) { // the odd last element
    const double wi = w[n - 1];
#pragma unroll
    for (int k = 0; k < GM_MAXV; k++) // (unrolled: the accumulators are registers)
      if (k < cnt)
        acc[k] += V[(size_t)k * ld + (n - 1)] * wi;
  }
  gm_wg_sum<GM_MAXV>(acc, sred);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < GM_MAXV; k++)
      partials[(size_t)blockIdx.x * GM_MAXV + k] = acc[k];
  }
}

// one workgroup: h[k] (+)= sum over blocks of partials[.][k]
__global__ __launch_bounds__(WG) void k_gm_multidot_final(const double *__restrict__ partials,
                                                          unsigned nparts, int cnt,
                                                          double *__restrict__ h, int accumulate,
                                                          const lsb_gmres_state *__restrict__ st) {
  if (gm_idle(st))
    return;
  __shared__ double sred[4];
  for (int k = 0; k < cnt; k++) {
    double v[1] = {0.0};
    for (unsigned i = threadIdx.x; i < nparts; i += WG)
      v[0] += partials[(size_t)i * GM_MAXV + k];
    gm_wg_sum<1>(v, sred);
    if (threadIdx.x == 0)
      h[k] = accumulate ? h[k] + v[0] : v[0];
  }
}

// w -= sum_k h[k] V[k]; partial sum of w^2 after the update (16 bytes per lane and operand, four
// columns' loads in flight together, subtracted in column order)
__global__ __launch_bounds__(WG) void k_gm_update_w(unsigned n, const double *__restrict__ V,
                                                    size_t ld, int cnt,
                                                    const double *__restrict__ h,
                                                    double *__restrict__ w,
                                                    double *__restrict__ partials,
                                                    const lsb_gmres_state *__restrict__ st) {
  if (gm_idle(st))
    return;
  __shared__ double sred[4];
  __shared__ double sh[GM_MAXV + 3];
  if (threadIdx.x < GM_MAXV + 3)
    sh[threadIdx.x] = (int)threadIdx.x < cnt ? h[threadIdx.x] : 0.0;
  __syncthreads();
  double acc[1] = {0.0};
  const size_t npair = n / 2;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < npair; i += (size_t)gridDim.x * WG) {
    gm_d2 wi = ((const gm_d2 *)w)[i];
    for (int k0 = 0; k0 < cnt; k0 += 4) {
      gm_d2 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++)
        v[u] = ((const gm_d2 *)(V + (size_t)(k0 + u < cnt ? k0 + u : cnt - 1) * ld))[i];
#pragma unroll
      for (int u = 0; u < 4; u++) { // (h is 0 past cnt)
        wi.x = fma(-sh[k0 + u], v[u].x, wi.x);
        wi.y = fma(-sh[k0 + u], v[u].y, wi.y);
      }
    }
    ((gm_d2 *)w)[i] = wi;
    acc[0] = fma(wi.y, wi.y, fma(wi.x, wi.x, acc[0]));
  }
  if ((n & 1u) && blockIdx.x == 0 && threadIdx.x == 0) {
    double wi = w[n - 1];
    for (int k = 0; k < cnt; k++)
      wi -= sh[k] * V[(size_t)k * ld + (n - 1)];
    w[n - 1] = wi;
    acc[0] += wi * wi;
  }
  gm_wg_sum<1>(acc, sred);
  if (threadIdx.x == 0)
    partials[blockIdx.x] = acc[0];
}

// one workgroup: finish inner step j -- h = h1 + h2, h[j+1] = ||w||, Givens
// rotations, residual estimate, stop decision
__global__ __launch_bounds__(WG) void k_gm_hess(lsb_gmres_state *__restrict__ st, int j,
                                                const double *__restrict__ h_in,
                                                const double *__restrict__ h2_in,
                                                const double *__restrict__ partials,
                                                unsigned nparts) {
  if (gm_idle(st))
    return;
  __shared__ double sred[4];
  double v[1] = {0.0};
  for (unsigned i = threadIdx.x; i < nparts; i += WG)
    v[0] += partials[i];
  gm_wg_sum<1>(v, sred);
  if (threadIdx.x != 0)
    return;
  double h[GM_MAXV + 1];
  for (int i = 0; i <= j; i++)
    h[i] = h_in[i] + h2_in[i]; // CGS2: both passes' coefficients
  const double hn = sqrt(v[0]);
  h[j + 1] = hn;
  st->hnorm = hn;
  for (int i = 0; i < j; i++) { // earlier rotations on the new column
    const double t = st->cs[i] * h[i] + st->sn[i] * h[i + 1];
    h[i + 1] = -st->sn[i] * h[i] + st->cs[i] * h[i + 1];
    h[i] = t;
  }
  const double a = h[j], b = h[j + 1], d = sqrt(a * a + b * b);
  double c = 1.0, s = 0.0;
  if (d != 0.0)
    c = a / d, s = b / d;
  st->cs[j] = c, st->sn[j] = s;
  h[j] = d;
  for (int i = 0; i <= j; i++)
    st->R[i * LSB_GMRES_MAX_RESTART + j] = h[i];
  st->g[j + 1] = -s * st->g[j];
  st->g[j] = c * st->g[j];
  const double resid = fabs(st->g[j + 1]);
  st->resid = resid;
  st->jlast = j + 1;
  const int it = st->iters + 1;
  st->iters = it;
  if (d == 0.0)
    st->status = LSB_STATUS_BREAKDOWN;
  else if (resid <= st->thresh || hn == 0.0) // hn == 0: the Krylov space is exhausted
    st->status = LSB_STATUS_CONVERGED;
  else if (it >= st->maxit)
    st->status = LSB_STATUS_MAXIT;
}

// one thread: y = R^-1 g for the jlast columns of this cycle
__global__ void k_gm_solve_y(lsb_gmres_state *__restrict__ st) {
  if (!st->cycle_open || threadIdx.x != 0 || blockIdx.x != 0)
    return;
  const int m = st->jlast;
  for (int i = m - 1; i >= 0; i--) {
    double s = st->g[i];
    for (int k = i + 1; k < m; k++)
      s -= st->R[i * LSB_GMRES_MAX_RESTART + k] * st->y[k];
    const double d = st->R[i * LSB_GMRES_MAX_RESTART + i];
    st->y[i] = d != 0.0 ? s / d : 0.0;
  }
}

// x += dinv .* sum_k y[k] V[k]
__global__ __launch_bounds__(WG) void k_gm_update_x(unsigned n, const double *__restrict__ V,
                                                    size_t ld, const double *__restrict__ dinv,
                                                    double *__restrict__ x,
                                                    const lsb_gmres_state *__restrict__ st) {
  if (!st->cycle_open)
    return;
  __shared__ double sy[GM_MAXV];
  const int m = st->jlast;
  if (threadIdx.x < GM_MAXV)
    sy[threadIdx.x] = (int)threadIdx.x < m ? st->y[threadIdx.x] : 0.0;
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
    double s = 0.0;
    for (int k = 0; k < m; k++)
      s += sy[k] * V[(size_t)k * ld + i];
    x[i] += dinv[i] * s;
  }
}

__global__ void k_gm_end_cycle(lsb_gmres_state *__restrict__ st) {
  if (threadIdx.x == 0 && blockIdx.x == 0)
    st->cycle_open = 0;
}

static unsigned gm_grid(unsigned n) {
  unsigned g = (n + WG * 4 - 1) / (WG * 4);
  if (g > LSB_GMRES_PARTIALS)
    g = LSB_GMRES_PARTIALS;
  if (g > 768u) /* three workgroups per CU stream vectors fastest out of HBM (LSB_STREAM_GRID_CAP) */
    g = 768u;
  return g ? g : 1;
}

extern "C" {

unsigned lsb_k_gm_grid(unsigned n) { return gm_grid(n); }

void lsb_k_gm_resid(unsigned n, const double *b, const double *ax, double *v0,
                    double *partials, const struct lsb_gmres_state *st, void *stream) {
  k_gm_resid<<<gm_grid(n), WG, 0, (hipStream_t)stream>>>(n, b, ax, v0, partials, st);
}

void lsb_k_gm_begin(struct lsb_gmres_state *st, const double *partials, unsigned nparts,
                    double tol, int maxit, int restart, int first, void *stream) {
  k_gm_begin<<<1, WG, 0, (hipStream_t)stream>>>(st, partials, nparts, tol, maxit, restart, first);
}

void lsb_k_gm_scale_prec(unsigned n, const double *w, double *v, const double *dinv, double *z,
                         const struct lsb_gmres_state *st, void *stream) {
  k_gm_scale_prec<<<gm_grid(n), WG, 0, (hipStream_t)stream>>>(n, w, v, dinv, z, st);
}

void lsb_k_gm_multidot(unsigned n, const double *V, size_t ld, int cnt, const double *w,
                       double *partials, double *h, int accumulate,
                       const struct lsb_gmres_state *st, void *stream) {
  const unsigned g = gm_grid(n);
  k_gm_multidot<<<g, WG, 0, (hipStream_t)stream>>>(n, V, ld, cnt, w, partials, st);
  k_gm_multidot_final<<<1, WG, 0, (hipStream_t)stream>>>(partials, g, cnt, h, accumulate, st);
}

void lsb_k_gm_update_w(unsigned n, const double *V, size_t ld, int cnt, const double *h,
                       double *w, double *partials, const struct lsb_gmres_state *st,
                       void *stream) {
  k_gm_update_w<<<gm_grid(n), WG, 0, (hipStream_t)stream>>>(n, V, ld, cnt, h, w, partials, st);
}

void lsb_k_gm_hess(struct lsb_gmres_state *st, int j, const double *h, const double *h2,
                   const double *partials, unsigned nparts, void *stream) {
  k_gm_hess<<<1, WG, 0, (hipStream_t)stream>>>(st, j, h, h2, partials, nparts);
}

void lsb_k_gm_finish_cycle(unsigned n, const double *V, size_t ld, const double *dinv, double *x,
                           struct lsb_gmres_state *st, void *stream) {
  k_gm_solve_y<<<1, 64, 0, (hipStream_t)stream>>>(st);
  k_gm_update_x<<<gm_grid(n), WG, 0, (hipStream_t)stream>>>(n, V, ld, dinv, x, st);
  k_gm_end_cycle<<<1, 64, 0, (hipStream_t)stream>>>(st);
}

} // extern "C"
