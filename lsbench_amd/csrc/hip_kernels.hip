// gfx950 (MI355X, CDNA4) kernels of the lsbench HIP backend + their C-ABI
// launchers ("the shim").  Host code (hip_cdna4.c, plain C) never sees a
// kernel symbol, only the extern "C" lsb_k_* functions at the bottom.
//
// None of this arithmetic exists in the reference's source: every reference
// backend hands its CSR to a third-party library; the
// nearest call site is the Krylov+Jacobi apply of src/ginkgo.cpp:55-69,91-99.
// Kernel inventory = SURVEY.md section 8 (a2):
//   a2-1  spmv (adaptive row-blocked / sub-wavefront / scalar), fused p.q
//   a2-2  dot, nrm2           two-stage, fixed-order => run-to-run identical
//   a2-3  axpy, xpay          scalars read from HBM, no host sync
//   a2-4  jacobi setup/apply/sweep
//   a2-5  fused PCG sweeps    (x,r update + r.z + r.r) and (p update)
//
// Everything here is HBM-bandwidth bound (0.125 flop/B for a 5-point row):
// 64-wide wavefronts, coalesced 8-16 B/lane streams, LDS staging of the
// per-non-zero products, XCD-contiguous work assignment so that the x-vector
// window a row block gathers from is re-used out of one XCD's L2.  No MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "hip_ar.h"
#include "lsb_impl.h"

#include "hip_wg.h"
static_assert(WG == AR_WG, "the folded all-reduce phases assume this workgroup size");

// End of an SpMV launch with the fused dot: the workgroup's partial sum goes to
// partials[blockIdx.x]; with a tail (sharded solve over the direct xGMI path)
// the workgroup that hands in last also runs the all-reduce's contribute phase.
__device__ __forceinline__ void spmv_publish(double *__restrict__ partials, double dot,
                                             double *sred, const lsb_ar_tail &tail) {
  if (!partials)
    return;
  double d[1] = {dot};
  wg_sum<1>(d, sred);
  if (tail.counter)
    ar_tail(partials, d[0], tail);
  else if (threadIdx.x == 0)
    partials[blockIdx.x] = d[0];
}

// --------------------------------------------------------------------------
// a2-1  SpMV, adaptive row blocks.
// rowblk[k]..rowblk[k+1] is a run of consecutive rows holding <= CAP
// non-zeros, or a single row longer than CAP (lsb_csr_row_blocks).
//   * short-row block: all 256 lanes stream the block's cols/vals (fully
//     coalesced, CAP/256 loads in flight per lane), gather x, park the products
//     in LDS; then L = blklanes[k] (1..64) lanes per row add a row's products
//     and finish with a wavefront shuffle tree;
//   * long row: the whole workgroup strides over it from global memory.
// Work assignment: the row blocks are cut into 8 contiguous chunks, one per
// XCD, and inside a chunk dealt CYCLICALLY to that XCD's workgroups.  The
// G/8 workgroups resident on one XCD therefore sweep G/8 ADJACENT blocks at any
// time and the window of x they gather from (a few grid lines of a stencil)
// fits that XCD's 4 MiB L2: measured 907 -> 732 MB fetched per launch on the
// 10M-row 5-point operator (720 MB is the minimum).  A contiguous range per
// workgroup keeps 256 far-apart windows alive per XCD and thrashes it.
// SP_PREFETCH: the next block's stream loads are issued before the current
// block is reduced.  SP_NT: the once-read cols/vals stream is loaded
// nontemporal so it does not evict x from L2.  Both are picked per operator by
// a timing pass at solver creation (hip_cdna4.c).
// --------------------------------------------------------------------------
enum { SP_PREFETCH = 1, SP_NT = 2 };

template <int FLAGS, class T>
__device__ __forceinline__ T stream_load(const T *p) {
  if (FLAGS & SP_NT)
    return __builtin_nontemporal_load(p);
  return *p;
}

#define LSB_ISSUE_BLOCK(kk)                                                    \
  do {                                                                         \
    r0 = rowblk[kk], r1 = rowblk[(kk) + 1];                                    \
    j0 = offs[r0], j1 = offs[r1];                                              \
    if (j1 - j0 <= CAP) {                                                      \
      _Pragma("unroll") for (int u = 0; u < U; u++) {                          \
        const int t = (int)tid + u * WG;                                       \
        if (t < j1 - j0) {                                                     \
          c[u] = stream_load<FLAGS>(cols + j0 + t);                            \
          v[u] = stream_load<FLAGS>(vals + j0 + t);                            \
        }                                                                      \
      }                                                                        \
    }                                                                          \
  } while (0)

// VT = double, or float: opts.precision = LSB_PREC_MIXED streams the matrix
// values as fp32 (4 B instead of 8 B per entry); products and sums stay fp64.
template <int CAP, int FLAGS, class VT = double>
__global__ __launch_bounds__(WG, 8) void k_spmv_adaptive(
    const int *__restrict__ rowblk, const unsigned char *__restrict__ blklanes,
    unsigned nblk, const int *__restrict__ offs, const int *__restrict__ cols,
    const VT *__restrict__ vals, const double *__restrict__ x,
    double *__restrict__ y, const double *__restrict__ xdot,
    double *__restrict__ partials, const lsb_pcg_state *__restrict__ st,
    const int *__restrict__ rowmap, const lsb_ar_tail tail) {
  // rowmap != NULL: column-panel mode -- the CSR's rows are (row, panel) pairs,
  // row r of it accumulates into y[rowmap[r]] (one launch per panel, so a y
  // entry is touched by one lane per launch; lsb_csr_panelize)
  __shared__ double sprod[CAP];
  __shared__ double sred[4];
  const unsigned tid = threadIdx.x;
  const unsigned gx = gridDim.x / NXCD; // workgroups per XCD
  const unsigned xcd = blockIdx.x % NXCD, slot = blockIdx.x / NXCD;
  const unsigned chunk = (nblk + NXCD - 1) / NXCD; // row blocks per XCD
  const unsigned kbeg = xcd * chunk, kend = min(kbeg + chunk, nblk);
  constexpr int U = CAP / WG;
  double dot = 0.0;
  int c[U];
  VT v[U];
  int r0 = 0, r1 = 0, j0 = 0, j1 = 0;
  unsigned k = kbeg + slot;
  if (k < kend)
    LSB_ISSUE_BLOCK(k);
  // the "solve already finished" test rides behind the first block's loads
  // instead of costing a memory round trip before them
  if (st && st->status)
    return;
  for (; k < kend; k += gx) {
    const int cr0 = r0, cj0 = j0, cnt = j1 - j0, nr = r1 - r0;
    if (cnt <= CAP) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int t = (int)tid + u * WG;
        if (t < cnt)
          sprod[t] = (double)v[u] * x[c[u]];
      }
      if ((FLAGS & SP_PREFETCH) && k + gx < kend)
        LSB_ISSUE_BLOCK(k + gx);
      __syncthreads();
      // ---- L lanes per row add the products up -----------------------
      unsigned L;
      if (blklanes) {
        L = blklanes[k];
      } else {
        L = 1;
        while (L < 64 && (unsigned)nr * (L * 2) <= WG)
          L <<= 1;
      }
      const unsigned sl = tid / L, l = tid % L, slots = WG / L;
      for (unsigned rb = 0; rb < (unsigned)nr; rb += slots) {
        const unsigned r = rb + sl;
        double s = 0.0;
        if (r < (unsigned)nr) {
          const int a = offs[cr0 + r] - cj0, b = offs[cr0 + r + 1] - cj0;
          for (int j = a + (int)l; j < b; j += (int)L)
            s += sprod[j];
        }
        for (unsigned off = L >> 1; off > 0; off >>= 1)
          s += __shfl_xor(s, off, 64);
        if (r < (unsigned)nr && l == 0) {
          if (rowmap) {
            y[rowmap[cr0 + r]] += s;
          } else {
            y[cr0 + r] = s;
            if (xdot)
              dot += s * xdot[cr0 + r];
          }
        }
      }
      __syncthreads(); // sprod is overwritten by the next block
    } else {
      // ---- one long row (nr == 1): the whole workgroup strides over it.
      // Kept deliberately simple: an unrolled version of this rare path
      // pushed the whole kernel over 64 VGPRs into scratch and cost the
      // common path 70 % (tests/test_build_resources.py guards that).
      double s[1] = {0.0};
      for (int j = cj0 + (int)tid; j < cj0 + cnt; j += WG)
        s[0] += (double)vals[j] * x[cols[j]];
      wg_sum<1>(s, sred);
      if (tid == 0) {
        if (rowmap) {
          y[rowmap[cr0]] += s[0];
        } else {
          y[cr0] = s[0];
          if (xdot)
            dot += s[0] * xdot[cr0];
        }
      }
      if ((FLAGS & SP_PREFETCH) && k + gx < kend)
        LSB_ISSUE_BLOCK(k + gx);
    }
    if (!(FLAGS & SP_PREFETCH) && k + gx < kend)
      LSB_ISSUE_BLOCK(k + gx);
  }
  spmv_publish(partials, dot, sred, tail);
}
#undef LSB_ISSUE_BLOCK

// (Round 4, measured and taken out: k_spmv_rows -- one row per lane straight off the CSR arrays for
// operators of short rows, eight entries per trip on clamped indices, tiles of 256 rows dealt like row
// blocks; the idea: no LDS round trip, no barriers, neighbouring lanes read neighbouring 40-60 byte runs
// and re-use their lines out of L1.  10 M-row 5-point pattern with general values: 223 us (378 us with
// nontemporal stream loads, which forgo exactly that re-use) against 160 us for k_spmv_adaptive in the
// same timing pass -- a wave instruction that touches 20 lines costs the L1 more than the barriers cost
// the row-blocked form.  The CSR figure of the bench line stays k_spmv_adaptive's.)
// --------------------------------------------------------------------------
// a2-1  SpMV, L lanes per row (L = 2..64; L = 64 is the classic
// one-wavefront-per-row kernel), rows dealt to workgroups in contiguous,
// XCD-contiguous chunks; shuffle (DPP) reduction of the L partial sums.
// --------------------------------------------------------------------------
template <int L, class VT = double>
__global__ __launch_bounds__(WG) void k_spmv_subwave(
    unsigned n, unsigned rows_per_wg, const int *__restrict__ offs,
    const int *__restrict__ cols, const VT *__restrict__ vals,
    const double *__restrict__ x, double *__restrict__ y,
    const double *__restrict__ xdot, double *__restrict__ partials,
    const lsb_pcg_state *__restrict__ st) {
  __shared__ double sred[4];
  const unsigned tid = threadIdx.x, slot = tid / L, l = tid % L;
  constexpr unsigned SLOTS = WG / L;
  const unsigned w = xcd_contiguous_wg();
  const unsigned ra = min(w * rows_per_wg, n), rb = min(ra + rows_per_wg, n);
  // the status word travels with the first row's loads (this kernel serves
  // the small, latency-bound operators); nothing is stored before it is tested
  const int stopped = st ? st->status : 0;
  double dot = 0.0;
  for (unsigned base = ra; base < rb; base += SLOTS) {
    const unsigned r = base + slot;
    double s = 0.0, xd = 0.0;
    if (r < rb) {
      const int j0 = offs[r], j1 = offs[r + 1];
      if (xdot)
        xd = xdot[r];
      for (int j = j0 + (int)l; j < j1; j += L)
        s += (double)vals[j] * x[cols[j]];
    }
#pragma unroll
    for (int off = L >> 1; off > 0; off >>= 1)
      s += __shfl_xor(s, off, 64);
    if (stopped)
      return;
    if (r < rb && l == 0) {
      y[r] = s;
      dot += s * xd;
    }
  }
  if (stopped)
    return;
  if (partials) {
    double d[1] = {dot};
    wg_sum<1>(d, sred);
    if (tid == 0)
      partials[w] = d[0];
  }
}

// The launch-bound operators' iteration with the direction update folded into
// the SpMV: TWO launches per classic PCG iteration instead of three.  The
// prologue is k_pcg_update_p's (partial sums of the previous sweep -> r.z, r.r,
// stop test, beta); then q = S p_new with p_new = D^-1 r + beta p_old formed on
// the fly for every gathered column (three gathers of vectors that sit in L2)
// and stored, for the rows of this workgroup, into the OTHER direction buffer.
// At a few thousand rows a launch costs more than all of that.
__device__ __forceinline__ double pnew_of(double d, double r, double beta, double p) {
  return __fma_rn(beta, p, d * r);
}
template <int L>
__global__ __launch_bounds__(WG) void k_spmv_subwave_p(
    unsigned n, unsigned rows_per_wg, const int *__restrict__ offs,
    const int *__restrict__ cols, const double *__restrict__ vals,
    const double *__restrict__ r, const double *__restrict__ dinv, double dc,
    const double *__restrict__ pold, double *__restrict__ pnew, double *__restrict__ y,
    double *__restrict__ partials, lsb_pcg_state *__restrict__ st, int parity,
    const double *__restrict__ parts2, unsigned nparts2) {
  __shared__ double sred[8];
  const unsigned tid = threadIdx.x, slot = tid / L, l = tid % L;
  constexpr unsigned SLOTS = WG / L;
  const unsigned w = xcd_contiguous_wg();
  const unsigned ra = min(w * rows_per_wg, n), rb = min(ra + rows_per_wg, n);
  const int stopped = st->status;
  const double rz_old = st->rz[parity], thresh2 = st->thresh2;
  double v[2];
  wg_sum_partials<2>(parts2, nparts2, v, sred);
  if (stopped)
    return;
  const double rz_new = v[0], rr = v[1];
  const bool conv = rr <= thresh2;
  if (blockIdx.x == 0 && threadIdx.x == 0) { // exactly k_pcg_update_p's bookkeeping
    const int it = st->iters + 1;
    st->iters = it;
    st->rr = rr;
    st->rz[parity ^ 1] = rz_new;
    if (conv)
      st->status = LSB_STATUS_CONVERGED;
    else if (it >= st->maxit)
      st->status = LSB_STATUS_MAXIT;
  }
  if (conv)
    return;
  const double beta = rz_new / rz_old;
  double dot = 0.0;
  for (unsigned base = ra; base < rb; base += SLOTS) {
    const unsigned row = base + slot;
    double s = 0.0, pi = 0.0;
    if (row < rb) {
      const int j0 = offs[row], j1 = offs[row + 1];
      pi = pnew_of(dinv ? dinv[row] : dc, r[row], beta, pold[row]);
      for (int j = j0 + (int)l; j < j1; j += L) {
        const int c = cols[j];
        s += vals[j] * pnew_of(dinv ? dinv[c] : dc, r[c], beta, pold[c]);
      }
    }
#pragma unroll
    for (int off = L >> 1; off > 0; off >>= 1)
      s += __shfl_xor(s, off, 64);
    if (row < rb && l == 0) {
      pnew[row] = pi;
      y[row] = s;
      dot += s * pi;
    }
  }
  double d[1] = {dot};
  wg_sum<1>(d, sred);
  if (tid == 0)
    partials[w] = d[0];
}

// one lane per row: debug / test baseline only
__global__ __launch_bounds__(WG) void k_spmv_scalar(
    unsigned n, unsigned rows_per_wg, const int *__restrict__ offs,
    const int *__restrict__ cols, const double *__restrict__ vals,
    const double *__restrict__ x, double *__restrict__ y,
    const double *__restrict__ xdot, double *__restrict__ partials,
    const lsb_pcg_state *__restrict__ st) {
  if (st && st->status)
    return;
  __shared__ double sred[4];
  const unsigned w = xcd_contiguous_wg();
  const unsigned ra = min(w * rows_per_wg, n), rb = min(ra + rows_per_wg, n);
  double dot = 0.0;
  for (unsigned r = ra + threadIdx.x; r < rb; r += WG) {
    double s = 0.0;
    for (int j = offs[r]; j < offs[r + 1]; j++)
      s += vals[j] * x[cols[j]];
    y[r] = s;
    if (xdot)
      dot += s * xdot[r];
  }
  if (partials) {
    double d[1] = {dot};
    wg_sum<1>(d, sred);
    if (threadIdx.x == 0)
      partials[w] = d[0];
  }
}

// --------------------------------------------------------------------------
// a2-2  second stage of every reduction: one workgroup, fixed order.
// out[k] = sum over records of parts[i*width+k]   (sqrt'ed for nrm2)
// --------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void k_reduce_final(
    const double *__restrict__ parts, unsigned nparts, unsigned width,
    double *__restrict__ out, int take_sqrt,
    const lsb_pcg_state *__restrict__ st) {
  if (st && st->status)
    return;
  __shared__ double sred[4];
  for (unsigned k = 0; k < width; k++) {
    double v[1] = {0.0};
    for (unsigned i = threadIdx.x; i < nparts; i += WG)
      v[0] += parts[(size_t)i * width + k];
    wg_sum<1>(v, sred);
    if (threadIdx.x == 0)
      out[k] = take_sqrt ? sqrt(v[0]) : v[0];
  }
}

// two of those in one launch (the RCCL path of the single-reduction iteration
// needs the SpMV's and the sweep's partial sums reduced before ONE all-reduce)
__global__ __launch_bounds__(WG) void k_reduce_final2(
    const double *__restrict__ pa, unsigned na, unsigned wa, double *__restrict__ outa,
    const double *__restrict__ pb, unsigned nb, unsigned wb, double *__restrict__ outb,
    const lsb_pcg_state *__restrict__ st) {
  if (st && st->status)
    return;
  __shared__ double sred[4];
  for (unsigned k = 0; k < wa + wb; k++) {
    const bool a = k < wa;
    const double *p = a ? pa : pb;
    const unsigned n = a ? na : nb, w = a ? wa : wb, c = a ? k : k - wa;
    double v[1] = {0.0};
    for (unsigned i = threadIdx.x; i < n; i += WG)
      v[0] += p[(size_t)i * w + c];
    wg_sum<1>(v, sred);
    if (threadIdx.x == 0)
      (a ? outa : outb)[c] = v[0];
  }
}

// first stage of dot / nrm2 (b == a gives sum a_i^2)
__global__ __launch_bounds__(WG) void k_dot(unsigned n,
                                            const double *__restrict__ a,
                                            const double *__restrict__ b,
                                            double *__restrict__ partials) {
  __shared__ double sred[4];
  double v[1] = {0.0};
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n;
       i += (size_t)gridDim.x * WG)
    v[0] += a[i] * b[i];
  wg_sum<1>(v, sred);
  if (threadIdx.x == 0)
    partials[blockIdx.x] = v[0];
}

// --------------------------------------------------------------------------
// a2-3  axpy / xpay with the scalar in HBM
// --------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void k_axpy(unsigned n,
                                             const double *__restrict__ alpha,
                                             const double *__restrict__ x,
                                             double *__restrict__ y) {
  const double a = alpha[0];
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n;
       i += (size_t)gridDim.x * WG)
    y[i] += a * x[i];
}

__global__ __launch_bounds__(WG) void k_xpay(unsigned n,
                                             const double *__restrict__ beta,
                                             const double *__restrict__ x,
                                             double *__restrict__ y) {
  const double b = beta[0];
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n;
       i += (size_t)gridDim.x * WG)
    y[i] = x[i] + b * y[i];
}

// --------------------------------------------------------------------------
// a2-4  Jacobi
// --------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void k_jacobi_setup(
    unsigned n, unsigned row_begin, const int *__restrict__ offs,
    const int *__restrict__ cols, const double *__restrict__ vals,
    double *__restrict__ dinv, int *__restrict__ nzero) {
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n;
       i += (size_t)gridDim.x * WG) {
    const int want = (int)(i + row_begin);
    double d = 0.0;
    for (int j = offs[i]; j < offs[i + 1]; j++)
      if (cols[j] == want)
        d = vals[j];
    if (d != 0.0) {
      dinv[i] = 1.0 / d;
    } else {
      dinv[i] = 0.0;
      atomicAdd(nzero, 1);
    }
  }
}

// l1-Jacobi: dinv[i] = 1 / sum_j |S_ij| (the whole row, also its entries in
// other shards' columns: independent of the partition)
__global__ __launch_bounds__(WG) void k_l1_setup(unsigned n, const int *__restrict__ offs,
                                                 const double *__restrict__ vals,
                                                 double *__restrict__ dinv,
                                                 int *__restrict__ nzero) {
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n;
       i += (size_t)gridDim.x * WG) {
    double d = 0.0;
    for (int j = offs[i]; j < offs[i + 1]; j++)
      d += fabs(vals[j]);
    if (d != 0.0) {
      dinv[i] = 1.0 / d;
    } else {
      dinv[i] = 0.0;
      atomicAdd(nzero, 1);
    }
  }
}

__global__ __launch_bounds__(WG) void k_jacobi_apply(
    unsigned n, const double *__restrict__ dinv, const double *__restrict__ r,
    double *__restrict__ z) {
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n;
       i += (size_t)gridDim.x * WG)
    z[i] = dinv[i] * r[i];
}

// x <- x + w * dinv .* (b - ax)      (ax = Op x from a preceding SpMV)
__global__ __launch_bounds__(WG) void k_jacobi_sweep(
    unsigned n, double w, const double *__restrict__ dinv,
    const double *__restrict__ b, const double *__restrict__ ax,
    double *__restrict__ x) {
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n;
       i += (size_t)gridDim.x * WG)
    x[i] += w * dinv[i] * (b[i] - ax[i]);
}

// dst[i] = src[perm[i]]  /  dst[perm[i]] = src[i]   (reordering, perm[new] = old)
__global__ __launch_bounds__(WG) void k_perm_gather(unsigned n, const int *__restrict__ perm,
                                                    const double *__restrict__ src,
                                                    double *__restrict__ dst) {
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
    const int j = perm[i]; // -1: a pad row of a line-padded grid (lsb_csr_pad_lines)
    dst[i] = j >= 0 ? src[j] : 0.0;
  }
}

__global__ __launch_bounds__(WG) void k_perm_scatter(unsigned n, const int *__restrict__ perm,
                                                     const double *__restrict__ src,
                                                     double *__restrict__ dst) {
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG) {
    const int j = perm[i];
    if (j >= 0)
      dst[j] = src[i];
  }
}

__global__ __launch_bounds__(WG) void k_fill_index(unsigned n, unsigned first,
                                                   double *__restrict__ v) {
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n;
       i += (size_t)gridDim.x * WG)
    v[i] = (double)(i + first);
}

// --------------------------------------------------------------------------
// a2-5  fused PCG sweeps.  Vector loads are 16 B/lane (double2) when every
// operand is 16-B aligned, 8 B/lane otherwise (a shard that starts on an odd
// row); the tail element, if any, is handled by the last thread.
// --------------------------------------------------------------------------

// x = 0, r = b, p = dinv.*b ; partials (r.z, b.b)
template <bool V2>
__global__ __launch_bounds__(WG) void k_pcg_init(
    unsigned n, const double *__restrict__ b, const double *__restrict__ dinv, double dc,
    double *__restrict__ x, double *__restrict__ r, double *__restrict__ p,
    double *__restrict__ partials2) {
  __shared__ double sred[8];
  double acc[2] = {0.0, 0.0};
  const size_t gtid = (size_t)blockIdx.x * WG + threadIdx.x;
  const size_t gsz = (size_t)gridDim.x * WG;
  if (V2) {
    const size_t n2 = n / 2;
    const double2 *b2 = (const double2 *)b, *d2 = (const double2 *)dinv;
    double2 *x2 = (double2 *)x, *r2 = (double2 *)r, *p2 = (double2 *)p;
    for (size_t i = gtid; i < n2; i += gsz) {
      const double2 bv = b2[i], dv = dinv ? d2[i] : double2{dc, dc};
      double2 pv;
      pv.x = dv.x * bv.x, pv.y = dv.y * bv.y;
      x2[i] = make_double2(0.0, 0.0);
      r2[i] = bv;
      p2[i] = pv;
      acc[0] += bv.x * pv.x;
      acc[0] += bv.y * pv.y;
      acc[1] += bv.x * bv.x;
      acc[1] += bv.y * bv.y;
    }
    if ((n & 1) && gtid == gsz - 1) {
      const size_t i = n - 1;
      const double bv = b[i], pv = (dinv ? dinv[i] : dc) * bv;
      x[i] = 0.0, r[i] = bv, p[i] = pv;
      acc[0] += bv * pv, acc[1] += bv * bv;
    }
  } else {
    for (size_t i = gtid; i < n; i += gsz) {
      const double bv = b[i], pv = (dinv ? dinv[i] : dc) * bv;
      x[i] = 0.0, r[i] = bv, p[i] = pv;
      acc[0] += bv * pv, acc[1] += bv * bv;
    }
  }
  wg_sum<2>(acc, sred);
  if (threadIdx.x == 0) {
    partials2[2 * blockIdx.x + 0] = acc[0];
    partials2[2 * blockIdx.x + 1] = acc[1];
  }
}

__global__ __launch_bounds__(WG) void k_pcg_init_state(
    lsb_pcg_state *__restrict__ st, const double *__restrict__ partials2,
    unsigned nparts, double tol, int maxit) {
  __shared__ double sred[8];
  double v[2];
  wg_sum_partials<2>(partials2, nparts, v, sred);
  if (threadIdx.x == 0) {
    st->rz[0] = v[0];
    st->rz[1] = 0.0;
    st->alpha[0] = st->alpha[1] = 0.0; // "no previous step" marker of k_cg1_update
    st->bb = v[1];
    st->thresh2 = tol * tol * v[1];
    st->rr = v[1];
    st->pq = 0.0;
    st->iters = 0;
    st->maxit = maxit;
    st->pad = 0; // "maxit-th update done, status pending" marker of k_cg1_update / k_pcg_col_px
    st->xpend = 0; // k_pcg_col_px: no x update pending
    // b == 0 => x = 0 is the solution; maxit == 0 => nothing to do
    st->status = (v[1] == 0.0) ? LSB_STATUS_CONVERGED
                               : (maxit <= 0 ? LSB_STATUS_MAXIT : LSB_STATUS_RUNNING);
  }
}

// 16-byte lane loads of the BLAS-1 sweeps.  NT = nontemporal: on MI355X a
// plain read-only stream tops out near 4.6-4.8 TB/s while the same loop with
// nontemporal loads reads 6.1-6.2 TB/s (tools/spmv_lab.hip, "read-only" probes);
// a 5-in/2-out sweep shaped like k_pcg_update_xr gains 27 %.
typedef double d2v __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ d2v ld2(const d2v *p) {
  if (NT)
    return __builtin_nontemporal_load(p);
  return *p;
}
// Stores of vectors nobody reads before the NEXT sweep (x; in the single-
// reduction form also p, s, r): nontemporal, so that they do not sit as dirty
// lines in L2 / Infinity Cache while the SpMV that follows streams the matrix.
// Jacobi diagonal: a vector, or -- d2 == nullptr -- ONE value for every row (an
// operator with a constant diagonal: the preconditioner is a scaling and its
// vector need not be read; same arithmetic, the factor comes from a register).
template <bool NT>
__device__ __forceinline__ d2v ldd(const d2v *d2, size_t i, double dc) {
  if (!d2)
    return d2v{dc, dc};
  return ld2<NT>(d2 + i);
}
template <bool NT>
__device__ __forceinline__ void st2(d2v *p, d2v v) {
  if (NT)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}

// alpha = rz/pq ; x += alpha p ; r -= alpha q ; partials (r.dinv.r, r.r)
// NTX / NTPQ / NTR: which operands are loaded nontemporal -- x (also stored so), p and q (and the
// Jacobi diagonal), r.  Which of them should bypass the caches is a matter of what the NEXT launches
// read again (LSBENCH_HIP_BLAS1_NT is the mask: bit 0 x, 1 p and q, 2 r here; 3 r, 4 p in
// k_pcg_update_p; 1 = all of them, the setting measured in rounds 1 and 2).
template <bool V2, bool NTX, bool NTPQ, bool NTR>
__global__ __launch_bounds__(WG) void k_pcg_update_xr(
    unsigned n, const double *__restrict__ p, const double *__restrict__ q,
    const double *__restrict__ dinv, double dc, double *__restrict__ x,
    double *__restrict__ r, lsb_pcg_state *__restrict__ st, int parity,
    const double *__restrict__ pq_parts, unsigned npq,
    double *__restrict__ partials2) {
  __shared__ double sred[8];
  const size_t gtid = (size_t)blockIdx.x * WG + threadIdx.x;
  const size_t gsz = (size_t)gridDim.x * WG;
  const size_t n2 = n / 2;
  const d2v *p2 = (const d2v *)p, *q2 = (const d2v *)q, *d2 = (const d2v *)dinv;
  d2v *x2 = (d2v *)x, *r2 = (d2v *)r;
  // Everything that does not depend on alpha is requested up front, so the
  // status word, the p.q partials, r.z and this lane's first operands are ONE
  // memory round trip, not four in a row (a small operator's sweep is nothing
  // but these latencies).
  const int stopped = st->status;
  const double rz = st->rz[parity];
  d2v pv = {0.0, 0.0}, qv = pv, dv = pv, xv = pv, rv = pv;
  const bool first = V2 && gtid < n2;
  if (first) {
    pv = ld2<NTPQ>(p2 + gtid), qv = ld2<NTPQ>(q2 + gtid), dv = ldd<NTPQ>(d2, gtid, dc);
    xv = ld2<NTX>(x2 + gtid), rv = ld2<NTR>(r2 + gtid);
  }
  double pqv[1];
  wg_sum_partials<1>(pq_parts, npq, pqv, sred);
  if (stopped)
    return;
  const double pq = pqv[0];
  if (!(pq != 0.0) || !isfinite(pq)) { // same decision in every workgroup
    if (blockIdx.x == 0 && threadIdx.x == 0)
      st->status = LSB_STATUS_BREAKDOWN;
    return;
  }
  const double alpha = rz / pq;
  if (blockIdx.x == 0 && threadIdx.x == 0)
    st->pq = pq;
  double acc[2] = {0.0, 0.0};
  if (V2) {
    if (first) {
      size_t i = gtid;
      for (;;) {
        xv.x += alpha * pv.x, xv.y += alpha * pv.y;
        rv.x -= alpha * qv.x, rv.y -= alpha * qv.y;
        st2<NTX>(x2 + i, xv), r2[i] = rv;
        acc[0] += rv.x * (dv.x * rv.x);
        acc[0] += rv.y * (dv.y * rv.y);
        acc[1] += rv.x * rv.x;
        acc[1] += rv.y * rv.y;
        i += gsz;
        if (i >= n2)
          break;
        pv = ld2<NTPQ>(p2 + i), qv = ld2<NTPQ>(q2 + i), dv = ldd<NTPQ>(d2, i, dc);
        xv = ld2<NTX>(x2 + i), rv = ld2<NTR>(r2 + i);
      }
    }
    if ((n & 1) && gtid == gsz - 1) {
      const size_t i = n - 1;
      x[i] += alpha * p[i];
      const double rs = r[i] - alpha * q[i];
      r[i] = rs;
      acc[0] += rs * ((dinv ? dinv[i] : dc) * rs), acc[1] += rs * rs;
    }
  } else {
    for (size_t i = gtid; i < n; i += gsz) {
      x[i] += alpha * p[i];
      const double rs = r[i] - alpha * q[i];
      r[i] = rs;
      acc[0] += rs * ((dinv ? dinv[i] : dc) * rs), acc[1] += rs * rs;
    }
  }
  wg_sum<2>(acc, sred);
  if (threadIdx.x == 0) {
    partials2[2 * blockIdx.x + 0] = acc[0];
    partials2[2 * blockIdx.x + 1] = acc[1];
  }
}

// (rz', rr) = sum partials ; stop test ; beta = rz'/rz ; p = dinv.*r + beta p
// X2: a lane keeps TWO 16-byte pairs per operand in flight (rows i and i + grid).  With one
// pair a lane has 32 bytes on their way (r and p; the constant diagonal is a register) --
// 2048 workgroups x 256 lanes x 32 B = 16.8 MB, about what 8 TB/s x 2 us of latency needs, and
// the sweep ran at 0.73 of peak where its five-operand sibling k_pcg_update_xr (64-80 B per
// lane) reaches 0.84 (profiles/r02_trace_kernel_stats.csv).
template <bool V2, bool NTR, bool NTP, bool X2>
__global__ __launch_bounds__(WG) void k_pcg_update_p(
    unsigned n, const double *__restrict__ r, const double *__restrict__ dinv, double dc,
    const double *pin, double *p, lsb_pcg_state *__restrict__ st, int parity,
    const double *__restrict__ parts2, unsigned nparts2) {
  // pin: where the previous direction is read from (== p, or the other buffer
  // of the launch-bound path that folds this update into the SpMV)
  __shared__ double sred[8];
  const size_t gtid = (size_t)blockIdx.x * WG + threadIdx.x;
  const size_t gsz = (size_t)gridDim.x * WG;
  const size_t n2 = n / 2;
  const d2v *r2 = (const d2v *)r, *d2 = (const d2v *)dinv;
  d2v *p2 = (d2v *)p;
  const d2v *pi2 = (const d2v *)pin;
  // as in k_pcg_update_xr: one round trip for status, scalars and operands
  const int stopped = st->status;
  const double rz_old = st->rz[parity], thresh2 = st->thresh2;
  if (blockIdx.x == 0 && threadIdx.x == 0)
    st->xpend = 0; // (two-launch column form: k_pcg_xfix, the launch before this one, has applied it; nobody
                   // reads the word in this launch)
  d2v rv = {0.0, 0.0}, dv = rv, pv = rv, rw = rv, dw = rv, pw = rv;
  const bool first = V2 && gtid < n2;
  bool second = X2 && V2 && gtid + gsz < n2;
  if (first)
    rv = ld2<NTR>(r2 + gtid), dv = ldd<NTR>(d2, gtid, dc), pv = ld2<NTP>(pi2 + gtid);
  if (second)
    rw = ld2<NTR>(r2 + gtid + gsz), dw = ldd<NTR>(d2, gtid + gsz, dc), pw = ld2<NTP>(pi2 + gtid + gsz);
  double v[2];
  wg_sum_partials<2>(parts2, nparts2, v, sred);
  if (stopped)
    return;
  const double rz_new = v[0], rr = v[1];
  const bool conv = rr <= thresh2;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // only this thread touches iters/rr/rz[parity^1]/status in this launch
    const int it = st->iters + 1;
    st->iters = it;
    st->rr = rr;
    st->rz[parity ^ 1] = rz_new;
    if (conv)
      st->status = LSB_STATUS_CONVERGED;
    else if (it >= st->maxit)
      st->status = LSB_STATUS_MAXIT;
  }
  if (conv)
    return;
  const double beta = rz_new / rz_old;
  if (V2) {
    if (first) {
      size_t i = gtid;
      const size_t step = X2 ? 2 * gsz : gsz;
      for (;;) {
        pv.x = pnew_of(dv.x, rv.x, beta, pv.x); // one expression for every kernel that forms p
        pv.y = pnew_of(dv.y, rv.y, beta, pv.y);
        p2[i] = pv;
        if (X2 && second) {
          pw.x = pnew_of(dw.x, rw.x, beta, pw.x);
          pw.y = pnew_of(dw.y, rw.y, beta, pw.y);
          p2[i + gsz] = pw;
        }
        i += step;
        if (i >= n2)
          break;
        rv = ld2<NTR>(r2 + i), dv = ldd<NTR>(d2, i, dc), pv = ld2<NTP>(pi2 + i);
        if (X2) {
          second = i + gsz < n2;
          if (second)
            rw = ld2<NTR>(r2 + i + gsz), dw = ldd<NTR>(d2, i + gsz, dc), pw = ld2<NTP>(pi2 + i + gsz);
        }
      }
    }
    if ((n & 1) && gtid == gsz - 1)
      p[n - 1] = pnew_of(dinv ? dinv[n - 1] : dc, r[n - 1], beta, pin[n - 1]);
  } else {
    for (size_t i = gtid; i < n; i += gsz)
      p[i] = pnew_of(dinv ? dinv[i] : dc, r[i], beta, pin[i]);
  }
}

// --------------------------------------------------------------------------
// Single-reduction CG (Chronopoulos & Gear 1989), LSB_KRYLOV_PCG1: the same
// Krylov iterates as PCG in exact arithmetic, arranged so that an iteration is
// TWO launches and ONE global reduction instead of three and two:
//     [this kernel]  beta = g'/g ; alpha = g' / (d - beta g'/alpha)
//                    p = u + beta p ; s = w + beta s ; x += alpha p ; r -= alpha s
//                    u = D^-1 r ; partials (g'' = r.u, r.r)
//     [SpMV]         w = S u ; partials d = w.u         (the fused-dot SpMV)
// with g' = r.u and r.r taken from this kernel's own previous launch and
// d = w.u from the SpMV in between.  For launch-latency-bound operators that is
// 2/3 of the launches; across GPUs it is one all-reduce (3 doubles) per
// iteration instead of two.  Costs one more vector (s) and 96 n instead of 88 n
// bytes per iteration, so the large single-GPU case keeps the classic form.
// --------------------------------------------------------------------------
// UI ("implicit u"): the Jacobi diagonal is the constant dc, so u = dc r is not
// kept at all -- r itself lives in the gather vector, the SpMV in between
// delivers t = S r and r.t, and w = dc t, w.u = dc^2 r.t are formed here:
// 9 vector passes per sweep instead of 11 (u neither read nor written).
template <bool V2, bool NT, bool UI>
__global__ __launch_bounds__(WG) void k_cg1_update(
    unsigned n, double *__restrict__ u, const double *__restrict__ w,
    const double *__restrict__ dinv, double dc, double *__restrict__ p, double *__restrict__ sv,
    double *__restrict__ x, double *__restrict__ r, lsb_pcg_state *__restrict__ st,
    int parity, const double *__restrict__ parts_gr, unsigned ngr,
    const double *__restrict__ parts_d, unsigned nd, const lsb_ar_collect col,
    double *__restrict__ partials2) {
  __shared__ double sred[8];
  const size_t gtid = (size_t)blockIdx.x * WG + threadIdx.x;
  const size_t gsz = (size_t)gridDim.x * WG;
  const size_t n2 = n / 2;
  // `pend`: the previous launch was the maxit-th update.  That launch does NOT
  // publish LSB_STATUS_MAXIT itself: its workgroups read the status word on
  // entry, and one that started after the leader's store would skip its slice
  // of x/r/p/s (a mix of two iterates).  It raises st->pad instead, a word
  // nobody tests in that launch; THIS launch promotes it to the final status --
  // every workgroup sees pad = 1 (written one launch ago) and returns, whatever
  // it reads in the status word.
  const int stopped = st->status, pend = st->pad;
  const double g_old = st->rz[parity], a_old = st->alpha[parity], thresh2 = st->thresh2;
  d2v *u2 = (d2v *)u, *p2 = (d2v *)p, *s2 = (d2v *)sv, *x2 = (d2v *)x, *r2 = (d2v *)r;
  const d2v *w2 = (const d2v *)w, *d2 = (const d2v *)dinv;
  d2v uv = {0.0, 0.0}, wv = uv, dv = uv, pv = uv, sw = uv, xv = uv, rv = uv;
  const bool first = V2 && gtid < n2;
  if (first) {
    wv = ld2<NT>(w2 + gtid), dv = ldd<NT>(d2, gtid, dc);
    pv = ld2<NT>(p2 + gtid), sw = ld2<NT>(s2 + gtid), xv = ld2<NT>(x2 + gtid);
    // (the vector the SpMV gathers next -- r with the implicit u, else u -- is loaded the plain
    // way: loaded nontemporal it is gone from the caches when the SpMV wants it, 40 instead of
    // 25 us on the 10 M-row operator, as with p in k_pcg_update_p)
    rv = ld2 < NT && !UI > (r2 + gtid);
    if (UI)
      uv = dc * rv, wv = dc * wv;
    else
      uv = ld2<false>(u2 + gtid);
  }
  double gr[2], dd[1];
  if (col.mbox) {
    // sharded solve over the direct xGMI path: the SpMV launch in front of this
    // one sent this rank's sums to every rank; take w.u, r.u, r.r from the
    // mailbox (rank order: the same bits everywhere) -- hip_ar.h
    if (stopped)
      return;
    if (threadIdx.x < 64) {
      double v[3];
      const bool ok = ar_collect<false>(col.mbox, col.R, col.epoch, col.timeout, 3, v);
      if (threadIdx.x == 0)
        sred[0] = v[0], sred[1] = v[1], sred[2] = v[2], sred[3] = ok ? 1.0 : 0.0;
    }
    __syncthreads();
    dd[0] = sred[0], gr[0] = sred[1], gr[1] = sred[2];
    if (sred[3] == 0.0) { // a peer did not arrive: every workgroup that notices says so
      if (threadIdx.x == 0)
        __hip_atomic_store(&st->status, (int)LSB_STATUS_COMM, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  } else {
    wg_sum_partials<2>(parts_gr, ngr, gr, sred);
    wg_sum_partials<1>(parts_d, nd, dd, sred);
    if (stopped)
      return;
  }
  const double g_new = gr[0], rr = gr[1], delta = UI ? dc * dc * dd[0] : dd[0];
  const bool leader = blockIdx.x == 0 && threadIdx.x == 0;
  if (rr <= thresh2 || pend) { // r of the previous update meets the tolerance, or it was the last allowed
    if (leader)
      st->status = rr <= thresh2 ? LSB_STATUS_CONVERGED : LSB_STATUS_MAXIT, st->rr = rr;
    return;
  }
  double beta = 0.0, alpha;
  if (a_old == 0.0) { // first iteration of the solve (k_pcg_init_state zeroes alpha)
    alpha = g_new / delta;
  } else {
    beta = g_new / g_old;
    alpha = g_new / (delta - beta * g_new / a_old);
  }
  if (!isfinite(alpha) || alpha == 0.0) { // same decision in every workgroup
    if (leader)
      st->status = LSB_STATUS_BREAKDOWN;
    return;
  }
  if (leader) {
    const int it = st->iters + 1;
    st->iters = it;
    st->rr = rr;
    st->pq = delta;
    st->rz[parity ^ 1] = g_new;
    st->alpha[parity ^ 1] = alpha;
    if (it >= st->maxit)
      st->pad = 1; // promoted to LSB_STATUS_MAXIT / CONVERGED by the next launch (see `pend`)
  }
  double acc[2] = {0.0, 0.0};
  if (V2) {
    if (first) {
      size_t i = gtid;
      for (;;) {
        pv.x = uv.x + beta * pv.x, pv.y = uv.y + beta * pv.y;
        sw.x = wv.x + beta * sw.x, sw.y = wv.y + beta * sw.y;
        xv.x += alpha * pv.x, xv.y += alpha * pv.y;
        rv.x -= alpha * sw.x, rv.y -= alpha * sw.y;
        uv.x = dv.x * rv.x, uv.y = dv.y * rv.y;
        st2<NT>(p2 + i, pv), st2<NT>(s2 + i, sw), st2<NT>(x2 + i, xv);
        if (UI) {
          r2[i] = rv; // the SpMV gathers it next: keep it cached
        } else {
          st2<NT>(r2 + i, rv);
          u2[i] = uv;
        }
        acc[0] += rv.x * uv.x;
        acc[0] += rv.y * uv.y;
        acc[1] += rv.x * rv.x;
        acc[1] += rv.y * rv.y;
        i += gsz;
        if (i >= n2)
          break;
        wv = ld2<NT>(w2 + i), dv = ldd<NT>(d2, i, dc);
        pv = ld2<NT>(p2 + i), sw = ld2<NT>(s2 + i), xv = ld2<NT>(x2 + i);
        rv = ld2 < NT && !UI > (r2 + i);
        if (UI)
          uv = dc * rv, wv = dc * wv;
        else
          uv = ld2<false>(u2 + i);
      }
    }
    if ((n & 1) && gtid == gsz - 1) {
      const size_t i = n - 1;
      const double ui0 = UI ? dc * r[i] : u[i], wi = UI ? dc * w[i] : w[i];
      const double pi = ui0 + beta * p[i], si = wi + beta * sv[i];
      p[i] = pi, sv[i] = si;
      x[i] += alpha * pi;
      const double ri = r[i] - alpha * si, ui = (dinv ? dinv[i] : dc) * ri;
      r[i] = ri;
      if (!UI)
        u[i] = ui;
      acc[0] += ri * ui, acc[1] += ri * ri;
    }
  } else {
    for (size_t i = gtid; i < n; i += gsz) {
      const double ui0 = UI ? dc * r[i] : u[i], wi = UI ? dc * w[i] : w[i];
      const double pi = ui0 + beta * p[i], si = wi + beta * sv[i];
      p[i] = pi, sv[i] = si;
      x[i] += alpha * pi;
      const double ri = r[i] - alpha * si, ui = (dinv ? dinv[i] : dc) * ri;
      r[i] = ri;
      if (!UI)
        u[i] = ui;
      acc[0] += ri * ui, acc[1] += ri * ri;
    }
  }
  wg_sum<2>(acc, sred);
  if (threadIdx.x == 0) {
    partials2[2 * blockIdx.x + 0] = acc[0];
    partials2[2 * blockIdx.x + 1] = acc[1];
  }
}

// Virtual-rank stand-in for the all-reduce: `nshard` shards on ONE device keep
// their scalars at base[q*stride + off .. +cnt); sum over q in rank order and
// hand every shard the same bits.
__global__ void k_vreduce(double *__restrict__ base, unsigned stride,
                          unsigned nshard, unsigned off, unsigned cnt) {
  const unsigned t = threadIdx.x;
  if (t < cnt) {
    double s = 0.0;
    for (unsigned q = 0; q < nshard; q++)
      s += base[(size_t)q * stride + off + t];
    for (unsigned q = 0; q < nshard; q++)
      base[(size_t)q * stride + off + t] = s;
  }
}

// --------------------------------------------------------------------------
// Launchers (C ABI)
// --------------------------------------------------------------------------
static inline unsigned div_up(unsigned a, unsigned b) { return (a + b - 1) / b; }
static inline unsigned round_up(unsigned a, unsigned b) { return div_up(a, b) * b; }
static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

static __thread int g_blas1_nt = 63; /* mask, see k_pcg_update_xr; bit 5: k_cg1_update.  Per host thread =
                                         per rank: set by the solver that enqueues (tune_blas1_nt) */

// --------------------------------------------------------------------------
// ONE ROUNDING RULE for the sliced-ELL kernels (k_spmv_sell, k_spmv_sell16, k_spmv_tmpl): a
// row's sum is the chain a = fma(value_j, x_j, a) over its slots in slot order, written with
// explicit fma() -- not left to the compiler's contraction, which fused some products of one
// source expression and not others (slot 0 of a group in k_spmv_sell16's per-slot path came
// out as v_mul + v_add, the rest as v_fma: found when the template kernel disagreed with it in
// the last bit).  With the rule in the source every layout of one operator -- 32-bit columns,
// 16-bit codes, constant slots, templates -- produces the same bits by construction.
// --------------------------------------------------------------------------
// a2-1, sliced-ELL form (LSB_SPMV_SELL, host side lsb_csr_sellize): slices of
// 128 rows stored column-major; lane l of the slice's wavefront owns rows 2l
// and 2l+1 and reads their j-th entries as ONE int2 + ONE double2, so every
// stream instruction of a wave moves a contiguous 512 B / 1 KiB, and for a
// stencil the gathers of a wave are contiguous runs of x as well.  No LDS, no
// barrier, no row offsets; a row's products are added in column order by one
// lane (the order of a sequential CSR loop).  10 M-row 5-point: 140 us against
// 152 us for the row-blocked kernel; 64 M-row 7-point: 1.23 ms against 1.55 ms
// (tools/spmv_lab.hip).  Groups of four slices are dealt to the XCDs like the
// row blocks of k_spmv_adaptive.  [s0, s0+ns) = the slices of this launch.
// --------------------------------------------------------------------------
// Which slice does wave `wave` of this workgroup take in its turn `it`?
// period = 0: the slices [0, ns) are cut into 8 contiguous chunks, one per XCD,
// groups of four dealt cyclically to the XCD's workgroups (turns it = slot,
// slot + gx, ... < turns).  period = P > 0 (slices per plane of a 3-D stencil,
// ns a multiple of P): XCD k takes the SAME eighth [P k/8, P (k+1)/8) of every
// plane, plane after plane -- the +-plane gathers of its resident workgroups
// then stay inside ~8 of its own sections (64 M-row 7-point: 1.4 MB of x instead
// of 3.6 MB next to the matrix stream in a 4 MiB L2; PMC 5.59 -> see DESIGN.md).
// Placement is a speed matter only; every slice is taken exactly once.
struct sell_deal {
  unsigned turns, base, L, period, full, total, colturns;
};
// (Tried in round 3 and taken out: MARCHING along z -- inside its eighth of every plane a wave keeps
// ONE position and walks plane after plane, so that the +-plane operands of a step are what the
// same positions touched one and two steps ago and come out of the L2 by time rather than by the
// concurrency of the workgroups that hold the neighbouring planes.  Bit-identical, and 622-685 us
// per launch of the 64 M-row 7-point operator against 322-327 us: gpurun_out/r3_march,
// profiles/r03_cfg4_dealing.txt.)
__device__ __forceinline__ sell_deal sell_deal_init(unsigned ns, unsigned period, unsigned xcd) {
  sell_deal d;
  d.period = period, d.full = 0, d.total = 0, d.colturns = 0;
  if (!period) {
    const unsigned ngrp = (ns + 3) / 4, chunk = (ngrp + NXCD - 1) / NXCD;
    const unsigned g0 = min(xcd * chunk, ngrp), g1 = min(g0 + chunk, ngrp);
    d.base = g0 * 4, d.turns = g1 - g0, d.L = 0;
  } else {
    // whole planes first, then this XCD's part of the ragged last one (a shard
    // need not hold whole planes)
    const unsigned qlo = period * xcd / NXCD, qhi = period * (xcd + 1) / NXCD;
    const unsigned planes = ns / period, rem = ns % period;
    d.L = qhi - qlo, d.base = qlo;
    d.full = d.L * planes;
    d.total = d.full + (rem > qlo ? min(qhi, rem) - qlo : 0u);
    // z-COLUMNS: while four whole planes are left, the four waves of a workgroup take the SAME
    // position of four consecutive planes -- their +-plane operands meet in one CU at one time
    // instead of relying on other workgroups of the XCD being at the neighbouring plane just then
    // (64 M-row 7-point, k_spmv_tmpl<2>: FETCH 1166 -> 704 MB per launch for 527 MB of compulsory
    // reads; the launch time does not move, profiles/r03_cfg4_dealing.txt)
    d.colturns = (planes / 4) * d.L;
    d.turns = d.colturns + (d.total - 4 * d.colturns + 3) / 4;
  }
  return d;
}
__device__ __forceinline__ unsigned sell_deal_slice(const sell_deal &d, unsigned it, unsigned wave,
                                                    unsigned ns) {
  if (!d.period) {
    const unsigned si = d.base + it * 4 + wave;
    return si < ns ? si : 0xFFFFFFFFu;
  }
  if (it < d.colturns)
    return ((it / d.L) * 4 + wave) * d.period + d.base + it % d.L;
  const unsigned m = it * 4 + wave;
  if (m >= d.total)
    return 0xFFFFFFFFu;
  if (m < d.full)
    return (m / d.L) * d.period + d.base + m % d.L;
  return (ns / d.period) * d.period + d.base + (m - d.full);
}

typedef int i2v __attribute__((ext_vector_type(2)));
typedef int i4v __attribute__((ext_vector_type(4)));
typedef double sell_d2u __attribute__((ext_vector_type(2), aligned(8))); // a 16-byte gather at an 8-byte boundary
typedef double sell_d2v __attribute__((ext_vector_type(2)));
#define SELL_U 5
template <class VT> struct vt2;
template <> struct vt2<double> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct vt2<float> { typedef float type __attribute__((ext_vector_type(2))); };
template <int FLAGS, class VT = double>
__global__ __launch_bounds__(WG, 6) void k_spmv_sell(
    const unsigned *__restrict__ sptr, unsigned s0, unsigned ns, unsigned period, unsigned n,
    const int *__restrict__ cols, const VT *__restrict__ vals,
    const double *__restrict__ x, double *__restrict__ y, const double *__restrict__ xdot,
    double *__restrict__ partials, const lsb_pcg_state *__restrict__ st, const lsb_ar_tail tail) {
  __shared__ double sred[4];
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const unsigned gx = gridDim.x / NXCD, xcd = blockIdx.x % NXCD, slot = blockIdx.x / NXCD;
  const sell_deal deal = sell_deal_init(ns, period, xcd);
  const int stopped = st ? st->status : 0; // tested behind the first loads
  double dot = 0.0;
  for (unsigned g = slot; g < deal.turns; g += gx) {
    const unsigned si = __builtin_amdgcn_readfirstlane(sell_deal_slice(deal, g, wave, ns));
    if (si != 0xFFFFFFFFu) {
      const unsigned s = s0 + si;
      const unsigned base = sptr[s], len = (sptr[s + 1] - base) / LSB_SELL_ROWS;
      typedef typename vt2<VT>::type v2t;
      const i2v *cp = (const i2v *)(cols + base) + lane;
      const v2t *vp = (const v2t *)(vals + base) + lane;
      double a0 = 0.0, a1 = 0.0;
      for (unsigned j0 = 0; j0 < len; j0 += SELL_U) {
        i2v c[SELL_U];
        v2t v[SELL_U];
#pragma unroll
        for (int u = 0; u < SELL_U; u++)
          if (j0 + u < len) {
            if (FLAGS & SP_NT) {
              c[u] = __builtin_nontemporal_load(cp + (size_t)(j0 + u) * 64);
              v[u] = __builtin_nontemporal_load(vp + (size_t)(j0 + u) * 64);
            } else {
              c[u] = cp[(size_t)(j0 + u) * 64];
              v[u] = vp[(size_t)(j0 + u) * 64];
            }
          }
        if (stopped)
          return;
#pragma unroll
        for (int u = 0; u < SELL_U; u++)
          if (j0 + u < len) {
            a0 = fma((double)v[u].x, x[c[u].x], a0); // explicit: see "one rounding rule" below
            a1 = fma((double)v[u].y, x[c[u].y], a1);
          }
      }
      const unsigned row = s * LSB_SELL_ROWS + 2 * lane;
      if (row + 1 < n) {
        const sell_d2v o = {a0, a1};
        *(sell_d2v *)(y + row) = o;
        if (xdot) {
          dot = fma(a0, xdot[row], dot);
          dot = fma(a1, xdot[row + 1], dot);
        }
      } else if (row < n) {
        y[row] = a0;
        if (xdot)
          dot = fma(a0, xdot[row], dot);
      }
    }
  }
  if (stopped)
    return;
  spmv_publish(partials, dot, sred, tail);
}

// The same with 16-bit column codes (lsb_csr_sellize16): the column of the entry
// in slot j of global row g is g + base + code, {base, k} = sbase[slice, j] one
// 8-byte scalar load per slice and slot; k >= 0 says where the slot's 128 codes
// are, k = -1 that the slot has none (all its entries lie on ONE diagonal,
// folded into the base -- every slot of a structured-grid operator).  10 instead of 12 bytes per entry: 10 M-row 5-point
// 146 -> 122 us on the same box (tools/spmv_lab.hip).  A slot holds entries of
// one diagonal band, so rows can have padding BETWEEN their entries; padding
// has value 0 and is recognised by that (no gather, contributes an exact 0).
// CHEB: a Chebyshev step in the epilogue (lsb_cheb_epi, lsb_impl.h) instead of the
// store of y: the polynomial preconditioner's S z products never travel to memory.
typedef short s2v __attribute__((ext_vector_type(2)));
template <int FLAGS, class VT = double, bool CHEB = false>
__global__ __launch_bounds__(WG, 6) void k_spmv_sell16(
    const unsigned *__restrict__ sptr, unsigned s0, unsigned ns, unsigned period, unsigned n,
    unsigned row_begin, unsigned xlen, const short *__restrict__ codes, const int *__restrict__ sbase,
    const VT *__restrict__ vals, const double *__restrict__ vconst, unsigned ulen,
    const double *__restrict__ x, double *__restrict__ y, const double *__restrict__ xdot,
    double *__restrict__ partials,
    const lsb_pcg_state *__restrict__ st, const lsb_ar_tail tail, const lsb_cheb_epi epi) {
  // vconst != NULL: the constant-slot layout (lsb_sell16_value_slots) -- sbase holds
  // {base, code slot, value slot, 0} per slot, and a slot whose value slot is -1 has the
  // one value vconst[slot] for all its 128 entries (every interior diagonal of a
  // constant-coefficient stencil): 8 bytes per SLOT instead of per entry.
  __shared__ double sred[4];
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const unsigned gx = gridDim.x / NXCD, xcd = blockIdx.x % NXCD, slot = blockIdx.x / NXCD;
  const sell_deal deal = sell_deal_init(ns, period, xcd);
  const int stopped = st ? st->status : 0; // tested behind the first loads
  double dot = 0.0;
  for (unsigned g = slot; g < deal.turns; g += gx) {
    const unsigned si = __builtin_amdgcn_readfirstlane(sell_deal_slice(deal, g, wave, ns));
    if (si != 0xFFFFFFFFu) {
      const unsigned s = s0 + si;
      typedef typename vt2<VT>::type v2t;
      // ulen != 0: every slice has ulen slots -- no look at sptr, one memory round trip
      // less in front of the gathers (with the values gone the kernel waits on its chain
      // of dependent loads, not on bandwidth)
      const unsigned base = ulen ? s * ulen * LSB_SELL_ROWS : sptr[s];
      const unsigned len = ulen ? ulen : (sptr[s + 1] - base) / LSB_SELL_ROWS;
      const unsigned q0 = base / LSB_SELL_ROWS; // first slot of the slice
      const unsigned row = s * LSB_SELL_ROWS + 2 * lane;
      const int grow = (int)(row + row_begin);
      double a0 = 0.0, a1 = 0.0;
      sell_d2v xd = {0.0, 0.0}; // the dot's operand: asked for up front, not behind the gathers
      if (!CHEB && xdot) {
        if (row + 1 < n)
          xd = *(const sell_d2v *)(xdot + row);
        else if (row < n)
          xd.x = xdot[row];
      }
      for (unsigned j0 = 0; j0 < len; j0 += SELL_U) {
        if (vconst) {
          // The common case of a constant-coefficient operator: every slot of this group is
          // constant (no values, no padding) with one common code.  Straight-line code: all
          // slot records and constants by scalar loads in one go, then one 16-byte gather per
          // lane and slot -- rows 2l and 2l + 1 read x at grow + b and grow + b + 1 -- all in
          // flight together (a branch per slot makes every one of them a round trip of its
          // own: 61 us instead of the 3x fewer bytes' worth on the 10 M-row operator).
          const unsigned cnt = len - j0 < SELL_U ? len - j0 : SELL_U;
          i4v rec[SELL_U];
          double cst[SELL_U];
#pragma unroll
          for (int u = 0; u < SELL_U; u++) { // (past the group's end: its last slot again)
            const unsigned q = q0 + j0 + ((unsigned)u < cnt ? (unsigned)u : cnt - 1u);
            rec[u] = ((const i4v *)sbase)[q];
            cst[u] = vconst[q];
          }
          bool fast = true;
#pragma unroll
          for (int u = 0; u < SELL_U; u++)
            fast = fast && (rec[u].y < 0 && rec[u].z < 0);
          if (fast) {
            if (stopped)
              return;
            sell_d2u t[SELL_U];
#pragma unroll
            for (int u = 0; u < SELL_U; u++)
              t[u] = *(const sell_d2u *)(x + (grow + rec[u].x));
#pragma unroll
            for (int u = 0; u < SELL_U; u++)
              if ((unsigned)u < cnt) { // the same products in the same order as below
                const VT cv = (VT)cst[u];
                a0 = fma((double)cv, t[u].x, a0);
                a1 = fma((double)cv, t[u].y, a1);
              }
            continue;
          }
        }
        // Per slot: record (scalar), codes, values AND the gathers, all asked for before any of them
        // is waited for.  The gather's address does not look at the value (it used to: padding has
        // value 0 and was given address 0, which made every gather wait for its value's trip to
        // HBM): the index is clamped into x instead and the operand of a padding entry is dropped
        // at the product, an exact 0 as before.  pair: both rows of every lane gather at the same
        // offset AND the wave's 129 operands lie inside x -- one 16-byte gather per lane.
        s2v c[SELL_U];
        v2t v[SELL_U];
        sell_d2u t[SELL_U];
        const long long g0 = (long long)s * LSB_SELL_ROWS + row_begin; // global row of lane 0's first row
#pragma unroll
        for (int u = 0; u < SELL_U; u++)
          if (j0 + u < len) {
            const unsigned q = q0 + j0 + u;
            int b, kc, kv; // wave-uniform: scalar loads
            if (vconst) {
              const i4v bk = ((const i4v *)sbase)[q];
              b = bk.x, kc = bk.y, kv = bk.z;
            } else {
              const i2v bk = ((const i2v *)sbase)[q]; // {base, code slot or -1}
              b = bk.x, kc = bk.y, kv = (int)q;
            }
            c[u] = (s2v){0, 0};
            if (kc >= 0) { // slots with one common code carry none
              const s2v *cp = (const s2v *)codes + (size_t)kc * 64 + lane;
              c[u] = (FLAGS & SP_NT) ? __builtin_nontemporal_load(cp) : *cp;
            }
            if (kv >= 0) {
              const v2t *vp = (const v2t *)(vals + (size_t)kv * LSB_SELL_ROWS) + lane;
              v[u] = (FLAGS & SP_NT) ? __builtin_nontemporal_load(vp) : *vp;
            } else { // one value for the slot's 128 entries
              const VT cv = (VT)vconst[q];
              v[u] = (v2t){cv, cv};
            }
            const bool pair = kc < 0 && g0 + b >= 0 && g0 + b + (long long)LSB_SELL_ROWS <= (long long)xlen;
            if (pair) {
              t[u] = *(const sell_d2u *)(x + (grow + b));
            } else {
              long long e0 = (long long)grow + b + (int)c[u].x, e1 = (long long)grow + 1 + b + (int)c[u].y;
              e0 = e0 < 0 ? 0 : (e0 >= (long long)xlen ? (long long)xlen - 1 : e0);
              e1 = e1 < 0 ? 0 : (e1 >= (long long)xlen ? (long long)xlen - 1 : e1);
              t[u].x = x[e0], t[u].y = x[e1];
            }
          }
        if (stopped)
          return;
#pragma unroll
        for (int u = 0; u < SELL_U; u++)
          if (j0 + u < len) { // value 0 = padding: no operand, an exact 0
            a0 = fma((double)v[u].x, v[u].x != (VT)0 ? t[u].x : 0.0, a0);
            a1 = fma((double)v[u].y, v[u].y != (VT)0 ? t[u].y : 0.0, a1);
          }
      }
      if (CHEB) { // d = a d + b D^-1 (r - w); z' = z + d -- k_cheb_step's expression
        if (row + 1 < n) {
          const sell_d2v rr = *(const sell_d2v *)(epi.r + row), dd = *(const sell_d2v *)(epi.d + row);
          const double i0 = epi.dinv ? epi.dinv[row] : epi.dc, i1 = epi.dinv ? epi.dinv[row + 1] : epi.dc;
          const double v0 = fma(epi.a, dd.x, epi.b * (i0 * (rr.x - a0)));
          const double v1 = fma(epi.a, dd.y, epi.b * (i1 * (rr.y - a1)));
          const sell_d2v dn = {v0, v1}, zn = {x[grow] + v0, x[grow + 1] + v1};
          *(sell_d2v *)(epi.d + row) = dn;
          *(sell_d2v *)(epi.zout + row) = zn;
        } else if (row < n) {
          const double v0 = fma(epi.a, epi.d[row], epi.b * ((epi.dinv ? epi.dinv[row] : epi.dc) * (epi.r[row] - a0)));
          epi.d[row] = v0;
          epi.zout[row] = x[grow] + v0;
        }
      } else if (row + 1 < n) {
        const sell_d2v o = {a0, a1};
        *(sell_d2v *)(y + row) = o;
        if (xdot) {
          dot = fma(a0, xd.x, dot);
          dot = fma(a1, xd.y, dot);
        }
      } else if (row < n) {
        y[row] = a0;
        if (xdot)
          dot = fma(a0, xd.x, dot);
      }
    }
  }
  if (stopped)
    return;
  spmv_publish(partials, dot, sred, tail);
}

// --------------------------------------------------------------------------
// a2-1, sliced-ELL with slice TEMPLATES (LSB_SP_TMPL; host side lsb_sell16_templates) -- the
// constant-slot layout of a structured grid, where the shipped kernel above is bound by its
// chain of dependent loads per slice (cold slot records -> gathers), not by bytes: 176 MB in
// 41-48 us.  Measured on the 10 M-row 5-point operator (tools/stencil_lab.hip, back to back):
//   slot records by scalar loads per slice, five gathers (k_spmv_sell16)        40.8-48.0 us
//   one record per slice + the template out of the scalar cache, five gathers   32.5 us
//   + the +-1 diagonals from the centre's pair by a lane shift (three aligned
//     gathers, the dot's operand = the centre pair), straight-line              24.6-26.9 us
//   y = 4 x with the fused dot (the HBM floor of these bytes)                   23.2-24.6 us
// A wave takes a slice per turn as above.  srec[slice] (16 bytes, one scalar load) names its
// template (255: none -- the slice keeps values in a far slot or has more than 8 slots, and goes
// the per-slot way), its first kept value slot and its first mask;
// a SHAPED template is [NF far slots][c-1, c, c+1][NF far slots]: lanes gather the far slots
// and the centre c, and form the operands of c-1 / c+1 from the centre pair of the
// neighbouring lane (DPP wave shift; lane 0 and lane 63 fetch the one element beyond the
// wave's 128 by a two-lane load).  Products are formed in slot order, i.e. in the order of
// k_spmv_sell16: bit-identical results (tests/test_sell.py).
// --------------------------------------------------------------------------
__device__ __forceinline__ double lane_above(double v) { // the value lane - 1 holds (lane 0: 0)
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false); // wave_shr:1
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_below(double v) { // the value lane + 1 holds (lane 63: 0)
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, false); // wave_shl:1
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// values of the slots c-1 / c+1 of a shaped template where they are not plain constants.
// Masked slots (kind 2): the slice's 128-bit mask arrives in SCALAR registers (mk[side][0..1],
// loaded by the caller beside the gathers -- as two dependent vector loads behind them they
// were half of a turn's latency on the 7-point grid, where every slice holds a line end).
__device__ __forceinline__ void tmpl_side_values(const lsb_sell_tmpl *T, int NF, unsigned vb, unsigned lane,
                                                 const int (&side_k)[2], const int (&side_kd)[2],
                                                 const unsigned long long (&mk)[2][2],
                                                 const void *__restrict__ vals, int f32, double &vm0, double &vm1,
                                                 double &vp0, double &vp1) {
#pragma unroll
  for (int side = 0; side < 2; side++) {
    const int j = NF + 2 * side, k = side_k[side], kd = side_kd[side];
    if (k < 0)
      continue;
    double v0, v1;
    if (kd == 2) { // bit r of the slice's mask: row r holds the template's number
      const unsigned long long w = (lane & 32u) ? mk[side][1] : mk[side][0];
      const unsigned sh = (2u * lane) & 63u;
      v0 = (w >> sh) & 1ull ? T->cst[j] : 0.0;
      v1 = (w >> (sh + 1u)) & 1ull ? T->cst[j] : 0.0;
    } else if (f32) {
      typedef vt2<float>::type f2;
      const f2 v = *((const f2 *)((const float *)vals + (size_t)(vb + (unsigned)k) * LSB_SELL_ROWS) + lane);
      v0 = (double)v.x, v1 = (double)v.y;
    } else {
      const sell_d2v v = *((const sell_d2v *)((const double *)vals + (size_t)(vb + (unsigned)k) * LSB_SELL_ROWS) + lane);
      v0 = v.x, v1 = v.y;
    }
    if (side == 0)
      vm0 = v0, vm1 = v1;
    else
      vp0 = v0, vp1 = v1;
  }
}

typedef unsigned u4v __attribute__((ext_vector_type(4)));
#ifndef DEFER_DEPTH
#define DEFER_DEPTH 1 // turns a wave's y waits in LDS (2: 272-274 against 276-287 us on the 64 M-row 7-point operator, 29.7 against 28.2 us on the 10 M-row 5-point one)
#endif
template <int NF, bool CHEB, bool DEFER = false>
__global__ __launch_bounds__(WG, 6) void k_spmv_tmpl(
    const unsigned *__restrict__ sptr, unsigned s0, unsigned ns, unsigned period, unsigned n,
    unsigned row_begin, unsigned xlen, const u4v *__restrict__ srec,
    const unsigned long long *__restrict__ mask, const lsb_sell_tmpl *__restrict__ td,
    const int *__restrict__ sbase, const void *__restrict__ vals, int f32,
    const double *__restrict__ vconst, const double *__restrict__ x, double *__restrict__ y,
    const double *__restrict__ xdot, int dot_is_x, double *__restrict__ partials,
    const lsb_pcg_state *__restrict__ st, const lsb_ar_tail tail, const lsb_cheb_epi epi) {
  __shared__ double sred[4];
  // DEFERRED STORE.  vmcnt counts loads and stores in ONE order (gfx9), so a wave's first wait for a
  // gather also waits until the 1 KB of y it stored at the end of the turn before has been
  // acknowledged -- quick where the vectors sit in the Infinity Cache, slow where every line
  // written pushes a dirty one out to HBM (a turn of the 64 M-row 7-point operator: 4.0 us, of its
  // 50-plane slab 2.75 us; neither less traffic nor look-ahead touches changed that,
  // profiles/r03_cfg4_dealing.txt).  So a turn parks its two results per lane in LDS and the NEXT
  // turn stores them, behind its own gathers: the store is then the youngest operation in flight
  // when the gathers are waited for, and has a whole turn to complete.
  __shared__ sell_d2v ypark[DEFER ? DEFER_DEPTH * WG : 1];
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const unsigned gx = gridDim.x / NXCD, xcd = blockIdx.x % NXCD, slot = blockIdx.x / NXCD;
  const sell_deal deal = sell_deal_init(ns, period, xcd);
  const int stopped = st ? st->status : 0; // tested behind the first loads
  double dot = 0.0;
  // first row of the slice whose results sit in ypark[k] (wave-uniform); slot = turn % DEFER_DEPTH
  unsigned parked[DEFER_DEPTH], pslot = 0;
#pragma unroll
  for (int k = 0; k < DEFER_DEPTH; k++)
    parked[k] = 0xFFFFFFFFu;
#define TMPL_UNPARK_SLOT(K)                                                                    \
  do {                                                                                         \
    if (DEFER && parked[K] != 0xFFFFFFFFu) {                                                   \
      *(sell_d2v *)(y + parked[K] + 2 * lane) = ypark[(K) * WG + tid];                         \
      parked[K] = 0xFFFFFFFFu;                                                                 \
    }                                                                                          \
  } while (0)
#define TMPL_UNPARK()                                                                          \
  do {                                                                                         \
    if (DEFER_DEPTH == 1 || pslot == 0)                                                        \
      TMPL_UNPARK_SLOT(0);                                                                     \
    else                                                                                       \
      TMPL_UNPARK_SLOT(DEFER_DEPTH - 1);                                                       \
  } while (0)
  for (unsigned g = slot; g < deal.turns; g += gx, pslot ^= 1u) {
    const unsigned si = __builtin_amdgcn_readfirstlane(sell_deal_slice(deal, g, wave, ns));
    if (si == 0xFFFFFFFFu)
      continue;
    const unsigned s = s0 + si;
    // the slice's record {template id (255: none), first kept value slot, first mask, 0}: ONE
    // 16-byte scalar load (a byte array of ids + a pair array cost two dependent round trips, the
    // byte one through the vector path).  (Asking for it one turn ahead, beside the gathers of the
    // turn before: 320 -> 304 us on the 64 M-row 7-point operator, 25 -> 28 us on the 10 M-row
    // 5-point one whose records stay in L2; on top of the deferred store 276-287 -> 291-295 us -- not kept.)
    const u4v rec = srec[s];
    const unsigned t = __builtin_amdgcn_readfirstlane(rec.x), vb = __builtin_amdgcn_readfirstlane(rec.y);
    const unsigned mb = __builtin_amdgcn_readfirstlane(rec.z);
    const unsigned row = s * LSB_SELL_ROWS + 2 * lane;
    const int grow = (int)(row + row_begin);
    double a0 = 0.0, a1 = 0.0;
    sell_d2v xd = {0.0, 0.0};
    bool have_xd = CHEB || !xdot;
    const lsb_sell_tmpl *T = td + (t == 255u ? 0u : t);
    if (t != 255u && T->shaped) {
      const int bc = T->base[NF + 1];
      // (everything the masks' addresses need, in the same batch of scalar loads as the bases)
      const int side_k[2] = {T->kidx[NF], T->kidx[NF + 2]}, side_kd[2] = {T->kind[NF], T->kind[NF + 2]};
      sell_d2u lo[NF > 0 ? NF : 1], hi[NF > 0 ? NF : 1];
#pragma unroll
      for (int k = 0; k < NF; k++)
        lo[k] = *(const sell_d2u *)(x + (grow + T->base[k]));
      const sell_d2u c = *(const sell_d2u *)(x + (grow + bc));
#pragma unroll
      for (int k = 0; k < NF; k++)
        hi[k] = *(const sell_d2u *)(x + (grow + T->base[NF + 3 + k]));
      // the element in front of the wave's first operand of slot c-1 / behind its last of slot
      // c+1.  Where the slot is constant it exists; where it keeps values the entry it belongs to
      // may be padding at the operator's first / last row: the index is clamped (the product is
      // an exact 0 there whatever is read)
      double edge = 0.0;
      if (lane == 0 || lane == 63) {
        long long e = (long long)grow + bc + (lane == 0 ? -1 : 2);
        e = e < 0 ? 0 : (e >= (long long)xlen ? (long long)xlen - 1 : e);
        edge = x[e];
      }
      if (!have_xd && !(dot_is_x && bc == 0)) {
        if (row + 1 < n)
          xd = *(const sell_d2v *)(xdot + row);
        else if (row < n)
          xd.x = xdot[row];
        have_xd = true;
      }
      // the masks of masked slots c-1 / c+1: wave-uniform, 16 bytes each, by scalar loads that
      // travel while the gathers do
      unsigned long long mk[2][2] = {{0ull, 0ull}, {0ull, 0ull}};
#pragma unroll
      for (int side = 0; side < 2; side++)
        if (side_k[side] >= 0 && side_kd[side] == 2) {
          const unsigned long long *mp = mask + 2 * ((size_t)mb + (unsigned)side_k[side]);
          mk[side][0] = mp[0], mk[side][1] = mp[1];
        }
      TMPL_UNPARK(); // the turn before's y: behind this turn's gathers (and behind the masks' scalar
                     // loads, whose counter the LDS read shares: in front of them measured no better)
      if (stopped)
        return;
      double up = lane_above(c.y), dn = lane_below(c.x);
      if (lane == 0)
        up = edge;
      if (lane == 63)
        dn = edge;
      if (!have_xd)
        xd.x = c.x, xd.y = c.y, have_xd = true; // the centre pair IS the dot's operand
#pragma unroll
      for (int k = 0; k < NF; k++) {
        const double v = T->cst[k];
        a0 = fma(v, lo[k].x, a0), a1 = fma(v, lo[k].y, a1);
      }
      {
        // slots c-1 / c+1 may keep their values (a grid line ends inside the slice: zeros there) or
        // be MASKED: the template's number where the slice's 128-bit mask says so, else zero
        // (looked up here, behind the far slots' products: their registers are free by now)
        double vm0 = T->cst[NF], vm1 = vm0, vp0 = T->cst[NF + 2], vp1 = vp0;
        if (side_k[0] >= 0 || side_k[1] >= 0)
          tmpl_side_values(T, NF, vb, lane, side_k, side_kd, mk, vals, f32, vm0, vm1, vp0, vp1);
        // value 0 = padding: no operand, an exact 0 (the rule of k_spmv_sell16)
        const double vc = T->cst[NF + 1];
        a0 = fma(vm0, vm0 != 0.0 ? up : 0.0, a0), a1 = fma(vm1, vm1 != 0.0 ? c.x : 0.0, a1); // slot c-1: x[. - 1], x[.]
        a0 = fma(vc, c.x, a0), a1 = fma(vc, c.y, a1);                                         // slot c
        a0 = fma(vp0, vp0 != 0.0 ? c.y : 0.0, a0), a1 = fma(vp1, vp1 != 0.0 ? dn : 0.0, a1); // slot c+1
      }
#pragma unroll
      for (int k = 0; k < NF; k++) {
        const double v = T->cst[NF + 3 + k];
        a0 = fma(v, hi[k].x, a0), a1 = fma(v, hi[k].y, a1);
      }
    } else {
      TMPL_UNPARK();
      if (!have_xd) {
        if (row + 1 < n)
          xd = *(const sell_d2v *)(xdot + row);
        else if (row < n)
          xd.x = xdot[row];
        have_xd = true;
      }
      if (t != 255u) { // a pure slice of another shape (first / last grid line): every slot gathered
        const int cnt = T->nslots;
        if (stopped)
          return;
        for (int u0 = 0; u0 < LSB_TMPL_SLOTS; u0 += 4) { // (rare slices: four gathers in flight will do)
          sell_d2u v[4];
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (u0 + u < cnt)
              v[u] = *(const sell_d2u *)(x + (grow + T->base[u0 + u]));
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (u0 + u < cnt) {
              const double k = T->cst[u0 + u];
              a0 = fma(k, v[u].x, a0), a1 = fma(k, v[u].y, a1);
            }
        }
      } else { // a slot keeps its values (or the slice has > 8 slots): slot by slot
        const unsigned q0 = sptr[s] / LSB_SELL_ROWS, len = (sptr[s + 1] - sptr[s]) / LSB_SELL_ROWS;
        if (stopped)
          return;
        for (unsigned j = 0; j < len; j++) {
          const i4v r = ((const i4v *)sbase)[q0 + j]; // {base, -1 (no codes here), value slot or -1, 0}
          if (r.z < 0) {
            const double k = vconst[q0 + j];
            const sell_d2u v = *(const sell_d2u *)(x + (grow + r.x));
            a0 = fma(k, v.x, a0), a1 = fma(k, v.y, a1);
          } else {
            double v0, v1;
            if (f32) {
              const vt2<float>::type v = *((const vt2<float>::type *)((const float *)vals + (size_t)r.z * LSB_SELL_ROWS) + lane);
              v0 = (double)v.x, v1 = (double)v.y;
            } else {
              const sell_d2v v = *((const sell_d2v *)((const double *)vals + (size_t)r.z * LSB_SELL_ROWS) + lane);
              v0 = v.x, v1 = v.y;
            }
            const bool p0 = v0 != 0.0, p1 = v1 != 0.0; // padding: no gather, an exact 0
            const double t0 = x[p0 ? grow + r.x : 0], t1 = x[p1 ? grow + 1 + r.x : 0];
            a0 = fma(v0, p0 ? t0 : 0.0, a0), a1 = fma(v1, p1 ? t1 : 0.0, a1);
          }
        }
      }
    }
    if (CHEB) { // d = a d + b D^-1 (r - w); z' = z + d -- k_cheb_step's expression
      if (row + 1 < n) {
        const sell_d2v rr = *(const sell_d2v *)(epi.r + row), dd = *(const sell_d2v *)(epi.d + row);
        const double i0 = epi.dinv ? epi.dinv[row] : epi.dc, i1 = epi.dinv ? epi.dinv[row + 1] : epi.dc;
        const double v0 = fma(epi.a, dd.x, epi.b * (i0 * (rr.x - a0)));
        const double v1 = fma(epi.a, dd.y, epi.b * (i1 * (rr.y - a1)));
        const sell_d2v dn2 = {v0, v1}, zn = {x[grow] + v0, x[grow + 1] + v1};
        *(sell_d2v *)(epi.d + row) = dn2;
        *(sell_d2v *)(epi.zout + row) = zn;
      } else if (row < n) {
        const double v0 = fma(epi.a, epi.d[row], epi.b * ((epi.dinv ? epi.dinv[row] : epi.dc) * (epi.r[row] - a0)));
        epi.d[row] = v0;
        epi.zout[row] = x[grow] + v0;
      }
    } else if (DEFER && s * LSB_SELL_ROWS + LSB_SELL_ROWS <= n) { // a whole slice: parked, stored by the next turn
      const sell_d2v o = {a0, a1};
      if (DEFER_DEPTH == 1 || pslot == 0)
        ypark[tid] = o, parked[0] = s * LSB_SELL_ROWS;
      else
        ypark[(DEFER_DEPTH - 1) * WG + tid] = o, parked[DEFER_DEPTH - 1] = s * LSB_SELL_ROWS;
      if (xdot) {
        dot = fma(a0, xd.x, dot);
        dot = fma(a1, xd.y, dot);
      }
    } else if (row + 1 < n) {
      const sell_d2v o = {a0, a1};
      *(sell_d2v *)(y + row) = o;
      if (xdot) {
        dot = fma(a0, xd.x, dot);
        dot = fma(a1, xd.y, dot);
      }
    } else if (row < n) {
      y[row] = a0;
      if (xdot)
        dot = fma(a0, xd.x, dot);
    }
  }
#pragma unroll
  for (int k = 0; k < DEFER_DEPTH; k++)
    TMPL_UNPARK_SLOT(k);
#undef TMPL_UNPARK
#undef TMPL_UNPARK_SLOT
  if (stopped)
    return;
  spmv_publish(partials, dot, sred, tail);
}

// --------------------------------------------------------------------------
// a2-1, the template layout walked in Z-COLUMNS (LSB_SP_COL; host side lsb_sell_tmpl_columns,
// include/lsbench_hip.h).  On a 3-D stencil whose planes are whole slices the outermost far slots
// of a slice reach exactly one plane down and up, so the pair a lane gathers for slot 0 of the
// slice one plane up IS the centre pair of this slice, and this slice's last slot is that one's
// centre: a wave that walks a column s, s + period, s + 2 period, ... keeps three centre pairs in
// registers and gathers ONE new plane per step -- 3 instead of 5 gathers per 7-point slice, and
// every element of x leaves memory (K + 2) / K times instead of "three times unless the L2 still
// has it" (k_spmv_tmpl on the 64 M-row 7-point operator: FETCH 1.30 GB for 0.53 GB of compulsory
// reads).  The template, its constants and the masks of the slots c-1 / c+1 are looked at once
// per column (the host has checked that all its slices share them): a step is ~40 vector
// instructions against ~110 + ~120 scalar ones per slice of k_spmv_tmpl.
// Pipeline of a column (vmcnt counts loads and stores in one order on gfx9): a step issues the
// NEXT step's gathers, then stores the PREVIOUS step's y out of registers, then waits for its own
// operands -- which are older than both, so neither the new gathers nor the store are waited for,
// and a store has a whole step to be acknowledged (the job of LSB_SP_DEFER's LDS parking).
// Items of one slice (first / last plane, ragged ends) go slot by slot off the slot records.
// The same operands and products in the same order as k_spmv_tmpl / k_spmv_sell16: y bit for
// bit; the fused dot's partial sums follow this kernel's own (fixed) dealing.
// DOT: 0 none, 1 the centre pair is the dot's operand (xdot = x + row_begin, centre base 0), 2 loaded.
// --------------------------------------------------------------------------
template <int NF> struct col_ops { // what a step needs besides the three centre pairs
  sell_d2u lo[NF > 1 ? NF - 1 : 1], hi[NF > 1 ? NF - 1 : 1];
  double edge;
  sell_d2v xd;
};
template <int NF> struct col_tmpl { // the column's template, in scalar registers
  int bc, bl[NF > 1 ? NF - 1 : 1], bh[NF > 1 ? NF - 1 : 1];
  double k0, kl[NF > 1 ? NF - 1 : 1], kc, kh[NF > 1 ? NF - 1 : 1], kL;
};
// gu = global row of lane 0's first row in this step's slice; lrow = its local row
template <int NF, int DOT>
__device__ __forceinline__ void col_issue(col_ops<NF> &o, sell_d2u &cnext, const col_tmpl<NF> &T, const double *__restrict__ x,
                                          const double *__restrict__ xdot, long long gu, unsigned lrow, unsigned P,
                                          unsigned xlen, unsigned lane, bool with_next) {
  if (with_next)
    cnext = *(const sell_d2u *)(x + (gu + T.bc + (long long)P) + 2 * lane);
#pragma unroll
  for (int k = 0; k < NF - 1; k++) {
    o.lo[k] = *(const sell_d2u *)(x + (gu + T.bl[k]) + 2 * lane);
    o.hi[k] = *(const sell_d2u *)(x + (gu + T.bh[k]) + 2 * lane);
  }
  // the element in front of the wave's 128 centre operands (lanes 0..31 ask for it) and the one
  // behind them (lanes 32..63): every lane loads, two addresses per wave -- no divergent branch
  // around a load, whose join would make the compiler wait for everything in flight
  // (clamped: at the operator's first / last row the slot c-1 / c+1 is masked there and whatever is
  // read is dropped at the product)
  long long el = gu + T.bc - 1, er = gu + T.bc + (long long)LSB_SELL_ROWS; // wave-uniform
  el = el < 0 ? 0 : el, er = er >= (long long)xlen ? (long long)xlen - 1 : er;
  o.edge = x[lane < 32u ? el : er];
  if (DOT == 2)
    o.xd = *(const sell_d2v *)(xdot + lrow + 2 * lane);
}
template <int NF, int DOT>
__device__ __forceinline__ void col_compute(const col_ops<NF> &o, const sell_d2u &cm, const sell_d2u &c, const sell_d2u &cp,
                                            const col_tmpl<NF> &T, double vm0, double vm1, double vp0, double vp1,
                                            unsigned lane, double &a0, double &a1, double &dot) {
  double up = lane_above(c.y), dn = lane_below(c.x);
  if (lane == 0)
    up = o.edge;
  if (lane == 63)
    dn = o.edge;
  a0 = fma(T.k0, cm.x, 0.0), a1 = fma(T.k0, cm.y, 0.0); // slot 0: one plane down = the centre of the step before
#pragma unroll
  for (int k = 0; k < NF - 1; k++)
    a0 = fma(T.kl[k], o.lo[k].x, a0), a1 = fma(T.kl[k], o.lo[k].y, a1);
  // value 0 = padding: no operand, an exact 0 (the rule of k_spmv_sell16)
  a0 = fma(vm0, vm0 != 0.0 ? up : 0.0, a0), a1 = fma(vm1, vm1 != 0.0 ? c.x : 0.0, a1); // slot c-1
  a0 = fma(T.kc, c.x, a0), a1 = fma(T.kc, c.y, a1);                                     // slot c
  a0 = fma(vp0, vp0 != 0.0 ? c.y : 0.0, a0), a1 = fma(vp1, vp1 != 0.0 ? dn : 0.0, a1); // slot c+1
#pragma unroll
  for (int k = 0; k < NF - 1; k++)
    a0 = fma(T.kh[k], o.hi[k].x, a0), a1 = fma(T.kh[k], o.hi[k].y, a1);
  a0 = fma(T.kL, cp.x, a0), a1 = fma(T.kL, cp.y, a1); // last slot: one plane up = the centre of the step after
  if (DOT == 1)
    dot = fma(a0, c.x, dot), dot = fma(a1, c.y, dot);
  else if (DOT == 2)
    dot = fma(a0, o.xd.x, dot), dot = fma(a1, o.xd.y, dot);
}

#define LSB_COL_HEAD 16 // unsigneds in front of the items: xbeg[NXCD + 1]
// (a loaded dot operand is four more registers per step in flight: five workgroups per CU instead of six
// rather than spills -- the walk runs best from three or four anyway)
template <int NF, int DOT>
__global__ __launch_bounds__(WG, DOT == 2 ? 5 : 6) void k_spmv_tmpl_col(
    const unsigned *__restrict__ plan, unsigned period, unsigned n, unsigned row_begin, unsigned xlen,
    const unsigned *__restrict__ sptr, const unsigned long long *__restrict__ mask,
    const lsb_sell_tmpl *__restrict__ td, const int *__restrict__ sbase, const void *__restrict__ vals, int f32,
    const double *__restrict__ vconst, const double *__restrict__ x, double *__restrict__ y,
    const double *__restrict__ xdot, double *__restrict__ partials, const lsb_pcg_state *__restrict__ st,
    const lsb_ar_tail tail) {
  static_assert(NF >= 1 && NF <= 2, "one or two far slots per side");
  __shared__ double sred[4];
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const unsigned gx = gridDim.x / NXCD, xcd = blockIdx.x % NXCD, slot = blockIdx.x / NXCD;
  const unsigned i0 = plan[xcd], i1 = plan[xcd + 1];
  const u4v *__restrict__ items = (const u4v *)(plan + LSB_COL_HEAD);
  const int stopped = st ? st->status : 0; // tested behind the first loads
  const unsigned P = period * LSB_SELL_ROWS; // rows of a plane
  double dot = 0.0;
  for (unsigned g = slot; i0 + 4 * g < i1; g += gx) {
    const unsigned it = __builtin_amdgcn_readfirstlane(i0 + 4 * g + wave);
    if (it >= i1)
      continue;
    const u4v rec = items[it];
    const unsigned s = __builtin_amdgcn_readfirstlane(rec.x), Kw = __builtin_amdgcn_readfirstlane(rec.y);
    const unsigned K = Kw & ~LSB_TMPL_COL_LOCKSTEP;
    // lockstep: the four items of this turn are columns of one length (the host says so in bit 31 of
    // all four): a barrier per plane keeps the workgroup's requests together
    const bool lockstep = (Kw & LSB_TMPL_COL_LOCKSTEP) != 0;
    if (K >= 2) {
      const unsigned t = __builtin_amdgcn_readfirstlane(rec.z), mb = __builtin_amdgcn_readfirstlane(rec.w);
      const lsb_sell_tmpl *T = td + t;
      col_tmpl<NF> C;
      C.bc = T->base[NF + 1];
      C.k0 = T->cst[0], C.kc = T->cst[NF + 1], C.kL = T->cst[2 * NF + 2];
#pragma unroll
      for (int k = 0; k < NF - 1; k++) {
        C.bl[k] = T->base[1 + k], C.bh[k] = T->base[NF + 3 + k];
        C.kl[k] = T->cst[1 + k], C.kh[k] = T->cst[NF + 3 + k];
      }
      const int side_k[2] = {T->kidx[NF], T->kidx[NF + 2]}, side_kd[2] = {T->kind[NF], T->kind[NF + 2]};
      long long gu = (long long)s * LSB_SELL_ROWS + row_begin;
      unsigned lrow = s * LSB_SELL_ROWS;
      // prologue: the operands of step 0 -- the plane below, the centre, the plane above
      col_ops<NF> cur, nxt;
      sell_d2u cm = *(const sell_d2u *)(x + (gu + C.bc - (long long)P) + 2 * lane);
      sell_d2u c0 = *(const sell_d2u *)(x + (gu + C.bc) + 2 * lane);
      sell_d2u cp, cn;
      col_issue<NF, DOT>(cur, cp, C, x, xdot, gu, lrow, P, xlen, lane, true);
      // the column's masks (wave-uniform, scalar loads beside the gathers) and side values, once
      unsigned long long mk[2][2] = {{0ull, 0ull}, {0ull, 0ull}};
#pragma unroll
      for (int side = 0; side < 2; side++)
        if (side_k[side] >= 0 && side_kd[side] == 2) {
          const unsigned long long *mp = mask + 2 * ((size_t)mb + (unsigned)side_k[side]);
          mk[side][0] = mp[0], mk[side][1] = mp[1];
        }
      double vm0 = T->cst[NF], vm1 = vm0, vp0 = T->cst[NF + 2], vp1 = vp0;
      if (side_k[0] >= 0 || side_k[1] >= 0)
        tmpl_side_values(T, NF, 0u, lane, side_k, side_kd, mk, vals, f32, vm0, vm1, vp0, vp1);
      if (stopped)
        return;
      // step 0 (K >= 2: there is a next one), steps 1 .. K-2, step K-1
      col_issue<NF, DOT>(nxt, cn, C, x, xdot, gu + (long long)P, lrow + P, P, xlen, lane, true);
      double a0, a1;
      col_compute<NF, DOT>(cur, cm, c0, cp, C, vm0, vm1, vp0, vp1, lane, a0, a1, dot);
      cm = c0, c0 = cp, cp = cn, cur = nxt;
      for (unsigned k = 1; k + 1 < K; k++) {
        gu += (long long)P, lrow += P;
        if (lockstep)
          __builtin_amdgcn_s_barrier();
        col_issue<NF, DOT>(nxt, cn, C, x, xdot, gu + (long long)P, lrow + P, P, xlen, lane, true);
        {
          const sell_d2v o = {a0, a1};
          *(sell_d2v *)(y + (lrow - P) + 2 * lane) = o;
        }
        col_compute<NF, DOT>(cur, cm, c0, cp, C, vm0, vm1, vp0, vp1, lane, a0, a1, dot);
        cm = c0, c0 = cp, cp = cn, cur = nxt;
      }
      gu += (long long)P, lrow += P;
      {
        const sell_d2v o = {a0, a1};
        *(sell_d2v *)(y + (lrow - P) + 2 * lane) = o;
      }
      col_compute<NF, DOT>(cur, cm, c0, cp, C, vm0, vm1, vp0, vp1, lane, a0, a1, dot);
      {
        const sell_d2v o = {a0, a1};
        *(sell_d2v *)(y + lrow + 2 * lane) = o;
      }
    } else { // one slice, slot by slot off the slot records (k_spmv_tmpl's way for a slice without a template)
      const unsigned row = s * LSB_SELL_ROWS + 2 * lane;
      const int grow = (int)(row + row_begin);
      const unsigned q0 = sptr[s] / LSB_SELL_ROWS, len = (sptr[s + 1] - sptr[s]) / LSB_SELL_ROWS;
      sell_d2v xd = {0.0, 0.0};
      if (DOT) {
        if (row + 1 < n)
          xd = *(const sell_d2v *)(xdot + row);
        else if (row < n)
          xd.x = xdot[row];
      }
      if (stopped)
        return;
      double a0 = 0.0, a1 = 0.0;
      for (unsigned j = 0; j < len; j++) {
        const i4v r = ((const i4v *)sbase)[q0 + j]; // {base, -1 (no codes where templates exist), value slot or -1, 0}
        if (r.z < 0) {
          const double k = vconst[q0 + j];
          const sell_d2u v = *(const sell_d2u *)(x + (grow + r.x));
          a0 = fma(k, v.x, a0), a1 = fma(k, v.y, a1);
        } else {
          double v0, v1;
          if (f32) {
            const vt2<float>::type v = *((const vt2<float>::type *)((const float *)vals + (size_t)r.z * LSB_SELL_ROWS) + lane);
            v0 = (double)v.x, v1 = (double)v.y;
          } else {
            const sell_d2v v = *((const sell_d2v *)((const double *)vals + (size_t)r.z * LSB_SELL_ROWS) + lane);
            v0 = v.x, v1 = v.y;
          }
          const bool p0 = v0 != 0.0, p1 = v1 != 0.0; // padding: no gather, an exact 0
          const double t0 = x[p0 ? grow + r.x : 0], t1 = x[p1 ? grow + 1 + r.x : 0];
          a0 = fma(v0, p0 ? t0 : 0.0, a0), a1 = fma(v1, p1 ? t1 : 0.0, a1);
        }
      }
      if (row + 1 < n) {
        const sell_d2v o = {a0, a1};
        *(sell_d2v *)(y + row) = o;
        if (DOT)
          dot = fma(a0, xd.x, dot), dot = fma(a1, xd.y, dot);
      } else if (row < n) {
        y[row] = a0;
        if (DOT)
          dot = fma(a0, xd.x, dot);
      }
    }
  }
  if (stopped)
    return;
  spmv_publish(partials, dot, sred, tail);
}

// --------------------------------------------------------------------------
// a2-5 on a z-column plan: the classic PCG iteration in TWO launches and 60 instead of 88 bytes per
// row (one shard, constant Jacobi diagonal dc).
//   [k_pcg_col_px]   (r.z, r.r) of the sweep before -> stop test, beta;  alpha, alpha2 = the steps of the two
//                    iterations before (st->alpha[0], [1], left there by k_pcg_col_r);  per row
//                      p' = dc r + beta p        (pnew_of: k_pcg_update_p's expression)
//                      q  = S p' -- NOT stored: only the partials of p'.q leave the launch
//                      every SECOND iteration of a run (XUPD):  x += alpha2 p'' + alpha p  -- the x half of
//                        k_pcg_update_xr, two iterations' worth: p is in registers here anyway, and p'', the
//                        direction before it, is what the buffer p' goes to still holds
//                    -- reads r, p (+ x, p'') and writes p' (the OTHER direction buffer) (+ x): 24 / 48 B per row
//   [k_pcg_col_r]    alpha = r.z / p'.q;  r -= alpha (S p') with S p' formed AGAIN by the same walk (the same
//                    operands and products in the same order: the same bits);  partials of (r.z', r.r)
//                    -- reads p', r and writes r: 24 B per row
// against k_spmv_tmpl_col (16) + k_pcg_update_xr (48) + k_pcg_update_p (24): the stencil is applied twice and
// q = S p never travels.  What makes the fold pay
// where k_spmv_tmpl_p's did not (round 3: every gathered operand of every slice formed p' anew, five
// or seven times per row): a column forms p' ONCE per plane and row and keeps it in registers for the
// three steps that use it as the plane above, the centre and the plane below; only the +-line
// operands of a 3-D stencil (NF = 2) are formed a second time, out of r and p lines that sit in L2.
// Ownership: a column loads, updates and stores x and p' for ITS planes only; the plane below its
// first and above its last slice belong to other columns -- p' is formed for them, nothing stored.
// What is pending at the end of a run (one update after an even iteration, two after an odd one) has no
// k_pcg_col_px behind it: st->xpend (set by k_pcg_col_r, cleared by an updating launch here) says so and k_pcg_xfix
// applies it (hip_pcg.c).  maxit: the launch
// that counts the maxit-th iteration sets st->pad and does all of its work (x included); the next
// k_pcg_col_r turns that into the status -- never set and tested in the same launch.
// Pipeline of a column: a step issues the NEXT step's loads, waits for its own operands (older) and stores
// x, p' right away -- younger than what the next step waits for.
// --------------------------------------------------------------------------
template <int NF> struct colp_c { // centre-side loads of one step, issued ONE step ahead
  sell_d2u rn, pn;                // r, p of the plane ahead (it becomes this step's "plane above")
  sell_d2v xn, sn;                // x of that plane and the direction of TWO iterations ago (what the buffer p' goes to
                                  // still holds), where the column owns the plane and this launch updates x
  double re, pe;                  // r, p of the element in front of / behind the wave's 128 centre operands
};
// the +-line operands of a step (3-D stencil), issued TWO steps ahead: the lines a column gathers for plane z are
// the centre lines of the columns three slices to either side, which fetch them one step before they use them as
// "the plane ahead".  Asked for at that same time the request meets theirs in L2; asked for a step later (as in
// k_spmv_tmpl_col, whose steps are a quarter as long) half of them had left it again: 20.98 M fabric read requests
// per launch on the 64 M-row 7-point operator where 13 M are compulsory (profiles/r04_px.txt)
template <int NF> struct colp_m {
  sell_d2u rlo[NF > 1 ? NF - 1 : 1], plo[NF > 1 ? NF - 1 : 1], rhi[NF > 1 ? NF - 1 : 1], phi[NF > 1 ? NF - 1 : 1];
};
struct colp_out { // results of a step
  sell_d2v q, x, p;
};
// gu: global row of lane 0's first row of the step's slice (one shard: = its local row)
template <int NF, bool WITH_X, int NT>
__device__ __forceinline__ void colp_issue_c(colp_c<NF> &o, const col_tmpl<NF> &T, const double *__restrict__ r,
                                             const double *__restrict__ pold, const double *__restrict__ x,
                                             const double *pstale, long long gu, unsigned P, unsigned xlen,
                                             unsigned lane) {
  const long long ga = gu + T.bc + (long long)P; // the plane ahead
  o.rn = *(const sell_d2u *)(r + ga + 2 * lane);
  o.pn = *(const sell_d2u *)(pold + ga + 2 * lane);
  if (WITH_X) { // x and the stale direction are touched once per launch: nontemporal (NT & 1) keeps them out of the
                // way of the r / p lines the neighbouring columns gather again
    if (NT & 1) {
      o.xn = __builtin_nontemporal_load((const sell_d2v *)(x + (gu + (long long)P) + 2 * lane));
      o.sn = __builtin_nontemporal_load((const sell_d2v *)(pstale + (gu + (long long)P) + 2 * lane));
    } else {
      o.xn = *(const sell_d2v *)(x + (gu + (long long)P) + 2 * lane);
      o.sn = *(const sell_d2v *)(pstale + (gu + (long long)P) + 2 * lane);
    }
  }
  long long el = gu + T.bc - 1, er = gu + T.bc + (long long)LSB_SELL_ROWS; // wave-uniform, clamped (see col_issue)
  el = el < 0 ? 0 : el, er = er >= (long long)xlen ? (long long)xlen - 1 : er;
  const long long e = lane < 32u ? el : er;
  o.re = r[e], o.pe = pold[e];
}
template <int NF>
__device__ __forceinline__ void colp_issue_m(colp_m<NF> &o, const col_tmpl<NF> &T, const double *__restrict__ r,
                                             const double *__restrict__ pold, long long gu, unsigned lane) {
#pragma unroll
  for (int k = 0; k < NF - 1; k++) {
    o.rlo[k] = *(const sell_d2u *)(r + (gu + T.bl[k]) + 2 * lane);
    o.plo[k] = *(const sell_d2u *)(pold + (gu + T.bl[k]) + 2 * lane);
    o.rhi[k] = *(const sell_d2u *)(r + (gu + T.bh[k]) + 2 * lane);
    o.phi[k] = *(const sell_d2u *)(pold + (gu + T.bh[k]) + 2 * lane);
  }
}
// one step: q of the centre plane from (below, centre, above) = (pm, p0, p' of the plane ahead); OWNED: the
// plane ahead is the column's own -- its x is updated and it and p' go to `out`
// XUPD: this launch applies the two pending x updates (alpha of the iteration before with its direction p, alpha2 of
// the one before that with the stale direction)
template <int NF, bool OWNED, bool XUPD>
__device__ __forceinline__ void colp_compute(const colp_c<NF> &o, const colp_m<NF> &m, const sell_d2u &pm,
                                             const sell_d2u &p0, sell_d2u &pp, const col_tmpl<NF> &T, double vm0,
                                             double vm1, double vp0, double vp1, double beta, double alpha, double alpha2,
                                             double dc, unsigned lane, colp_out &out, double &dot) {
  pp.x = pnew_of(dc, o.rn.x, beta, o.pn.x), pp.y = pnew_of(dc, o.rn.y, beta, o.pn.y);
  if (OWNED) {
    if (XUPD) { // x += alpha2 p'' + alpha p: k_pcg_update_xr's x += alpha p, two iterations' worth in their order
      out.x.x = (o.xn.x + alpha2 * o.sn.x) + alpha * o.pn.x, out.x.y = (o.xn.y + alpha2 * o.sn.y) + alpha * o.pn.y;
    }
    out.p.x = pp.x, out.p.y = pp.y;
  }
  const double edge = pnew_of(dc, o.re, beta, o.pe);
  double up = lane_above(p0.y), dn = lane_below(p0.x);
  if (lane == 0)
    up = edge;
  if (lane == 63)
    dn = edge;
  double a0 = fma(T.k0, pm.x, 0.0), a1 = fma(T.k0, pm.y, 0.0);
#pragma unroll
  for (int k = 0; k < NF - 1; k++) {
    const double l0 = pnew_of(dc, m.rlo[k].x, beta, m.plo[k].x), l1 = pnew_of(dc, m.rlo[k].y, beta, m.plo[k].y);
    a0 = fma(T.kl[k], l0, a0), a1 = fma(T.kl[k], l1, a1);
  }
  a0 = fma(vm0, vm0 != 0.0 ? up : 0.0, a0), a1 = fma(vm1, vm1 != 0.0 ? p0.x : 0.0, a1);
  a0 = fma(T.kc, p0.x, a0), a1 = fma(T.kc, p0.y, a1);
  a0 = fma(vp0, vp0 != 0.0 ? p0.y : 0.0, a0), a1 = fma(vp1, vp1 != 0.0 ? dn : 0.0, a1);
#pragma unroll
  for (int k = 0; k < NF - 1; k++) {
    const double h0 = pnew_of(dc, m.rhi[k].x, beta, m.phi[k].x), h1 = pnew_of(dc, m.rhi[k].y, beta, m.phi[k].y);
    a0 = fma(T.kh[k], h0, a0), a1 = fma(T.kh[k], h1, a1);
  }
  a0 = fma(T.kL, pp.x, a0), a1 = fma(T.kL, pp.y, a1);
  out.q.x = a0, out.q.y = a1;
  dot = fma(a0, p0.x, dot), dot = fma(a1, p0.y, dot);
}
// one slice, slot by slot off the slot records, every operand's p' formed on the fly (first / last
// plane, ragged ends, columns of two)
__device__ __forceinline__ void colp_single(unsigned s, unsigned n, unsigned lane, const unsigned *__restrict__ sptr,
                                            const int *__restrict__ sbase, const double *__restrict__ vals,
                                            const double *__restrict__ vconst, const double *__restrict__ r,
                                            const double *__restrict__ pold, double *pnew,
                                            double *__restrict__ x, double *__restrict__ q, double beta, double alpha,
                                            double alpha2, bool xupd, double dc, double &dot) {
  const unsigned row = s * LSB_SELL_ROWS + 2 * lane;
  const unsigned q0 = sptr[s] / LSB_SELL_ROWS, len = (sptr[s + 1] - sptr[s]) / LSB_SELL_ROWS;
  const bool l0 = row < n, l1 = row + 1 < n;
  const double r0 = l0 ? r[row] : 0.0, r1 = l1 ? r[row + 1] : 0.0;
  const double o0 = l0 ? pold[row] : 0.0, o1 = l1 ? pold[row + 1] : 0.0;
  const double x0 = xupd && l0 ? x[row] : 0.0, x1 = xupd && l1 ? x[row + 1] : 0.0;
  const double s0 = xupd && l0 ? pnew[row] : 0.0, s1 = xupd && l1 ? pnew[row + 1] : 0.0; // the direction of two iterations ago
  double a0 = 0.0, a1 = 0.0;
  for (unsigned j = 0; j < len; j++) {
    const i4v rec = ((const i4v *)sbase)[q0 + j]; // {base, -1 (no codes where templates exist), value slot or -1, 0}
    if (rec.z < 0) { // a constant slot holds 128 real entries (lsb_tmpl_check): unguarded pairs
      const double k = vconst[q0 + j];
      const sell_d2u rv = *(const sell_d2u *)(r + ((int)row + rec.x)), pv = *(const sell_d2u *)(pold + ((int)row + rec.x));
      a0 = fma(k, pnew_of(dc, rv.x, beta, pv.x), a0), a1 = fma(k, pnew_of(dc, rv.y, beta, pv.y), a1);
    } else {
      const sell_d2v v = *((const sell_d2v *)(vals + (size_t)rec.z * LSB_SELL_ROWS) + lane);
      const bool p0 = v.x != 0.0, p1 = v.y != 0.0; // padding: no gather, an exact 0
      const int c0 = p0 ? (int)row + rec.x : 0, c1 = p1 ? (int)row + 1 + rec.x : 0;
      const double t0 = pnew_of(dc, r[c0], beta, pold[c0]), t1 = pnew_of(dc, r[c1], beta, pold[c1]);
      a0 = fma(v.x, p0 ? t0 : 0.0, a0), a1 = fma(v.y, p1 ? t1 : 0.0, a1);
    }
  }
  const double n0 = pnew_of(dc, r0, beta, o0), n1 = pnew_of(dc, r1, beta, o1);
  if (l0) {
    pnew[row] = n0;
    if (xupd)
      x[row] = (x0 + alpha2 * s0) + alpha * o0;
    if (q)
      q[row] = a0;
    dot = fma(a0, n0, dot);
  }
  if (l1) {
    pnew[row + 1] = n1;
    if (xupd)
      x[row + 1] = (x1 + alpha2 * s1) + alpha * o1;
    if (q)
      q[row + 1] = a1;
    dot = fma(a1, n1, dot);
  }
}

#define COLP_ST(PTR, V, BIT)                                                                   \
  do {                                                                                         \
    if (NT & (BIT))                                                                            \
      __builtin_nontemporal_store((V), (sell_d2v *)(PTR));                                     \
    else                                                                                       \
      *(sell_d2v *)(PTR) = (V);                                                                \
  } while (0)
#define COLP_STORE(OUT, LROW, WITH_Q, WITH_XP)                                                 \
  do {                                                                                         \
    if (WITH_Q)                                                                                \
      COLP_ST(q + (LROW) + 2 * lane, (OUT).q, 2);                                              \
    if (WITH_XP) {                                                                             \
      if (XUPD)                                                                                \
        COLP_ST(x + ((LROW) + P) + 2 * lane, (OUT).x, 1);                                      \
      COLP_ST(pnew + ((LROW) + P) + 2 * lane, (OUT).p, 2);                                     \
    }                                                                                          \
  } while (0)

// NT: bit 0 x loaded and stored nontemporal, bit 1 p' and q stored nontemporal
// q = S p' is NOT stored: only its dot with p' leaves the launch -- k_pcg_col_r forms the same q again out of p' when
// it updates r, and the vector never travels
// XUPD: x is updated every SECOND iteration of a run, with two directions at once -- x += alpha2 p'' + alpha p: p is
// the direction this launch reads anyway, p'' the one of the iteration before it, which is what the buffer p' goes
// to still holds (each lane reads its own rows' p'' just before it overwrites them; nobody else reads that
// buffer in this launch).  Launches of odd run index leave x alone: x is read and written half as often.
template <int NF, int NT, bool XUPD>
__global__ __launch_bounds__(WG, NF == 2 ? 3 : 4) void k_pcg_col_px(
    const unsigned *__restrict__ plan, unsigned period, unsigned n, const unsigned *__restrict__ sptr,
    const unsigned long long *__restrict__ mask, const lsb_sell_tmpl *__restrict__ td, const int *__restrict__ sbase,
    const double *__restrict__ vals, const double *__restrict__ vconst, const double *__restrict__ r,
    const double *__restrict__ pold, double *pnew, double *__restrict__ x, double dc,
    double *__restrict__ partials, lsb_pcg_state *__restrict__ st, int parity,
    const double *__restrict__ parts2, unsigned nparts2) {
  static_assert(NF >= 1 && NF <= 2, "one or two far slots per side");
  __shared__ double sred[8];
  double *const q = nullptr; // (COLP_STORE's q leg is compiled out)
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const unsigned gx = gridDim.x / NXCD, xcd = blockIdx.x % NXCD, slot = blockIdx.x / NXCD;
  // ---- k_pcg_update_p's prologue: the sweep's partial sums -> r.z, r.r, stop test, beta
  const int stopped = st->status;
  const double rz_old = st->rz[parity], thresh2 = st->thresh2, alpha = st->alpha[0], alpha2 = st->alpha[1];
  const unsigned i0 = plan[xcd], i1 = plan[xcd + 1];
  double v[2];
  wg_sum_partials<2>(parts2, nparts2, v, sred);
  if (stopped)
    return;
  const double rz_new = v[0], rr = v[1];
  const bool conv = rr <= thresh2;
  if (blockIdx.x == 0 && threadIdx.x == 0) { // only this thread touches these words in this launch
    const int it = st->iters + 1;
    st->iters = it;
    st->rr = rr;
    st->rz[parity ^ 1] = rz_new;
    if (conv)
      st->status = LSB_STATUS_CONVERGED; // (the pending x updates stay pending: k_pcg_xfix)
    else {
      if (XUPD)
        st->xpend = 0; // this launch applies them
      if (it >= st->maxit)
        st->pad = 1; // the next k_pcg_col_r makes it the status; this launch still does all its work
    }
  }
  if (conv)
    return;
  const double beta = rz_new / rz_old;
  const u4v *__restrict__ items = (const u4v *)(plan + LSB_COL_HEAD);
  const unsigned P = period * LSB_SELL_ROWS;
  double dot = 0.0;
  for (unsigned g = slot; i0 + 4 * g < i1; g += gx) {
    const unsigned it = __builtin_amdgcn_readfirstlane(i0 + 4 * g + wave);
    if (it >= i1)
      continue;
    const u4v rec = items[it];
    const unsigned s = __builtin_amdgcn_readfirstlane(rec.x);
    const unsigned K = __builtin_amdgcn_readfirstlane(rec.y) & ~LSB_TMPL_COL_LOCKSTEP;
    if (K >= 3) {
      const unsigned t = __builtin_amdgcn_readfirstlane(rec.z), mb = __builtin_amdgcn_readfirstlane(rec.w);
      const lsb_sell_tmpl *T = td + t;
      col_tmpl<NF> C;
      C.bc = T->base[NF + 1];
      C.k0 = T->cst[0], C.kc = T->cst[NF + 1], C.kL = T->cst[2 * NF + 2];
#pragma unroll
      for (int k = 0; k < NF - 1; k++) {
        C.bl[k] = T->base[1 + k], C.bh[k] = T->base[NF + 3 + k];
        C.kl[k] = T->cst[1 + k], C.kh[k] = T->cst[NF + 3 + k];
      }
      const int side_k[2] = {T->kidx[NF], T->kidx[NF + 2]}, side_kd[2] = {T->kind[NF], T->kind[NF + 2]};
      long long gu = (long long)s * LSB_SELL_ROWS;
      unsigned lrow = s * LSB_SELL_ROWS;
      // prologue: r, p of the plane below (another column's) and of the first plane with its x; step 0's loads
      const long long gc = gu + C.bc;
      const sell_d2u rb = *(const sell_d2u *)(r + (gc - (long long)P) + 2 * lane);
      const sell_d2u pb = *(const sell_d2u *)(pold + (gc - (long long)P) + 2 * lane);
      const sell_d2u r0 = *(const sell_d2u *)(r + gc + 2 * lane), o0 = *(const sell_d2u *)(pold + gc + 2 * lane);
      sell_d2v x0 = {0.0, 0.0}, s0 = {0.0, 0.0};
      if (XUPD) // (a compile-time branch: no join in front of the loads)
        x0 = *(const sell_d2v *)(x + lrow + 2 * lane), s0 = *(const sell_d2v *)(pnew + lrow + 2 * lane);
      colp_c<NF> c0, c1;
      colp_m<NF> m0, m1, m2;
      colp_issue_c<NF, XUPD, NT>(c0, C, r, pold, x, pnew, gu, P, n, lane);
      colp_issue_m<NF>(m0, C, r, pold, gu, lane);
      colp_issue_m<NF>(m1, C, r, pold, gu + (long long)P, lane);
      unsigned long long mk[2][2] = {{0ull, 0ull}, {0ull, 0ull}};
#pragma unroll
      for (int side = 0; side < 2; side++)
        if (side_k[side] >= 0 && side_kd[side] == 2) {
          const unsigned long long *mp = mask + 2 * ((size_t)mb + (unsigned)side_k[side]);
          mk[side][0] = mp[0], mk[side][1] = mp[1];
        }
      double vm0 = T->cst[NF], vm1 = vm0, vp0 = T->cst[NF + 2], vp1 = vp0;
      if (side_k[0] >= 0 || side_k[1] >= 0)
        tmpl_side_values(T, NF, 0u, lane, side_k, side_kd, mk, vals, 0, vm0, vm1, vp0, vp1);
      sell_d2u pm, p0, pp;
      pm.x = pnew_of(dc, rb.x, beta, pb.x), pm.y = pnew_of(dc, rb.y, beta, pb.y);
      p0.x = pnew_of(dc, r0.x, beta, o0.x), p0.y = pnew_of(dc, r0.y, beta, o0.y);
      // A step issues the NEXT step's loads, waits for its own (older) and stores its results right away: the stores
      // are younger than the loads the next step waits for, so they are never waited for either (vmcnt counts
      // loads and stores in one order) -- no parking of results across a step.
      colp_out res;
      res.x.x = (x0.x + alpha2 * s0.x) + alpha * o0.x, res.x.y = (x0.y + alpha2 * s0.y) + alpha * o0.y;
      res.p.x = p0.x, res.p.y = p0.y;
      res.q = res.p; // (not stored)
      COLP_STORE(res, lrow - P, false, true); // the first plane's x and p'
      // steps 0 .. K-3 (K >= 3): the plane after next is the column's own
      for (unsigned k = 0; k + 2 < K; k++) {
        colp_issue_c<NF, XUPD, NT>(c1, C, r, pold, x, pnew, gu + (long long)P, P, n, lane);
        colp_issue_m<NF>(m2, C, r, pold, gu + 2 * (long long)P, lane);
        colp_compute<NF, true, XUPD>(c0, m0, pm, p0, pp, C, vm0, vm1, vp0, vp1, beta, alpha, alpha2, dc, lane, res, dot);
        COLP_STORE(res, lrow, false, true);
        pm = p0, p0 = pp, c0 = c1, m0 = m1, m1 = m2;
        gu += (long long)P, lrow += P;
      }
      // step K-2: the plane after next is the one above the column (no x, no +-line operands to ask for)
      colp_issue_c<NF, false, NT>(c1, C, r, pold, x, pnew, gu + (long long)P, P, n, lane);
      colp_compute<NF, true, XUPD>(c0, m0, pm, p0, pp, C, vm0, vm1, vp0, vp1, beta, alpha, alpha2, dc, lane, res, dot);
      COLP_STORE(res, lrow, false, true);
      pm = p0, p0 = pp, c0 = c1, m0 = m1;
      gu += (long long)P, lrow += P;
      // step K-1: nothing to load; the plane ahead is not the column's
      colp_compute<NF, false, XUPD>(c0, m0, pm, p0, pp, C, vm0, vm1, vp0, vp1, beta, alpha, alpha2, dc, lane, res, dot);
      COLP_STORE(res, lrow, false, false);
    } else {
      for (unsigned k = 0; k < K; k++)
        colp_single(s + k * period, n, lane, sptr, sbase, vals, vconst, r, pold, pnew, x, nullptr, beta, alpha, alpha2,
                    XUPD, dc, dot);
    }
  }
  if (partials) {
    double d[1] = {dot};
    wg_sum<1>(d, sred);
    if (threadIdx.x == 0)
      partials[blockIdx.x] = d[0];
  }
}
#undef COLP_STORE
#undef COLP_ST

// The r half of the first sweep WITHOUT q: alpha = r.z / p.q (p.q from k_pcg_col_px's partials), then per row
// r -= alpha (S p) with S p formed again by the z-column walk of k_spmv_tmpl_col -- the same operands and products in
// the same order as in k_pcg_col_px, so the same bits -- and the partials of (r.z', r.r).  Reads p and r, writes r:
// three passes, as a sweep over a stored q and r would make, but k_pcg_col_px need not write q.  Leaves alpha and "x is one update behind, the direction is in buffer pbuf" (xpend) in
// the state and promotes a pending maxit (st->pad) to the status.
template <int NF>
__global__ __launch_bounds__(WG, 5) void k_pcg_col_r(
    const unsigned *__restrict__ plan, unsigned period, unsigned n, const unsigned *__restrict__ sptr,
    const unsigned long long *__restrict__ mask, const lsb_sell_tmpl *__restrict__ td, const int *__restrict__ sbase,
    const double *__restrict__ vals, const double *__restrict__ vconst, const double *__restrict__ p,
    double *__restrict__ r, double dc, lsb_pcg_state *__restrict__ st, int parity, int pbuf, int xtwo,
    const double *__restrict__ pq_parts, unsigned npq, double *__restrict__ partials2) {
  static_assert(NF >= 1 && NF <= 2, "one or two far slots per side");
  __shared__ double sred[8];
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const unsigned gx = gridDim.x / NXCD, xcd = blockIdx.x % NXCD, slot = blockIdx.x / NXCD;
  const int stopped = st->status, pad = st->pad;
  const double rz = st->rz[parity];
  const unsigned i0 = plan[xcd], i1 = plan[xcd + 1];
  double pqv[1];
  wg_sum_partials<1>(pq_parts, npq, pqv, sred);
  if (stopped)
    return;
  if (pad) { // the launch before counted the maxit-th iteration (and did its x update): stop here
    if (blockIdx.x == 0 && threadIdx.x == 0)
      st->status = LSB_STATUS_MAXIT;
    return;
  }
  const double pq = pqv[0];
  if (!(pq != 0.0) || !isfinite(pq)) { // same decision in every workgroup
    if (blockIdx.x == 0 && threadIdx.x == 0)
      st->status = LSB_STATUS_BREAKDOWN;
    return;
  }
  const double alpha = rz / pq;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // x is now one update behind (the run's even iterations: k_pcg_col_px has just applied the two before) or two
    // (odd ones: the launch before left x alone): xpend = 1 + pbuf / 3 + pbuf, the steps in alpha[0] and alpha[1]
    st->pq = pq, st->alpha[1] = st->alpha[0], st->alpha[0] = alpha, st->xpend = (xtwo ? 3 : 1) + pbuf;
  }
  const u4v *__restrict__ items = (const u4v *)(plan + LSB_COL_HEAD);
  const unsigned P = period * LSB_SELL_ROWS;
  double acc[2] = {0.0, 0.0}, nodot = 0.0;
#define COLR_FINISH(A0, A1, RV, LROW)                                                          \
  do {                                                                                         \
    sell_d2v rn_;                                                                              \
    rn_.x = (RV).x - alpha * (A0), rn_.y = (RV).y - alpha * (A1);                              \
    *(sell_d2v *)(r + (LROW) + 2 * lane) = rn_;                                                \
    acc[0] += rn_.x * (dc * rn_.x);                                                            \
    acc[0] += rn_.y * (dc * rn_.y);                                                            \
    acc[1] += rn_.x * rn_.x;                                                                   \
    acc[1] += rn_.y * rn_.y;                                                                   \
  } while (0)
  for (unsigned g = slot; i0 + 4 * g < i1; g += gx) {
    const unsigned it = __builtin_amdgcn_readfirstlane(i0 + 4 * g + wave);
    if (it >= i1)
      continue;
    const u4v rec = items[it];
    const unsigned s = __builtin_amdgcn_readfirstlane(rec.x);
    const unsigned K = __builtin_amdgcn_readfirstlane(rec.y) & ~LSB_TMPL_COL_LOCKSTEP;
    if (K >= 3) { // (columns of two go slice by slice, as in k_pcg_col_px: the same sums either way)
      const unsigned t = __builtin_amdgcn_readfirstlane(rec.z), mb = __builtin_amdgcn_readfirstlane(rec.w);
      const lsb_sell_tmpl *T = td + t;
      col_tmpl<NF> C;
      C.bc = T->base[NF + 1];
      C.k0 = T->cst[0], C.kc = T->cst[NF + 1], C.kL = T->cst[2 * NF + 2];
#pragma unroll
      for (int k = 0; k < NF - 1; k++) {
        C.bl[k] = T->base[1 + k], C.bh[k] = T->base[NF + 3 + k];
        C.kl[k] = T->cst[1 + k], C.kh[k] = T->cst[NF + 3 + k];
      }
      const int side_k[2] = {T->kidx[NF], T->kidx[NF + 2]}, side_kd[2] = {T->kind[NF], T->kind[NF + 2]};
      long long gu = (long long)s * LSB_SELL_ROWS;
      unsigned lrow = s * LSB_SELL_ROWS;
      col_ops<NF> cur, nxt;
      sell_d2u cm = *(const sell_d2u *)(p + (gu + C.bc - (long long)P) + 2 * lane);
      sell_d2u c0 = *(const sell_d2u *)(p + (gu + C.bc) + 2 * lane);
      sell_d2u cp, cn;
      col_issue<NF, 2>(cur, cp, C, p, r, gu, lrow, P, n, lane, true); // (the "dot operand" slot carries r's pair)
      unsigned long long mk[2][2] = {{0ull, 0ull}, {0ull, 0ull}};
#pragma unroll
      for (int side = 0; side < 2; side++)
        if (side_k[side] >= 0 && side_kd[side] == 2) {
          const unsigned long long *mp = mask + 2 * ((size_t)mb + (unsigned)side_k[side]);
          mk[side][0] = mp[0], mk[side][1] = mp[1];
        }
      double vm0 = T->cst[NF], vm1 = vm0, vp0 = T->cst[NF + 2], vp1 = vp0;
      if (side_k[0] >= 0 || side_k[1] >= 0)
        tmpl_side_values(T, NF, 0u, lane, side_k, side_kd, mk, vals, 0, vm0, vm1, vp0, vp1);
      for (unsigned k = 0; k + 1 < K; k++) {
        col_issue<NF, 2>(nxt, cn, C, p, r, gu + (long long)P, lrow + P, P, n, lane, true);
        double a0, a1;
        col_compute<NF, 0>(cur, cm, c0, cp, C, vm0, vm1, vp0, vp1, lane, a0, a1, nodot);
        COLR_FINISH(a0, a1, cur.xd, lrow);
        cm = c0, c0 = cp, cp = cn, cur = nxt;
        gu += (long long)P, lrow += P;
      }
      double a0, a1;
      col_compute<NF, 0>(cur, cm, c0, cp, C, vm0, vm1, vp0, vp1, lane, a0, a1, nodot);
      COLR_FINISH(a0, a1, cur.xd, lrow);
    } else {
      for (unsigned k = 0; k < K; k++) { // one slice, slot by slot off the slot records
        const unsigned sl = s + k * period, row = sl * LSB_SELL_ROWS + 2 * lane;
        const unsigned q0 = sptr[sl] / LSB_SELL_ROWS, len = (sptr[sl + 1] - sptr[sl]) / LSB_SELL_ROWS;
        const bool l0 = row < n, l1 = row + 1 < n;
        const double r0 = l0 ? r[row] : 0.0, r1 = l1 ? r[row + 1] : 0.0;
        double a0 = 0.0, a1 = 0.0;
        for (unsigned j = 0; j < len; j++) {
          const i4v rec2 = ((const i4v *)sbase)[q0 + j];
          if (rec2.z < 0) {
            const double kk = vconst[q0 + j];
            const sell_d2u pv = *(const sell_d2u *)(p + ((int)row + rec2.x));
            a0 = fma(kk, pv.x, a0), a1 = fma(kk, pv.y, a1);
          } else {
            const sell_d2v v = *((const sell_d2v *)(vals + (size_t)rec2.z * LSB_SELL_ROWS) + lane);
            const bool p0 = v.x != 0.0, p1 = v.y != 0.0; // padding: no gather, an exact 0
            const double t0 = p[p0 ? (int)row + rec2.x : 0], t1 = p[p1 ? (int)row + 1 + rec2.x : 0];
            a0 = fma(v.x, p0 ? t0 : 0.0, a0), a1 = fma(v.y, p1 ? t1 : 0.0, a1);
          }
        }
        if (l0) {
          const double rn0 = r0 - alpha * a0;
          r[row] = rn0;
          acc[0] += rn0 * (dc * rn0), acc[1] += rn0 * rn0;
        }
        if (l1) {
          const double rn1 = r1 - alpha * a1;
          r[row + 1] = rn1;
          acc[0] += rn1 * (dc * rn1), acc[1] += rn1 * rn1;
        }
      }
    }
  }
#undef COLR_FINISH
  wg_sum<2>(acc, sred);
  if (threadIdx.x == 0) {
    partials2[2 * blockIdx.x + 0] = acc[0];
    partials2[2 * blockIdx.x + 1] = acc[1];
  }
}

// the x update a run's last k_pcg_col_r left pending (no k_pcg_col_px came behind it, or that one
// found the solve converged): x += alpha p with p in the buffer st->xpend names.  Runs whatever the
// status; the stand-alone k_pcg_update_p behind it clears st->xpend.
__global__ __launch_bounds__(WG) void k_pcg_xfix(unsigned n, const double *__restrict__ p0, const double *__restrict__ p1,
                                                 double *__restrict__ x, const lsb_pcg_state *__restrict__ st) {
  const int pend = st->xpend;
  if (!pend)
    return;
  const double alpha = st->alpha[0], alpha2 = st->alpha[1];
  if (pend <= 2) { // one update behind: the direction is in buffer pend - 1
    const double *__restrict__ p = pend == 1 ? p0 : p1;
    for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG)
      x[i] += alpha * p[i];
  } else { // two: the last direction in buffer pend - 3, the one before it in the other
    const double *__restrict__ p = pend == 3 ? p0 : p1, *__restrict__ pp = pend == 3 ? p1 : p0;
    for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += (size_t)gridDim.x * WG)
      x[i] = (x[i] + alpha2 * pp[i]) + alpha * p[i];
  }
}

// (Round 4, measured and taken out again: k_spmv_tmpl_deep -- a wave takes 2 or 4 of its turns AT ONCE,
// all their slice records in one batch of scalar loads, all 10 / 20 gathers, edge elements and mask
// words in flight together, the stores of the group before behind them; 90 / 136 VGPRs, 5 / 3 waves per
// SIMD, 40 / 48 slices in flight per CU instead of 24; bit-identical, fused dot included.  64 M-row
// 7-point operator, same box: 332 / 336 us back to back against 302 us for the deferred-store form
// below and 345 us for the plain one; 10 M-row 5-point: 28.3 / 32.5 against 25.9 us.  So the launch
// is NOT short of bytes in flight -- round 3's reading -- and the counters agree: its fabric reads wait
// 1200-1450 L2 cycles where the sweeps' wait 2300-2700 at 1.4-2x the requests outstanding.  What a
// slice costs is mostly there when everything sits in the Infinity Cache too (0.45 of 0.60 ns per
// slice): ~110 vector and ~120 scalar instructions per slice and wave, five L1->L2 round trips of
// which three hit L2.  profiles/r04_cfg4_spmv.txt; the kernel is in the history
// at commit "Vectors of a shard carved out of one allocation".)
// --------------------------------------------------------------------------
// a2-1, binned form (LSB_SPMV_BINNED, host side lsb_csr_binize) -- for operators
// whose rows scatter over far more of x than an XCD's L2 holds (power-law
// config 5: PMC shows 2.7e8 L2 misses per launch of the row-major kernel, one
// 128-byte line per non-zero, 34 GB of fabric traffic for 3.2 GB of algorithmic
// bytes).  One launch per BIN = per 2 MiB window of x, so every gather of the
// launch hits L2 after first touch; inside the bin the entries come as a
// row-sorted (row, col, value) stream and are cut into chunks of whole rows.
// A workgroup takes one chunk:
//   1. all 256 lanes stream 8 entries each (coalesced), gather x, and park
//      product and row id in LDS;
//   2. lane t then owns the 8 CONSECUTIVE entries [8t, 8t+8): it adds up the
//      runs of equal row ids in them -- a run that begins and ends inside is
//      added to y at once -- and what is left open at either end goes through
//      a segmented scan over the lanes (wavefront shuffles, then the four
//      waves through LDS);
//   3. the lane in which a multi-lane run ends adds carry + its own part to y.
// Every lane does the same amount of work whatever the row lengths are (the
// row-blocked kernel's reduce phase serialises on a block's longest row), each
// y entry is touched by exactly one lane per launch (no atomics), and the order
// of the additions is fixed by the layout alone: bit-identical run to run.
// --------------------------------------------------------------------------
__device__ __forceinline__ unsigned bin_pad(unsigned i) { return i + (i >> 3); }

template <int FLAGS, int BIN_U>
__global__ __launch_bounds__(WG) void k_spmv_binned(
    const unsigned *__restrict__ chunk_begin, unsigned nchunk, const unsigned *__restrict__ rows,
    const unsigned *__restrict__ cols, const double *__restrict__ vals,
    const double *__restrict__ x, double *__restrict__ y, const lsb_pcg_state *__restrict__ st) {
  constexpr unsigned CHUNK = BIN_U * WG; // entries of a chunk: 1024, 1536 or 2048
  __shared__ double sprod[CHUNK + CHUNK / 8 + 8];
  __shared__ unsigned skey[CHUNK + CHUNK / 8 + 8];
  __shared__ double swv[4];
  __shared__ int swf[4];
  __shared__ double sred[4];
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  // XCD-contiguous chunk ids: an XCD's workgroups sweep neighbouring rows of y
  const unsigned per = gridDim.x / NXCD;
  const unsigned w = (blockIdx.x % NXCD) * per + blockIdx.x / NXCD;
  if (w >= nchunk)
    return;
  const unsigned e0 = chunk_begin[w], cnt = chunk_begin[w + 1] - e0;
  unsigned r[BIN_U], c[BIN_U];
  double v[BIN_U];
  if (cnt <= CHUNK) {
#pragma unroll
    for (int u = 0; u < BIN_U; u++) {
      const unsigned t = tid + u * WG;
      if (t < cnt) {
        r[u] = stream_load<FLAGS>(rows + e0 + t);
        c[u] = stream_load<FLAGS>(cols + e0 + t);
        v[u] = stream_load<FLAGS>(vals + e0 + t);
      }
    }
  }
  if (st && st->status)
    return;
  if (cnt > CHUNK) { // ONE run longer than a chunk: the workgroup strides over it
    double s[1] = {0.0};
    for (unsigned j = e0 + tid; j < e0 + cnt; j += WG)
      s[0] += vals[j] * x[cols[j]];
    wg_sum<1>(s, sred);
    if (tid == 0)
      y[rows[e0]] += s[0];
    return;
  }
#pragma unroll
  for (int u = 0; u < BIN_U; u++) {
    const unsigned t = tid + u * WG;
    if (t < cnt) {
      // how the window of x is gathered out of L2 is a tuning flag: plain loads
      // fill a 128-byte L1 line per gather for 8 bytes used
      double xv;
      if (FLAGS & 8)
        xv = __hip_atomic_load(x + c[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (FLAGS & 16)
        xv = __builtin_nontemporal_load(x + c[u]);
      else
        xv = x[c[u]];
      sprod[bin_pad(t)] = v[u] * xv;
      skey[bin_pad(t)] = r[u];
    }
  }
  __syncthreads();
  // ---- lane t: entries [8t, 8t+8) -------------------------------------------
  const unsigned i0 = tid * BIN_U;
  const int nloc = i0 < cnt ? (int)min(cnt - i0, (unsigned)BIN_U) : 0;
  double first_sum = 0.0, acc = 0.0;
  unsigned first_key = 0, prevk = 0xFFFFFFFFu, nextk = 0xFFFFFFFEu;
  bool cont = false, has_boundary = false, closes = true;
  // runs that END inside this lane and did not begin in an earlier one are added
  // to y by this lane alone: collect them first, then ALL the loads of y, then
  // the stores -- a load-add-store chain per run would serialise up to eight
  // memory round trips per lane (short rows: nearly every entry ends a run)
  unsigned ok_[BIN_U];
  double ov[BIN_U];
  unsigned omask = 0;
  if (nloc > 0) {
    if (i0 > 0)
      prevk = skey[bin_pad(i0 - 1)];
    if (i0 + BIN_U < cnt)
      nextk = skey[bin_pad(i0 + BIN_U)];
    unsigned k[BIN_U + 1];
    double p[BIN_U];
#pragma unroll
    for (int j = 0; j < BIN_U; j++) {
      k[j] = j < nloc ? skey[bin_pad(i0 + j)] : 0u;
      p[j] = j < nloc ? sprod[bin_pad(i0 + j)] : 0.0;
    }
    cont = k[0] == prevk;
    first_key = k[0];
#pragma unroll
    for (int j = 0; j < BIN_U; j++)
      if (j < nloc) {
        acc += p[j];
        const bool last = j == nloc - 1;
        const bool ends = last ? (k[j] != nextk) : (k[j + 1] != k[j]);
        if (last)
          closes = ends;
        if (ends) {
          if (cont && !has_boundary) {
            first_sum = acc; // the head run: needs the carry of the lanes before
          } else {
            omask |= 1u << j;
            ok_[j] = k[j], ov[j] = acc;
          }
          if (!last)
            has_boundary = true, acc = 0.0;
        }
      }
    // (has_boundary = a run begins after this lane's first entry)
    double yv[BIN_U];
#pragma unroll
    for (int j = 0; j < BIN_U; j++)
      if (omask & (1u << j))
        yv[j] = y[ok_[j]];
#pragma unroll
    for (int j = 0; j < BIN_U; j++)
      if (omask & (1u << j))
        y[ok_[j]] = yv[j] + ov[j];
  }
  // ---- what is open at the lane ends: segmented scan over the lanes --------
  // value = sum of the lane's last run, flag = "a run begins in this lane"
  double sv = acc;
  int sf = (has_boundary || !cont) ? 1 : 0;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double v2 = __shfl_up(sv, d, 64);
    const int f2 = __shfl_up(sf, d, 64);
    if ((int)lane >= d) {
      if (!sf)
        sv += v2;
      sf |= f2;
    }
  }
  if (lane == 63)
    swv[wave] = sv, swf[wave] = sf;
  __syncthreads();
  double carry_w = 0.0;
  for (unsigned q = 0; q < wave; q++)
    carry_w = swf[q] ? swv[q] : carry_w + swv[q];
  const double incl = sv + (sf ? 0.0 : carry_w); // sum of the open run up to and including this lane
  double carry_in = __shfl_up(incl, 1, 64);
  if (lane == 0)
    carry_in = carry_w;
  // the head run of a lane that continues a run of the lanes before, where it ends
  // in this lane (inside it, or with its last entry)
  if (nloc > 0 && cont && (has_boundary || closes))
    y[first_key] += carry_in + first_sum;
}

extern "C" {

void lsb_k_set_blas1_nt(int on) { g_blas1_nt = on == 1 ? 63 : on; } /* 1 = every operand (bits 0..5) */
int lsb_k_get_blas1_nt(void) { return g_blas1_nt; }

/* one bin of the binned form: chunks [c0, c0 + nchunk) */
void lsb_k_spmv_binned(unsigned flags, unsigned chunk_cap, const unsigned *chunk_begin, unsigned c0,
                       unsigned nchunk, const unsigned *rows, const unsigned *cols, const double *vals,
                       const double *x, double *y, const struct lsb_pcg_state *st, void *stream) {
  if (!nchunk)
    return;
  const unsigned g = (nchunk + NXCD - 1) / NXCD * NXCD;
#define LSB_BINNED_U(FL, U)                                                                    \
  k_spmv_binned<FL, U><<<g, WG, 0, (hipStream_t)stream>>>(chunk_begin + c0, nchunk, rows, cols,   \
                                                          vals, x, y, st)
#define LSB_BINNED(FL)                                                                         \
  case FL:                                                                                     \
    if (chunk_cap == 1024)                                                                     \
      LSB_BINNED_U(FL, 4);                                                                     \
    else if (chunk_cap == 1536)                                                                \
      LSB_BINNED_U(FL, 6);                                                                     \
    else                                                                                       \
      LSB_BINNED_U(FL, 8);                                                                     \
    break;
  switch (flags & (SP_NT | 8u | 16u)) {
    LSB_BINNED(0)
    LSB_BINNED(2)
    LSB_BINNED(8)
    LSB_BINNED(10)
    LSB_BINNED(16)
    LSB_BINNED(18)
  default:
    LSB_BINNED_U(SP_NT, 8);
  }
#undef LSB_BINNED_U
#undef LSB_BINNED
}

unsigned lsb_k_blas1_grid(unsigned n) {
  // 16 B/lane => WG*2 elements per workgroup per trip.  Up to 256 workgroups
  // one trip each (small operators: all latency); beyond that four trips per
  // lane before the grid grows -- every workgroup of the NEXT kernel re-reduces
  // this kernel's partial sums, so a grid of 2048 on a 1 M-row shard costs more
  // in its consumers than it gains (1.25 M rows: sweep 26.5 -> 20.8 us); cap at
  // MAX_PARTIALS.
  unsigned g = div_up(n, WG * 2);
  if (g > 256) {
    g = div_up(n, WG * 2 * 4);
    if (g < 256)
      g = 256;
  }
  // ... and at three workgroups per CU: beyond that the sweeps get SLOWER the more of them stream
  // at once -- round 3, iteration of the 64 M-row 7-point operator (vectors of 512 MB: nothing
  // comes out of the Infinity Cache) 1211-1238 us with 2048 workgroups, 1075-1077 us with 768
  // (256 / 512 / 1024: 1081-1104 / 1082-1090 / 1103-1110), the 10 M-row 5-point one 140-142 ->
  // 136-137 us; y = 4 x over 512 MB vectors: 178 us with 1024 workgroups, 204-207 us with
  // 2048 / 4096 (profiles/r03_sweep_grid.txt).
  if (g > LSB_STREAM_GRID_CAP)
    g = LSB_STREAM_GRID_CAP;
  return g ? g : 1;
}

// number of workgroups (== number of dot partials) a given SpMV launch uses
unsigned lsb_k_spmv_grid(int variant, unsigned n, unsigned nblk,
                         unsigned lanes_per_row, unsigned grid_cap) {
  unsigned items;
  if (variant == LSB_SPMV_SELL) /* four slices per workgroup step */
    items = div_up(nblk, 4);
  else if (variant == LSB_SPMV_ADAPTIVE)
    items = nblk;
  else if (variant == LSB_SPMV_SUBWAVE)
    items = div_up(n, WG / (lanes_per_row ? lanes_per_row : 1));
  else
    items = div_up(n, WG);
  unsigned g = round_up(items ? items : 1, NXCD);
  if (grid_cap == 0 || grid_cap > LSB_MAX_PARTIALS)
    grid_cap = LSB_MAX_PARTIALS;
  grid_cap = grid_cap / NXCD * NXCD;
  if (grid_cap < NXCD)
    grid_cap = NXCD;
  if (g > grid_cap) {
    g = grid_cap;
    /* The persistent kernels cut the items into NXCD chunks and deal a chunk
     * cyclically to the XCD's g/NXCD workgroups: size the grid so that every
     * workgroup gets the same number of items (2440 items on 1536 workgroups
     * is two rounds with the second 59 % full; on 1224 it is two full ones,
     * and the consumers re-reduce 1224 partial sums instead of 1536). */
    if (variant == LSB_SPMV_SELL || variant == LSB_SPMV_ADAPTIVE) {
      const unsigned chunk = div_up(items, NXCD), cap = grid_cap / NXCD;
      const unsigned per = div_up(chunk, cap);
      g = div_up(chunk, per) * NXCD;
    }
  }
  return g;
}

/* the kernels with spmv_publish at their end */
int lsb_k_spmv_has_tail(int variant) { return variant == LSB_SPMV_ADAPTIVE || variant == LSB_SPMV_SELL; }

static lsb_ar_tail tail_for(const struct lsb_ar_tail *t, const double *partials) {
  lsb_ar_tail none;
  memset(&none, 0, sizeof none);
  if (!t || !t->counter)
    return none;
  if (!partials)
    errx(EXIT_FAILURE, "an SpMV launch without dot partials cannot carry the all-reduce");
  return *t;
}

void lsb_k_spmv(int variant, unsigned n, const int *offs, const int *cols,
                const double *vals, const int *rowblk,
                const unsigned char *blklanes, unsigned nblk,
                unsigned lanes_per_row, unsigned flags, unsigned grid_cap,
                const double *x, double *y, const double *xdot,
                double *partials, unsigned *npartials,
                const struct lsb_pcg_state *st, const int *rowmap,
                const struct lsb_ar_tail *tail_in, void *stream) {
  /* flags & LSB_SP_F32: `vals` points at fp32 values (mixed precision) */
  hipStream_t s = (hipStream_t)stream;
  const lsb_ar_tail tail = tail_for(tail_in, partials);
  if (tail.counter && !lsb_k_spmv_has_tail(variant))
    errx(EXIT_FAILURE, "lsb_k_spmv: SpMV form %d cannot carry the all-reduce", variant);
  const unsigned g = lsb_k_spmv_grid(variant, n, nblk, lanes_per_row, grid_cap);
  const float *vals32 = (const float *)(const void *)vals;
  const bool f32 = (flags & LSB_SP_F32) != 0;
  if (npartials)
    *npartials = g;
  if (variant == LSB_SPMV_ADAPTIVE) {
#define LSB_ADAPTIVE(FL)                                                                 \
  case FL:                                                                               \
    if (f32)                                                                             \
      k_spmv_adaptive<LSB_BLOCK_NNZ, FL, float><<<g, WG, 0, s>>>(                        \
          rowblk, blklanes, nblk, offs, cols, vals32, x, y, xdot, partials, st, rowmap,  \
          tail);                                                                         \
    else                                                                                 \
      k_spmv_adaptive<LSB_BLOCK_NNZ, FL, double><<<g, WG, 0, s>>>(                       \
          rowblk, blklanes, nblk, offs, cols, vals, x, y, xdot, partials, st, rowmap,    \
          tail);                                                                         \
    break;
    switch (flags & 3u) {
      LSB_ADAPTIVE(0)
      LSB_ADAPTIVE(1)
      LSB_ADAPTIVE(2)
      LSB_ADAPTIVE(3)
    }
#undef LSB_ADAPTIVE
  } else if (variant == LSB_SPMV_SUBWAVE) {
    const unsigned L = lanes_per_row;
    const unsigned slots = WG / L;
    const unsigned rpw = round_up(div_up(n, g), slots);
#define LSB_SUBWAVE(LL)                                                                  \
  case LL:                                                                               \
    if (f32)                                                                             \
      k_spmv_subwave<LL, float><<<g, WG, 0, s>>>(n, rpw, offs, cols, vals32, x, y, xdot, \
                                                 partials, st);                          \
    else                                                                                 \
      k_spmv_subwave<LL, double><<<g, WG, 0, s>>>(n, rpw, offs, cols, vals, x, y, xdot,  \
                                                  partials, st);                         \
    break;
    switch (L) {
      LSB_SUBWAVE(2)
      LSB_SUBWAVE(4)
      LSB_SUBWAVE(8)
      LSB_SUBWAVE(16)
      LSB_SUBWAVE(32)
    default:
      if (f32)
        k_spmv_subwave<64, float><<<g, WG, 0, s>>>(n, round_up(div_up(n, g), 4), offs, cols, vals32,
                                                   x, y, xdot, partials, st);
      else
        k_spmv_subwave<64, double><<<g, WG, 0, s>>>(n, round_up(div_up(n, g), 4), offs, cols, vals,
                                                    x, y, xdot, partials, st);
    }
#undef LSB_SUBWAVE
  } else {
    const unsigned rpw = div_up(n, g);
    k_spmv_scalar<<<g, WG, 0, s>>>(n, rpw, offs, cols, vals, x, y, xdot, partials, st);
  }
}

/* sub-wavefront SpMV with the direction update of the previous iteration folded
 * in (k_spmv_subwave_p); same grid as the plain sub-wavefront launch */
void lsb_k_spmv_subwave_p(unsigned n, const int *offs, const int *cols, const double *vals,
                          unsigned lanes_per_row, const double *r, const double *dinv, double dc,
                          const double *pold, double *pnew, double *y, double *partials,
                          unsigned *npartials, struct lsb_pcg_state *st, int parity,
                          const double *parts2, unsigned nparts2, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const unsigned L = lanes_per_row;
  const unsigned g = lsb_k_spmv_grid(LSB_SPMV_SUBWAVE, n, 0, L, 0);
  *npartials = g;
  const unsigned rpw = round_up(div_up(n, g), WG / L);
#define LSB_SWP(LL)                                                                         \
  case LL:                                                                                  \
    k_spmv_subwave_p<LL><<<g, WG, 0, s>>>(n, rpw, offs, cols, vals, r, dinv, dc, pold, pnew, y, \
                                          partials, st, parity, parts2, nparts2);           \
    break;
  switch (L) {
    LSB_SWP(2)
    LSB_SWP(4)
    LSB_SWP(8)
    LSB_SWP(16)
    LSB_SWP(32)
  default:
    k_spmv_subwave_p<64><<<g, WG, 0, s>>>(n, round_up(div_up(n, g), 4), offs, cols, vals, r, dinv,
                                          dc, pold, pnew, y, partials, st, parity, parts2, nparts2);
  }
#undef LSB_SWP
}

/* Sliced-ELL launch over the slices [s0, s0+ns).  flags & LSB_SP_C16: `cols` is
 * the 16-bit code array and `sbase` the slot bases (row_begin = global index
 * of local row 0); else `cols` holds 32-bit column ids and sbase is unused. */
void lsb_k_spmv_sell(unsigned flags, unsigned grid_cap, unsigned period, const unsigned *sptr,
                     unsigned s0, unsigned ns, unsigned n, unsigned row_begin, unsigned xlen,
                     const void *cols,
                     const int *sbase, const double *vals, const double *vconst, unsigned ulen,
                     const double *x, double *y, const double *xdot, double *partials,
                     unsigned *npartials,
                     const struct lsb_pcg_state *st, const struct lsb_ar_tail *tail_in,
                     const struct lsb_cheb_epi *epi_in, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const lsb_ar_tail tail = tail_for(tail_in, partials);
  lsb_cheb_epi epi;
  memset(&epi, 0, sizeof epi);
  if (epi_in && epi_in->zout) {
    if (!(flags & LSB_SP_C16) || partials || ((row_begin | s0) & 1u))
      errx(EXIT_FAILURE, "lsb_k_spmv_sell: the Chebyshev epilogue rides in the 16-bit kernel only");
    epi = *epi_in;
  }
  const unsigned g = lsb_k_spmv_grid(LSB_SPMV_SELL, n, ns, 0, grid_cap ? grid_cap : 1536);
  if (npartials)
    *npartials = g;
  const int nt = (flags & SP_NT) != 0;
  const float *vals32 = (const float *)(const void *)vals; /* flags & LSB_SP_F32 */
  if (period && (period < NXCD || ns < period))
    period = 0; /* less than a plane: contiguous dealing */
#define LSB_SELL16(FL, VT, V)                                                                  \
  k_spmv_sell16<FL, VT><<<g, WG, 0, s>>>(sptr, s0, ns, period, n, row_begin, xlen, (const short *)cols, \
                                         sbase, V, vconst, ulen, x, y, xdot, partials, st, tail, epi)
#define LSB_SELL32(FL, VT, V)                                                                  \
  k_spmv_sell<FL, VT><<<g, WG, 0, s>>>(sptr, s0, ns, period, n, (const int *)cols, V, x, y, xdot, \
                                       partials, st, tail)
  if (epi.zout) {
#define LSB_SELL16C(FL, VT, V)                                                                            \
  k_spmv_sell16<FL, VT, true><<<g, WG, 0, s>>>(sptr, s0, ns, period, n, row_begin, xlen, (const short *)cols, \
                                               sbase, V, vconst, ulen, x, y, xdot, partials, st, tail, epi)
    if (flags & LSB_SP_F32) {
      if (nt)
        LSB_SELL16C(SP_NT, float, vals32);
      else
        LSB_SELL16C(0, float, vals32);
    } else if (nt)
      LSB_SELL16C(SP_NT, double, vals);
    else
      LSB_SELL16C(0, double, vals);
#undef LSB_SELL16C
  } else if (flags & LSB_SP_F32) {
    if (flags & LSB_SP_C16) {
      if (nt)
        LSB_SELL16(SP_NT, float, vals32);
      else
        LSB_SELL16(0, float, vals32);
    } else if (nt)
      LSB_SELL32(SP_NT, float, vals32);
    else
      LSB_SELL32(0, float, vals32);
  } else if (flags & LSB_SP_C16) {
    if (nt)
      LSB_SELL16(SP_NT, double, vals);
    else
      LSB_SELL16(0, double, vals);
  } else if (nt)
    LSB_SELL32(SP_NT, double, vals);
  else
    LSB_SELL32(0, double, vals);
#undef LSB_SELL16
#undef LSB_SELL32
}

/* The constant-slot layout through slice templates (k_spmv_tmpl).  vals: the kept value slots
 * (fp32 when flags & LSB_SP_F32; the templates' constants are then fp32-rounded by the caller). */
void lsb_k_spmv_tmpl(unsigned flags, unsigned grid_cap, unsigned period, const unsigned *sptr, unsigned s0,
                     unsigned ns, unsigned n, unsigned row_begin, unsigned xlen, const unsigned *srec,
                     const unsigned long long *mask, const struct lsb_sell_tmpl *td,
                     unsigned nfar, const int *sbase, const void *vals,
                     const double *vconst, const double *x, double *y, const double *xdot,
                     double *partials, unsigned *npartials, const struct lsb_pcg_state *st,
                     const struct lsb_ar_tail *tail_in, const struct lsb_cheb_epi *epi_in, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const lsb_ar_tail tail = tail_for(tail_in, partials);
  lsb_cheb_epi epi;
  memset(&epi, 0, sizeof epi);
  if (epi_in && epi_in->zout) {
    if (partials || ((row_begin | s0) & 1u))
      errx(EXIT_FAILURE, "lsb_k_spmv_tmpl: the Chebyshev epilogue takes no dot and even row offsets");
    epi = *epi_in;
  }
  const unsigned g = lsb_k_spmv_grid(LSB_SPMV_SELL, n, ns, 0, grid_cap ? grid_cap : 1536);
  if (npartials)
    *npartials = g;
  if (period && (period < NXCD || ns < period))
    period = 0;
  const int f32 = (flags & LSB_SP_F32) != 0, dot_is_x = xdot && xdot == x + row_begin;
#define LSB_TMPL(NF)                                                                                 \
  do {                                                                                               \
    if (epi.zout)                                                                                    \
      k_spmv_tmpl<NF, true><<<g, WG, 0, s>>>(sptr, s0, ns, period, n, row_begin, xlen, (const u4v *)srec, mask, td, sbase, vals, f32, \
                                             vconst, x, y, xdot, dot_is_x, partials, st, tail, epi);   \
    else if (flags & LSB_SP_DEFER)                                                                   \
      k_spmv_tmpl<NF, false, true><<<g, WG, 0, s>>>(sptr, s0, ns, period, n, row_begin, xlen, (const u4v *)srec, mask, td, sbase, vals, f32, \
                                                    vconst, x, y, xdot, dot_is_x, partials, st, tail, epi); \
    else                                                                                             \
      k_spmv_tmpl<NF, false><<<g, WG, 0, s>>>(sptr, s0, ns, period, n, row_begin, xlen, (const u4v *)srec, mask, td, sbase, vals, f32, \
                                              vconst, x, y, xdot, dot_is_x, partials, st, tail, epi);  \
  } while (0)
  switch (nfar) {
  case 0: LSB_TMPL(0); break;
  case 1: LSB_TMPL(1); break;
  case 2: LSB_TMPL(2); break;
  default: errx(EXIT_FAILURE, "lsb_k_spmv_tmpl: %u far slots per side", nfar);
  }
#undef LSB_TMPL
}

/* The template layout walked in z-columns (k_spmv_tmpl_col): plan = xbeg[NXCD + 1], padding to 16
 * unsigneds, then nitem 16-byte items (lsb_sell_tmpl_columns).  Whole launches only (every slice of
 * the shard is in exactly one item). */
void lsb_k_spmv_tmpl_col(unsigned flags, unsigned grid_cap, unsigned period, const unsigned *plan, unsigned nitem,
                         int centre0, unsigned n, unsigned row_begin, unsigned xlen, const unsigned *sptr,
                         const unsigned long long *mask, const struct lsb_sell_tmpl *td, unsigned nfar,
                         const int *sbase, const void *vals, const double *vconst, const double *x, double *y,
                         const double *xdot, double *partials, unsigned *npartials,
                         const struct lsb_pcg_state *st, const struct lsb_ar_tail *tail_in, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const lsb_ar_tail tail = tail_for(tail_in, partials);
  const unsigned g = lsb_k_spmv_grid(LSB_SPMV_SELL, n, nitem, 0, grid_cap ? grid_cap : 1536);
  if (npartials)
    *npartials = g;
  if (period < NXCD || !plan)
    errx(EXIT_FAILURE, "lsb_k_spmv_tmpl_col: no column plan (period %u)", period);
  const int f32 = (flags & LSB_SP_F32) != 0;
  // the centre pair is the dot's operand where the dot is with the gathered vector itself (and the
  // centre base is 0: true of every operator with a diagonal; the kernel's DOT = 1 relies on it,
  // the plan builder says so in lsb_tmpl_cols.centre0)
  const int dot = !partials || !xdot ? 0 : (xdot == x + row_begin && centre0) ? 1 : 2;
#define LSB_COL(NF, D)                                                                                \
  k_spmv_tmpl_col<NF, D><<<g, WG, 0, s>>>(plan, period, n, row_begin, xlen, sptr, mask, td, sbase, vals, f32, vconst, x, y, \
                                          xdot, partials, st, tail)
#define LSB_COLD(NF)                                                                                  \
  do {                                                                                                \
    if (dot == 0)                                                                                     \
      LSB_COL(NF, 0);                                                                                 \
    else if (dot == 1)                                                                                \
      LSB_COL(NF, 1);                                                                                 \
    else                                                                                              \
      LSB_COL(NF, 2);                                                                                 \
  } while (0)
  switch (nfar) {
  case 1: LSB_COLD(1); break;
  case 2: LSB_COLD(2); break;
  default: errx(EXIT_FAILURE, "lsb_k_spmv_tmpl_col: %u far slots per side", nfar);
  }
#undef LSB_COLD
#undef LSB_COL
}

void lsb_k_reduce_final(const double *partials, unsigned nparts, unsigned width,
                        double *out, int take_sqrt,
                        const struct lsb_pcg_state *st, void *stream) {
  k_reduce_final<<<1, WG, 0, (hipStream_t)stream>>>(partials, nparts, width, out,
                                                    take_sqrt, st);
}

void lsb_k_reduce_final2(const double *pa, unsigned na, unsigned wa, double *outa,
                         const double *pb, unsigned nb, unsigned wb, double *outb,
                         const struct lsb_pcg_state *st, void *stream) {
  k_reduce_final2<<<1, WG, 0, (hipStream_t)stream>>>(pa, na, wa, outa, pb, nb, wb, outb, st);
}

void lsb_k_dot(unsigned n, const double *a, const double *b, double *partials,
               unsigned *npartials, void *stream) {
  unsigned g = div_up(n ? n : 1, WG * 4);
  if (g > LSB_STREAM_GRID_CAP)
    g = LSB_STREAM_GRID_CAP;
  *npartials = g;
  k_dot<<<g, WG, 0, (hipStream_t)stream>>>(n, a, b, partials);
}

static unsigned ew_grid(unsigned n) {
  unsigned g = div_up(n ? n : 1, WG * 4);
  return g > LSB_STREAM_GRID_CAP ? LSB_STREAM_GRID_CAP : g;
}

void lsb_k_axpy(unsigned n, const double *alpha, const double *x, double *y,
                void *stream) {
  k_axpy<<<ew_grid(n), WG, 0, (hipStream_t)stream>>>(n, alpha, x, y);
}

void lsb_k_xpay(unsigned n, const double *beta, const double *x, double *y,
                void *stream) {
  k_xpay<<<ew_grid(n), WG, 0, (hipStream_t)stream>>>(n, beta, x, y);
}

void lsb_k_jacobi_setup(unsigned n, unsigned row_begin, const int *offs,
                        const int *cols, const double *vals, double *dinv,
                        int *nzero, void *stream) {
  k_jacobi_setup<<<ew_grid(n), WG, 0, (hipStream_t)stream>>>(n, row_begin, offs, cols,
                                                             vals, dinv, nzero);
}

void lsb_k_l1_setup(unsigned n, const int *offs, const double *vals, double *dinv, int *nzero,
                    void *stream) {
  k_l1_setup<<<ew_grid(n), WG, 0, (hipStream_t)stream>>>(n, offs, vals, dinv, nzero);
}

void lsb_k_jacobi_apply(unsigned n, const double *dinv, const double *r,
                        double *z, void *stream) {
  k_jacobi_apply<<<ew_grid(n), WG, 0, (hipStream_t)stream>>>(n, dinv, r, z);
}

void lsb_k_jacobi_sweep(unsigned n, double w, const double *dinv,
                        const double *b, const double *ax, double *x,
                        void *stream) {
  k_jacobi_sweep<<<ew_grid(n), WG, 0, (hipStream_t)stream>>>(n, w, dinv, b, ax, x);
}

void lsb_k_vreduce(double *base, unsigned stride, unsigned nshard, unsigned off,
                   unsigned cnt, void *stream) {
  k_vreduce<<<1, 64, 0, (hipStream_t)stream>>>(base, stride, nshard, off, cnt);
}

void lsb_k_perm_gather(unsigned n, const int *perm, const double *src, double *dst,
                       void *stream) {
  k_perm_gather<<<ew_grid(n), WG, 0, (hipStream_t)stream>>>(n, perm, src, dst);
}

void lsb_k_perm_scatter(unsigned n, const int *perm, const double *src, double *dst,
                        void *stream) {
  k_perm_scatter<<<ew_grid(n), WG, 0, (hipStream_t)stream>>>(n, perm, src, dst);
}

void lsb_k_fill_index(unsigned n, unsigned first, double *v, void *stream) {
  k_fill_index<<<ew_grid(n), WG, 0, (hipStream_t)stream>>>(n, first, v);
}

void lsb_k_pcg_init(unsigned n, const double *b, const double *dinv, double dc, double *x,
                    double *r, double *p, double *partials2,
                    unsigned *npartials, void *stream) {
  const unsigned g = lsb_k_blas1_grid(n);
  *npartials = g;
  if (aligned16(b) && aligned16(dinv) && aligned16(x) && aligned16(r) && aligned16(p))
    k_pcg_init<true><<<g, WG, 0, (hipStream_t)stream>>>(n, b, dinv, dc, x, r, p, partials2);
  else
    k_pcg_init<false><<<g, WG, 0, (hipStream_t)stream>>>(n, b, dinv, dc, x, r, p, partials2);
}

void lsb_k_pcg_init_state(struct lsb_pcg_state *st, const double *partials2,
                          unsigned nparts, double tol, int maxit,
                          void *stream) {
  k_pcg_init_state<<<1, WG, 0, (hipStream_t)stream>>>(st, partials2, nparts, tol, maxit);
}

void lsb_k_pcg_update_xr(unsigned n, const double *p, const double *q,
                         const double *dinv, double dc, double *x, double *r,
                         struct lsb_pcg_state *st, int parity,
                         const double *pq_parts, unsigned npq,
                         double *partials2, unsigned *npartials, void *stream) {
  const unsigned g = lsb_k_blas1_grid(n);
  *npartials = g;
  if (aligned16(p) && aligned16(q) && aligned16(dinv) && aligned16(x) && aligned16(r)) {
#define LSB_XR(A, B, C)                                                                           \
  k_pcg_update_xr<true, A, B, C><<<g, WG, 0, (hipStream_t)stream>>>(n, p, q, dinv, dc, x, r, st, parity, pq_parts, \
                                                                  npq, partials2)
    switch (g_blas1_nt & 7) {
    case 0: LSB_XR(false, false, false); break;
    case 1: LSB_XR(true, false, false); break;
    case 2: LSB_XR(false, true, false); break;
    case 3: LSB_XR(true, true, false); break;
    case 4: LSB_XR(false, false, true); break;
    case 5: LSB_XR(true, false, true); break;
    case 6: LSB_XR(false, true, true); break;
    default: LSB_XR(true, true, true); break;
    }
#undef LSB_XR
  } else {
    k_pcg_update_xr<false, false, false, false><<<g, WG, 0, (hipStream_t)stream>>>(
        n, p, q, dinv, dc, x, r, st, parity, pq_parts, npq, partials2);
  }
}

void lsb_k_pcg_col_px(unsigned grid_cap, unsigned period, const unsigned *plan, unsigned nitem, unsigned n,
                      const unsigned *sptr, const unsigned long long *mask, const struct lsb_sell_tmpl *td,
                      unsigned nfar, const int *sbase, const double *vals, const double *vconst, const double *r,
                      const double *pold, double *pnew, double *x, int xupd, double dc, double *partials,
                      unsigned *npartials, struct lsb_pcg_state *st, int parity, const double *parts2,
                      unsigned nparts2, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  /* every workgroup resident: three per CU with two far slots per side (142 VGPRs), five with one (96) */
  const unsigned res = nfar == 2 ? 768u : 1280u;
  const unsigned g = lsb_k_spmv_grid(LSB_SPMV_SELL, n, nitem, 0, grid_cap && grid_cap < res ? grid_cap : res);
  if (npartials)
    *npartials = g;
  if (period < NXCD || !plan || pold == pnew)
    errx(EXIT_FAILURE, "lsb_k_pcg_col_px: no column plan (period %u) or one direction buffer", period);
  /* x, p' and q streamed nontemporally (NT = 3; measured against 0 / 1 / 2 on config 4: 950.5 / 950.3 / 905.7 /
   * 894.1 us per iteration, profiles/r04_px.txt) */
#define LSB_PX(NF)                                                                                    \
  do {                                                                                                \
    if (xupd)                                                                                         \
      k_pcg_col_px<NF, 3, true><<<g, WG, 0, s>>>(plan, period, n, sptr, mask, td, sbase, vals, vconst, r, pold, pnew, x, dc, \
                                                 partials, st, parity, parts2, nparts2);              \
    else                                                                                              \
      k_pcg_col_px<NF, 3, false><<<g, WG, 0, s>>>(plan, period, n, sptr, mask, td, sbase, vals, vconst, r, pold, pnew, x, \
                                                  dc, partials, st, parity, parts2, nparts2);         \
  } while (0)
  switch (nfar) {
  case 1: LSB_PX(1); break;
  case 2: LSB_PX(2); break;
  default: errx(EXIT_FAILURE, "lsb_k_pcg_col_px: %u far slots per side", nfar);
  }
#undef LSB_PX
}

void lsb_k_pcg_col_r(unsigned grid_cap, unsigned period, const unsigned *plan, unsigned nitem, unsigned n,
                     const unsigned *sptr, const unsigned long long *mask, const struct lsb_sell_tmpl *td, unsigned nfar,
                     const int *sbase, const double *vals, const double *vconst, const double *p, double *r, double dc,
                     struct lsb_pcg_state *st, int parity, int pbuf, int xtwo, const double *pq_parts, unsigned npq,
                     double *partials2, unsigned *npartials, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const unsigned g = lsb_k_spmv_grid(LSB_SPMV_SELL, n, nitem, 0, grid_cap && grid_cap < 1280u ? grid_cap : 1280u);
  *npartials = g;
  if (period < NXCD || !plan)
    errx(EXIT_FAILURE, "lsb_k_pcg_col_r: no column plan (period %u)", period);
  switch (nfar) {
  case 1:
    k_pcg_col_r<1><<<g, WG, 0, s>>>(plan, period, n, sptr, mask, td, sbase, vals, vconst, p, r, dc, st, parity, pbuf,
                                    xtwo, pq_parts, npq, partials2);
    break;
  case 2:
    k_pcg_col_r<2><<<g, WG, 0, s>>>(plan, period, n, sptr, mask, td, sbase, vals, vconst, p, r, dc, st, parity, pbuf,
                                    xtwo, pq_parts, npq, partials2);
    break;
  default: errx(EXIT_FAILURE, "lsb_k_pcg_col_r: %u far slots per side", nfar);
  }
}

void lsb_k_pcg_xfix(unsigned n, const double *p0, const double *p1, double *x, const struct lsb_pcg_state *st,
                    void *stream) {
  k_pcg_xfix<<<lsb_k_blas1_grid(n), WG, 0, (hipStream_t)stream>>>(n, p0, p1, x, st);
}

void lsb_k_cg1_update(unsigned n, double *u, const double *w, const double *dinv, double dc,
                      double *p,
                      double *s, double *x, double *r, struct lsb_pcg_state *st, int parity,
                      const double *parts_gr, unsigned ngr, const double *parts_d, unsigned nd,
                      const struct lsb_ar_collect *collect, double *partials2,
                      unsigned *npartials, void *stream) {
  const unsigned g = lsb_k_blas1_grid(n);
  *npartials = g;
  hipStream_t hs = (hipStream_t)stream;
  lsb_ar_collect col;
  memset(&col, 0, sizeof col);
  if (collect)
    col = *collect;
  const bool v2 = aligned16(u) && aligned16(w) && aligned16(dinv) && aligned16(p) &&
                  aligned16(s) && aligned16(x) && aligned16(r);
#define LSB_CG1(V, N, U)                                                                  \
  k_cg1_update<V, N, U><<<g, WG, 0, hs>>>(n, u, w, dinv, dc, p, s, x, r, st, parity, parts_gr, \
                                          ngr, parts_d, nd, col, partials2)
  if (!u) { /* implicit u = dc r: r is the gather vector (needs the constant diagonal) */
    if (dinv)
      errx(EXIT_FAILURE, "lsb_k_cg1_update: implicit u needs a constant diagonal");
    if (v2 && (g_blas1_nt & 32))
      LSB_CG1(true, true, true);
    else if (v2)
      LSB_CG1(true, false, true);
    else
      LSB_CG1(false, false, true);
  } else if (v2 && (g_blas1_nt & 32))
    LSB_CG1(true, true, false);
  else if (v2)
    LSB_CG1(true, false, false);
  else
    LSB_CG1(false, false, false);
#undef LSB_CG1
}

void lsb_k_pcg_update_p(unsigned n, const double *r, const double *dinv, double dc,
                        const double *pin, double *p, struct lsb_pcg_state *st, int parity,
                        const double *parts2, unsigned nparts2, void *stream) {
  const unsigned g = lsb_k_blas1_grid(n);
  const int x2 = 1; /* two pairs per operand in flight (one: 40.6 against 40.4 us, round 3 -- no difference) */
  hipStream_t s = (hipStream_t)stream;
#define LSB_UPD_P(V2, NTR, NTP, X2) \
  k_pcg_update_p<V2, NTR, NTP, X2><<<g, WG, 0, s>>>(n, r, dinv, dc, pin, p, st, parity, parts2, nparts2)
#define LSB_UPD_P2(NTR, NTP)          \
  do {                                \
    if (big)                          \
      LSB_UPD_P(true, NTR, NTP, true); \
    else                              \
      LSB_UPD_P(true, NTR, NTP, false); \
  } while (0)
  if (aligned16(r) && aligned16(dinv) && aligned16(p) && aligned16(pin)) {
    const bool big = x2 && (size_t)n / 2 > (size_t)g * WG; /* a second pair exists at all */
    switch ((g_blas1_nt >> 3) & 3) {
    case 0: LSB_UPD_P2(false, false); break;
    case 1: LSB_UPD_P2(true, false); break;
    case 2: LSB_UPD_P2(false, true); break;
    default: LSB_UPD_P2(true, true); break;
    }
  } else {
    LSB_UPD_P(false, false, false, false);
  }
#undef LSB_UPD_P2
#undef LSB_UPD_P
}

} // extern "C"
