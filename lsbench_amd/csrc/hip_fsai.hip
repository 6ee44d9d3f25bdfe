// Factorised sparse approximate inverse (LSB_PRECOND_FSAI; SURVEY.md section 8(f) rank 2,
// VERDICT r2 item 7): M^-1 = G^T G, G lower triangular on a prescribed pattern, minimising
// ||I - G L||_F for the (unknown) Cholesky factor L of S (Kolotilina & Yeremin 1993).  Row i of
// G comes from ONE small dense SPD system: with J = the pattern of row i (columns <= i, i last),
//     S[J, J] y = e_last,      g_i = y / sqrt(y_last)
// -- all rows independent, which is what makes the set-up a GPU job.  The reference's own
// answer to "expensive set-up once, cheap solves" is CHOLMOD's factorisation in csr_init,
// outside the timed loop (src/cholmod-impl.h:25-26, 59-62); this is the sparse counterpart
// for the iterative path: set-up untimed, an application = two SpMVs (t = G r, z = G^T t), no
// triangular solve, no reduction.
//
// k_fsai_rows: a workgroup (rows of up to 128 pattern entries, 131 KB of LDS) or a single
// wavefront (up to 32, 8.6 KB) per row: gather S[J, J] out of the CSR into LDS (binary search
// of each stored entry of the rows J in the sorted list J), Cholesky in place, one back
// substitution, scale, store.  A pivot <= 0 (S not positive definite on J) is reported.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hip_wg.h"
#include "lsb_impl.h"

template <int NT>
__global__ __launch_bounds__(NT) void k_fsai_rows(
    const unsigned *__restrict__ rows, unsigned mcap, const int *__restrict__ offs,
    const int *__restrict__ cols, const double *__restrict__ vals, unsigned row_begin,
    const unsigned *__restrict__ poffs, const unsigned *__restrict__ pcols,
    double *__restrict__ gvals, int *__restrict__ bad) {
  extern __shared__ double sm[];
  const unsigned tid = threadIdx.x;
  const unsigned i = rows[blockIdx.x];
  const unsigned p0 = poffs[i], m = poffs[i + 1] - p0; // m <= mcap (host)
  double *A = sm;                     // m x m, row-major, lower triangle used
  double *w = A + (size_t)mcap * mcap; // mcap
  int *J = (int *)(w + mcap);         // mcap
  for (unsigned t = tid; t < m; t += NT)
    J[t] = (int)pcols[p0 + t];
  for (unsigned t = tid; t < m * m; t += NT)
    A[t] = 0.0;
  __syncthreads();
  // A[a][b] = S[J[a], J[b]] for b <= a: every stored entry of row J[a] looks itself up in J[0..a]
  for (unsigned a = tid / 8; a < m; a += NT / 8) {
    const unsigned ra = (unsigned)J[a] - row_begin; // the pattern's rows are rows of this shard
    for (int e = offs[ra] + (int)(tid % 8); e < offs[ra + 1]; e += 8) {
      const int c = cols[e];
      int lo = 0, hi = (int)a; // J ascending
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (J[mid] < c)
          lo = mid + 1;
        else
          hi = mid;
      }
      if (J[lo] == c)
        A[(size_t)a * m + lo] = vals[e];
    }
  }
  __syncthreads();
  // Cholesky, right-looking, in place (lower)
  for (unsigned k = 0; k < m; k++) {
    if (tid == 0) {
      const double d = A[(size_t)k * m + k];
      if (!(d > 0.0)) {
        *bad = 1;
        A[(size_t)k * m + k] = 1.0; // keep going with finite numbers; the host refuses the result
      } else {
        A[(size_t)k * m + k] = sqrt(d);
      }
    }
    __syncthreads();
    const double lkk = A[(size_t)k * m + k];
    for (unsigned a = k + 1 + tid; a < m; a += NT)
      A[(size_t)a * m + k] /= lkk;
    __syncthreads();
    const unsigned cnt = m - k - 1;
    for (unsigned idx = tid; idx < cnt * cnt; idx += NT) {
      const unsigned a = k + 1 + idx / cnt, b = k + 1 + idx % cnt;
      if (b <= a)
        A[(size_t)a * m + b] -= A[(size_t)a * m + k] * A[(size_t)b * m + k];
    }
    __syncthreads();
  }
  // L u = e_last gives u = (0, ..., 0, 1 / L_mm); back substitution L^T y = u
  for (unsigned t = tid; t < m; t += NT)
    w[t] = t + 1 == m ? 1.0 / A[(size_t)(m - 1) * m + (m - 1)] : 0.0;
  __syncthreads();
  for (unsigned kk = m; kk-- > 0;) {
    if (tid == 0)
      w[kk] /= A[(size_t)kk * m + kk];
    __syncthreads();
    const double yk = w[kk];
    for (unsigned j = tid; j < kk; j += NT)
      w[j] -= A[(size_t)kk * m + j] * yk; // (L^T)[j][kk] = L[kk][j]
    __syncthreads();
  }
  const double ylast = w[m - 1];
  if (tid == 0 && !(ylast > 0.0))
    *bad = 1;
  const double scale = ylast > 0.0 ? 1.0 / sqrt(ylast) : 0.0;
  for (unsigned t = tid; t < m; t += NT)
    gvals[p0 + t] = w[t] * scale;
}

// ---- the launch-bound operators' FSAI-PCG iteration in THREE launches -----------------------
// A generic preconditioned iteration is six launches (S p, the x / r sweep, G r, G^T t, the
// two dots, the direction sweep): at 3.4 us a launch 69 iterations of tests/xn3b_A_18.txt
// cost 1.44 ms, 693 solves/s.  Everything fits the L2, so sweeps ride in the SpMVs that
// gather their operands anyway:
//   A  k_spmv_subwave_p (hip_kernels.hip, as it is, with z in the place of D^-1 r):
//      beta, stop test, p = z + beta p formed in the gather, q = S p, p.q
//   B  k_fsai_xr_gr:   alpha = r.z / p.q;  x += alpha p;  r' = r - alpha q formed in the gather
//      of G's rows and stored for the own row into the OTHER residual buffer;  t = G r'
//   C  k_fsai_gt_dots: z = G^T t;  partial sums of (r'.z, r'.r') -- what A's head consumes.
// L lanes per row, rows dealt in contiguous XCD-contiguous chunks, as in k_spmv_subwave.
template <int L>
__global__ __launch_bounds__(WG) void k_fsai_xr_gr(
    unsigned n, unsigned rows_per_wg, const int *__restrict__ goffs, const int *__restrict__ gcols,
    const double *__restrict__ gvals, const double *__restrict__ p, const double *__restrict__ q,
    double *__restrict__ x, const double *__restrict__ rold, double *__restrict__ rnew,
    double *__restrict__ t, lsb_pcg_state *__restrict__ st, int parity,
    const double *__restrict__ pq_parts, unsigned npq) {
  __shared__ double sred[4];
  const unsigned tid = threadIdx.x, slot = tid / L, l = tid % L;
  constexpr unsigned SLOTS = WG / L;
  const unsigned w = xcd_contiguous_wg();
  const unsigned ra = min(w * rows_per_wg, n), rb = min(ra + rows_per_wg, n);
  const int stopped = st->status;
  const double rz = st->rz[parity];
  double pqv[1];
  wg_sum_partials<1>(pq_parts, npq, pqv, sred);
  if (stopped)
    return;
  const double pq = pqv[0];
  if (!(pq != 0.0) || !isfinite(pq)) { // the same decision in every workgroup (k_pcg_update_xr's)
    if (blockIdx.x == 0 && threadIdx.x == 0)
      st->status = LSB_STATUS_BREAKDOWN;
    return;
  }
  const double alpha = rz / pq;
  if (blockIdx.x == 0 && threadIdx.x == 0)
    st->pq = pq;
  for (unsigned base = ra; base < rb; base += SLOTS) {
    const unsigned r = base + slot;
    double s = 0.0;
    if (r < rb) {
      const int j0 = goffs[r], j1 = goffs[r + 1];
      for (int j = j0 + (int)l; j < j1; j += L) {
        const int c = gcols[j];
        s = fma(gvals[j], fma(-alpha, q[c], rold[c]), s);
      }
    }
#pragma unroll
    for (int off = L >> 1; off > 0; off >>= 1)
      s += __shfl_xor(s, off, 64);
    if (r < rb && l == 0) {
      t[r] = s;
      rnew[r] = fma(-alpha, q[r], rold[r]);
      x[r] = fma(alpha, p[r], x[r]);
    }
  }
}

template <int L>
__global__ __launch_bounds__(WG) void k_fsai_gt_dots(
    unsigned n, unsigned rows_per_wg, const int *__restrict__ offs, const int *__restrict__ cols,
    const double *__restrict__ vals, const double *__restrict__ t, double *__restrict__ z,
    const double *__restrict__ r, double *__restrict__ partials2, const lsb_pcg_state *__restrict__ st) {
  __shared__ double sred[8];
  const unsigned tid = threadIdx.x, slot = tid / L, l = tid % L;
  constexpr unsigned SLOTS = WG / L;
  const unsigned w = xcd_contiguous_wg();
  const unsigned ra = min(w * rows_per_wg, n), rb = min(ra + rows_per_wg, n);
  const int stopped = st->status;
  double acc[2] = {0.0, 0.0};
  for (unsigned base = ra; base < rb; base += SLOTS) {
    const unsigned row = base + slot;
    double s = 0.0, rv = 0.0;
    if (row < rb) {
      const int j0 = offs[row], j1 = offs[row + 1];
      rv = r[row];
      for (int j = j0 + (int)l; j < j1; j += L)
        s = fma(vals[j], t[cols[j]], s);
    }
#pragma unroll
    for (int off = L >> 1; off > 0; off >>= 1)
      s += __shfl_xor(s, off, 64);
    if (stopped)
      return;
    if (row < rb && l == 0) {
      z[row] = s;
      acc[0] = fma(rv, s, acc[0]);
      acc[1] = fma(rv, rv, acc[1]);
    }
  }
  if (stopped)
    return;
  wg_sum<2>(acc, sred);
  if (tid == 0) {
    partials2[2 * w + 0] = acc[0];
    partials2[2 * w + 1] = acc[1];
  }
}

extern "C" {

static unsigned fsai_div_up(unsigned a, unsigned b) { return (a + b - 1) / b; }

/* grid and rows per workgroup of the sub-wavefront launches (the same rule as lsb_k_spmv's) */
static unsigned fsai_grid(unsigned n, unsigned L, unsigned *rpw) {
  const unsigned g = lsb_k_spmv_grid(LSB_SPMV_SUBWAVE, n, 0, L, 0);
  const unsigned per = WG / L, rows = fsai_div_up(n, g);
  *rpw = fsai_div_up(rows, per) * per;
  return g;
}

void lsb_k_fsai_xr_gr(unsigned n, const int *goffs, const int *gcols, const double *gvals, unsigned lanes,
                      const double *p, const double *q, double *x, const double *rold, double *rnew, double *t,
                      struct lsb_pcg_state *st, int parity, const double *pq_parts, unsigned npq, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  unsigned rpw;
  const unsigned g = fsai_grid(n, lanes, &rpw);
#define LSB_FS(LL)                                                                                          \
  case LL:                                                                                                  \
    k_fsai_xr_gr<LL><<<g, WG, 0, s>>>(n, rpw, goffs, gcols, gvals, p, q, x, rold, rnew, t, st, parity, pq_parts, npq); \
    break;
  switch (lanes) {
    LSB_FS(2) LSB_FS(4) LSB_FS(8) LSB_FS(16) LSB_FS(32)
  default:
    k_fsai_xr_gr<64><<<g, WG, 0, s>>>(n, rpw, goffs, gcols, gvals, p, q, x, rold, rnew, t, st, parity, pq_parts, npq);
  }
#undef LSB_FS
}

void lsb_k_fsai_gt_dots(unsigned n, const int *offs, const int *cols, const double *vals, unsigned lanes,
                        const double *t, double *z, const double *r, double *partials2, unsigned *npartials,
                        const struct lsb_pcg_state *st, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  unsigned rpw;
  const unsigned g = fsai_grid(n, lanes, &rpw);
  *npartials = g;
#define LSB_FS(LL)                                                                            \
  case LL:                                                                                    \
    k_fsai_gt_dots<LL><<<g, WG, 0, s>>>(n, rpw, offs, cols, vals, t, z, r, partials2, st);   \
    break;
  switch (lanes) {
    LSB_FS(2) LSB_FS(4) LSB_FS(8) LSB_FS(16) LSB_FS(32)
  default:
    k_fsai_gt_dots<64><<<g, WG, 0, s>>>(n, rpw, offs, cols, vals, t, z, r, partials2, st);
  }
#undef LSB_FS
}


/* Rows `rows[0..nrows)` of the pattern, each with at most mcap entries.  mcap <= 32: one
 * wavefront per row; else one workgroup per row (mcap <= LSB_FSAI_CAP = 128). */
void lsb_k_fsai_rows(const unsigned *rows, unsigned nrows, unsigned mcap, const int *offs, const int *cols,
                     const double *vals, unsigned row_begin, const unsigned *poffs, const unsigned *pcols,
                     double *gvals, int *bad, void *stream) {
  if (!nrows)
    return;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)mcap * mcap * 8 + (size_t)mcap * 8 + (size_t)mcap * 4;
  if (mcap > LSB_FSAI_CAP)
    errx(EXIT_FAILURE, "lsb_k_fsai_rows: %u pattern entries per row, at most %d", mcap, LSB_FSAI_CAP);
  static __thread int attr_set = 0;
  if (mcap > 32 && !attr_set) { /* the workgroup class only: more than 64 KB of LDS per workgroup has to be
                                   asked for, and the device has to have it */
    const int want = LSB_FSAI_CAP * LSB_FSAI_CAP * 8 + LSB_FSAI_CAP * 12;
    int dev = 0, have = 0;
    LSB_CHK_HIP(hipGetDevice(&dev));
    LSB_CHK_HIP(hipDeviceGetAttribute(&have, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
    if (have < want)
      errx(EXIT_FAILURE, "lsb_k_fsai_rows: FSAI rows of up to %d entries need %d bytes of LDS per workgroup, this "
                         "device offers %d (use a smaller --fsai-power)", LSB_FSAI_CAP, want, have);
    LSB_CHK_HIP(hipFuncSetAttribute((const void *)k_fsai_rows<256>, hipFuncAttributeMaxDynamicSharedMemorySize, want));
    attr_set = 1;
  }
  if (mcap <= 32)
    k_fsai_rows<64><<<nrows, 64, lds, s>>>(rows, mcap, offs, cols, vals, row_begin, poffs, pcols, gvals, bad);
  else
    k_fsai_rows<256><<<nrows, 256, lds, s>>>(rows, mcap, offs, cols, vals, row_begin, poffs, pcols, gvals, bad);
}

} // extern "C"
