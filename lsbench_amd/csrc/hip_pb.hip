// a2-1, two-phase form (LSB_SPMV_TWOPHASE, host side lsb_csr_pbize) -- for operators
// whose rows scatter over far more of x than any cache level holds (power-law
// config 5).  The row-major kernels fetch a 128-byte line per 8-byte gather (PMC,
// 8 M rows: 2.7e8 L2 misses, 34 GB of fabric traffic for 3.2 GB of algorithmic
// bytes); the binned form (k_spmv_binned) keeps the window of x in L2 but still
// moves a 128-byte L2->L1 line per gather, 33 GB per launch at ~11.7 TB/s = its
// 2.5-2.8 ms.  Here no gather leaves the compute unit ("propagation blocking"):
//   phase 1 (k_pb_products)  entries ordered by column chunk (8192 columns: the
//       window of x is copied into LDS, 64 KB), streamed as value + 16-bit column
//       offset; one product per entry is written out, in entry order;
//   phase 2 (k_pb_reduce)    inside a chunk the entries are ordered by row bin
//       (2048 rows) and row, so each (chunk, bin) pair is a contiguous row-sorted
//       run of products.  A workgroup owns a bin; wave w adds runs w, w+4, ... --
//       product + 16-bit row offset -- into ITS copy of the bin's 2048 rows in LDS
//       (equal neighbouring rows of a step are combined by a segmented shuffle
//       scan first, so one lane adds per row and step); the four copies are
//       summed in a fixed order and y is written once, coalesced, by one lane per
//       row -- no y = 0 pass, no atomics, the same bits every run.
// Traffic per non-zero: 10 B read + 8 B written (phase 1), 10 B read (phase 2).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lsb_impl.h"

#define PB_WG 256
#define PB_U 8 // entries per lane and trip of phase 1

__global__ __launch_bounds__(PB_WG) void k_pb_products(
    const unsigned *__restrict__ item, const double *__restrict__ vals,
    const unsigned short *__restrict__ colw, const double *__restrict__ x, unsigned xlen,
    unsigned col_lo, double *__restrict__ prod, const lsb_pcg_state *__restrict__ st) {
  __shared__ double sx[LSB_PB_COLS];
  if (st && st->status)
    return;
  const unsigned tid = threadIdx.x;
  const unsigned c = item[3 * blockIdx.x], e0 = item[3 * blockIdx.x + 1],
                 e1 = item[3 * blockIdx.x + 2];
  const unsigned x0 = col_lo + c * LSB_PB_COLS;
  { // the chunk's window of x: 32 coalesced loads per lane, all in flight, then LDS
    double t[LSB_PB_COLS / PB_WG];
#pragma unroll
    for (int k = 0; k < LSB_PB_COLS / PB_WG; k++) {
      const unsigned i = x0 + tid + (unsigned)k * PB_WG;
      t[k] = i < xlen ? x[i] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < LSB_PB_COLS / PB_WG; k++)
      sx[tid + k * PB_WG] = t[k];
  }
  __syncthreads();
  // software pipeline over trips of PB_U x 256 entries: the loads of trip t+1 are
  // issued before the products of trip t are formed and stored
  double v[2][PB_U];
  unsigned short cw[2][PB_U];
#define PB_LOAD(buf, ebase)                                                    \
  _Pragma("unroll") for (int u = 0; u < PB_U; u++) {                           \
    const unsigned i_ = (ebase) + (unsigned)u * PB_WG;                         \
    if (i_ < e1) {                                                             \
      v[buf][u] = __builtin_nontemporal_load(vals + i_);                       \
      cw[buf][u] = __builtin_nontemporal_load(colw + i_);                      \
    }                                                                          \
  }
#define PB_STORE(buf, ebase)                                                   \
  _Pragma("unroll") for (int u = 0; u < PB_U; u++) {                           \
    const unsigned i_ = (ebase) + (unsigned)u * PB_WG;                         \
    if (i_ < e1)                                                               \
      __builtin_nontemporal_store(v[buf][u] * sx[cw[buf][u]], prod + i_);      \
  }
  unsigned e = e0 + tid;
  if (e < e1)
    PB_LOAD(0, e);
  while (e < e1) {
    const unsigned en = e + PB_WG * PB_U;
    if (en < e1)
      PB_LOAD(1, en);
    PB_STORE(0, e);
    e = en;
    if (e >= e1)
      break;
    const unsigned en2 = e + PB_WG * PB_U;
    if (en2 < e1)
      PB_LOAD(0, en2);
    PB_STORE(1, e);
    e = en2;
  }
#undef PB_LOAD
#undef PB_STORE
}

#define PB_G 8 // runs a wave has in flight
__global__ __launch_bounds__(PB_WG) void k_pb_reduce(
    const unsigned *__restrict__ bin_run, const unsigned *__restrict__ run,
    const double *__restrict__ prod, const unsigned short *__restrict__ roww, unsigned n,
    double *__restrict__ y, const double *__restrict__ xdot, double *__restrict__ partials,
    const lsb_pcg_state *__restrict__ st) {
  __shared__ double sy[4][LSB_PB_ROWS]; // 64 KB: everything LDS a workgroup may name statically
  if (st && st->status)
    return;
  const unsigned tid = threadIdx.x, lane = tid & 63u, bin = blockIdx.x;
  const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6); // uniform: run descriptors go through scalar loads
  double *my = sy[wave];
#pragma unroll
  for (int k = 0; k < LSB_PB_ROWS / 64; k++)
    my[lane + k * 64] = 0.0;
  const unsigned k0 = bin_run[bin], k1 = bin_run[bin + 1];
  // Groups of PB_G runs, two groups in flight: while group g is added up, the
  // products and rows of group g+1 are loading (a wave that waits out one memory
  // round trip per run reads a few GB/s; the bin's runs lie scattered over the
  // product array).
  unsigned start[2][PB_G], len[2][PB_G]; // wave-uniform: scalar registers
  double p[2][PB_G];
  unsigned r[2][PB_G];
#define PB_DESC(slot, kb_)                                                     \
  _Pragma("unroll") for (int g = 0; g < PB_G; g++) {                           \
    const unsigned k = (kb_) + 4u * (unsigned)g;                               \
    start[slot][g] = 0, len[slot][g] = 0;                                      \
    if (k < k1)                                                                \
      start[slot][g] = __builtin_amdgcn_readfirstlane(run[2 * (size_t)k]),     \
      len[slot][g] = __builtin_amdgcn_readfirstlane(run[2 * (size_t)k + 1]);   \
  }
#define PB_DATA(buf, slot)                                                     \
  _Pragma("unroll") for (int g = 0; g < PB_G; g++) {                           \
    p[buf][g] = 0.0, r[buf][g] = 0xFFFFu;                                      \
    if (lane < len[slot][g]) {                                                 \
      p[buf][g] = __builtin_nontemporal_load(prod + start[slot][g] + lane);    \
      r[buf][g] = __builtin_nontemporal_load(roww + start[slot][g] + lane);    \
    }                                                                          \
  }
#define PB_ADD(buf, slot)                                                      \
  _Pragma("unroll") for (int g = 0; g < PB_G; g++) {                           \
    for (unsigned off = 0; off < len[slot][g]; off += 64) {                    \
      double pv = p[buf][g];                                                   \
      unsigned rv = r[buf][g];                                                 \
      if (off) { /* a run longer than a wavefront: its further steps, as they come */ \
        pv = 0.0, rv = 0xFFFFu;                                                \
        if (off + lane < len[slot][g]) {                                       \
          pv = prod[start[slot][g] + off + lane];                              \
          rv = roww[start[slot][g] + off + lane];                              \
        }                                                                      \
      }                                                                        \
      /* rows ascend inside a run: equal rows are neighbours.  Segmented        \
       * inclusive scan (Hillis-Steele on the key), the LAST lane of a row adds */ \
      _Pragma("unroll") for (int d = 1; d < 64; d <<= 1) {                     \
        const double p2 = __shfl_up(pv, d, 64);                                \
        const unsigned r2 = __shfl_up(rv, d, 64);                              \
        if ((int)lane >= d && r2 == rv)                                        \
          pv += p2;                                                            \
      }                                                                        \
      const unsigned rn = __shfl_down(rv, 1, 64);                              \
      if (rv != 0xFFFFu && (lane == 63 || rn != rv))                           \
        my[rv] += pv;                                                          \
    }                                                                          \
  }
  const unsigned stride = 4 * PB_G;
  unsigned kb = k0 + wave;
  if (kb < k1) {
    PB_DESC(0, kb);
    PB_DATA(0, 0);
    for (;;) {
      PB_DESC(1, kb + stride); // (scalar loads: a few hundred cycles, not a vector round trip)
      PB_DATA(1, 1);
      PB_ADD(0, 0);
      kb += stride;
      if (kb >= k1)
        break;
      PB_DESC(0, kb + stride);
      PB_DATA(0, 0);
      PB_ADD(1, 1);
      kb += stride;
      if (kb >= k1)
        break;
    }
  }
#undef PB_DESC
#undef PB_DATA
#undef PB_ADD
  __syncthreads();
  double dot = 0.0;
  for (unsigned i = tid; i < LSB_PB_ROWS; i += PB_WG) {
    const unsigned row = bin * LSB_PB_ROWS + i;
    if (row < n) {
      const double s = (sy[0][i] + sy[1][i]) + (sy[2][i] + sy[3][i]);
      y[row] = s;
      if (xdot)
        dot += s * xdot[row];
    }
  }
  if (partials) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
      dot += __shfl_xor(dot, off, 64);
    __syncthreads(); // every read of sy above is done: its first words carry the wave sums
    if (lane == 0)
      sy[0][wave] = dot;
    __syncthreads();
    if (tid == 0)
      partials[bin] = (sy[0][0] + sy[0][1]) + (sy[0][2] + sy[0][3]);
  }
}

// more bins than partial-sum slots: fold the per-bin partial sums
__global__ __launch_bounds__(PB_WG) void k_pb_fold(const double *__restrict__ in, unsigned nin,
                                                   double *__restrict__ out, unsigned nout) {
  const unsigned j = blockIdx.x * PB_WG + threadIdx.x;
  if (j >= nout)
    return;
  double s = 0.0;
  for (unsigned i = j; i < nin; i += nout)
    s += in[i];
  out[j] = s;
}

extern "C" {

/* y = A x in two launches; prod: nnz doubles of scratch.  partials != NULL:
 * *npartials partial sums of y . xdot are left there (binparts: nbins doubles of
 * scratch when nbins exceeds the partial buffer). */
void lsb_k_spmv_twophase(unsigned nitems, const unsigned *item, const double *vals,
                         const unsigned short *colw, const unsigned short *roww, unsigned col_lo,
                         unsigned nbins, const unsigned *bin_run, const unsigned *run, double *prod,
                         unsigned n, const double *x, unsigned xlen, double *y, const double *xdot,
                         double *partials, unsigned *npartials, double *binparts,
                         const struct lsb_pcg_state *st, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (nitems)
    k_pb_products<<<nitems, PB_WG, 0, s>>>(item, vals, colw, x, xlen, col_lo, prod, st);
  const bool fold = partials && nbins > LSB_MAX_PARTIALS;
  k_pb_reduce<<<nbins, PB_WG, 0, s>>>(bin_run, run, prod, roww, n, y, partials ? xdot : NULL,
                                      partials ? (fold ? binparts : partials) : NULL, st);
  if (npartials)
    *npartials = fold ? LSB_MAX_PARTIALS : nbins;
  if (fold)
    k_pb_fold<<<(LSB_MAX_PARTIALS + PB_WG - 1) / PB_WG, PB_WG, 0, s>>>(binparts, nbins, partials,
                                                                       LSB_MAX_PARTIALS);
}

} // extern "C"
