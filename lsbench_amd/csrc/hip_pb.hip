// a2-1, two-phase form (LSB_SPMV_TWOPHASE, host side lsb_csr_pbize) -- for operators
// whose rows scatter over far more of x than any cache level holds (power-law
// config 5).  The row-major kernels fetch a 128-byte line per 8-byte gather (PMC,
// 8 M rows: 2.7e8 L2 misses, 34 GB of fabric traffic for 3.2 GB of algorithmic
// bytes); the binned form (k_spmv_binned) keeps the window of x in L2 but still
// moves a 128-byte L2->L1 line per gather, 33 GB per launch at ~11.7 TB/s = its
// 2.5-2.8 ms.  Here no gather leaves the compute unit ("propagation blocking"):
//   phase 1 (k_pb_products)  entries ordered by column chunk (4096 columns: the
//       window of x is copied into LDS, 32 KB), streamed as value + 16-bit column
//       offset + 32-bit target slot; each product is stored into its slot of the
//       row-bin-major product array (the products of one (chunk, bin) pair go to
//       consecutive slots: contiguous pieces);
//   phase 2 (k_pb_reduce)    a workgroup owns a row bin (2048 rows) and streams the
//       bin's slots, wave w the steps w, w+4, ... of 64 slots: product + 16-bit row
//       offset, added into the wave's own LDS copy of the bin's rows.  The host
//       laid the slots out so that equal rows of a step are neighbours: a segmented
//       shuffle scan combines them and one lane per row adds -- 64 lanes, distinct
//       LDS words, nothing to resolve.  The four copies are summed in a
//       fixed order and y is written once, coalesced, by one lane per row -- no
//       y = 0 pass, no atomics, the same bits every run.
// Traffic per non-zero: 14 B read + 8 B written (phase 1), 10 B read (phase 2).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lsb_impl.h"

#define PB_WG 256
#define PB_U 8 // entries per lane and trip of phase 1

__global__ __launch_bounds__(PB_WG) void k_pb_products(
    const unsigned *__restrict__ item, const double *__restrict__ vals,
    const unsigned short *__restrict__ colw, const unsigned *__restrict__ pos,
    const double *__restrict__ x, unsigned xlen, unsigned col_lo, double *__restrict__ prod,
    const lsb_pcg_state *__restrict__ st) {
  __shared__ double sx[LSB_PB_COLS];
  if (st && st->status)
    return;
  const unsigned tid = threadIdx.x;
  const unsigned c = item[3 * blockIdx.x], e0 = item[3 * blockIdx.x + 1],
                 e1 = item[3 * blockIdx.x + 2];
  const unsigned x0 = col_lo + c * LSB_PB_COLS;
  { // the chunk's window of x: coalesced loads, all in flight, then LDS
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
  for (unsigned e = e0 + tid; e < e1; e += PB_WG * PB_U) {
    double v[PB_U];
    unsigned short cw[PB_U];
    unsigned ps[PB_U];
#pragma unroll
    for (int u = 0; u < PB_U; u++) {
      const unsigned i = e + (unsigned)u * PB_WG;
      if (i < e1) {
        v[u] = __builtin_nontemporal_load(vals + i);
        cw[u] = __builtin_nontemporal_load(colw + i);
        ps[u] = __builtin_nontemporal_load(pos + i);
      }
    }
#pragma unroll
    for (int u = 0; u < PB_U; u++) {
      const unsigned i = e + (unsigned)u * PB_WG;
      if (i < e1)
        prod[ps[u]] = v[u] * sx[cw[u]];
    }
  }
}

#define PB_S 8 // steps a wave has in flight
__global__ __launch_bounds__(PB_WG) void k_pb_reduce(
    const unsigned *__restrict__ bin_ptr, const double *__restrict__ prod,
    const unsigned short *__restrict__ roww, unsigned n, double *__restrict__ y,
    const double *__restrict__ xdot, double *__restrict__ partials,
    const lsb_pcg_state *__restrict__ st) {
  __shared__ double sy[4][LSB_PB_ROWS]; // 64 KB: everything LDS a workgroup may name statically
  if (st && st->status)
    return;
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, bin = blockIdx.x;
  double *my = sy[wave];
#pragma unroll
  for (int k = 0; k < LSB_PB_ROWS / 64; k++)
    my[lane + k * 64] = 0.0;
  const unsigned s0 = bin_ptr[bin], s1 = bin_ptr[bin + 1]; // multiples of 64
  // wave w: steps w, w+4, ...; PB_S of them loaded before any is added
  for (unsigned sb = s0 + wave * 64; sb < s1; sb += 4 * 64 * PB_S) {
    double p[PB_S];
    unsigned short r[PB_S];
#pragma unroll
    for (int g = 0; g < PB_S; g++) {
      const unsigned i = sb + (unsigned)g * 256u + lane;
      p[g] = 0.0, r[g] = 0xFFFFu;
      if (i < s1) {
        r[g] = __builtin_nontemporal_load(roww + i);
        p[g] = __builtin_nontemporal_load(prod + i);
      }
    }
#pragma unroll
    for (int g = 0; g < PB_S; g++) {
      // equal rows of a step are neighbours (host layout): segmented inclusive scan
      // on the row id, then the LAST lane of a row adds -- one lane per LDS word
      double pv = p[g];
      const unsigned rv = r[g];
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const double p2 = __shfl_up(pv, d, 64);
        const unsigned r2 = __shfl_up(rv, d, 64);
        if ((int)lane >= d && r2 == rv)
          pv += p2;
      }
      const unsigned rn = __shfl_down(rv, 1, 64);
      if (rv != 0xFFFFu && (lane == 63 || rn != rv)) // (padding: never written, never used)
        my[rv] += pv;
    }
  }
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

/* y = A x in two launches; prod: nslots doubles of scratch.  partials != NULL:
 * *npartials partial sums of y . xdot are left there (binparts: nbins doubles of
 * scratch when nbins exceeds the partial buffer). */
void lsb_k_spmv_twophase(unsigned nitems, const unsigned *item, const double *vals,
                         const unsigned short *colw, const unsigned *pos, const unsigned short *roww,
                         unsigned col_lo, unsigned nbins, const unsigned *bin_ptr, double *prod,
                         unsigned n, const double *x, unsigned xlen, double *y, const double *xdot,
                         double *partials, unsigned *npartials, double *binparts,
                         const struct lsb_pcg_state *st, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (nitems)
    k_pb_products<<<nitems, PB_WG, 0, s>>>(item, vals, colw, pos, x, xlen, col_lo, prod, st);
  const bool fold = partials && nbins > LSB_MAX_PARTIALS;
  k_pb_reduce<<<nbins, PB_WG, 0, s>>>(bin_ptr, prod, roww, n, y, partials ? xdot : NULL,
                                      partials ? (fold ? binparts : partials) : NULL, st);
  if (npartials)
    *npartials = fold ? LSB_MAX_PARTIALS : nbins;
  if (fold)
    k_pb_fold<<<(LSB_MAX_PARTIALS + PB_WG - 1) / PB_WG, PB_WG, 0, s>>>(binparts, nbins, partials,
                                                                       LSB_MAX_PARTIALS);
}

} // extern "C"
