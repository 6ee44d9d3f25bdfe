// a2-1, two-phase form (LSB_SPMV_TWOPHASE, host side lsb_csr_pbize) -- for operators
// whose rows scatter over far more of x than any cache level holds (power-law
// config 5).  The row-major kernels fetch a 128-byte line per 8-byte gather (PMC,
// 8 M rows: 2.7e8 L2 misses, 34 GB of fabric traffic for 3.2 GB of algorithmic
// bytes); the binned form (k_spmv_binned) keeps the window of x in L2 but still
// moves a 128-byte L2->L1 line per gather, 33 GB per launch at ~11.7 TB/s = its
// 2.8 ms.  Here no gather leaves the compute unit ("propagation blocking"):
//   phase 1 (k_pb_products)  entries ordered by column chunk (the chunk's window
//       of x is copied into LDS: 8192 columns = 64 KB), streamed as value +
//       16-bit column offset; the product goes to entry index + delta[piece] of
//       the row-bin-major product array -- a piece = the entries of one (chunk,
//       bin) pair, contiguous and in the same order on both sides, so the stores
//       of a piece are one contiguous run; the piece of an entry = grp_first of
//       its group of 64 + the set bits of grp_mask up to its lane (two scalar
//       loads per wavefront and trip);
//   phase 2 (k_pb_reduce)    a WAVEFRONT owns a row bin (2048 rows, 16 KB of LDS)
//       and streams the bin's slots in pairs: products + 16-bit row offsets, ds_add_f64
//       into its own LDS copy of the bin's rows -- nobody else adds there, so the
//       order of the additions is the wavefront's program order -- then writes
//       the bin's rows of y once, coalesced.  No y = 0 pass, no global atomics.
// Traffic per non-zero: 10.2 B read + 8 B written (phase 1), 10 B read (phase 2).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lsb_impl.h"

#define PB1_U 8   // entries per lane and trip of phase 1
#define PB2_WG 256
#define PB2_S 4   // steps of 128 slots a wavefront has in flight in phase 2

template <int WG>
__global__ __launch_bounds__(WG) void k_pb_products(
    const unsigned *__restrict__ item, const double *__restrict__ vals,
    const unsigned short *__restrict__ colw, const unsigned *__restrict__ grp_first,
    const unsigned long long *__restrict__ grp_mask, const unsigned *__restrict__ delta,
    const double *__restrict__ x, unsigned xlen, unsigned col_lo, unsigned cols,
    double *__restrict__ prod, int nt_store, const lsb_pcg_state *__restrict__ st) {
  extern __shared__ double sx[]; // cols doubles
  if (st && st->status)
    return;
  const unsigned tid = threadIdx.x, lane = tid & 63u;
  const unsigned c = item[3 * blockIdx.x], e0 = item[3 * blockIdx.x + 1],
                 e1 = item[3 * blockIdx.x + 2];
  const unsigned x0 = col_lo + c * cols;
  for (unsigned b = tid; b < cols; b += WG * 8) { // the chunk's window of x: coalesced, 8 loads in flight
    double t[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const unsigned i = b + (unsigned)k * WG;
      t[k] = i < cols && x0 + i < xlen ? x[x0 + i] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const unsigned i = b + (unsigned)k * WG;
      if (i < cols)
        sx[i] = t[k];
    }
  }
  __syncthreads();
  const unsigned long long le = lane == 63 ? ~0ull : (2ull << lane) - 1ull; // lanes <= mine
  if (e1 <= e0)
    return;
  // Loads on CLAMPED indices, no branches around them: all 2 * PB1_U loads of a lane
  // are in flight before the first product is formed (a trip's wavefronts read the
  // item's last entry again where they reach past its end; only the stores are
  // predicated).  e0 and WG are multiples of 64, so a wavefront = a group of 64.
  for (unsigned e = e0 + tid; e - lane < e1; e += WG * PB1_U) {
    double v[PB1_U];
    unsigned short cw[PB1_U];
    unsigned pc[PB1_U];
#pragma unroll
    for (int u = 0; u < PB1_U; u++) {
      const unsigned i = min(e + (unsigned)u * WG, e1 - 1u);
      v[u] = __builtin_nontemporal_load(vals + i);
      cw[u] = __builtin_nontemporal_load(colw + i);
    }
#pragma unroll
    for (int u = 0; u < PB1_U; u++) { // wave-uniform: the group's two words by scalar loads
      const unsigned i = min(e + (unsigned)u * WG, e1 - 1u);
      const unsigned g = __builtin_amdgcn_readfirstlane(i >> 6);
      pc[u] = grp_first[g] + (unsigned)__popcll(grp_mask[g] & le);
    }
    unsigned d[PB1_U];
#pragma unroll
    for (int u = 0; u < PB1_U; u++)
      d[u] = delta[pc[u]];
#pragma unroll
    for (int u = 0; u < PB1_U; u++) {
      const unsigned i = e + (unsigned)u * WG;
      if (i < e1) { // (index mod 2^32, like the host's delta)
        if (nt_store)
          __builtin_nontemporal_store(v[u] * sx[cw[u]], prod + (unsigned)(i + d[u]));
        else
          prod[i + d[u]] = v[u] * sx[cw[u]];
      }
    }
  }
}

__global__ __launch_bounds__(PB2_WG) void k_pb_reduce(
    const unsigned *__restrict__ bin_ptr, const double *__restrict__ prod,
    const unsigned short *__restrict__ roww, unsigned n, unsigned rows, unsigned nbins,
    double *__restrict__ y, const double *__restrict__ xdot, double *__restrict__ partials,
    const lsb_pcg_state *__restrict__ st) {
  extern __shared__ double sy[]; // (PB2_WG / 64) * rows doubles: a bin per wavefront
  __shared__ double sdot[PB2_WG / 64];
  if (st && st->status)
    return;
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const unsigned bin = blockIdx.x * (PB2_WG / 64) + wave;
  double *my = sy + (size_t)wave * rows;
  double dot = 0.0;
  if (bin < nbins) {
    for (unsigned k = lane; k < rows; k += 64)
      my[k] = 0.0;
    const unsigned s0 = bin_ptr[bin], s1 = bin_ptr[bin + 1];
    // a lane takes PAIRS of consecutive slots -- one 16-byte and one 4-byte load
    // (8-byte lane loads stream at 0.54-0.70x the 16-byte rate, MI355X_MICROARCH.md) --
    // from the even slot at or below s0 on; loads on clamped indices (see phase 1;
    // prod and roww are allocated two slots longer than they are used)
    typedef double d2 __attribute__((ext_vector_type(2)));
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const unsigned a0 = s0 & ~1u, last = (s1 - 1u) & ~1u;
    for (unsigned sb = a0; sb < s1; sb += 128 * PB2_S) {
      d2 p[PB2_S];
      us2 r[PB2_S];
#pragma unroll
      for (int g = 0; g < PB2_S; g++) {
        const unsigned i = min(sb + (unsigned)g * 128u + 2u * lane, last);
        r[g] = __builtin_nontemporal_load((const us2 *)(roww + i));
        p[g] = __builtin_nontemporal_load((const d2 *)(prod + i));
      }
#pragma unroll
      for (int g = 0; g < PB2_S; g++) {
        const unsigned i = sb + (unsigned)g * 128u + 2u * lane;
        if (i >= s0 && i < s1)
          unsafeAtomicAdd(my + r[g].x, p[g].x); // ds_add_f64 on this wavefront's own copy
        if (i + 1u < s1) // (i + 1 > a0 >= s0 - 1, and i + 1 == s0 only for i == a0 < s0)
          if (i + 1u >= s0)
            unsafeAtomicAdd(my + r[g].y, p[g].y);
      }
    }
    // the wavefront's LDS operations complete in order: the sums are final here
    const unsigned row0 = bin * rows;
    for (unsigned k = lane; k < rows; k += 64) {
      const unsigned row = row0 + k;
      if (row < n) {
        const double s = my[k];
        y[row] = s;
        if (xdot)
          dot += s * xdot[row];
      }
    }
  }
  if (partials) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
      dot += __shfl_xor(dot, off, 64);
    if (lane == 0)
      sdot[wave] = dot;
    __syncthreads();
    if (tid == 0)
      partials[blockIdx.x] = (sdot[0] + sdot[1]) + (sdot[2] + sdot[3]);
  }
}

// more workgroups than partial-sum slots: fold the per-workgroup partial sums
__global__ __launch_bounds__(256) void k_pb_fold(const double *__restrict__ in, unsigned nin,
                                                 double *__restrict__ out, unsigned nout) {
  const unsigned j = blockIdx.x * 256 + threadIdx.x;
  if (j >= nout)
    return;
  double s = 0.0;
  for (unsigned i = j; i < nin; i += nout)
    s += in[i];
  out[j] = s;
}

extern "C" {

/* workgroups of phase 2 = partial sums it leaves (binparts must hold that many) */
unsigned lsb_k_twophase_groups(unsigned nbins) { return (nbins + PB2_WG / 64 - 1) / (PB2_WG / 64); }

/* y = A x in two launches; prod: nnz doubles of scratch.  partials != NULL:
 * *npartials partial sums of y . xdot are left there (binparts: scratch for the
 * per-workgroup sums when there are more of them than the partial buffer holds).
 * cols <= 16384 (128 KB of LDS for the window of x), rows <= 4096. */
void lsb_k_spmv_twophase(unsigned nitems, const unsigned *item, const double *vals,
                         const unsigned short *colw, const unsigned *grp_first,
                         const unsigned long long *grp_mask, const unsigned *delta,
                         const unsigned short *roww, unsigned col_lo, unsigned cols, unsigned rows,
                         unsigned nbins, const unsigned *bin_ptr, double *prod, unsigned n,
                         const double *x, unsigned xlen, double *y, const double *xdot,
                         double *partials, unsigned *npartials, double *binparts,
                         const struct lsb_pcg_state *st, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  static __thread int attr_set = 0; /* a rank = a host thread with its own device */
  if (!attr_set) { /* more than 64 KB of LDS per workgroup has to be asked for */
    LSB_CHK_HIP(hipFuncSetAttribute((const void *)k_pb_products<1024>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8));
    LSB_CHK_HIP(hipFuncSetAttribute((const void *)k_pb_products<512>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8));
    LSB_CHK_HIP(hipFuncSetAttribute((const void *)k_pb_reduce,
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (PB2_WG / 64) * 4096 * 8));
    attr_set = 1;
  }
  if (cols > 16384 || rows > 4096)
    errx(EXIT_FAILURE, "lsb_k_spmv_twophase: tiling %u x %u does not fit the LDS", cols, rows);
  /* products leave through nontemporal stores: nobody reads them before 2 GB more have
   * gone by (1.55-1.70 against 1.61-1.71 ms on the 8 M-row power-law operator) */
  const int nt = 1;
  if (nitems) {
    if (cols > 8192)
      k_pb_products<1024><<<nitems, 1024, (size_t)cols * 8, s>>>(item, vals, colw, grp_first, grp_mask,
                                                                 delta, x, xlen, col_lo, cols, prod, nt, st);
    else
      k_pb_products<512><<<nitems, 512, (size_t)cols * 8, s>>>(item, vals, colw, grp_first, grp_mask,
                                                               delta, x, xlen, col_lo, cols, prod, nt, st);
  }
  const unsigned g = lsb_k_twophase_groups(nbins);
  const bool fold = partials && g > LSB_MAX_PARTIALS;
  k_pb_reduce<<<g, PB2_WG, (size_t)(PB2_WG / 64) * rows * 8, s>>>(
      bin_ptr, prod, roww, n, rows, nbins, y, partials ? xdot : NULL,
      partials ? (fold ? binparts : partials) : NULL, st);
  if (npartials)
    *npartials = fold ? LSB_MAX_PARTIALS : g;
  if (fold)
    k_pb_fold<<<(LSB_MAX_PARTIALS + 255) / 256, 256, 0, s>>>(binparts, g, partials, LSB_MAX_PARTIALS);
}

} // extern "C"
