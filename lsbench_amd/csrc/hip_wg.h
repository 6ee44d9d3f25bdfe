// Workgroup-level helpers shared by the kernel files (hip_kernels.hip, hip_fsai.hip): fixed-order
// reductions (every thread returns with the same bits; a result depends on the number of partial
// records only, never on timing) and the XCD-contiguous workgroup numbering.
#ifndef LSB_HIP_WG_H
#define LSB_HIP_WG_H
#include <hip/hip_runtime.h>

#define WG 256      // threads per workgroup = 4 wavefronts of 64
#define NXCD 8      // XCDs per MI355X; blocks are dealt round-robin over them

// --------------------------------------------------------------------------
// Reductions: wavefront butterfly -> LDS across the 4 waves, fixed order.
// Every thread of the workgroup returns with the same bits.
// --------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ void wg_sum(double (&v)[W], double *sred /*4*W*/) {
#pragma unroll
  for (int k = 0; k < W; k++) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
      v[k] += __shfl_xor(v[k], off, 64);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads(); // sred may still be read from a previous call
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

// Sum `nparts` partial records of width W (written by an earlier launch, one
// record per workgroup) in an order that depends only on nparts.
template <int W>
__device__ __forceinline__ void wg_sum_partials(const double *__restrict__ parts,
                                                unsigned nparts, double (&v)[W],
                                                double *sred) {
#pragma unroll
  for (int k = 0; k < W; k++)
    v[k] = 0.0;
  for (unsigned i = threadIdx.x; i < nparts; i += WG) {
#pragma unroll
    for (int k = 0; k < W; k++)
      v[k] += parts[(size_t)i * W + k];
  }
  wg_sum<W>(v, sred);
}

// Logical workgroup id such that each XCD owns a contiguous range of logical
// ids (gridDim.x is a multiple of NXCD).  Placement is a speed matter only.
__device__ __forceinline__ unsigned xcd_contiguous_wg() {
  const unsigned b = blockIdx.x, per = gridDim.x / NXCD;
  return (b % NXCD) * per + b / NXCD;
}

#endif
