// Device side of the direct-xGMI all-reduce (hip_p2p.hip), shared with the
// kernels it can be folded into (hip_kernels.hip): the CONTRIBUTE phase --
// reduce this rank's partial sums in a fixed order, store the 1-3 results into
// slot [epoch & 1][rank] of every peer's mailbox -- runs either in
// k_p2p_allreduce or as the tail of the SpMV launch that wrote the partial
// sums; the COLLECT phase -- wait for the R slots of the own mailbox, add them
// in rank order -- either there or at the head of k_cg1_update.  One launch
// fewer per sharded iteration; the same loads, adds and stores in the same
// order, so the iterates keep their bits whichever kernel does the work
// (ranks may even differ in that).
#ifndef LSB_HIP_AR_H
#define LSB_HIP_AR_H
#include <hip/hip_runtime.h>

#include "lsb_impl.h"

#define AR_WG 256 // both phases are written for workgroups of 4 wavefronts
#define AR_MAX_RANKS 64
typedef unsigned long long ar_u64;

__device__ __forceinline__ ar_u64 ar_ld_sys(const ar_u64 *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double ar_ld_sys(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// true when *flag reached `epoch` before the deadline.  Epochs only grow, so
// "reached" is >=: a waiter that arrives late for epoch k must not spin on a
// flag its peer has meanwhile moved on to k+1.
__device__ __forceinline__ bool ar_wait_flag(const ar_u64 *flag, ar_u64 epoch, long long timeout) {
  const long long t0 = wall_clock64();
  while (ar_ld_sys(flag) < epoch) {
    __builtin_amdgcn_s_sleep(1);
    if (wall_clock64() - t0 > timeout)
      return false;
  }
  return true;
}

// slot of rank r, parity b, inside a mailbox: 3 doubles + the epoch
__device__ __forceinline__ char *ar_slot(char *mbox, unsigned b, int r) {
  return mbox + ((size_t)b * AR_MAX_RANKS + (size_t)r) * 32;
}

// AGENT: the partial sums were written by other workgroups of the RUNNING
// launch (agent-scope stores): load them past the L1 as well.
template <bool AGENT>
__device__ __forceinline__ double ar_ld_part(const double *p) {
  if (AGENT)
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}

// Column sums of recs[n][W] over this thread's records t, t + AR_WG, ... in
// index order; eight records' loads are in flight at a time (the loads of a
// plain `v += recs[i]` loop come back one by one, ~1 us each).
template <bool AGENT, int W>
__device__ __forceinline__ void ar_colsum(const double *recs, unsigned n, double *v) {
  for (unsigned base = threadIdx.x; base < n; base += 8 * AR_WG) {
    double a[8][W];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const unsigned i = base + (unsigned)j * AR_WG;
#pragma unroll
      for (int c = 0; c < W; c++)
        a[j][c] = i < n ? ar_ld_part<AGENT>(recs + (size_t)i * W + c) : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (base + (unsigned)j * AR_WG < n) {
#pragma unroll
        for (int c = 0; c < W; c++)
          v[c] += a[j][c];
      }
  }
}
template <bool AGENT>
__device__ __forceinline__ void ar_colsum_w(const double *recs, unsigned n, unsigned width,
                                            double *v) {
  if (width == 1)
    ar_colsum<AGENT, 1>(recs, n, v);
  else if (width == 2)
    ar_colsum<AGENT, 2>(recs, n, v);
  else if (width == 3)
    ar_colsum<AGENT, 3>(recs, n, v);
}

// CONTRIBUTE, whole workgroup (AR_WG threads): column sums of
// parts[nparts][width] and parts2[nparts2][width2], then extra[0..nextra), go
// to every peer.  sred: 3 * AR_WG/64 doubles, sval: 3 doubles of LDS.
// FOLDED (tail of an SpMV launch): width = 1, width2 = 2, no extras -- fixed at
// compile time there, so that the tail stays small in registers.
template <bool AGENT, bool FOLDED>
__device__ __forceinline__ void ar_contribute(const double *parts, unsigned nparts, unsigned width,
                                              const double *parts2, unsigned nparts2,
                                              unsigned width2, const double *extra,
                                              unsigned nextra, char *const *peer, int R, int me,
                                              ar_u64 epoch, double *sred, double *sval) {
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const unsigned nvals = width + width2 + nextra, b = (unsigned)(epoch & 1);
  // my partial sums, fixed order (the same in both forms); one workgroup
  // reduction for all (up to three) columns -- this sits on the critical path
  // of every sharded iteration
  double v[3] = {0.0, 0.0, 0.0};
  if (FOLDED) {
    ar_colsum<AGENT, 1>(parts, nparts, v);
    ar_colsum<AGENT, 2>(parts2, nparts2, v + 1);
  } else {
    ar_colsum_w<AGENT>(parts, nparts, width, v);
    ar_colsum_w<AGENT>(parts2, nparts2, width2, v + width);
  }
#pragma unroll
  for (int k = 0; k < 3; k++)
    for (int off = 32; off > 0; off >>= 1)
      v[k] += __shfl_xor(v[k], off, 64);
  if (lane == 0)
    for (int k = 0; k < 3; k++)
      sred[wave * 3 + k] = v[k];
  __syncthreads();
  if (tid < 3) {
    double t = 0.0;
    for (unsigned w = 0; w < AR_WG / 64; w++)
      t += sred[w * 3 + tid];
    sval[tid] = t;
  }
  __syncthreads();
  if (tid < nextra)
    sval[width + width2 + tid] = extra[tid];
  __syncthreads();
  if (tid < (unsigned)R) { // lane r serves peer r (own mailbox included)
    double *slot = (double *)ar_slot(peer[tid], b, me);
    for (unsigned k = 0; k < nvals; k++)
      slot[k] = sval[k];
    __threadfence_system();
    __hip_atomic_store((ar_u64 *)(slot + 3), epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// COLLECT, ONE wavefront (all 64 lanes call it): out[k] = sum over ranks, in
// rank order -- identical bits on every rank.  false: a peer did not arrive.
// FENCE = false (every workgroup of a sweep collects): no acquire fence between
// the flag and the values -- all of them are system-scope loads that bypass
// the caches, and the values' loads are issued after the flag's has returned.
template <bool FENCE>
__device__ __forceinline__ bool ar_collect(char *mbox, int R, ar_u64 epoch, long long timeout,
                                           unsigned nvals, double (&out)[3]) {
  const unsigned lane = threadIdx.x & 63u, b = (unsigned)(epoch & 1);
  const double *slot = (const double *)ar_slot(mbox, b, lane < (unsigned)R ? (int)lane : 0);
  bool ok = true;
  if (lane < (unsigned)R)
    ok = ar_wait_flag((const ar_u64 *)(slot + 3), epoch, timeout);
  if (!__all(ok))
    return false;
  if (FENCE)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  double a[3];
#pragma unroll
  for (int k = 0; k < 3; k++) // all three loads in flight
    a[k] = lane < (unsigned)R && (unsigned)k < nvals ? ar_ld_sys(slot + k) : 0.0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    double s = 0.0;
    for (int r = 0; r < R; r++)
      s += __shfl(a[k], r, 64);
    out[k] = s;
  }
  return true;
}

// Tail of an SpMV launch (hip_kernels.hip): every workgroup hands in its dot
// partial; the one whose hand-in came last reduces ALL of them (the earlier
// launches' of a split SpMV first) together with the sweep's records and
// contributes.  Hand-off between workgroups of one launch without fences (the
// L2s of different XCDs are not coherent for plain accesses): agent-scope
// store of the partial, s_waitcnt, agent-scope add; the last workgroup loads
// agent-scope after its add has returned (MI355X_MICROARCH.md, hand-off table).
__device__ __forceinline__ void ar_tail(double *partials, double mine, const lsb_ar_tail &t) {
  __shared__ double sred[3 * (AR_WG / 64)];
  __shared__ double sval[3];
  __shared__ unsigned last;
  if (threadIdx.x == 0) {
    __hip_atomic_store(partials + blockIdx.x, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned old =
        __hip_atomic_fetch_add(t.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = old + 1 == gridDim.x;
    if (last) // the next launch with a tail finds it at zero
      __hip_atomic_store(t.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!last)
    return;
  ar_contribute<true, true>(t.parts, t.nparts_before + gridDim.x, 1, t.parts2, t.nparts2, 2, NULL, 0,
                            t.peer, t.R, t.me, t.epoch, sred, sval);
}
#endif
