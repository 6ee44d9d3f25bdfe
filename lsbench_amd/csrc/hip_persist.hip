// Launch-bound operators: the WHOLE Jacobi-PCG solve as ONE kernel launch.
//
// tests/xn3b_A_18.txt (BASELINE configs[1]: n = 3461, 76.6 k non-zeros, 267
// iterations at tol 1e-12) is 0.9 MB of matrix: an SpMV is 0.2 us of HBM time and
// an iteration of the launch-per-kernel form is two dependent ~4 us launches.
// Here G workgroups (one per CU: each reserves > 80 KB of LDS) stay resident for
// the whole solve:
//   * workgroup g owns a contiguous row range; its rows of the CSR are copied
//     into LDS once, and x, r, u = D^-1 r, p, s of its rows live in registers;
//   * an iteration is the single-reduction form of hip_kernels.hip's
//     k_cg1_update (Chronopoulos-Gear: same iterates as PCG), so that only TWO
//     grid-wide synchronisations are needed:
//        [all]  copy the shared vector u into LDS; w = S u on own rows;
//               publish partial sums (r.u, r.r of the previous update, w.u)
//        -- barrier 1 --
//        [all]  add the G partial records in rank order (identical bits in every
//               workgroup: everybody takes the same stop decision), beta, alpha,
//               p = u + beta p, s = w + beta s, x += alpha p, r -= alpha s,
//               u = D^-1 r; publish own rows of u
//        -- barrier 2 --
//   * everything that crosses workgroups (u, the partial records, the barrier
//     counter) is written with agent-scope `sc1` stores, drained
//     (s_waitcnt vmcnt(0) in every storing wave, then a workgroup barrier) before
//     ONE lane adds to the counter; consumers poll the counter with sc1 loads
//     from one lane, join a workgroup barrier, and read with sc1 loads -- the
//     hand-off form MI355X_MICROARCH.md tabulates as valid without L2
//     write-back / L1 invalidate fences (and the L2s of different XCDs are not
//     coherent for anything else).
// Every spin is bounded (wall_clock64): a workgroup that never arrives -- the
// grid not co-resident because something else holds the CUs -- ends the solve
// with LSB_STATUS_COMM instead of hanging the device.
//
// opts.persistent: 1 = use it where the operator qualifies, 0 = never, -1 = time
// both forms at solver creation and keep the faster one (hip_pcg.c).  The
// reference has no counterpart (its iterative call site is src/ginkgo.cpp:55-69).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lsb_impl.h"

#define PS_WG 256
#define PS_NZMAX 6144  // non-zeros of one workgroup's rows (LDS: 12 B each)
#define PS_RMAX 512    // rows of one workgroup (2 per thread)
#define PS_NMAX 8192   // rows of the operator (the shared vector sits in LDS)
#define PS_GMAX 64

typedef unsigned long long u64;

struct ps_shared { // global memory, zeroed before every launch
  unsigned counter; // barrier arrivals, only grows
  unsigned fail;    // a spin timed out somewhere
  unsigned pad[14];
  double parts[PS_GMAX * 4]; // per workgroup: r.u, r.r, w.u, (unused)
};

__device__ __forceinline__ double ld_agent(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(double *p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// grid barrier: all of this workgroup's sc1 stores are drained, one lane
// arrives and polls.  false = timed out (or somebody else did).
__device__ __forceinline__ bool ps_barrier(ps_shared *sh, unsigned target, long long timeout,
                                           int *sflag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&sh->counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the poll is ONE load per turn: the clock and the fail word (each a memory
    // round trip of its own) are looked at every 256 turns only
    long long t0 = 0;
    unsigned spins = 0;
    int ok = 1;
    while (__hip_atomic_load(&sh->counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if ((++spins & 255u) == 0) {
        const long long now = wall_clock64();
        if (t0 == 0)
          t0 = now;
        if (now - t0 > timeout ||
            __hip_atomic_load(&sh->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(&sh->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
      }
    }
    *sflag = ok;
  }
  __syncthreads();
  return *sflag != 0;
}

template <int L>
__device__ __forceinline__ void ps_spmv(int nr, const int *soffs, const int *scol,
                                        const double *sval, const double *su, double *sw) {
  const unsigned tid = threadIdx.x, slot = tid / L, l = tid % L;
  constexpr unsigned SLOTS = PS_WG / L;
  for (unsigned base = 0; base < (unsigned)nr; base += SLOTS) {
    const unsigned r = base + slot;
    double s = 0.0;
    if (r < (unsigned)nr) {
      const int j0 = soffs[r], j1 = soffs[r + 1];
      for (int j = j0 + (int)l; j < j1; j += L)
        s += sval[j] * su[scol[j]];
    }
#pragma unroll
    for (int off = L >> 1; off > 0; off >>= 1)
      s += __shfl_xor(s, off, 64);
    if (r < (unsigned)nr && l == 0)
      sw[r] = s;
  }
}

__global__ __launch_bounds__(PS_WG) void k_pcg_persist(
    unsigned n, unsigned G, unsigned stride, const unsigned *__restrict__ wg_row,
    const int *__restrict__ offs, const int *__restrict__ cols, const double *__restrict__ vals,
    const double *__restrict__ dinv, const double *__restrict__ b, double *__restrict__ x,
    double *ug, ps_shared *sh, lsb_pcg_state *__restrict__ st, double tol, int maxit,
    unsigned lanes, long long timeout) {
  extern __shared__ unsigned char smem[];
  double *sval = (double *)smem;                 // PS_NZMAX
  double *su = sval + PS_NZMAX;                  // PS_NMAX
  double *sw = su + PS_NMAX;                     // PS_RMAX
  double *sred = sw + PS_RMAX;                   // 16
  int *scol = (int *)(sred + 16);                // PS_NZMAX
  int *soffs = scol + PS_NZMAX;                  // PS_RMAX + 1
  int *sflag = soffs + PS_RMAX + 1;
  // workgroups are dealt round-robin over the 8 XCDs: with stride = 8 the G
  // working ones (blockIdx % 8 == 0) share one XCD, the others leave at once
  if (blockIdx.x % stride)
    return;
  const unsigned g = blockIdx.x / stride, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const unsigned ra = wg_row[g], rb = wg_row[g + 1];
  const int nr = (int)(rb - ra), j0 = offs[ra];
  // ---- once per solve: my rows of the matrix into LDS ------------------------
  for (int i = (int)tid; i <= nr; i += PS_WG)
    soffs[i] = offs[ra + i] - j0;
  const int nz = offs[rb] - j0;
  for (int j = (int)tid; j < nz; j += PS_WG) {
    scol[j] = cols[j0 + j];
    sval[j] = vals[j0 + j];
  }
  // my rows of the vectors: thread t holds rows t and t + 256
  double xr[2] = {0.0, 0.0}, rr_[2], ur[2], pr[2] = {0.0, 0.0}, sr[2] = {0.0, 0.0}, dr[2], wr[2];
  double acc[3] = {0.0, 0.0, 0.0}; // r.u, r.r (b.b at first), w.u
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int i = (int)tid + k * PS_WG;
    rr_[k] = ur[k] = dr[k] = 0.0;
    if (i < nr) {
      const double bi = b[ra + i];
      dr[k] = dinv[ra + i];
      rr_[k] = bi, ur[k] = dr[k] * bi;
      st_agent(ug + ra + i, ur[k]);
      acc[0] += bi * ur[k];
      acc[1] += bi * bi;
    }
  }
  unsigned epoch = 0;
  if (!ps_barrier(sh, G * ++epoch, timeout, sflag)) // u is complete
    goto failed;
  {
    double g_old = 0.0, a_old = 0.0, thresh2 = 0.0, bb = 0.0, rrs = 0.0;
    int it = 0, status = LSB_STATUS_RUNNING;
    for (;;) {
      // ---- w = S u on my rows, partial sums --------------------------------
      { // all loads first (up to 32 in flight per lane), then the LDS stores: a
        // load-store loop waits out one memory round trip per element
        double tmp[PS_NMAX / PS_WG];
#pragma unroll
        for (int k = 0; k < PS_NMAX / PS_WG; k++) {
          const unsigned i = tid + (unsigned)k * PS_WG;
          tmp[k] = i < n ? ld_agent(ug + i) : 0.0;
        }
#pragma unroll
        for (int k = 0; k < PS_NMAX / PS_WG; k++) {
          const unsigned i = tid + (unsigned)k * PS_WG;
          if (i < n)
            su[i] = tmp[k];
        }
      }
      __syncthreads();
      switch (lanes) {
      case 2: ps_spmv<2>(nr, soffs, scol, sval, su, sw); break;
      case 4: ps_spmv<4>(nr, soffs, scol, sval, su, sw); break;
      case 8: ps_spmv<8>(nr, soffs, scol, sval, su, sw); break;
      case 16: ps_spmv<16>(nr, soffs, scol, sval, su, sw); break;
      case 32: ps_spmv<32>(nr, soffs, scol, sval, su, sw); break;
      default: ps_spmv<64>(nr, soffs, scol, sval, su, sw); break;
      }
      __syncthreads();
      acc[2] = 0.0;
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int i = (int)tid + k * PS_WG;
        wr[k] = i < nr ? sw[i] : 0.0;
        acc[2] += wr[k] * ur[k];
      }
      // workgroup sums (fixed order), published by one lane
#pragma unroll
      for (int c = 0; c < 3; c++)
        for (int off = 32; off > 0; off >>= 1)
          acc[c] += __shfl_xor(acc[c], off, 64);
      if (lane == 0)
        for (int c = 0; c < 3; c++)
          sred[wave * 3 + c] = acc[c];
      __syncthreads();
      if (tid < 3)
        st_agent(&sh->parts[g * 4 + tid],
                 (sred[tid] + sred[3 + tid]) + (sred[6 + tid] + sred[9 + tid]));
      if (!ps_barrier(sh, G * ++epoch, timeout, sflag))
        goto failed;
      // ---- the G records, in rank order: the same bits in every workgroup ---
      double tot[3], pv[3];
#pragma unroll
      for (int c = 0; c < 3; c++)
        pv[c] = lane < G ? ld_agent(&sh->parts[lane * 4 + c]) : 0.0;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        double s = 0.0;
        for (unsigned q = 0; q < G; q++)
          s += __shfl(pv[c], (int)q, 64);
        tot[c] = s;
      }
      const double g_new = tot[0], delta = tot[2];
      rrs = tot[1];
      if (it == 0) // first pass: r = b, tot[1] is b.b
        bb = rrs, thresh2 = tol * tol * bb;
      if (rrs <= thresh2) { // also b = 0
        status = LSB_STATUS_CONVERGED;
        break;
      }
      if (it >= maxit) {
        status = LSB_STATUS_MAXIT;
        break;
      }
      double beta = 0.0, alpha;
      if (a_old == 0.0) {
        alpha = g_new / delta;
      } else {
        beta = g_new / g_old;
        alpha = g_new / (delta - beta * g_new / a_old);
      }
      if (!isfinite(alpha) || alpha == 0.0) {
        status = LSB_STATUS_BREAKDOWN;
        break;
      }
      g_old = g_new, a_old = alpha, it++;
      acc[0] = acc[1] = 0.0;
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int i = (int)tid + k * PS_WG;
        if (i < nr) {
          pr[k] = ur[k] + beta * pr[k];
          sr[k] = wr[k] + beta * sr[k];
          xr[k] += alpha * pr[k];
          rr_[k] -= alpha * sr[k];
          ur[k] = dr[k] * rr_[k];
          st_agent(ug + ra + i, ur[k]);
          acc[0] += rr_[k] * ur[k];
          acc[1] += rr_[k] * rr_[k];
        }
      }
      if (!ps_barrier(sh, G * ++epoch, timeout, sflag))
        goto failed;
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int i = (int)tid + k * PS_WG;
      if (i < nr)
        x[ra + i] = xr[k];
    }
    if (g == 0 && tid == 0) {
      st->iters = it, st->status = status, st->rr = rrs, st->bb = bb, st->thresh2 = thresh2;
      st->maxit = maxit, st->pad = 0;
    }
    return;
  }
failed:
  if (g == 0 && tid == 0)
    st->status = LSB_STATUS_COMM, st->iters = 0;
}

extern "C" {

size_t lsb_k_persist_shared_bytes(void) { return sizeof(ps_shared); }
unsigned lsb_k_persist_limits(unsigned *nzmax, unsigned *rmax, unsigned *gmax) {
  *nzmax = PS_NZMAX, *rmax = PS_RMAX, *gmax = PS_GMAX;
  return PS_NMAX;
}

static size_t ps_lds_bytes(void) {
  return (size_t)(PS_NZMAX + PS_NMAX + PS_RMAX + 16) * sizeof(double) +
         (size_t)(PS_NZMAX + PS_RMAX + 1 + 4) * sizeof(int);
}

/* one solve; `shared` (lsb_k_persist_shared_bytes) must be zero on entry */
int lsb_k_pcg_persist(unsigned n, unsigned G, unsigned stride, const unsigned *wg_row,
                      const int *offs, const int *cols, const double *vals, const double *dinv,
                      const double *b, double *x, double *ug, void *shared,
                      struct lsb_pcg_state *st, double tol, int maxit, unsigned lanes,
                      long long timeout_ticks, void *stream) {
  static int attr_set = 0;
  const size_t lds = ps_lds_bytes();
  if (!attr_set) {
    if (hipFuncSetAttribute((const void *)k_pcg_persist, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return 1;
    attr_set = 1;
  }
  k_pcg_persist<<<G * stride, PS_WG, lds, (hipStream_t)stream>>>(
      n, G, stride, wg_row, offs, cols, vals, dinv, b, x, ug, (ps_shared *)shared, st, tol, maxit,
      lanes, timeout_ticks);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

} // extern "C"
