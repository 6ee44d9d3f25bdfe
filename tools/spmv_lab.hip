// SpMV kernel laboratory (development tool, not shipped in the library).
// Builds a synthetic operator on the host, runs N kernel variants interleaved
// in ONE process (cdna_hip_programming.md section 5.4 rule 24), checks each
// against a host reference and prints median microseconds and algorithmic GB/s
// (12*nnz + 20*n + 4 bytes per launch).
//   hipcc --offload-arch=gfx950 -O3 tools/spmv_lab.hip -o tools/spmv_lab
//   tools/spmv_lab lap2d 3162 | lap3d 400 | powerlaw 8000000
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHK(x)                                                                 \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

#define WG 256
#define NXCD 8

template <int W>
__device__ __forceinline__ void wg_sum(double (&v)[W], double *sred) {
#pragma unroll
  for (int k = 0; k < W; k++)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
      v[k] += __shfl_xor(v[k], off, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0)
    for (int k = 0; k < W; k++)
      sred[wave * W + k] = v[k];
  __syncthreads();
  for (int k = 0; k < W; k++)
    v[k] = (sred[0 * W + k] + sred[1 * W + k]) + (sred[2 * W + k] + sred[3 * W + k]);
}

enum { F_NOREMAP = 1, F_FAKEGATHER = 2, F_NODOT = 4, F_NOX = 8, F_CYCLIC = 16, F_NT = 32, F_PREFETCH = 64, F_HOIST = 128, F_TWOROW = 256, F_ROWCAP = 512, F_PERIOD = 1024, F_C16 = 2048, F_C16W = 4096 };
// F_C16: `cols` points at int16 deltas relative to the block's first row.
// F_C16W: `cols` points at uint16 codes, 2 bits window + 14 bits offset; the
// four window bases of block b are g_wbase[4b..4b+3] (passed through `per`'s
// neighbour argument wbase).
__device__ const int *g_wbase;

// BLAS-1 probe shaped like k_pcg_update_xr: 5 streams in, 2 out, 16 B/lane.
// NT bit 0: nontemporal loads, bit 1: nontemporal stores.
typedef double lab2_d2v __attribute__((ext_vector_type(2)));
template <int NT>
__global__ __launch_bounds__(256) void k_blas1_probe(size_t n2, const lab2_d2v *__restrict__ p,
                                                     const lab2_d2v *__restrict__ q,
                                                     const lab2_d2v *__restrict__ d,
                                                     lab2_d2v *__restrict__ x,
                                                     lab2_d2v *__restrict__ r, double alpha,
                                                     double *__restrict__ out) {
  double acc = 0.0;
  const size_t g = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += g) {
    lab2_d2v pv, qv, dv, xv, rv;
    if (NT & 1) {
      pv = __builtin_nontemporal_load(p + i), qv = __builtin_nontemporal_load(q + i);
      dv = __builtin_nontemporal_load(d + i), xv = __builtin_nontemporal_load(x + i);
      rv = __builtin_nontemporal_load(r + i);
    } else {
      pv = p[i], qv = q[i], dv = d[i], xv = x[i], rv = r[i];
    }
    xv += alpha * pv;
    rv -= alpha * qv;
    if (NT & 2) {
      __builtin_nontemporal_store(xv, x + i);
      __builtin_nontemporal_store(rv, r + i);
    } else {
      x[i] = xv, r[i] = rv;
    }
    acc += rv.x * (dv.x * rv.x) + rv.y * (dv.y * rv.y);
  }
  if (acc == 123.456)
    out[0] = acc;
}
template <int FLAGS, class T> __device__ __forceinline__ T ldg(const T *p) {
  if (FLAGS & F_NT) return __builtin_nontemporal_load(p);
  return *p;
}


template <int FLAGS>
__device__ __forceinline__ unsigned logical_wg() {
  if (FLAGS & F_NOREMAP)
    return blockIdx.x;
  const unsigned b = blockIdx.x, per = gridDim.x / NXCD;
  return (b % NXCD) * per + b / NXCD;
}

// ---- V0: the library's adaptive kernel (8 B / 4 B per lane loads) ----------
template <int CAP, int FLAGS>
__global__ __launch_bounds__(WG) void k_adaptive(const int *__restrict__ rowblk, unsigned nblk,
                                                 unsigned per, const int *__restrict__ offs,
                                                 const int *__restrict__ cols,
                                                 const double *__restrict__ vals,
                                                 const double *__restrict__ x,
                                                 double *__restrict__ y,
                                                 double *__restrict__ partials) {
  __shared__ double sprod[CAP];
  __shared__ double sred[4];
  const unsigned tid = threadIdx.x, w = logical_wg<FLAGS>();
  const unsigned k0 = w * per, k1 = min(k0 + per, nblk);
  constexpr int U = CAP / WG;
  double dot = 0.0;
  for (unsigned k = k0; k < k1; k++) {
    const int r0 = rowblk[k], r1 = rowblk[k + 1];
    const int j0 = offs[r0], j1 = offs[r1];
    const int cnt = j1 - j0, nr = r1 - r0;
    if (cnt <= CAP) {
      int c[U];
      double v[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int t = tid + u * WG;
        if (t < cnt) {
          c[u] = cols[j0 + t];
          v[u] = vals[j0 + t];
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int t = tid + u * WG;
        if (t < cnt) {
          if (FLAGS & F_NOX)
            sprod[t] = v[u] * (double)c[u];
          else if (FLAGS & F_FAKEGATHER)
            sprod[t] = v[u] * x[c[u] & 1023];
          else
            sprod[t] = v[u] * x[c[u]];
        }
      }
      __syncthreads();
      unsigned L = 1;
      while (L < 64 && (unsigned)nr * (L * 2) <= WG)
        L <<= 1;
      const unsigned slot = tid / L, l = tid % L, slots = WG / L;
      for (unsigned rb = 0; rb < (unsigned)nr; rb += slots) {
        const unsigned r = rb + slot;
        double s = 0.0;
        if (r < (unsigned)nr) {
          const int a = offs[r0 + r] - j0, b = offs[r0 + r + 1] - j0;
          for (int j = a + (int)l; j < b; j += (int)L)
            s += sprod[j];
        }
        for (unsigned off = L >> 1; off > 0; off >>= 1)
          s += __shfl_xor(s, off, 64);
        if (r < (unsigned)nr && l == 0) {
          y[r0 + r] = s;
          if (!(FLAGS & F_NODOT))
            dot += s * x[r0 + r];
        }
      }
      __syncthreads();
    } else {
      double s[1] = {0.0};
      for (int j = j0 + (int)tid; j < j1; j += WG)
        s[0] += vals[j] * x[cols[j]];
      wg_sum<1>(s, sred);
      if (tid == 0) {
        y[r0] = s[0];
        dot += s[0] * x[r0];
      }
    }
  }
  if (!(FLAGS & F_NODOT)) {
    double d[1] = {dot};
    wg_sum<1>(d, sred);
    if (tid == 0)
      partials[w] = d[0];
  }
}

// ---- V1: offsets staged in LDS too; reduce phase reads no global memory -----
// and the next block's loads are issued before the current block is reduced.
template <int CAP, int FLAGS>
__global__ __launch_bounds__(WG) void k_adaptive_pipe(const int *__restrict__ rowblk,
                                                      unsigned nblk, unsigned per,
                                                      const int *__restrict__ offs,
                                                      const int *__restrict__ cols,
                                                      const double *__restrict__ vals,
                                                      const double *__restrict__ x,
                                                      double *__restrict__ y,
                                                      double *__restrict__ partials) {
  __shared__ double sprod[CAP];
  __shared__ int soffs[CAP + 1]; // a block has at most CAP rows with >=1 nnz; empty rows: fallback
  __shared__ double sred[4];
  const unsigned tid = threadIdx.x, w = logical_wg<FLAGS>();
  const unsigned k0 = w * per, k1 = min(k0 + per, nblk);
  constexpr int U = CAP / WG;
  double dot = 0.0;
  int c[U];
  double v[U];
  int r0 = 0, r1 = 0, j0 = 0, j1 = 0;
  if (k0 < k1) {
    r0 = rowblk[k0], r1 = rowblk[k0 + 1];
    j0 = offs[r0], j1 = offs[r1];
    if (j1 - j0 <= CAP) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int t = tid + u * WG;
        if (t < j1 - j0) {
          c[u] = cols[j0 + t];
          v[u] = vals[j0 + t];
        }
      }
    }
  }
  for (unsigned k = k0; k < k1; k++) {
    const int cnt = j1 - j0, nr = r1 - r0;
    const int cr0 = r0, cj0 = j0;
    if (cnt <= CAP && nr <= CAP) {
      // gather + park products
      double xv[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int t = tid + u * WG;
        if (t < cnt)
          xv[u] = x[c[u]];
      }
      // row offsets of this block -> LDS (coalesced)
      for (int t = tid; t <= nr; t += WG)
        soffs[t] = offs[cr0 + t] - cj0;
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int t = tid + u * WG;
        if (t < cnt)
          sprod[t] = v[u] * xv[u];
      }
      // prefetch the next block's stream before reducing this one
      if (k + 1 < k1) {
        r0 = rowblk[k + 1], r1 = rowblk[k + 2];
        j0 = offs[r0], j1 = offs[r1];
        if (j1 - j0 <= CAP) {
#pragma unroll
          for (int u = 0; u < U; u++) {
            const int t = tid + u * WG;
            if (t < j1 - j0) {
              c[u] = cols[j0 + t];
              v[u] = vals[j0 + t];
            }
          }
        }
      }
      __syncthreads();
      unsigned L = 1;
      while (L < 64 && (unsigned)nr * (L * 2) <= WG)
        L <<= 1;
      const unsigned slot = tid / L, l = tid % L, slots = WG / L;
      for (unsigned rb = 0; rb < (unsigned)nr; rb += slots) {
        const unsigned r = rb + slot;
        double s = 0.0;
        if (r < (unsigned)nr) {
          const int a = soffs[r], b = soffs[r + 1];
          for (int j = a + (int)l; j < b; j += (int)L)
            s += sprod[j];
        }
        for (unsigned off = L >> 1; off > 0; off >>= 1)
          s += __shfl_xor(s, off, 64);
        if (r < (unsigned)nr && l == 0) {
          y[cr0 + r] = s;
          if (!(FLAGS & F_NODOT))
            dot += s * x[cr0 + r];
        }
      }
      __syncthreads();
    } else {
      double s[1] = {0.0};
      const int ej = cj0 + cnt;
      // long row (or a block of > CAP rows, all but one empty): generic path
      for (int r = cr0; r < cr0 + nr; r++) {
        s[0] = 0.0;
        const int a = offs[r], b = offs[r + 1];
        for (int j = a + (int)tid; j < b; j += WG)
          s[0] += vals[j] * x[cols[j]];
        wg_sum<1>(s, sred);
        if (tid == 0) {
          y[r] = s[0];
          dot += s[0] * x[r];
        }
      }
      (void)ej;
      if (k + 1 < k1) {
        r0 = rowblk[k + 1], r1 = rowblk[k + 2];
        j0 = offs[r0], j1 = offs[r1];
        if (j1 - j0 <= CAP) {
#pragma unroll
          for (int u = 0; u < U; u++) {
            const int t = tid + u * WG;
            if (t < j1 - j0) {
              c[u] = cols[j0 + t];
              v[u] = vals[j0 + t];
            }
          }
        }
      }
    }
  }
  if (!(FLAGS & F_NODOT)) {
    double d[1] = {dot};
    wg_sum<1>(d, sred);
    if (tid == 0)
      partials[w] = d[0];
  }
}

// ---- V2: 16 B/lane loads.  The block's nnz range is widened to a multiple of
// 4 at both ends (cols as int4, vals as 2 x double2); the extra entries are
// masked.  Offsets via LDS as in V1, no prefetch.
template <int CAP, int FLAGS>
__global__ __launch_bounds__(WG) void k_adaptive_v4(const int *__restrict__ rowblk,
                                                    unsigned nblk, unsigned per,
                                                    const int *__restrict__ offs,
                                                    const int *__restrict__ cols,
                                                    const double *__restrict__ vals,
                                                    const double *__restrict__ x,
                                                    double *__restrict__ y,
                                                    double *__restrict__ partials,
                                                    int nnz_total) {
  __shared__ double sprod[CAP + 8];
  __shared__ int soffs[CAP + 1];
  __shared__ double sred[4];
  const unsigned tid = threadIdx.x, w = logical_wg<FLAGS>();
  const unsigned k0 = w * per, k1 = min(k0 + per, nblk);
  constexpr int Q = (CAP / 4 + WG) / WG; // quads per thread (+1 quad of slack)
  double dot = 0.0;
  for (unsigned k = k0; k < k1; k++) {
    const int r0 = rowblk[k], r1 = rowblk[k + 1];
    const int j0 = offs[r0], j1 = offs[r1];
    const int cnt = j1 - j0, nr = r1 - r0;
    if (cnt <= CAP && nr <= CAP) {
      const int ja = j0 & ~3;           // aligned start
      const int nq = (j1 - ja + 3) / 4; // quads to fetch
      int4 c4[Q];
      double2 va[Q], vb[Q];
#pragma unroll
      for (int u = 0; u < Q; u++) {
        const int q = tid + u * WG;
        if (q < nq && ja + 4 * q + 4 <= ((nnz_total + 3) & ~3)) {
          c4[u] = *(const int4 *)(cols + ja + 4 * q);
          va[u] = *(const double2 *)(vals + ja + 4 * q);
          vb[u] = *(const double2 *)(vals + ja + 4 * q + 2);
        }
      }
      for (int t = tid; t <= nr; t += WG)
        soffs[t] = offs[r0 + t] - j0;
      const int sh = j0 - ja; // 0..3 leading entries to drop
#pragma unroll
      for (int u = 0; u < Q; u++) {
        const int q = tid + u * WG;
        if (q < nq) {
          const int e = 4 * q - sh; // index of the quad's first entry in the block
          const int cc[4] = {c4[u].x, c4[u].y, c4[u].z, c4[u].w};
          const double vv[4] = {va[u].x, va[u].y, vb[u].x, vb[u].y};
#pragma unroll
          for (int i = 0; i < 4; i++)
            if (e + i >= 0 && e + i < cnt)
              sprod[e + i] = vv[i] * ((FLAGS & F_NOX) ? (double)cc[i] : x[cc[i]]);
        }
      }
      __syncthreads();
      unsigned L = 1;
      while (L < 64 && (unsigned)nr * (L * 2) <= WG)
        L <<= 1;
      const unsigned slot = tid / L, l = tid % L, slots = WG / L;
      for (unsigned rb = 0; rb < (unsigned)nr; rb += slots) {
        const unsigned r = rb + slot;
        double s = 0.0;
        if (r < (unsigned)nr) {
          const int a = soffs[r], b = soffs[r + 1];
          for (int j = a + (int)l; j < b; j += (int)L)
            s += sprod[j];
        }
        for (unsigned off = L >> 1; off > 0; off >>= 1)
          s += __shfl_xor(s, off, 64);
        if (r < (unsigned)nr && l == 0) {
          y[r0 + r] = s;
          if (!(FLAGS & F_NODOT))
            dot += s * x[r0 + r];
        }
      }
      __syncthreads();
    } else {
      double s[1];
      for (int r = r0; r < r1; r++) {
        s[0] = 0.0;
        const int a = offs[r], b = offs[r + 1];
        for (int j = a + (int)tid; j < b; j += WG)
          s[0] += vals[j] * x[cols[j]];
        wg_sum<1>(s, sred);
        if (tid == 0) {
          y[r] = s[0];
          dot += s[0] * x[r];
        }
      }
    }
  }
  if (!(FLAGS & F_NODOT)) {
    double d[1] = {dot};
    wg_sum<1>(d, sred);
    if (tid == 0)
      partials[w] = d[0];
  }
}

// ---- V3: blocks dealt CYCLICALLY to the workgroups of one XCD (the XCD's 256
// resident workgroups then sweep 256 ADJACENT row blocks together, so the x
// window they gather from fits the XCD's 4 MiB L2), optional register prefetch
// of the next block's stream and nontemporal stream loads.
template <int CAP, int FLAGS, int MINW = 8>
__global__ __launch_bounds__(WG, MINW) void k_adaptive_cyc(const int *__restrict__ rowblk, unsigned nblk,
                                                 unsigned per, const int *__restrict__ offs,
                                                 const int *__restrict__ cols,
                                                 const double *__restrict__ vals,
                                                 const double *__restrict__ x,
                                                 double *__restrict__ y,
                                                 double *__restrict__ partials) {
  __shared__ double sprod[CAP];
  __shared__ double sred[4];
  const unsigned tid = threadIdx.x;
  const unsigned gx = gridDim.x / NXCD;           // workgroups per XCD
  const unsigned xcd = blockIdx.x % NXCD, slot = blockIdx.x / NXCD;
  // contiguous mode: XCD c owns blocks [c*chunk, (c+1)*chunk).  F_PERIOD: the
  // block sequence is cut into periods of T = `per` blocks (one stencil plane,
  // i.e. the matrix bandwidth) and XCD c owns the c-th eighth of EVERY period,
  // so the +-bandwidth neighbours of its blocks are its own blocks one period
  // away: reuse distance T/8 blocks instead of T.
  const unsigned T = (FLAGS & F_PERIOD) ? per : nblk;
  const unsigned sg = (T + NXCD - 1) / NXCD;      // blocks per XCD per period
  const unsigned nper = (nblk + T - 1) / T;
  const unsigned kend = nper * sg;                // virtual index space of this XCD
  constexpr int U = CAP / WG;
  double dot = 0.0;
  int c[U];
  double v[U];
  unsigned k = slot;
  unsigned kb = 0; // actual block id of the block in flight
  int r0 = 0, r1 = 0, j0 = 0, j1 = 0;
#define ISSUE_WG(kk)                                                           \
  do {                                                                         \
    const unsigned pos_ = xcd * sg + (kk) % sg;                                \
    kb = ((kk) / sg) * T + pos_;                                               \
    if (pos_ >= T || kb >= nblk) {                                             \
      r0 = r1 = j0 = j1 = 0;                                                   \
      break;                                                                   \
    }                                                                          \
    r0 = rowblk[kb], r1 = rowblk[kb + 1];                                      \
    j0 = offs[r0], j1 = offs[r1];                                              \
    int wb0_ = 0, wb1_ = 0, wb2_ = 0, wb3_ = 0;                                \
    if (FLAGS & F_C16W)                                                        \
      wb0_ = g_wbase[4 * kb], wb1_ = g_wbase[4 * kb + 1], wb2_ = g_wbase[4 * kb + 2], wb3_ = g_wbase[4 * kb + 3]; \
    if (j1 - j0 <= CAP) {                                                      \
      _Pragma("unroll") for (int u = 0; u < U; u++) {                          \
        const int t = tid + u * WG;                                            \
        if (t < j1 - j0) {                                                     \
          if (FLAGS & F_C16)                                                   \
            c[u] = r0 + (int)ldg<FLAGS>((const short *)cols + j0 + t);         \
          else if (FLAGS & F_C16W) {                                           \
            const unsigned code_ = ldg<FLAGS>((const unsigned short *)cols + j0 + t); \
            const unsigned w_ = code_ >> 14;                                   \
            const int lo_ = (w_ & 1) ? wb1_ : wb0_, hi_ = (w_ & 1) ? wb3_ : wb2_; \
            c[u] = ((w_ & 2) ? hi_ : lo_) + (int)(code_ & 16383u);             \
          } else                                                               \
            c[u] = ldg<FLAGS>(cols + j0 + t);                                  \
          v[u] = ldg<FLAGS>(vals + j0 + t);                                    \
        }                                                                      \
      }                                                                        \
    }                                                                          \
  } while (0)
  if (k < kend)
    ISSUE_WG(k);
  for (; k < kend; k += gx) {
    const int cr0 = r0, cj0 = j0, cnt = j1 - j0, nr = r1 - r0;
    if (cnt <= CAP) {
      unsigned L = 1;
      while (L < 64 && (unsigned)nr * (L * 2) <= WG)
        L <<= 1;
      // F_HOIST: row offsets and the dot operand of this lane's (up to two)
      // rows are requested together with the gathers, not after the barrier
      const bool fast = (FLAGS & (F_HOIST | F_TWOROW)) && L == 1 && nr <= 2 * WG;
      int oa0 = 0, ob0 = 0, oa1 = 0, ob1 = 0;
      double xd0 = 0.0, xd1 = 0.0;
      if (fast && (FLAGS & F_HOIST)) {
        if ((int)tid < nr) {
          oa0 = offs[cr0 + tid] - cj0, ob0 = offs[cr0 + tid + 1] - cj0;
          xd0 = x[cr0 + tid];
        }
        if ((int)tid + WG < nr) {
          oa1 = offs[cr0 + tid + WG] - cj0, ob1 = offs[cr0 + tid + WG + 1] - cj0;
          xd1 = x[cr0 + tid + WG];
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int t = tid + u * WG;
        if (t < cnt) {
          if (FLAGS & F_NOX)
            sprod[t] = v[u] * (double)c[u];
          else if (FLAGS & F_FAKEGATHER)
            sprod[t] = v[u] * x[c[u] & 1023];
          else
            sprod[t] = v[u] * x[c[u]];
        }
      }
      if ((FLAGS & F_PREFETCH) && k + gx < kend)
        ISSUE_WG(k + gx);
      __syncthreads();
      if (fast && (FLAGS & F_TWOROW)) { // both rows of this lane in ONE pass
        if ((int)tid < nr) {
          oa0 = offs[cr0 + tid] - cj0, ob0 = offs[cr0 + tid + 1] - cj0;
          xd0 = x[cr0 + tid];
        }
        if ((int)tid + WG < nr) {
          oa1 = offs[cr0 + tid + WG] - cj0, ob1 = offs[cr0 + tid + WG + 1] - cj0;
          xd1 = x[cr0 + tid + WG];
        }
      }
      if (fast) {
        double s0 = 0.0, s1 = 0.0;
        for (int j = oa0; j < ob0; j++)
          s0 += sprod[j];
        for (int j = oa1; j < ob1; j++)
          s1 += sprod[j];
        if ((int)tid < nr)
          y[cr0 + tid] = s0, dot += s0 * xd0;
        if ((int)tid + WG < nr)
          y[cr0 + tid + WG] = s1, dot += s1 * xd1;
      }
      const unsigned sl = tid / L, l = tid % L, slots = WG / L;
      for (unsigned rb = 0; !fast && rb < (unsigned)nr; rb += slots) {
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
          y[cr0 + r] = s;
          if (!(FLAGS & F_NODOT))
            dot += s * x[cr0 + r];
        }
      }
      __syncthreads();
    } else {
      double s[1] = {0.0};
      for (int j = cj0 + (int)tid; j < cj0 + cnt; j += WG)
        s[0] += vals[j] * x[cols[j]];
      wg_sum<1>(s, sred);
      if (tid == 0) {
        y[cr0] = s[0];
        dot += s[0] * x[cr0];
      }
      if ((FLAGS & F_PREFETCH) && k + gx < kend)
        ISSUE_WG(k + gx);
    }
    if (!(FLAGS & F_PREFETCH) && k + gx < kend)
      ISSUE_WG(k + gx);
  }
  if (!(FLAGS & F_NODOT)) {
    double d[1] = {dot};
    wg_sum<1>(d, sred);
    if (tid == 0)
      partials[xcd * gx + slot] = d[0];
  }
}

// ---- SELL-64: rows in slices of 64 (one wavefront), a slice stored column-
// major and padded to its longest row: lane i owns row 64s+i, every stream load
// of a wave is one contiguous 256 B / 512 B run, the gathers of a stencil row
// group are contiguous too, no LDS, no barrier, no row offsets.  cols/vals point
// at the SELL arrays, g_sell_ptr at the slice offsets; `per` carries n.
__device__ const unsigned *g_sell_ptr;
template <int FLAGS, int U>
__global__ __launch_bounds__(WG, 8) void k_sell(const int *, unsigned nslice, unsigned n, const int *,
                                                const int *__restrict__ cols,
                                                const double *__restrict__ vals,
                                                const double *__restrict__ x,
                                                double *__restrict__ y,
                                                double *__restrict__ partials) {
  __shared__ double sred[4];
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const unsigned gx = gridDim.x / NXCD, xcd = blockIdx.x % NXCD, slot = blockIdx.x / NXCD;
  const unsigned ngrp = (nslice + 3) / 4, chunk = (ngrp + NXCD - 1) / NXCD;
  const unsigned g0 = xcd * chunk, g1 = g0 + chunk < ngrp ? g0 + chunk : ngrp;
  const unsigned *sp = g_sell_ptr;
  double dot = 0.0;
  for (unsigned g = g0 + slot; g < g1; g += gx) {
    const unsigned s = __builtin_amdgcn_readfirstlane(g * 4 + wave);
    if (s < nslice) {
      const unsigned base = sp[s], len = (sp[s + 1] - base) >> 6;
      const int *cp = cols + base + lane;
      const double *vp = vals + base + lane;
      double acc = 0.0;
      int c[U];
      double v[U];
#pragma unroll
      for (int u = 0; u < U; u++)
        if ((unsigned)u < len) {
          c[u] = ldg<FLAGS>(cp + u * 64);
          v[u] = ldg<FLAGS>(vp + u * 64);
        }
#pragma unroll
      for (int u = 0; u < U; u++)
        if ((unsigned)u < len)
          acc += v[u] * x[c[u]];
      for (unsigned j = U; j < len; j++)
        acc += ldg<FLAGS>(vp + j * 64) * x[ldg<FLAGS>(cp + j * 64)];
      const unsigned row = s * 64 + lane;
      if (row < n) {
        y[row] = acc;
        if (!(FLAGS & F_NODOT))
          dot += acc * x[row];
      }
    }
  }
  if (!(FLAGS & F_NODOT)) {
    double d[1] = {dot};
    wg_sum<1>(d, sred);
    if (tid == 0)
      partials[xcd * gx + slot] = d[0];
  }
}

// ---- SELL-128x2: slices of 128 rows, lane l owns rows 2l and 2l+1, whose j-th
// entries sit side by side: 8 B column / 16 B value loads per lane.
typedef int lab_i2v __attribute__((ext_vector_type(2)));
typedef double lab_d2vb __attribute__((ext_vector_type(2)));
__device__ const unsigned *g_sell2_ptr;
template <int FLAGS, int U, int MINW>
__global__ __launch_bounds__(WG, MINW) void k_sell2(const int *, unsigned nslice, unsigned n, const int *,
                                                 const int *__restrict__ cols,
                                                 const double *__restrict__ vals,
                                                 const double *__restrict__ x,
                                                 double *__restrict__ y,
                                                 double *__restrict__ partials) {
  __shared__ double sred[4];
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const unsigned gx = gridDim.x / NXCD, xcd = blockIdx.x % NXCD, slot = blockIdx.x / NXCD;
  const unsigned ngrp = (nslice + 3) / 4, chunk = (ngrp + NXCD - 1) / NXCD;
  const unsigned g0 = xcd * chunk, g1 = g0 + chunk < ngrp ? g0 + chunk : ngrp;
  const unsigned *sp = g_sell2_ptr;
  double dot = 0.0;
  for (unsigned g = g0 + slot; g < g1; g += gx) {
    const unsigned s = __builtin_amdgcn_readfirstlane(g * 4 + wave);
    if (s < nslice) {
      const unsigned base = sp[s], len = (sp[s + 1] - base) >> 7;
      const lab_i2v *cp = (const lab_i2v *)(cols + base) + lane;
      const lab_d2vb *vp = (const lab_d2vb *)(vals + base) + lane;
      double a0 = 0.0, a1 = 0.0;
      lab_i2v c[U];
      lab_d2vb v[U];
#pragma unroll
      for (int u = 0; u < U; u++)
        if ((unsigned)u < len) {
          c[u] = ldg<FLAGS>(cp + u * 64);
          v[u] = ldg<FLAGS>(vp + u * 64);
        }
#pragma unroll
      for (int u = 0; u < U; u++)
        if ((unsigned)u < len) {
          a0 += v[u].x * x[c[u].x];
          a1 += v[u].y * x[c[u].y];
        }
      for (unsigned j = U; j < len; j++) {
        const lab_i2v cc = ldg<FLAGS>(cp + j * 64);
        const lab_d2vb vv = ldg<FLAGS>(vp + j * 64);
        a0 += vv.x * x[cc.x];
        a1 += vv.y * x[cc.y];
      }
      const unsigned row = s * 128 + 2 * lane;
      if (row + 1 < n) {
        lab_d2vb o = {a0, a1};
        *(lab_d2vb *)(y + row) = o;
        if (!(FLAGS & F_NODOT)) {
          const lab_d2vb xx = *(const lab_d2vb *)(x + row);
          dot += a0 * xx.x;
          dot += a1 * xx.y;
        }
      } else if (row < n) {
        y[row] = a0;
        if (!(FLAGS & F_NODOT))
          dot += a0 * x[row];
      }
    }
  }
  if (!(FLAGS & F_NODOT)) {
    double d[1] = {dot};
    wg_sum<1>(d, sred);
    if (tid == 0)
      partials[xcd * gx + slot] = d[0];
  }
}

// ---- SELL-128x2 with 16-bit row-relative column codes (probe: how much of the
// 4 B/nnz of column traffic converts into time).  cols = short deltas col - row.
typedef short lab_s2v __attribute__((ext_vector_type(2)));
template <int FLAGS, int U, int MINW>
__global__ __launch_bounds__(WG, MINW) void k_sell2c16(const int *, unsigned nslice, unsigned n, const int *,
                                                    const int *__restrict__ cols,
                                                    const double *__restrict__ vals,
                                                    const double *__restrict__ x,
                                                    double *__restrict__ y,
                                                    double *__restrict__ partials) {
  __shared__ double sred[4];
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const unsigned gx = gridDim.x / NXCD, xcd = blockIdx.x % NXCD, slot = blockIdx.x / NXCD;
  const unsigned ngrp = (nslice + 3) / 4, chunk = (ngrp + NXCD - 1) / NXCD;
  const unsigned g0 = xcd * chunk, g1 = g0 + chunk < ngrp ? g0 + chunk : ngrp;
  const unsigned *sp = g_sell2_ptr;
  double dot = 0.0;
  for (unsigned g = g0 + slot; g < g1; g += gx) {
    const unsigned s = __builtin_amdgcn_readfirstlane(g * 4 + wave);
    if (s < nslice) {
      const unsigned base = sp[s], len = (sp[s + 1] - base) >> 7;
      const lab_s2v *cp = (const lab_s2v *)((const short *)cols + base) + lane;
      const lab_d2vb *vp = (const lab_d2vb *)(vals + base) + lane;
      const int row = (int)(s * 128 + 2 * lane);
      double a0 = 0.0, a1 = 0.0;
      for (unsigned j0 = 0; j0 < len; j0 += U) {
        lab_s2v c[U];
        lab_d2vb v[U];
#pragma unroll
        for (int u = 0; u < U; u++)
          if (j0 + u < len) {
            c[u] = ldg<FLAGS>(cp + (j0 + u) * 64);
            v[u] = ldg<FLAGS>(vp + (j0 + u) * 64);
          }
#pragma unroll
        for (int u = 0; u < U; u++)
          if (j0 + u < len) {
            a0 += v[u].x * x[row + (int)c[u].x];
            a1 += v[u].y * x[row + 1 + (int)c[u].y];
          }
      }
      if ((unsigned)row + 1 < n) {
        lab_d2vb o = {a0, a1};
        *(lab_d2vb *)(y + row) = o;
        const lab_d2vb xx = *(const lab_d2vb *)(x + row);
        dot += a0 * xx.x;
        dot += a1 * xx.y;
      } else if ((unsigned)row < n) {
        y[row] = a0;
        dot += a0 * x[row];
      }
    }
  }
  double d[1] = {dot};
  wg_sum<1>(d, sred);
  if (tid == 0)
    partials[xcd * gx + slot] = d[0];
}

// ---- probe: the p update folded into the sliced-ELL SpMV.  Every gathered
// column computes p_new = c r + beta p_old on the fly (two gathers instead of
// one), the row owner also stores p_new; q = S p_new.  Compare with
// "S2C16" + a separate 240 MB sweep (p = c r + beta p).
__device__ const double *g_fuse_r;
__device__ double *g_fuse_pnew;
template <int FLAGS, int U, int MINW>
__global__ __launch_bounds__(WG, MINW) void k_sell2c16p(const int *, unsigned nslice, unsigned n, const int *,
                                                     const int *__restrict__ cols,
                                                     const double *__restrict__ vals,
                                                     const double *__restrict__ x,
                                                     double *__restrict__ y,
                                                     double *__restrict__ partials) {
  __shared__ double sred[4];
  const double cc = 0.25, beta = 0.37;
  const double *__restrict__ rr = g_fuse_r;
  double *__restrict__ pn = g_fuse_pnew;
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const unsigned gx = gridDim.x / NXCD, xcd = blockIdx.x % NXCD, slot = blockIdx.x / NXCD;
  const unsigned ngrp = (nslice + 3) / 4, chunk = (ngrp + NXCD - 1) / NXCD;
  const unsigned g0 = xcd * chunk, g1 = g0 + chunk < ngrp ? g0 + chunk : ngrp;
  const unsigned *sp = g_sell2_ptr;
  double dot = 0.0;
  for (unsigned g = g0 + slot; g < g1; g += gx) {
    const unsigned s = __builtin_amdgcn_readfirstlane(g * 4 + wave);
    if (s < nslice) {
      const unsigned base = sp[s], len = (sp[s + 1] - base) >> 7;
      const lab_s2v *cp = (const lab_s2v *)((const short *)cols + base) + lane;
      const lab_d2vb *vp = (const lab_d2vb *)(vals + base) + lane;
      const int row = (int)(s * 128 + 2 * lane);
      double a0 = 0.0, a1 = 0.0;
      for (unsigned j0 = 0; j0 < len; j0 += U) {
        lab_s2v c[U];
        lab_d2vb v[U];
#pragma unroll
        for (int u = 0; u < U; u++)
          if (j0 + u < len) {
            c[u] = ldg<FLAGS>(cp + (j0 + u) * 64);
            v[u] = ldg<FLAGS>(vp + (j0 + u) * 64);
          }
#pragma unroll
        for (int u = 0; u < U; u++)
          if (j0 + u < len) {
            const int i0 = row + (int)c[u].x, i1 = row + 1 + (int)c[u].y;
            a0 += v[u].x * __fma_rn(beta, x[i0], cc * rr[i0]);
            a1 += v[u].y * __fma_rn(beta, x[i1], cc * rr[i1]);
          }
      }
      if ((unsigned)row + 1 < n) {
        lab_d2vb o = {a0, a1};
        *(lab_d2vb *)(y + row) = o;
        const lab_d2vb xx = *(const lab_d2vb *)(x + row), r2 = *(const lab_d2vb *)(rr + row);
        lab_d2vb pv = {__fma_rn(beta, xx.x, cc * r2.x), __fma_rn(beta, xx.y, cc * r2.y)};
        *(lab_d2vb *)(pn + row) = pv;
        dot += a0 * pv.x;
        dot += a1 * pv.y;
      } else if ((unsigned)row < n) {
        y[row] = a0;
        const double pv = __fma_rn(beta, x[row], cc * rr[row]);
        pn[row] = pv;
        dot += a0 * pv;
      }
    }
  }
  double d[1] = {dot};
  wg_sum<1>(d, sred);
  if (tid == 0)
    partials[xcd * gx + slot] = d[0];
}

// ---- V5: V3 with 16 B/lane stream loads.  A block's nnz range [j0,j1) is
// widened to 4-aligned [j0&~3, ...); each lane owns QPT quads of 4 consecutive
// non-zeros (cols as int4, vals as 2 x double2).  Row blocks are built with
// cap CAP-8 so the widened range always fits CAP.  cols/vals must be readable
// up to the next multiple of 4 past nnz (the lab over-allocates).
template <int CAP, int FLAGS>
__global__ __launch_bounds__(WG, 8) void k_cyc_v4(const int *__restrict__ rowblk, unsigned nblk,
                                                  unsigned per, const int *__restrict__ offs,
                                                  const int *__restrict__ cols,
                                                  const double *__restrict__ vals,
                                                  const double *__restrict__ x,
                                                  double *__restrict__ y,
                                                  double *__restrict__ partials) {
  __shared__ double sprod[CAP];
  __shared__ double sred[4];
  const unsigned tid = threadIdx.x;
  const unsigned gx = gridDim.x / NXCD;
  const unsigned xcd = blockIdx.x % NXCD, slot = blockIdx.x / NXCD;
  const unsigned chunk = (nblk + NXCD - 1) / NXCD;
  const unsigned kbeg = xcd * chunk, kend = min(kbeg + chunk, nblk);
  constexpr int QPT = CAP / 4 / WG; // quads per thread
  double dot = 0.0;
  typedef int i4v __attribute__((ext_vector_type(4)));
  typedef double d2v __attribute__((ext_vector_type(2)));
  i4v c4[QPT];
  d2v va[QPT], vb[QPT];
  unsigned k = kbeg + slot;
  int r0 = 0, r1 = 0, j0 = 0, j1 = 0;
#define ISSUE_V4(kk)                                                           \
  do {                                                                         \
    r0 = rowblk[kk], r1 = rowblk[(kk) + 1];                                    \
    j0 = offs[r0], j1 = offs[r1];                                              \
    if (j1 - j0 <= CAP - 8) {                                                  \
      const int ja_ = j0 & ~3, nq_ = (j1 - ja_ + 3) >> 2;                      \
      _Pragma("unroll") for (int u = 0; u < QPT; u++) {                        \
        const int q = (int)tid + u * WG;                                       \
        if (q < nq_) {                                                         \
          c4[u] = ldg<FLAGS>((const i4v *)(cols + ja_) + q);                   \
          va[u] = ldg<FLAGS>((const d2v *)(vals + ja_) + 2 * q);               \
          vb[u] = ldg<FLAGS>((const d2v *)(vals + ja_) + 2 * q + 1);           \
        }                                                                      \
      }                                                                        \
    }                                                                          \
  } while (0)
  if (k < kend)
    ISSUE_V4(k);
  for (; k < kend; k += gx) {
    const int cr0 = r0, cj0 = j0, cnt = j1 - j0, nr = r1 - r0;
    if (cnt <= CAP - 8) {
      const int sh = cj0 & 3, nq = (cnt + sh + 3) >> 2;
#pragma unroll
      for (int u = 0; u < QPT; u++) {
        const int q = (int)tid + u * WG;
        if (q < nq) {
          const int e = 4 * q - sh;
          const int cc[4] = {c4[u].x, c4[u].y, c4[u].z, c4[u].w};
          const double vv[4] = {va[u].x, va[u].y, vb[u].x, vb[u].y};
#pragma unroll
          for (int i = 0; i < 4; i++)
            if (e + i >= 0 && e + i < cnt)
              sprod[e + i] = vv[i] * ((FLAGS & F_NOX) ? (double)cc[i] : x[cc[i]]);
        }
      }
      if ((FLAGS & F_PREFETCH) && k + gx < kend)
        ISSUE_V4(k + gx);
      __syncthreads();
      unsigned L = 1;
      while (L < 64 && (unsigned)nr * (L * 2) <= WG)
        L <<= 1;
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
          y[cr0 + r] = s;
          if (!(FLAGS & F_NODOT))
            dot += s * x[cr0 + r];
        }
      }
      __syncthreads();
    } else {
      double s[1] = {0.0};
      for (int r = cr0; r < cr0 + nr; r++) {
        s[0] = 0.0;
        for (int j = offs[r] + (int)tid; j < offs[r + 1]; j += WG)
          s[0] += vals[j] * x[cols[j]];
        wg_sum<1>(s, sred);
        if (tid == 0) {
          y[r] = s[0];
          dot += s[0] * x[r];
        }
      }
      if ((FLAGS & F_PREFETCH) && k + gx < kend)
        ISSUE_V4(k + gx);
    }
    if (!(FLAGS & F_PREFETCH) && k + gx < kend)
      ISSUE_V4(k + gx);
  }
  if (!(FLAGS & F_NODOT)) {
    double d[1] = {dot};
    wg_sum<1>(d, sred);
    if (tid == 0)
      partials[xcd * gx + slot] = d[0];
  }
}

// read-only ceiling: 16 B/lane loads, 4 independent streams per lane
typedef double lab_d2v __attribute__((ext_vector_type(2)));
template <int NT>
__global__ __launch_bounds__(WG) void k_read_ceiling(const lab_d2v *__restrict__ p, size_t n2,
                                                     double *__restrict__ out) {
  double acc = 0.0;
  const size_t g = (size_t)gridDim.x * WG;
  size_t i = (size_t)blockIdx.x * WG + threadIdx.x;
  for (; i + 3 * g < n2; i += 4 * g) {
    lab_d2v a, b, c, d;
    if (NT) {
      a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + g);
      c = __builtin_nontemporal_load(p + i + 2 * g), d = __builtin_nontemporal_load(p + i + 3 * g);
    } else {
      a = p[i], b = p[i + g], c = p[i + 2 * g], d = p[i + 3 * g];
    }
    acc += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
  }
  if (acc == 123.456)
    out[0] = acc;
}

// ---- V4: wave-private pipeline.  Each of the 4 wavefronts of a workgroup owns
// CAPW LDS doubles and walks its own (cyclically dealt) row blocks of <= CAPW
// non-zeros: no __syncthreads in the main loop, the waves never wait for each
// other.  DS operations of one wavefront execute in order, so a wave-scope
// fence (compiler ordering only) is all that separates the product writes from
// the row sums.
template <int CAPW, int FLAGS, int MINW = 8>
__global__ __launch_bounds__(WG, MINW) void k_wave(const int *__restrict__ rowblk, unsigned nblk,
                                                unsigned per, const int *__restrict__ offs,
                                                const int *__restrict__ cols,
                                                const double *__restrict__ vals,
                                                const double *__restrict__ x,
                                                double *__restrict__ y,
                                                double *__restrict__ partials) {
  __shared__ double sprod_all[4 * CAPW];
  __shared__ double sred[4];
  const unsigned tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double *sprod = sprod_all + wave * CAPW;
  const unsigned gx = gridDim.x / NXCD;
  const unsigned xcd = blockIdx.x % NXCD, slot = blockIdx.x / NXCD;
  const unsigned chunk = (nblk + NXCD - 1) / NXCD;
  const unsigned kbeg = xcd * chunk, kend = min(kbeg + chunk, nblk);
  const unsigned stride = gx * 4;
  constexpr int U = CAPW / 64;
  double dot = 0.0;
  int c[U];
  double v[U];
  int r0 = 0, r1 = 0, j0 = 0, j1 = 0;
#define ISSUE_WAVE(kk)                                                         \
  do {                                                                         \
    r0 = __builtin_amdgcn_readfirstlane(rowblk[kk]);                           \
    r1 = __builtin_amdgcn_readfirstlane(rowblk[(kk) + 1]);                     \
    j0 = __builtin_amdgcn_readfirstlane(offs[r0]);                             \
    j1 = __builtin_amdgcn_readfirstlane(offs[r1]);                             \
    if (j1 - j0 <= CAPW) {                                                     \
      _Pragma("unroll") for (int u = 0; u < U; u++) {                          \
        const int t = lane + u * 64;                                           \
        if (t < j1 - j0) {                                                     \
          c[u] = ldg<FLAGS>(cols + j0 + t);                                    \
          v[u] = ldg<FLAGS>(vals + j0 + t);                                    \
        }                                                                      \
      }                                                                        \
    }                                                                          \
  } while (0)
  unsigned k = kbeg + slot * 4 + wave;
  if (k < kend)
    ISSUE_WAVE(k);
  for (; k < kend; k += stride) {
    const int cr0 = r0, cj0 = j0, cnt = j1 - j0, nr = r1 - r0;
    if (cnt <= CAPW) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int t = lane + u * 64;
        if (t < cnt)
          sprod[t] = v[u] * x[c[u]];
      }
      if ((FLAGS & F_PREFETCH) && k + stride < kend)
        ISSUE_WAVE(k + stride);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      unsigned L = 1;
      while (L < 64 && (unsigned)nr * (L * 2) <= 64)
        L <<= 1;
      const unsigned sl = lane / L, l = lane % L, slots = 64 / L;
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
          y[cr0 + r] = s;
          if (!(FLAGS & F_NODOT))
            dot += s * x[cr0 + r];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    } else {
      // long row: this wavefront alone strides over it
      for (int r = cr0; r < cr0 + nr; r++) {
        double s = 0.0;
        for (int j = offs[r] + (int)lane; j < offs[r + 1]; j += 64)
          s += vals[j] * x[cols[j]];
        for (int off = 32; off > 0; off >>= 1)
          s += __shfl_xor(s, off, 64);
        if (lane == 0) {
          y[r] = s;
          dot += s * x[r];
        }
      }
      if ((FLAGS & F_PREFETCH) && k + stride < kend)
        ISSUE_WAVE(k + stride);
    }
    if (!(FLAGS & F_PREFETCH) && k + stride < kend)
      ISSUE_WAVE(k + stride);
  }
  if (!(FLAGS & F_NODOT)) {
    double d[1] = {dot};
    wg_sum<1>(d, sred);
    if (tid == 0)
      partials[xcd * gx + slot] = d[0];
  }
}

// ---- ceiling probes ---------------------------------------------------------
// stream: read cols+vals (16 B/lane), write one double per 5 nnz-ish (n of them)
__global__ __launch_bounds__(WG) void k_stream_ceiling(const int4 *__restrict__ cols4,
                                                       const double2 *__restrict__ vals2,
                                                       size_t nq, double *__restrict__ y,
                                                       size_t n) {
  double acc = 0.0;
  const size_t g = (size_t)gridDim.x * WG;
  for (size_t q = (size_t)blockIdx.x * WG + threadIdx.x; q < nq; q += g) {
    const int4 c = cols4[q];
    const double2 a = vals2[2 * q], b = vals2[2 * q + 1];
    acc += a.x * c.x + a.y * c.y + b.x * c.z + b.y * c.w;
  }
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n; i += g)
    y[i] = acc;
}

// ---------------------------------------------------------------------------
struct Csr {
  int n;
  std::vector<int> offs, cols;
  std::vector<double> vals;
};

static Csr lap2d(int nx, int ny) {
  Csr A;
  A.n = nx * ny;
  A.offs.resize(A.n + 1);
  A.cols.reserve((size_t)5 * A.n);
  A.vals.reserve((size_t)5 * A.n);
  for (int j = 0; j < ny; j++)
    for (int i = 0; i < nx; i++) {
      const int r = j * nx + i;
      A.offs[r] = (int)A.cols.size();
      if (j > 0) A.cols.push_back(r - nx), A.vals.push_back(-1);
      if (i > 0) A.cols.push_back(r - 1), A.vals.push_back(-1);
      A.cols.push_back(r), A.vals.push_back(4);
      if (i + 1 < nx) A.cols.push_back(r + 1), A.vals.push_back(-1);
      if (j + 1 < ny) A.cols.push_back(r + nx), A.vals.push_back(-1);
    }
  A.offs[A.n] = (int)A.cols.size();
  return A;
}

static Csr lap3d(int nx, int ny, int nz) {
  Csr A;
  A.n = nx * ny * nz;
  A.offs.resize((size_t)A.n + 1);
  A.cols.reserve((size_t)7 * A.n);
  A.vals.reserve((size_t)7 * A.n);
  const int nxy = nx * ny;
  for (int k = 0; k < nz; k++)
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++) {
        const int r = (k * ny + j) * nx + i;
        A.offs[r] = (int)A.cols.size();
        if (k > 0) A.cols.push_back(r - nxy), A.vals.push_back(-1);
        if (j > 0) A.cols.push_back(r - nx), A.vals.push_back(-1);
        if (i > 0) A.cols.push_back(r - 1), A.vals.push_back(-1);
        A.cols.push_back(r), A.vals.push_back(6);
        if (i + 1 < nx) A.cols.push_back(r + 1), A.vals.push_back(-1);
        if (j + 1 < ny) A.cols.push_back(r + nx), A.vals.push_back(-1);
        if (k + 1 < nz) A.cols.push_back(r + nxy), A.vals.push_back(-1);
      }
  A.offs[A.n] = (int)A.cols.size();
  return A;
}

static uint64_t mix(uint64_t s, uint64_t a, uint64_t b) {
  uint64_t z = s + 0x9E3779B97F4A7C15ull * (a + 1) + 0xC2B2AE3D27D4EB4Full * (b + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static Csr powerlaw(int n, double gamma, int dmax) {
  std::vector<double> cdf(dmax);
  double tot = 0, acc = 0;
  for (int d = 1; d <= dmax; d++) tot += pow(d, -gamma);
  for (int d = 1; d <= dmax; d++) acc += pow(d, -gamma) / tot, cdf[d - 1] = acc;
  Csr A;
  A.n = n;
  A.offs.resize((size_t)n + 1);
  size_t z = 0;
  for (int r = 0; r < n; r++) {
    A.offs[r] = (int)z;
    const double u = (double)(mix(7, r, 0) >> 11) / 9007199254740992.0;
    int d = (int)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin()) + 1;
    d = std::min(d, dmax);
    z += d;
  }
  A.offs[n] = (int)z;
  A.cols.resize(z);
  A.vals.resize(z);
#pragma omp parallel for schedule(dynamic, 4096)
  for (int r = 0; r < n; r++) {
    const int d = A.offs[r + 1] - A.offs[r];
    for (int k = 0; k < d; k++) {
      const uint64_t uk = mix(7, r, 2 * k + 1) % (uint64_t)(n - d + 1);
      A.cols[A.offs[r] + k] = (int)(((unsigned __int128)k * n + uk) / d);
      A.vals[A.offs[r] + k] = (double)(mix(7, r, 2 * k + 2) >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    }
  }
  return A;
}

static std::vector<int> row_blocks(const Csr &A, int cap) {
  std::vector<int> rb{0};
  int r = 0;
  int rowcap = 0x7fffffff;
  if (cap < 0) rowcap = -cap, cap = 2048;
  while (r < A.n) {
    const long lim = (long)A.offs[r] + cap;
    int e = r + 1;
    if (A.offs[e] <= lim)
      e = (int)(std::upper_bound(A.offs.begin() + r + 1, A.offs.end(), (int)std::min<long>(lim, 0x7fffffff)) -
                A.offs.begin()) - 1;
    if (e - r > rowcap) e = r + rowcap;
    rb.push_back(e);
    r = e;
  }
  return rb;
}

template <class T>
static T *upload(const std::vector<T> &h) {
  T *d;
  CHK(hipMalloc(&d, h.size() * sizeof(T) + 64));
  CHK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

struct Variant {
  std::string name;
  int cap;
  void (*launch)(const Variant &, unsigned g, unsigned per, unsigned nblk, const int *rb,
                 const int *offs, const int *cols, const double *vals, const double *x, double *y,
                 double *parts, int nnz);
  unsigned maxgrid;
  std::vector<float> us;
  const int *d_rb = nullptr;
  unsigned nblk = 0;
  bool check = true;
};

#define LAUNCHER(fn, KERNEL, ...)                                                             \
  static void fn(const Variant &v, unsigned g, unsigned per, unsigned nblk, const int *rb,    \
                 const int *offs, const int *cols, const double *vals, const double *x,       \
                 double *y, double *parts, int nnz) {                                         \
    KERNEL<<<g, WG>>>(rb, nblk, per, offs, cols, vals, x, y, parts __VA_ARGS__);              \
  }

LAUNCHER(l_base2048, (k_adaptive<2048, 0>))
LAUNCHER(l_base1024, (k_adaptive<1024, 0>))
LAUNCHER(l_base4096, (k_adaptive<4096, 0>))
LAUNCHER(l_noremap, (k_adaptive<2048, F_NOREMAP>))
LAUNCHER(l_fakeg, (k_adaptive<2048, F_FAKEGATHER>))
LAUNCHER(l_nox, (k_adaptive<2048, F_NOX>))
LAUNCHER(l_nodot, (k_adaptive<2048, F_NODOT>))
LAUNCHER(l_pipe2048, (k_adaptive_pipe<2048, 0>))
LAUNCHER(l_pipe1024, (k_adaptive_pipe<1024, 0>))
LAUNCHER(l_cyc, (k_adaptive_cyc<2048, F_CYCLIC>))
LAUNCHER(l_cyc_pf, (k_adaptive_cyc<2048, F_CYCLIC | F_PREFETCH>))
LAUNCHER(l_cyc_pf7, (k_adaptive_cyc<2048, F_CYCLIC | F_PREFETCH, 7>))
LAUNCHER(l_cyc_pf6, (k_adaptive_cyc<2048, F_CYCLIC | F_PREFETCH, 6>))
LAUNCHER(l_wave512_pf7, (k_wave<512, F_PREFETCH, 7>))
LAUNCHER(l_wave512_pf6, (k_wave<512, F_PREFETCH, 6>))
LAUNCHER(l_cyc_pf_nt, (k_adaptive_cyc<2048, F_CYCLIC | F_PREFETCH | F_NT>))
LAUNCHER(l_cyc_nt, (k_adaptive_cyc<2048, F_CYCLIC | F_NT>))
LAUNCHER(l_sell_nt8, (k_sell<F_NT, 8>))
LAUNCHER(l_sell2_nt8, (k_sell2<F_NT, 8, 4>))
LAUNCHER(l_s2c16_u5, (k_sell2c16<F_NT, 5, 8>))
LAUNCHER(l_s2c16p_u5, (k_sell2c16p<F_NT, 5, 8>))
LAUNCHER(l_s2c16p_u5o6, (k_sell2c16p<F_NT, 5, 6>))
LAUNCHER(l_s2c16_u8, (k_sell2c16<F_NT, 8, 6>))
LAUNCHER(l_sell2_nt5, (k_sell2<F_NT, 5, 6>))
LAUNCHER(l_sell2_nt7, (k_sell2<F_NT, 7, 5>))
LAUNCHER(l_sell2_nt4, (k_sell2<F_NT, 4, 8>))
LAUNCHER(l_sell_8, (k_sell<0, 8>))
LAUNCHER(l_sell_nt4, (k_sell<F_NT, 4>))
LAUNCHER(l_sell_nt8_nodot, (k_sell<F_NT | F_NODOT, 8>))
LAUNCHER(l_c16, (k_adaptive_cyc<2048, F_CYCLIC | F_NT | F_PREFETCH | F_C16>))
LAUNCHER(l_c16_nopf, (k_adaptive_cyc<2048, F_CYCLIC | F_NT | F_C16>))
LAUNCHER(l_c16w, (k_adaptive_cyc<2048, F_CYCLIC | F_NT | F_PREFETCH | F_C16W>))
LAUNCHER(l_c16w_nopf, (k_adaptive_cyc<2048, F_CYCLIC | F_NT | F_C16W>))
LAUNCHER(l_cyc_period, (k_adaptive_cyc<2048, F_CYCLIC | F_NT | F_PREFETCH | F_PERIOD>))
LAUNCHER(l_cyc_period_nont, (k_adaptive_cyc<2048, F_CYCLIC | F_PREFETCH | F_PERIOD>))
LAUNCHER(l_cyc_two, (k_adaptive_cyc<2048, F_CYCLIC | F_NT | F_PREFETCH | F_TWOROW>))
LAUNCHER(l_cyc_two_nont, (k_adaptive_cyc<2048, F_CYCLIC | F_PREFETCH | F_TWOROW>))
LAUNCHER(l_cyc_hoist, (k_adaptive_cyc<2048, F_CYCLIC | F_NT | F_PREFETCH | F_HOIST>))
LAUNCHER(l_cyc_hoist_nopf, (k_adaptive_cyc<2048, F_CYCLIC | F_NT | F_HOIST>))
LAUNCHER(l_cyc_hoist7, (k_adaptive_cyc<2048, F_CYCLIC | F_NT | F_PREFETCH | F_HOIST, 7>))
LAUNCHER(l_cyc_hoist_nonT, (k_adaptive_cyc<2048, F_CYCLIC | F_PREFETCH | F_HOIST>))
LAUNCHER(l_cyc_fake, (k_adaptive_cyc<2048, F_CYCLIC | F_NT | F_PREFETCH | F_FAKEGATHER>))
LAUNCHER(l_cyc_nox, (k_adaptive_cyc<2048, F_CYCLIC | F_NT | F_PREFETCH | F_NOX>))
LAUNCHER(l_cyc_nox_nodot, (k_adaptive_cyc<2048, F_CYCLIC | F_NT | F_PREFETCH | F_NOX | F_NODOT>))
LAUNCHER(l_v4_pf_nt, (k_cyc_v4<2048, F_NT | F_PREFETCH>))
LAUNCHER(l_v4_nt, (k_cyc_v4<2048, F_NT>))
LAUNCHER(l_v4_pf, (k_cyc_v4<2048, F_PREFETCH>))
LAUNCHER(l_v4_4096_pf_nt, (k_cyc_v4<4096, F_NT | F_PREFETCH>))
LAUNCHER(l_v4_nox, (k_cyc_v4<2048, F_NT | F_PREFETCH | F_NOX>))
LAUNCHER(l_cyc1024_pf, (k_adaptive_cyc<1024, F_CYCLIC | F_PREFETCH>))
LAUNCHER(l_wave512, (k_wave<512, 0>))
LAUNCHER(l_wave512_pf, (k_wave<512, F_PREFETCH>))
LAUNCHER(l_wave512_pf_nt, (k_wave<512, F_PREFETCH | F_NT>))
LAUNCHER(l_wave256_pf, (k_wave<256, F_PREFETCH>))
LAUNCHER(l_wave1024_pf, (k_wave<1024, F_PREFETCH>))
LAUNCHER(l_v4_2048, (k_adaptive_v4<2048, 0>), , nnz)
LAUNCHER(l_v4_1024, (k_adaptive_v4<1024, 0>), , nnz)
LAUNCHER(l_v4_4096, (k_adaptive_v4<4096, 0>), , nnz)

int main(int argc, char **argv) {
  const std::string kind = argc > 1 ? argv[1] : "lap2d";
  const int size = argc > 2 ? atoi(argv[2]) : 3162;
  const int rounds = argc > 3 ? atoi(argv[3]) : 15;
  Csr A = kind == "lap3d" ? lap3d(size, size, size)
                          : (kind == "powerlaw" ? powerlaw(size, 1.585350372615855, 4096) : lap2d(size, size));
  const size_t nnz = A.cols.size();
  const double bytes = 12.0 * nnz + 20.0 * A.n + 4;
  printf("%s %d: n=%d nnz=%zu algorithmic bytes=%.0f\n", kind.c_str(), size, A.n, nnz, bytes);
  std::vector<double> hx(A.n), href(A.n);
  for (int i = 0; i < A.n; i++)
    hx[i] = (double)(mix(1, i, 0) >> 11) / 9007199254740992.0 - 0.5;
#pragma omp parallel for
  for (int i = 0; i < A.n; i++) {
    double s = 0;
    for (int j = A.offs[i]; j < A.offs[i + 1]; j++) s += A.vals[j] * hx[A.cols[j]];
    href[i] = s;
  }
  int *d_offs = upload(A.offs), *d_cols = upload(A.cols);
  double *d_vals = upload(A.vals), *d_x = upload(hx), *d_y, *d_parts;
  CHK(hipMalloc(&d_y, (size_t)A.n * 8));
  CHK(hipMalloc(&d_parts, 4096 * 8));
  {
    double *d_r = upload(hx), *d_pn;
    CHK(hipMalloc(&d_pn, (size_t)A.n * 8 + 64));
    CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_fuse_r), &d_r, sizeof(d_r)));
    CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_fuse_pnew), &d_pn, sizeof(d_pn)));
  }

  std::vector<Variant> vs = {
      {"adaptive cap2048 g2048 (round-1a lib)", 2048, l_base2048, 2048},
      {"cyc+prefetch+nt cap2048 (library)", 2048, l_cyc_pf_nt, 2048},
      {"cyc+nt cap2048", 2048, l_cyc_nt, 2048},
      {"SELL64 nt U8", 2048, l_sell_nt8, 2048},
      {"S2C16 nt U5 g2048", 2048, l_s2c16_u5, 2048},
      {"S2C16 nt U5 g1536", 2048, l_s2c16_u5, 1536},
      {"probe: S2C16 + fused p update g1536", 2048, l_s2c16p_u5, 1536},
      {"probe: S2C16 + fused p update g2048", 2048, l_s2c16p_u5, 2048},
      {"probe: S2C16 + fused p update occ6", 2048, l_s2c16p_u5o6, 1536},
      {"S2C16 nt U8 g1536", 2048, l_s2c16_u8, 1536},
      {"SELL128x2 nt U8 occ4", 2048, l_sell2_nt8, 1024},
      {"SELL128x2 nt U7 occ5", 2048, l_sell2_nt7, 1280},
      {"SELL128x2 nt U5 occ6", 2048, l_sell2_nt5, 1536},
      {"SELL128x2 nt U4 occ8", 2048, l_sell2_nt4, 2048},
      {"SELL128x2 nt U8 occ4 g2048", 2048, l_sell2_nt8, 2048},
      {"SELL64 nt U4", 2048, l_sell_nt4, 2048},

      {"probe: SELL64 nt U8 no dot", 2048, l_sell_nt8_nodot, 2048},
      {"C16 cyc+pf+nt (int16 row-relative cols)", 2048, l_c16, 2048},
      {"C16 cyc+nt", 2048, l_c16_nopf, 2048},
      {"C16W cyc+pf+nt (4 windows x 14 bit)", 2048, l_c16w, 2048},
      {"C16W cyc+nt", 2048, l_c16w_nopf, 2048},
      {"cyc+prefetch cap2048", 2048, l_cyc_pf, 2048},
      {"cyc+pf+nt PERIOD=bandwidth", 2048, l_cyc_period, 2048},
      {"cyc+pf PERIOD=bandwidth (no nt)", 2048, l_cyc_period_nont, 2048},
      {"probe: cyc+pf+nt, gather from 8KB", 2048, l_cyc_fake, 2048},
      {"probe: cyc+pf+nt, no gather", 2048, l_cyc_nox, 2048},
      {"probe: v4 cyc+pf+nt, no gather", 2040, l_v4_nox, 2048},
      {"probe: cyc+pf+nt, no gather, no dot", 2048, l_cyc_nox_nodot, 2048},
  };
  for (auto &v : vs)
    if (v.name.rfind("probe", 0) == 0) v.check = false;
  // row blocks per distinct cap
  std::vector<std::pair<int, std::pair<int *, unsigned>>> rbs;
  for (auto &v : vs) {
    bool found = false;
    for (auto &e : rbs)
      if (e.first == v.cap) v.d_rb = e.second.first, v.nblk = e.second.second, found = true;
    if (!found) {
      auto rb = row_blocks(A, v.cap);
      int *d = upload(rb);
      rbs.push_back({v.cap, {d, (unsigned)rb.size() - 1}});
      v.d_rb = d, v.nblk = (unsigned)rb.size() - 1;
    }
  }
  // SELL-64
  int *d_scols = nullptr;
  double *d_svals = nullptr;
  unsigned nslice = (unsigned)((A.n + 63) / 64);
  {
    std::vector<unsigned> sptr(nslice + 1, 0);
    for (unsigned sl = 0; sl < nslice; sl++) {
      int len = 0;
      for (int r = sl * 64; r < std::min<long>(A.n, (long)sl * 64 + 64); r++)
        len = std::max(len, A.offs[r + 1] - A.offs[r]);
      sptr[sl + 1] = sptr[sl] + 64u * (unsigned)len;
    }
    const size_t tot = sptr[nslice];
    std::vector<int> sc(tot + 64, 0);
    std::vector<double> sv(tot + 64, 0.0);
#pragma omp parallel for
    for (unsigned sl = 0; sl < nslice; sl++) {
      const unsigned len = (sptr[sl + 1] - sptr[sl]) / 64;
      for (unsigned l = 0; l < 64; l++) {
        const long r = (long)sl * 64 + l;
        const int a = r < A.n ? A.offs[r] : 0, b = r < A.n ? A.offs[r + 1] : 0;
        const int padc = b > a ? A.cols[b - 1] : 0;
        for (unsigned j = 0; j < len; j++) {
          const size_t at = (size_t)sptr[sl] + (size_t)j * 64 + l;
          if (a + (int)j < b) sc[at] = A.cols[a + j], sv[at] = A.vals[a + j];
          else sc[at] = padc, sv[at] = 0.0;
        }
      }
    }
    printf("SELL-64: %u slices, %zu stored entries for %zu non-zeros (padding %.3f%%)\n", nslice, tot, nnz,
           100.0 * ((double)tot - (double)nnz) / (double)nnz);
    d_scols = upload(sc), d_svals = upload(sv);
    unsigned *d_sp = upload(sptr);
    CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_sell_ptr), &d_sp, sizeof(d_sp)));
  }
  // SELL-128x2
  int *d_s2cols = nullptr;
  short *d_s2c16 = nullptr;
  double *d_s2vals = nullptr;
  unsigned nslice2 = (unsigned)((A.n + 127) / 128);
  {
    std::vector<unsigned> sptr(nslice2 + 1, 0);
    for (unsigned sl = 0; sl < nslice2; sl++) {
      int len = 0;
      for (long r = (long)sl * 128; r < std::min<long>(A.n, (long)sl * 128 + 128); r++)
        len = std::max(len, A.offs[r + 1] - A.offs[r]);
      sptr[sl + 1] = sptr[sl] + 128u * (unsigned)len;
    }
    const size_t tot = sptr[nslice2];
    std::vector<int> sc(tot + 128, 0);
    std::vector<double> sv(tot + 128, 0.0);
#pragma omp parallel for
    for (unsigned sl = 0; sl < nslice2; sl++) {
      const unsigned len = (sptr[sl + 1] - sptr[sl]) / 128;
      for (unsigned l = 0; l < 128; l++) {
        const long r = (long)sl * 128 + l;
        const int a = r < A.n ? A.offs[r] : 0, b = r < A.n ? A.offs[r + 1] : 0;
        const int padc = b > a ? A.cols[b - 1] : 0;
        for (unsigned j = 0; j < len; j++) {
          const size_t at = (size_t)sptr[sl] + (size_t)j * 128 + l; // rows 2l,2l+1 adjacent
          if (a + (int)j < b) sc[at] = A.cols[a + j], sv[at] = A.vals[a + j];
          else sc[at] = padc, sv[at] = 0.0;
        }
      }
    }
    {
      std::vector<short> s16(tot + 128, 0);
      bool fits = true;
      for (unsigned sl = 0; sl < nslice2; sl++) {
        const unsigned len = (sptr[sl + 1] - sptr[sl]) / 128;
        for (unsigned j = 0; j < len; j++)
          for (unsigned l = 0; l < 128; l++) {
            const size_t at = (size_t)sptr[sl] + (size_t)j * 128 + l;
            const long r_ = (long)sl * 128 + l;
            long d = (long)sc[at] - r_;
            if (sv[at] == 0.0 && (d < -32768 || d > 32767)) d = (r_ < A.n ? 0 : (long)A.n - 1 - r_);
            if (d < -32768 || d > 32767) fits = false;
            s16[at] = (short)d;
          }
      }
      printf("SELL row-relative int16 column codes: %s\n", fits ? "fit" : "DO NOT FIT");
      if (fits) d_s2c16 = upload(s16);
    }
    d_s2cols = upload(sc), d_s2vals = upload(sv);
    unsigned *d_sp = upload(sptr);
    CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_sell2_ptr), &d_sp, sizeof(d_sp)));
  }
  // 16-bit column encodings over the cap-2048 row blocks
  short *d_c16 = nullptr;
  unsigned short *d_c16w = nullptr;
  {
    auto rb = row_blocks(A, 2048);
    const unsigned nb = (unsigned)rb.size() - 1;
    std::vector<short> c16(nnz + 8);
    std::vector<unsigned short> c16w(nnz + 8);
    std::vector<int> wbase(4 * (size_t)nb, 0);
    bool ok16 = true, okw = true;
    for (unsigned b = 0; b < nb; b++) {
      const int r0 = rb[b], j0 = A.offs[rb[b]], j1 = A.offs[rb[b + 1]];
      std::vector<int> cs(A.cols.begin() + j0, A.cols.begin() + j1);
      std::sort(cs.begin(), cs.end());
      int nw = 0, base[4] = {0, 0, 0, 0};
      for (int c : cs)
        if (nw == 0 || c - base[nw - 1] >= 16384) {
          if (nw == 4) { okw = false; break; }
          base[nw++] = c;
        }
      for (int w = 0; w < 4; w++) wbase[4 * (size_t)b + w] = base[w < nw ? w : 0];
      for (int j = j0; j < j1; j++) {
        const int d = A.cols[j] - r0;
        if (d < -32768 || d > 32767) ok16 = false;
        c16[j] = (short)d;
        int w = nw - 1;
        while (w > 0 && A.cols[j] < base[w]) w--;
        c16w[j] = (unsigned short)((w << 14) | (A.cols[j] - base[w]));
      }
    }
    printf("16-bit column codes: row-relative %s, 4-window %s\n", ok16 ? "fits" : "DOES NOT FIT", okw ? "fits" : "DOES NOT FIT");
    if (ok16) d_c16 = upload(c16);
    if (okw) {
      d_c16w = upload(c16w);
      int *d_wb = upload(wbase);
      CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_wbase), &d_wb, sizeof(d_wb)));
    }
  }
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  std::vector<double> hy(A.n);
  long bw = 0;
  for (int i = 0; i < A.n; i += 97)
    for (int j = A.offs[i]; j < A.offs[i + 1]; j++) bw = std::max<long>(bw, labs((long)A.cols[j] - i));
  printf("bandwidth (sampled) = %ld rows\n", bw);
  auto run = [&](Variant &v) {
    unsigned g = std::min(v.maxgrid, ((v.nblk + 7) / 8) * 8);
    unsigned per = (v.nblk + g - 1) / g;
    if (v.name.find("PERIOD") != std::string::npos) {
      const double rows_per_blk = (double)A.n / v.nblk;
      per = (unsigned)std::max(8.0, std::round(bw / rows_per_blk));
    }
    const int *cp = d_cols;
    if (v.name.find("S2C16") != std::string::npos) {
      if (!d_s2c16) return;
      unsigned gs = std::min(v.maxgrid, ((((nslice2 + 3) / 4) + 7) / 8) * 8);
      v.launch(v, gs, (unsigned)A.n, nslice2, v.d_rb, d_offs, (const int *)d_s2c16, d_s2vals, d_x, d_y, d_parts, (int)nnz);
      return;
    }
    if (v.name.find("SELL128x2") != std::string::npos) {
      unsigned gs = std::min(v.maxgrid, ((((nslice2 + 3) / 4) + 7) / 8) * 8);
      v.launch(v, gs, (unsigned)A.n, nslice2, v.d_rb, d_offs, d_s2cols, d_s2vals, d_x, d_y, d_parts, (int)nnz);
      return;
    }
    if (v.name.find("SELL64") != std::string::npos) {
      unsigned gs = std::min(v.maxgrid, ((((nslice + 3) / 4) + 7) / 8) * 8);
      v.launch(v, gs, (unsigned)A.n, nslice, v.d_rb, d_offs, d_scols, d_svals, d_x, d_y, d_parts, (int)nnz);
      return;
    }
    if (v.name.rfind("C16W", 0) == 0) cp = (const int *)d_c16w;
    else if (v.name.rfind("C16", 0) == 0) cp = (const int *)d_c16;
    if (v.name.rfind("C16", 0) == 0 && !cp) return;
    v.launch(v, g, per, v.nblk, v.d_rb, d_offs, cp, d_vals, d_x, d_y, d_parts, (int)nnz);
  };
  // correctness first
  for (auto &v : vs) {
    CHK(hipMemset(d_y, 0xff, (size_t)A.n * 8));
    run(v);
    CHK(hipDeviceSynchronize());
    if (!v.check) continue;
    if (v.name.rfind("C16W", 0) == 0 ? !d_c16w : (v.name.rfind("C16", 0) == 0 && !d_c16)) { v.check = false; continue; }
    if (v.name.find("S2C16") != std::string::npos && !d_s2c16) { v.check = false; continue; }
    CHK(hipMemcpy(hy.data(), d_y, (size_t)A.n * 8, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int i = 0; i < A.n; i++) {
      const double d = fabs(hy[i] - href[i]);
      if (!(d <= 1e-9)) worst = std::max(worst, std::isnan(d) ? 1e300 : d);
    }
    printf("check %-36s %s (worst %.2e)\n", v.name.c_str(), worst == 0 ? "ok" : "MISMATCH", worst);
  }
  // stream ceiling probe
  std::vector<float> ceil_us, read_us[2];
  for (int r = 0; r < rounds + 2; r++) {
    for (auto &v : vs) {
      CHK(hipEventRecord(e0));
      run(v);
      CHK(hipEventRecord(e1));
      CHK(hipEventSynchronize(e1));
      float ms;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) v.us.push_back(ms * 1e3f);
    }
    CHK(hipEventRecord(e0));
    k_stream_ceiling<<<2048, WG>>>((const int4 *)d_cols, (const double2 *)d_vals, nnz / 4, d_y, A.n);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 2) ceil_us.push_back(ms * 1e3f);
    for (int nt = 0; nt < 2; nt++) {
      CHK(hipEventRecord(e0));
      if (nt)
        k_read_ceiling<1><<<2048, WG>>>((const lab_d2v *)d_vals, nnz / 2, d_parts);
      else
        k_read_ceiling<0><<<2048, WG>>>((const lab_d2v *)d_vals, nnz / 2, d_parts);
      CHK(hipEventRecord(e1));
      CHK(hipEventSynchronize(e1));
      CHK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) read_us[nt].push_back(ms * 1e3f);
    }
  }
  auto med = [](std::vector<float> v) {
    std::sort(v.begin(), v.end());
    return std::make_pair(v[v.size() / 2], v[0]);
  };
  for (auto &v : vs) {
    auto m = med(v.us);
    printf("%-36s median %8.1f us  min %8.1f us  => %6.0f GB/s  (%.1f%% of 8 TB/s)\n", v.name.c_str(),
           m.first, m.second, bytes / m.first / 1e3, bytes / m.first / 1e3 / 80.0);
  }
  auto m = med(ceil_us);
  const double sb = 12.0 * nnz + 8.0 * A.n;
  printf("%-36s median %8.1f us  min %8.1f us  => %6.0f GB/s of its own %0.f bytes\n",
         "ceiling: stream cols+vals, write y", m.first, m.second, sb / m.first / 1e3, sb);
  {
    // BLAS-1 probe: 5 vectors of n doubles (carved out of d_vals, which is
    // 5n+ doubles long for the Laplacians), 2 written back
    const size_t n2 = (size_t)A.n / 2;
    if (nnz >= (size_t)5 * A.n - 8) {
      lab2_d2v *b = (lab2_d2v *)d_vals;
      for (int nt = 0; nt < 4; nt++) {
        std::vector<float> us;
        for (int r = 0; r < rounds + 2; r++) {
          CHK(hipEventRecord(e0));
          switch (nt) {
          case 0: k_blas1_probe<0><<<2048, 256>>>(n2, b, b + n2, b + 2 * n2, b + 3 * n2, b + 4 * n2 - 8, 1e-9, d_parts); break;
          case 1: k_blas1_probe<1><<<2048, 256>>>(n2, b, b + n2, b + 2 * n2, b + 3 * n2, b + 4 * n2 - 8, 1e-9, d_parts); break;
          case 2: k_blas1_probe<2><<<2048, 256>>>(n2, b, b + n2, b + 2 * n2, b + 3 * n2, b + 4 * n2 - 8, 1e-9, d_parts); break;
          default: k_blas1_probe<3><<<2048, 256>>>(n2, b, b + n2, b + 2 * n2, b + 3 * n2, b + 4 * n2 - 8, 1e-9, d_parts); break;
          }
          CHK(hipEventRecord(e1));
          CHK(hipEventSynchronize(e1));
          float ms;
          CHK(hipEventElapsedTime(&ms, e0, e1));
          if (r >= 2) us.push_back(ms * 1e3f);
        }
        auto mb = med(us);
        printf("blas1 probe (5 in, 2 out) nt-load=%d nt-store=%d  median %8.1f us  => %6.0f GB/s of %.0f bytes\n",
               nt & 1, nt >> 1, mb.first, 56.0 * A.n / mb.first / 1e3, 56.0 * A.n);
      }
    }
  }
  for (int nt = 0; nt < 2; nt++) {
    auto mr = med(read_us[nt]);
    printf("ceiling: read-only vals%s            median %8.1f us  => %6.0f GB/s of its own %.0f bytes\n",
           nt ? " (nt)" : "     ", mr.first, 8.0 * nnz / mr.first / 1e3, 8.0 * nnz);
  }
  return 0;
}
