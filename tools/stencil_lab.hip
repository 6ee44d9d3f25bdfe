// Lab for the constant-slot path of the 16-bit sliced-ELL SpMV:
// the product's layout (built by the product's own host code, linked from
// liblsbench_hip.so) under kernel variants that remove memory round trips:
//   prod   the shipped kernel, through lsb_k_spmv_sell
//   T5     slice TEMPLATES: slices whose slot records and constants are identical share one
//          record set (a handful on a constant-coefficient grid), a byte per slice says
//          which; the records then come out of the scalar cache instead of 120 B of cold
//          scalar loads per slice
//   T3     + slots whose base is a neighbour's base +-1 take their operands from that
//          neighbour's 16-byte pair by a lane shift (3 gathers instead of 5 on a 5-point row,
//          none of them misaligned), the dot's operand is the centre pair
//   T3x2   + two slices per wave in flight
//   hard   the same arithmetic with the stencil hard-coded (no metadata at all; wrong at grid
//          line ends): the floor of this access pattern
//   copy   y = 4 x + fused dot: the HBM floor of 80 MB in + 80 MB out
// build:  hipcc -O3 --offload-arch=gfx950 -fopenmp -Iinclude -Ilsbench_amd/csrc \
//           -o tools/stencil_lab.bin tools/stencil_lab.hip -Llsbench_amd/csrc -llsbench_hip \
//           -Wl,-rpath,'$ORIGIN/../lsbench_amd/csrc'
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "lsbench_hip.h"
#include "lsb_impl.h"

#define CHK(c)                                                                                 \
  do {                                                                                         \
    hipError_t e_ = (c);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                \
      exit(1);                                                                                 \
    }                                                                                          \
  } while (0)

#define WG 256
#define NXCD 8
#define ROWS 128
typedef int i4v __attribute__((ext_vector_type(4)));
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
typedef double d2v __attribute__((ext_vector_type(2)));

#define TMAX 8
struct tmpl { // a pure slice: nslots constant, code-free slots, bases ascending
  int nslots;
  int centre;     // slot c with neighbours derived from it by lane shifts, -1 = none (all gathered)
  int base[TMAX];
  int kind[TMAX]; // 0 gather; 1 = base[centre] - 1 (the centre pair shifted up); 2 = base[centre] + 1
  double cst[TMAX];
};

__device__ __forceinline__ void wg_sum1(double &v, double *sred) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    v += __shfl_xor(v, off, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0)
    sred[wave] = v;
  __syncthreads();
  v = (sred[0] + sred[1]) + (sred[2] + sred[3]);
}

// slices of turn `it` of this workgroup: XCD-contiguous chunks, groups of PER slices dealt
// cyclically (the product's dealing with PER = 4)
struct deal {
  unsigned base, turns;
};
template <int PER>
__device__ __forceinline__ deal deal_init(unsigned ns, unsigned xcd) {
  const unsigned ngrp = (ns + PER - 1) / PER, chunk = (ngrp + NXCD - 1) / NXCD;
  const unsigned g0 = min(xcd * chunk, ngrp), g1 = min(g0 + chunk, ngrp);
  deal d;
  d.base = g0 * PER, d.turns = g1 - g0;
  return d;
}

// generic slice (a slot keeps its values, e.g. where a grid line ends): the product's
// per-slot path without codes (structured grids have none)
__device__ __forceinline__ void slice_generic(unsigned s, const unsigned *__restrict__ sptr, const i4v *__restrict__ rec,
                                              const double *__restrict__ vconst,
                                              const double *__restrict__ vals,
                                              const double *__restrict__ x, int grow, unsigned lane,
                                              double &a0, double &a1) {
  const unsigned q0 = sptr[s] / ROWS, ulen = (sptr[s + 1] - sptr[s]) / ROWS;
  for (unsigned j = 0; j < ulen; j++) {
    const i4v r = rec[q0 + j];
    if (r.z < 0) {
      const double c = vconst[q0 + j];
      const d2u t = *(const d2u *)(x + (grow + r.x));
      a0 += c * t.x, a1 += c * t.y;
    } else {
      const d2v v = *((const d2v *)(vals + (size_t)r.z * ROWS) + lane);
      const bool p0 = v.x != 0.0, p1 = v.y != 0.0;
      const double t0 = x[p0 ? grow + r.x : 0], t1 = x[p1 ? grow + 1 + r.x : 0];
      a0 += v.x * (p0 ? t0 : 0.0), a1 += v.y * (p1 ? t1 : 0.0);
    }
  }
}

// ---- T5: templates, every slot gathered ------------------------------------------------------
template <int MINW, int TM>
__global__ __launch_bounds__(WG, MINW) void k_t5(unsigned ns, unsigned n, const unsigned *__restrict__ sptr,
                                                const unsigned char *__restrict__ tid8,
                                                const tmpl *__restrict__ td,
                                                const i4v *__restrict__ rec,
                                                const double *__restrict__ vconst,
                                                const double *__restrict__ vals,
                                                const double *__restrict__ x, double *__restrict__ y,
                                                double *__restrict__ partials) {
  __shared__ double sred[4];
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned gx = gridDim.x / NXCD, xcd = blockIdx.x % NXCD, slot = blockIdx.x / NXCD;
  const deal d = deal_init<4>(ns, xcd);
  double dot = 0.0;
  for (unsigned g = slot; g < d.turns; g += gx) {
    const unsigned s = __builtin_amdgcn_readfirstlane(d.base + g * 4 + wave);
    if (s >= ns)
      continue;
    const unsigned t = __builtin_amdgcn_readfirstlane((unsigned)tid8[s]);
    const unsigned row = s * ROWS + 2 * lane;
    const int grow = (int)row;
    double a0 = 0.0, a1 = 0.0;
    d2v xd = {0.0, 0.0};
    if (row + 1 < n)
      xd = *(const d2v *)(x + row);
    else if (row < n)
      xd.x = x[row];
    if (t != 255u) {
      const tmpl *T = td + t;
      const int cnt = T->nslots;
      d2u v[TM];
#pragma unroll
      for (int u = 0; u < TM; u++)
        if (u < cnt)
          v[u] = *(const d2u *)(x + (grow + T->base[u]));
#pragma unroll
      for (int u = 0; u < TM; u++)
        if (u < cnt) {
          const double c = T->cst[u];
          a0 += c * v[u].x, a1 += c * v[u].y;
        }
    } else {
      slice_generic(s, sptr, rec, vconst, vals, x, grow, lane, a0, a1);
    }
    if (row + 1 < n) {
      const d2v o = {a0, a1};
      *(d2v *)(y + row) = o;
      dot += a0 * xd.x, dot += a1 * xd.y;
    } else if (row < n) {
      y[row] = a0;
      dot += a0 * xd.x;
    }
  }
  wg_sum1(dot, sred);
  if (threadIdx.x == 0)
    partials[blockIdx.x] = dot;
}

// one pure slice through the shift path: gathers for kind 0 slots, lane shifts for the
// neighbours of a gathered slot; the two elements beyond the wave's 128 come by one
// two-lane load.  Products in slot order (the order of the shipped kernel).
template <bool DPP>
__device__ __forceinline__ double lane_up(double v) { // value of lane - 1
  if (DPP) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false); // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
  }
  return __shfl_up(v, 1, 64);
}
template <bool DPP>
__device__ __forceinline__ double lane_down(double v) { // value of lane + 1
  if (DPP) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, false); // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
  }
  return __shfl_down(v, 1, 64);
}

template <int TM>
struct pure_loads {
  d2u v[TM];
  double edge;
};
// issue the loads of a pure slice: one 16-byte gather per kind-0 slot; where slots are derived from
// the centre pair, the two elements beyond the wave's 128 come by one two-lane load
template <int TM>
__device__ __forceinline__ void pure_issue(const tmpl *T, const double *__restrict__ x, int grow,
                                           unsigned lane, int wave_row0, unsigned n_cols, pure_loads<TM> &L) {
  const int cnt = T->nslots;
#pragma unroll
  for (int u = 0; u < TM; u++)
    if (u < cnt && T->kind[u] == 0)
      L.v[u] = *(const d2u *)(x + (grow + T->base[u]));
  L.edge = 0.0;
  const int c = T->centre;
  if (c >= 0 && (lane == 0 || lane == 63)) {
    // lane 0: x[first row + b - 1]; lane 63: x[first row + b + 128], b = base[centre] (clamped: a
    // pure slice never USES an element outside the operator, the clamp keeps the address legal)
    long long e = (long long)wave_row0 + T->base[c] + (lane == 0 ? -1 : ROWS);
    e = e < 0 ? 0 : (e >= (long long)n_cols ? (long long)n_cols - 1 : e);
    L.edge = x[e];
  }
}
template <int TM, bool DPP>
__device__ __forceinline__ void pure_fma(const tmpl *T, pure_loads<TM> &L, unsigned lane, double &a0, double &a1,
                                         d2v &centre_pair) {
  const int cnt = T->nslots, ci = T->centre;
  d2u c = {0.0, 0.0};
#pragma unroll
  for (int u = 0; u < TM; u++)
    if (u == ci)
      c = L.v[u];
  double up = 0.0, dn = 0.0;
  if (ci >= 0) {
    up = lane_up<DPP>(c.y), dn = lane_down<DPP>(c.x);
    if (lane == 0)
      up = L.edge;
    if (lane == 63)
      dn = L.edge;
  }
  centre_pair.x = c.x, centre_pair.y = c.y;
#pragma unroll
  for (int u = 0; u < TM; u++)
    if (u < cnt) {
      const int kd = T->kind[u];
      d2u v = L.v[u];
      if (kd == 1) // base - 1: rows 2l, 2l + 1 read x[2l - 1], x[2l]
        v.x = up, v.y = c.x;
      else if (kd == 2) // base + 1: x[2l + 1], x[2l + 2]
        v.x = c.y, v.y = dn;
      const double k = T->cst[u];
      a0 += k * v.x, a1 += k * v.y; // slot order: the order of the shipped kernel
    }
}

// ---- T3: templates + lane shifts; PER slices per wave and turn --------------------------------
template <int MINW, int PER, bool DPP, int TM>
__global__ __launch_bounds__(WG, MINW) void k_t3(unsigned ns, unsigned n, const unsigned *__restrict__ sptr,
                                                const unsigned char *__restrict__ tid8,
                                                const tmpl *__restrict__ td,
                                                const i4v *__restrict__ rec,
                                                const double *__restrict__ vconst,
                                                const double *__restrict__ vals,
                                                const double *__restrict__ x, double *__restrict__ y,
                                                double *__restrict__ partials) {
  __shared__ double sred[4];
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned gx = gridDim.x / NXCD, xcd = blockIdx.x % NXCD, slot = blockIdx.x / NXCD;
  const deal d = deal_init<4 * PER>(ns, xcd);
  double dot = 0.0;
  for (unsigned g = slot; g < d.turns; g += gx) {
    const unsigned s0 = __builtin_amdgcn_readfirstlane(d.base + (g * 4 + wave) * PER);
    unsigned t[PER];
    pure_loads<TM> L[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) // 254: no such slice
      t[k] = s0 + k < ns ? __builtin_amdgcn_readfirstlane((unsigned)tid8[s0 + k]) : 254u;
#pragma unroll
    for (int k = 0; k < PER; k++)
      if (t[k] < 254u)
        pure_issue<TM>(td + t[k], x, (int)((s0 + k) * ROWS + 2 * lane), lane, (int)((s0 + k) * ROWS), n, L[k]);
#pragma unroll
    for (int k = 0; k < PER; k++) {
      if (t[k] == 254u)
        continue;
      const unsigned s = s0 + k, row = s * ROWS + 2 * lane;
      double a0 = 0.0, a1 = 0.0;
      d2v xd = {0.0, 0.0};
      bool have_xd = false;
      if (t[k] != 255u) {
        pure_fma<TM, DPP>(td + t[k], L[k], lane, a0, a1, xd);
        have_xd = td[t[k]].centre >= 0 && td[t[k]].base[td[t[k]].centre] == 0; // the centre pair IS the dot's operand
      } else {
        slice_generic(s, sptr, rec, vconst, vals, x, (int)row, lane, a0, a1);
      }
      if (!have_xd) {
        if (row + 1 < n)
          xd = *(const d2v *)(x + row);
        else if (row < n)
          xd.x = x[row];
      }
      if (row + 1 < n) {
        const d2v o = {a0, a1};
        *(d2v *)(y + row) = o;
        dot += a0 * xd.x, dot += a1 * xd.y;
      } else if (row < n) {
        y[row] = a0;
        dot += a0 * xd.x;
      }
    }
  }
  wg_sum1(dot, sred);
  if (threadIdx.x == 0)
    partials[blockIdx.x] = dot;
}

// ---- hard: the 5-point stencil hard-coded (floor; wrong at grid-line ends and faces) ----------
template <int MINW, int PER>
__global__ __launch_bounds__(WG, MINW) void k_hard(unsigned ns, unsigned n, int nx,
                                                  const double *__restrict__ x, double *__restrict__ y,
                                                  double *__restrict__ partials) {
  __shared__ double sred[4];
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned gx = gridDim.x / NXCD, xcd = blockIdx.x % NXCD, slot = blockIdx.x / NXCD;
  const deal d = deal_init<4 * PER>(ns, xcd);
  double dot = 0.0;
  for (unsigned g = slot; g < d.turns; g += gx) {
    const unsigned s0 = __builtin_amdgcn_readfirstlane(d.base + (g * 4 + wave) * PER);
    d2u lo[PER], c[PER], hi[PER];
    double e[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const long long row = (long long)(s0 + k) * ROWS + 2 * lane;
      const long long rl = row - nx < 0 ? 0 : row - nx, rh = row + nx + 1 < (long long)n ? row + nx : row;
      const bool ok = row + 1 < (long long)n;
      lo[k] = *(const d2u *)(x + (ok ? rl : 0));
      c[k] = *(const d2u *)(x + (ok ? row : 0));
      hi[k] = *(const d2u *)(x + (ok ? rh : 0));
      e[k] = 0.0;
      if (lane == 0 || lane == 63) {
        long long q = (long long)(s0 + k) * ROWS + (lane == 0 ? -1 : ROWS);
        q = q < 0 ? 0 : (q >= (long long)n ? (long long)n - 1 : q);
        e[k] = x[q];
      }
    }
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const unsigned row = (s0 + k) * ROWS + 2 * lane;
      double up = __shfl_up(c[k].y, 1, 64), dn = __shfl_down(c[k].x, 1, 64);
      if (lane == 0)
        up = e[k];
      if (lane == 63)
        dn = e[k];
      double a0 = 0.0, a1 = 0.0;
      a0 += -1.0 * lo[k].x, a1 += -1.0 * lo[k].y;
      a0 += -1.0 * up, a1 += -1.0 * c[k].x;
      a0 += 4.0 * c[k].x, a1 += 4.0 * c[k].y;
      a0 += -1.0 * c[k].y, a1 += -1.0 * dn;
      a0 += -1.0 * hi[k].x, a1 += -1.0 * hi[k].y;
      if (row + 1 < n) {
        const d2v o = {a0, a1};
        *(d2v *)(y + row) = o;
        dot += a0 * c[k].x, dot += a1 * c[k].y;
      }
    }
  }
  wg_sum1(dot, sred);
  if (threadIdx.x == 0)
    partials[blockIdx.x] = dot;
}

// ---- copy: y = 4 x with the fused dot (HBM floor) ---------------------------------------------
__global__ __launch_bounds__(WG) void k_copy(unsigned n2, const d2v *__restrict__ x, d2v *__restrict__ y,
                                            double *__restrict__ partials) {
  __shared__ double sred[4];
  double dot = 0.0;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n2; i += (size_t)gridDim.x * WG) {
    const d2v v = x[i];
    const d2v o = {4.0 * v.x, 4.0 * v.y};
    y[i] = o;
    dot += o.x * v.x, dot += o.y * v.y;
  }
  wg_sum1(dot, sred);
  if (threadIdx.x == 0)
    partials[blockIdx.x] = dot;
}

// ------------------------------------------------------------------------------------------------
template <class T>
static T *up(const T *h, size_t cnt) {
  T *d;
  CHK(hipMalloc((void **)&d, (cnt ? cnt : 1) * sizeof(T)));
  CHK(hipMemcpy(d, h, cnt * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

static unsigned grid_for(unsigned ns, unsigned per, unsigned cap) {
  const unsigned items = (ns + per - 1) / per, chunk = (items + NXCD - 1) / NXCD, c8 = cap / NXCD;
  if (chunk <= c8)
    return chunk * NXCD;
  const unsigned each = (chunk + c8 - 1) / c8;
  return ((chunk + each - 1) / each) * NXCD;
}

int main(int argc, char **argv) {
  const int nx = argc > 1 ? atoi(argv[1]) : 3162, ny = argc > 2 ? atoi(argv[2]) : nx;
  const int reps = argc > 3 ? atoi(argv[3]) : 200;
  char spec[128];
  snprintf(spec, sizeof spec, "lap2d:nx=%d,ny=%d", nx, ny);
  unsigned ng = 0;
  struct csr *A = lsbench_matrix_synth(spec, 0, 0, &ng);
  const unsigned n = A->nrows;
  struct lsb_sell *H = lsb_csr_sellize16(A, 0);
  if (!H) {
    fprintf(stderr, "no 16-bit sliced-ELL form\n");
    return 1;
  }
  struct lsb_sell_vc *V = lsb_sell16_value_slots(H);
  const unsigned ns = H->nslice;
  printf("%s: n=%u nnz=%u slices=%u slots=%llu kept=%u code_slots=%u\n", spec, n, A->offs[n], ns,
         V->nslots, V->nval_slots, H->ncode_slots);
  if (H->ncode_slots) {
    fprintf(stderr, "lab handles code-free slices only\n");
    return 1;
  }
  // templates: pure slices (every slot constant and code-free, at most TMAX of them) with identical
  // records share one; a slot c whose neighbours c-1 / c+1 hold base[c] -+ 1 becomes the centre
  std::map<std::string, int> seen;
  std::vector<tmpl> T;
  std::vector<unsigned char> tid(ns + 8, 255);
  unsigned pure = 0, shaped = 0, maxslots = 0;
  for (unsigned s = 0; s < ns; s++) {
    const unsigned q0 = H->sptr[s] / ROWS, len = (H->sptr[s + 1] - H->sptr[s]) / ROWS;
    if (len > TMAX || len == 0)
      continue;
    bool ok = true;
    tmpl t;
    memset(&t, 0, sizeof t);
    t.nslots = (int)len, t.centre = -1;
    for (unsigned j = 0; j < len && ok; j++) {
      const int *r = V->slots + 4 * ((size_t)q0 + j);
      ok = r[1] < 0 && r[2] < 0;
      t.base[j] = r[0], t.cst[j] = V->vconst[(size_t)q0 + j];
    }
    if (!ok)
      continue;
    // centre: prefer an even base (16-byte aligned gather) with both neighbours, then with one
    int best = -1, bestscore = 0;
    for (int j = 0; j < (int)len; j++) {
      const int lo = j > 0 && t.base[j - 1] == t.base[j] - 1, hi = j + 1 < (int)len && t.base[j + 1] == t.base[j] + 1;
      const int score = (lo + hi) * 2 + ((lo + hi) && !(t.base[j] & 1));
      if (score > bestscore)
        best = j, bestscore = score;
    }
    if (best >= 0 && !getenv("LAB_NO_SHIFT")) {
      t.centre = best;
      if (best > 0 && t.base[best - 1] == t.base[best] - 1)
        t.kind[best - 1] = 1;
      if (best + 1 < (int)len && t.base[best + 1] == t.base[best] + 1)
        t.kind[best + 1] = 2;
    }
    std::string key((const char *)&t, sizeof t);
    auto it = seen.find(key);
    int id;
    if (it == seen.end()) {
      if (T.size() >= 254)
        continue; // stays generic
      id = (int)T.size(), seen[key] = id, T.push_back(t);
    } else
      id = it->second;
    tid[s] = (unsigned char)id, pure++;
    shaped += t.centre >= 0;
    if (len > maxslots)
      maxslots = len;
  }
  printf("templates: %zu; pure slices %u of %u (%u with a centre), at most %u slots\n", T.size(), pure, ns, shaped,
         maxslots);
  for (size_t k = 0; k < T.size() && k < 6; k++) {
    printf("  T%zu:", k);
    for (int j = 0; j < T[k].nslots; j++)
      printf(" (%d,%g,k%d)", T[k].base[j], T[k].cst[j], T[k].kind[j]);
    printf(" centre %d\n", T[k].centre);
  }

  // device data
  std::vector<double> hx(n), yref(n);
  srand(1);
  for (unsigned i = 0; i < n; i++)
    hx[i] = (double)rand() / RAND_MAX - 0.5;
  double dref = 0.0;
#pragma omp parallel for reduction(+ : dref)
  for (long long i = 0; i < (long long)n; i++) {
    double a = 0.0;
    for (unsigned j = A->offs[i]; j < A->offs[i + 1]; j++)
      a += A->vals[j] * hx[A->cols[j]];
    yref[i] = a, dref += a * hx[i];
  }
  double *dx = up(hx.data(), n), *dy, *dparts;
  CHK(hipMalloc((void **)&dy, (size_t)(n + ROWS) * sizeof(double)));
  CHK(hipMalloc((void **)&dparts, 4096 * sizeof(double)));
  unsigned *d_sptr = up(H->sptr, (size_t)ns + 1);
  short *d_codes = up(H->codes, (size_t)(H->ncode_slots + 1) * ROWS);
  int *d_rec = up(V->slots, 4 * ((size_t)V->nslots + 1));
  double *d_vc = up(V->vconst, (size_t)V->nslots + 1);
  double *d_vals = up(V->vals, ((size_t)V->nval_slots + 1) * ROWS);
  unsigned char *d_tid = up(tid.data(), tid.size());
  tmpl *d_T = up(T.data(), T.size());
  hipStream_t st;
  CHK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  std::vector<double> hy(n), hp(4096);
  struct lsb_ar_tail notail;
  memset(&notail, 0, sizeof notail);

  auto run = [&](const char *name, unsigned g, auto launch, bool exact) {
    CHK(hipMemsetAsync(dy, 0xff, (size_t)n * sizeof(double), st));
    for (int i = 0; i < 5; i++)
      launch();
    CHK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; i++)
      launch();
    CHK(hipEventRecord(e1, st));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    CHK(hipMemcpy(hy.data(), dy, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    CHK(hipMemcpy(hp.data(), dparts, g * sizeof(double), hipMemcpyDeviceToHost));
    double dot = 0.0, maxd = 0.0;
    for (unsigned i = 0; i < g; i++)
      dot += hp[i];
    size_t bad = 0;
    for (unsigned i = 0; i < n; i++) {
      const double dd = fabs(hy[i] - yref[i]);
      if (!(dd <= 1e-12))
        bad++;
      if (dd > maxd)
        maxd = dd;
    }
    const double us = ms * 1e3 / reps;
    printf("%-28s grid %5u  %7.2f us  %6.2f TB/s on 176 MB  %s bad=%zu maxdiff=%.1e dot rel %.1e\n", name, g, us,
           176.0e6 / us / 1e6, exact ? (bad ? "WRONG" : "ok") : "(floor, not exact)", bad, maxd,
           fabs(dot - dref) / fabs(dref));
    fflush(stdout);
  };

  for (unsigned cap : {1536u, 2048u}) {
    unsigned np = 0;
    const unsigned g = lsb_k_spmv_grid(LSB_SPMV_SELL, n, ns, 0, cap);
    for (unsigned fl : {LSB_SP_C16 | LSB_SP_NT, (unsigned)LSB_SP_C16}) {
      char nm[64];
      snprintf(nm, sizeof nm, "prod flags=%u cap=%u", fl, cap);
      run(nm, g, [&] {
        lsb_k_spmv_sell(fl, cap, 0, d_sptr, 0, ns, n, 0, n, d_codes, d_rec, d_vals, d_vc, 0, dx, dy, dx, dparts, &np,
                        NULL, &notail, NULL, st);
      }, true);
    }
  }
#define TMK 5
#define ARGS ns, n, d_sptr, d_tid, d_T, (const i4v *)d_rec, d_vc, d_vals, dx, dy, dparts
  for (unsigned cap : {1024u, 1536u, 2048u}) {
    char nm[64];
    unsigned g = grid_for(ns, 4, cap);
    snprintf(nm, sizeof nm, "T5 minw6 cap=%u", cap);
    run(nm, g, [&] { k_t5<6, TMK><<<g, WG, 0, st>>>(ARGS); }, true);
    snprintf(nm, sizeof nm, "T5 minw8 cap=%u", cap);
    run(nm, g, [&] { k_t5<8, TMK><<<g, WG, 0, st>>>(ARGS); }, true);
    snprintf(nm, sizeof nm, "T3 shfl minw6 cap=%u", cap);
    run(nm, g, [&] { k_t3<6, 1, false, TMK><<<g, WG, 0, st>>>(ARGS); }, true);
    snprintf(nm, sizeof nm, "T3 shfl minw8 cap=%u", cap);
    run(nm, g, [&] { k_t3<8, 1, false, TMK><<<g, WG, 0, st>>>(ARGS); }, true);
    snprintf(nm, sizeof nm, "T3 dpp minw8 cap=%u", cap);
    run(nm, g, [&] { k_t3<8, 1, true, TMK><<<g, WG, 0, st>>>(ARGS); }, true);
    g = grid_for(ns, 8, cap);
    snprintf(nm, sizeof nm, "T3x2 shfl minw6 cap=%u", cap);
    run(nm, g, [&] { k_t3<6, 2, false, TMK><<<g, WG, 0, st>>>(ARGS); }, true);
    snprintf(nm, sizeof nm, "T3x2 dpp minw6 cap=%u", cap);
    run(nm, g, [&] { k_t3<6, 2, true, TMK><<<g, WG, 0, st>>>(ARGS); }, true);
    snprintf(nm, sizeof nm, "T3x2 dpp minw8 cap=%u", cap);
    run(nm, g, [&] { k_t3<8, 2, true, TMK><<<g, WG, 0, st>>>(ARGS); }, true);
    g = grid_for(ns, 16, cap);
    snprintf(nm, sizeof nm, "T3x4 dpp minw4 cap=%u", cap);
    run(nm, g, [&] { k_t3<4, 4, true, TMK><<<g, WG, 0, st>>>(ARGS); }, true);
    g = grid_for(ns, 4, cap);
    snprintf(nm, sizeof nm, "hard x1 minw8 cap=%u", cap);
    run(nm, g, [&] { k_hard<8, 1><<<g, WG, 0, st>>>(ns, n, nx, dx, dy, dparts); }, false);
    g = grid_for(ns, 8, cap);
    snprintf(nm, sizeof nm, "hard x2 minw8 cap=%u", cap);
    run(nm, g, [&] { k_hard<8, 2><<<g, WG, 0, st>>>(ns, n, nx, dx, dy, dparts); }, false);
    g = grid_for(ns, 16, cap);
    snprintf(nm, sizeof nm, "hard x4 minw4 cap=%u", cap);
    run(nm, g, [&] { k_hard<4, 4><<<g, WG, 0, st>>>(ns, n, nx, dx, dy, dparts); }, false);
  }
  for (unsigned g : {1024u, 2048u, 4096u})
    run("copy y=4x + dot", g, [&] { k_copy<<<g, WG, 0, st>>>(n / 2, (const d2v *)dx, (d2v *)dy, dparts); }, false);
  return 0;
}
