/*
 * Synthetic operators of BASELINE.json configs 3-5,
 * generated on the host straight into the reference's CSR layout
 * (src/lsbench-impl.h:22-26) -- a 50 M..447 M-nnz matrix cannot sensibly go
 * through the text loader.  Any row range [r0,r1) can
 * be produced on its own, which is how each rank of a row-partitioned run
 * builds only its shard.  Column ids are global and 0-based.
 *
 *   lap2d   5-point Laplacian on an nx x ny grid, row = j*nx + i, diagonal 4,
 *           neighbours -1, Dirichlet boundary (missing neighbours dropped)
 *   lap3d   7-point Laplacian on nx x ny x nz, row = (k*ny + j)*nx + i,
 *           diagonal 6, neighbours -1
 *   lap2d / lap3d with ",coef=K" (K != 0): the same pattern with GENERAL values --
 *           the stencil of -div(w grad u) with one weight per grid edge,
 *           w(a,b) = 1/2 + (mix(K, a, b) >> 11) * 2^-53 in [1/2, 3/2) for the
 *           edge between rows a < b, and w = 1/2 + (mix(K ^ GHOST, row, d) >> 11)
 *           * 2^-53 for the edge that leaves the grid in direction d (0..5 =
 *           k-, j-, i-, i+, j+, k+; Dirichlet).  Off-diagonal = -w, diagonal =
 *           the weights of the row's 4 (6) edges added in direction order from
 *           0.0.  Exactly symmetric, weakly diagonally dominant with strict rows
 *           at the boundary, irreducible => SPD.  The operator the roofline line
 *           "fp64 CSR SpMV with values that must be streamed" is measured on.
 *   powerlaw  n x n, row degree d_i from a truncated discrete power law
 *           P(d) ~ d^-gamma on [1,max] by inverse CDF on a 2^53-scaled integer
 *           table; the k-th entry of row i sits in column
 *           floor((k*n + u_k)/d_i), u_k = mix(seed,i,2k+1) mod (n-d_i+1)
 *           (one per stratum => sorted, distinct); value uniform in [-1,1)
 *           from mix(seed,i,2k+2).  mix = splitmix64 finaliser, counter-based,
 *           so any row is reproducible in isolation.  Unsymmetric: a SpMV /
 *           load-balance stress, not a CG operator.
 */
#define _GNU_SOURCE
#include "lsb_impl.h"
#include <math.h>
#include <string.h>

static int spec_get(const char *spec, const char *key, double *out) {
  /* spec = "name:k1=v1,k2=v2" */
  const char *p = strchr(spec, ':');
  size_t kl = strlen(key);
  p = p ? p + 1 : spec;
  while (*p) {
    if (strncmp(p, key, kl) == 0 && p[kl] == '=') {
      *out = strtod(p + kl + 1, NULL);
      return 1;
    }
    p = strchr(p, ',');
    if (!p)
      break;
    p++;
  }
  return 0;
}

static struct csr *alloc_rows(unsigned nrows) {
  struct csr *A = lsb_calloc(struct csr, 1);
  A->nrows = nrows, A->base = 0;
  A->offs = lsb_calloc(unsigned, (size_t)nrows + 1);
  return A;
}

static void alloc_entries(struct csr *A, const unsigned long long *cnt) {
  unsigned long long acc = 0;
  for (unsigned i = 0; i < A->nrows; i++) {
    A->offs[i] = (unsigned)acc;
    acc += cnt ? cnt[i] : 0;
  }
  if (acc > 0xFFFFFFFFull)
    errx(EXIT_FAILURE, "synthetic shard with %llu non-zeros exceeds 32-bit offsets", acc);
  A->offs[A->nrows] = (unsigned)acc;
  A->cols = (unsigned *)malloc((size_t)(acc ? acc : 1) * sizeof(unsigned));
  A->vals = (double *)malloc((size_t)(acc ? acc : 1) * sizeof(double));
  if (!A->cols || !A->vals)
    errx(EXIT_FAILURE, "out of host memory for %llu synthetic non-zeros", acc);
}

static unsigned long long mix64(unsigned long long seed, unsigned long long a,
                                unsigned long long b) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (a + 1) +
                         0xC2B2AE3D27D4EB4Full * (b + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

#define TWO53 9007199254740992.0
#define COEF_GHOST 0x5851F42D4C957F2Dull

/* weight of the grid edge between rows a < b / of the edge leaving the grid */
static double edge_w(unsigned long long cseed, unsigned long long a, unsigned long long b) {
  return 0.5 + (double)(mix64(cseed, a, b) >> 11) * (1.0 / TWO53);
}
static double ghost_w(unsigned long long cseed, unsigned long long row, unsigned dir) {
  return 0.5 + (double)(mix64(cseed ^ COEF_GHOST, row, dir) >> 11) * (1.0 / TWO53);
}

static struct csr *gen_lap(unsigned long long nx, unsigned long long ny,
                           unsigned long long nz, int three,
                           unsigned long long r0, unsigned long long r1,
                           unsigned long long cseed) {
  const unsigned long long nxy = nx * ny;
  const double diag = three ? 6.0 : 4.0;
  struct csr *A = alloc_rows((unsigned)(r1 - r0));
  unsigned long long *cnt = (unsigned long long *)malloc((size_t)(r1 - r0 + 1) * sizeof *cnt);
#pragma omp parallel for schedule(static)
  for (long long rr = (long long)r0; rr < (long long)r1; rr++) {
    const unsigned long long row = (unsigned long long)rr;
    const unsigned long long i = row % nx, j = (row / nx) % ny, k = row / nxy;
    cnt[row - r0] = 1 + (i > 0) + (i + 1 < nx) + (j > 0) + (j + 1 < ny) +
                    (three ? (k > 0) + (k + 1 < nz) : 0);
  }
  alloc_entries(A, cnt);
  free(cnt);
#pragma omp parallel for schedule(static)
  for (long long rr = (long long)r0; rr < (long long)r1; rr++) {
    const unsigned long long row = (unsigned long long)rr;
    const unsigned long long i = row % nx, j = (row / nx) % ny, k = row / nxy;
    unsigned z = A->offs[row - r0], zd;
    if (!cseed) {
      if (three && k > 0)
        A->cols[z] = (unsigned)(row - nxy), A->vals[z++] = -1.0;
      if (j > 0)
        A->cols[z] = (unsigned)(row - nx), A->vals[z++] = -1.0;
      if (i > 0)
        A->cols[z] = (unsigned)(row - 1), A->vals[z++] = -1.0;
      A->cols[z] = (unsigned)row, A->vals[z++] = diag;
      if (i + 1 < nx)
        A->cols[z] = (unsigned)(row + 1), A->vals[z++] = -1.0;
      if (j + 1 < ny)
        A->cols[z] = (unsigned)(row + nx), A->vals[z++] = -1.0;
      if (three && k + 1 < nz)
        A->cols[z] = (unsigned)(row + nxy), A->vals[z++] = -1.0;
      continue;
    }
    /* general values: one weight per edge, the diagonal = their sum in direction order */
    double w, d = 0.0;
    if (three) {
      w = k > 0 ? edge_w(cseed, row - nxy, row) : ghost_w(cseed, row, 0);
      d += w;
      if (k > 0)
        A->cols[z] = (unsigned)(row - nxy), A->vals[z++] = -w;
    }
    w = j > 0 ? edge_w(cseed, row - nx, row) : ghost_w(cseed, row, 1);
    d += w;
    if (j > 0)
      A->cols[z] = (unsigned)(row - nx), A->vals[z++] = -w;
    w = i > 0 ? edge_w(cseed, row - 1, row) : ghost_w(cseed, row, 2);
    d += w;
    if (i > 0)
      A->cols[z] = (unsigned)(row - 1), A->vals[z++] = -w;
    zd = z++;
    A->cols[zd] = (unsigned)row;
    w = i + 1 < nx ? edge_w(cseed, row, row + 1) : ghost_w(cseed, row, 3);
    d += w;
    if (i + 1 < nx)
      A->cols[z] = (unsigned)(row + 1), A->vals[z++] = -w;
    w = j + 1 < ny ? edge_w(cseed, row, row + nx) : ghost_w(cseed, row, 4);
    d += w;
    if (j + 1 < ny)
      A->cols[z] = (unsigned)(row + nx), A->vals[z++] = -w;
    if (three) {
      w = k + 1 < nz ? edge_w(cseed, row, row + nxy) : ghost_w(cseed, row, 5);
      d += w;
      if (k + 1 < nz)
        A->cols[z] = (unsigned)(row + nxy), A->vals[z++] = -w;
    }
    A->vals[zd] = d;
  }
  return A;
}

/* thr[d-1] = floor(2^53 * CDF(d)) of P(d) ~ d^-gamma on [1,dmax] */
static double pl_table(double gamma, unsigned dmax, unsigned long long *thr) {
  double tot = 0.0, acc = 0.0, mean = 0.0;
  for (unsigned d = 1; d <= dmax; d++)
    tot += pow((double)d, -gamma);
  for (unsigned d = 1; d <= dmax; d++) {
    const double pd = pow((double)d, -gamma) / tot;
    acc += pd, mean += d * pd;
    thr[d - 1] = (unsigned long long)floor((acc >= 1.0 ? 1.0 : acc) * TWO53);
  }
  thr[dmax - 1] = (unsigned long long)TWO53;
  return mean;
}

static unsigned pl_degree(const unsigned long long *thr, unsigned dmax,
                          unsigned long long u53) {
  unsigned lo = 0, hi = dmax - 1;
  while (lo < hi) {
    const unsigned mid = (lo + hi) / 2;
    if (u53 < thr[mid])
      hi = mid;
    else
      lo = mid + 1;
  }
  return lo + 1;
}

/* exponent whose truncated power law has the requested mean (bisection) */
static double pl_gamma_for_mean(double avg, unsigned dmax) {
  unsigned long long *thr = (unsigned long long *)malloc((size_t)dmax * sizeof *thr);
  double lo = 0.0, hi = 8.0;
  for (int it = 0; it < 100; it++) {
    const double mid = 0.5 * (lo + hi);
    if (pl_table(mid, dmax, thr) > avg)
      lo = mid;
    else
      hi = mid;
  }
  free(thr);
  return 0.5 * (lo + hi);
}

static struct csr *gen_powerlaw(unsigned long long n, double gamma, unsigned dmax,
                                unsigned long long seed, unsigned long long r0,
                                unsigned long long r1) {
  unsigned long long *thr = (unsigned long long *)malloc((size_t)dmax * sizeof *thr);
  pl_table(gamma, dmax, thr);
  struct csr *A = alloc_rows((unsigned)(r1 - r0));
  unsigned long long *cnt = (unsigned long long *)malloc((size_t)(r1 - r0 + 1) * sizeof *cnt);
#pragma omp parallel for schedule(static)
  for (long long rr = (long long)r0; rr < (long long)r1; rr++) {
    unsigned long long d = pl_degree(thr, dmax, mix64(seed, (unsigned long long)rr, 0) >> 11);
    cnt[rr - (long long)r0] = d > n ? n : d;
  }
  alloc_entries(A, cnt);
  free(cnt);
#pragma omp parallel for schedule(dynamic, 4096)
  for (long long rr = (long long)r0; rr < (long long)r1; rr++) {
    const unsigned long long row = (unsigned long long)rr;
    unsigned z = A->offs[row - r0];
    const unsigned long long d = A->offs[row - r0 + 1] - z;
    for (unsigned long long k = 0; k < d; k++, z++) {
      const unsigned long long uk = mix64(seed, row, 2 * k + 1) % (n - d + 1);
      A->cols[z] = (unsigned)(((unsigned __int128)k * n + uk) / d);
      A->vals[z] = (double)(mix64(seed, row, 2 * k + 2) >> 11) * (2.0 / TWO53) - 1.0;
    }
  }
  free(thr);
  return A;
}

/* "spd=1": the symmetric positive definite operator SURVEY.md section 8(d) cfg5
 * defines for CG runs on the power-law structure,
 *     S = (B + B^T) + diag(1 + sum_j |(B + B^T)_ij|),
 * strictly diagonally dominant with a positive diagonal.  A row of B^T needs
 * every row of B, so the whole of B is generated (and freed) whatever the
 * requested row range is; rows [r0, r1) of S are returned. */
static struct csr *gen_powerlaw_spd(unsigned long long n, double gamma, unsigned dmax,
                                    unsigned long long seed, unsigned long long r0,
                                    unsigned long long r1) {
  struct csr *B = gen_powerlaw(n, gamma, dmax, seed, 0, n);
  const unsigned long long nnz = B->offs[n];
  if (2 * nnz + n > 0xFFFFFFFEull)
    errx(EXIT_FAILURE, "powerlaw spd=1: operator too large for 32-bit offsets");
  /* B^T by counting sort: columns of a row of B^T come out ascending */
  unsigned *toffs = lsb_calloc(unsigned, (size_t)n + 2);
  for (unsigned long long j = 0; j < nnz; j++)
    toffs[B->cols[j] + 2]++;
  for (unsigned long long i = 0; i < n; i++)
    toffs[i + 2] += toffs[i + 1];
  unsigned *tcols = (unsigned *)malloc((size_t)(nnz ? nnz : 1) * sizeof(unsigned));
  double *tvals = (double *)malloc((size_t)(nnz ? nnz : 1) * sizeof(double));
  if (!tcols || !tvals)
    errx(EXIT_FAILURE, "out of host memory for the transposed power-law operator");
  for (unsigned long long i = 0; i < n; i++)
    for (unsigned j = B->offs[i]; j < B->offs[i + 1]; j++) {
      const unsigned e = toffs[B->cols[j] + 1]++;
      tcols[e] = (unsigned)i, tvals[e] = B->vals[j];
    }
  /* toffs[i] .. toffs[i+1] is now row i of B^T */
  struct csr *S = alloc_rows((unsigned)(r1 - r0));
  unsigned long long *cnt = (unsigned long long *)malloc((size_t)(r1 - r0 + 1) * sizeof *cnt);
#pragma omp parallel for schedule(dynamic, 4096)
  for (long long rr = (long long)r0; rr < (long long)r1; rr++) {
    unsigned a = B->offs[rr], ae = B->offs[rr + 1], b = toffs[rr], be = toffs[rr + 1];
    unsigned long long c = 0;
    int diag = 0;
    while (a < ae || b < be) {
      const unsigned ca = a < ae ? B->cols[a] : 0xFFFFFFFFu, cb = b < be ? tcols[b] : 0xFFFFFFFFu;
      const unsigned col = ca < cb ? ca : cb;
      a += ca == col, b += cb == col;
      diag |= col == (unsigned)rr;
      c++;
    }
    cnt[rr - (long long)r0] = c + !diag;
  }
  alloc_entries(S, cnt);
  free(cnt);
#pragma omp parallel for schedule(dynamic, 4096)
  for (long long rr = (long long)r0; rr < (long long)r1; rr++) {
    unsigned a = B->offs[rr], ae = B->offs[rr + 1], b = toffs[rr], be = toffs[rr + 1];
    unsigned z = S->offs[rr - (long long)r0], zd = 0xFFFFFFFFu;
    double sum = 0.0;
    while (a < ae || b < be) {
      const unsigned ca = a < ae ? B->cols[a] : 0xFFFFFFFFu, cb = b < be ? tcols[b] : 0xFFFFFFFFu;
      const unsigned col = ca < cb ? ca : cb;
      if (zd == 0xFFFFFFFFu && col > (unsigned)rr) /* the diagonal's place, if B + B^T has none */
        zd = z, S->cols[z] = (unsigned)rr, S->vals[z++] = 0.0;
      double v = 0.0;
      if (ca == col)
        v += B->vals[a++];
      if (cb == col)
        v += tvals[b++];
      sum += fabs(v);
      if (col == (unsigned)rr)
        zd = z;
      S->cols[z] = col, S->vals[z++] = v;
    }
    if (zd == 0xFFFFFFFFu)
      zd = z, S->cols[z] = (unsigned)rr, S->vals[z++] = 0.0;
    S->vals[zd] += 1.0 + sum;
  }
  free(toffs), free(tcols), free(tvals);
  lsb_csr_free(B);
  return S;
}

struct csr *lsbench_matrix_synth(const char *spec, unsigned r0, unsigned r1,
                                 unsigned *n_global) {
  double v;
  unsigned long long n = 0;
  int kind = 0;
  unsigned long long nx = 1, ny = 1, nz = 1, seed = 20240607ull;
  double gamma = 0.0;
  unsigned dmax = 4096;
  if (strncmp(spec, "lap2d", 5) == 0) {
    kind = 1;
    if (!spec_get(spec, "nx", &v))
      return NULL;
    nx = (unsigned long long)v;
    ny = spec_get(spec, "ny", &v) ? (unsigned long long)v : nx;
    n = nx * ny;
  } else if (strncmp(spec, "lap3d", 5) == 0) {
    kind = 2;
    if (!spec_get(spec, "nx", &v))
      return NULL;
    nx = (unsigned long long)v;
    ny = spec_get(spec, "ny", &v) ? (unsigned long long)v : nx;
    nz = spec_get(spec, "nz", &v) ? (unsigned long long)v : nx;
    n = nx * ny * nz;
  } else if (strncmp(spec, "powerlaw", 8) == 0) {
    kind = 3;
    if (!spec_get(spec, "n", &v))
      return NULL;
    n = (unsigned long long)v;
    if (spec_get(spec, "max", &v))
      dmax = (unsigned)v;
    if (spec_get(spec, "seed", &v))
      seed = (unsigned long long)v;
    if (spec_get(spec, "gamma", &v))
      gamma = v;
    else if (spec_get(spec, "avg", &v))
      gamma = pl_gamma_for_mean(v, dmax);
    else
      return NULL;
  } else {
    return NULL;
  }
  if (n == 0 || n > 0xFFFFFFFFull || dmax == 0)
    return NULL;
  if (r1 == 0 || r1 > n)
    r1 = (unsigned)n;
  if (r0 > r1)
    return NULL;
  if (n_global)
    *n_global = (unsigned)n;
  if (kind == 3 && spec_get(spec, "spd", &v) && v != 0.0)
    return gen_powerlaw_spd(n, gamma, dmax, seed, r0, r1);
  if (kind == 3)
    return gen_powerlaw(n, gamma, dmax, seed, r0, r1);
  unsigned long long cseed = 0;
  if (spec_get(spec, "coef", &v) && v != 0.0)
    cseed = (unsigned long long)v;
  return gen_lap(nx, ny, kind == 2 ? nz : 1, kind == 2, r0, r1, cseed);
}
