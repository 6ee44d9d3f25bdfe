/*
 * Host-side preparation of the operator the HIP backend solves with
 * (SURVEY.md section 8 a2-8): everything between the `struct csr` the caller
 * hands over and the int32 CSR shards that are uploaded to HBM.  Pure C, no
 * GPU needed -- unit-tested on the CPU.
 */
#include "lsb_impl.h"
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#else
static inline int omp_get_max_threads(void) { return 1; }
#endif

static struct csr *csr_alloc(unsigned nrows, unsigned long long nnz) {
  if (nnz > 0xFFFFFFFFull)
    errx(EXIT_FAILURE, "CSR with %llu non-zeros does not fit 32-bit offsets "
                       "(reference layout, src/lsbench-impl.h:22-26)", nnz);
  struct csr *S = lsb_calloc(struct csr, 1);
  S->nrows = nrows, S->base = 0;
  S->offs = lsb_calloc(unsigned, (size_t)nrows + 1);
  S->cols = lsb_calloc(unsigned, (size_t)nnz);
  S->vals = lsb_calloc(double, (size_t)nnz);
  if (!S->offs || !S->cols || !S->vals)
    errx(EXIT_FAILURE, "out of host memory for a %u-row / %llu-nnz CSR", nrows, nnz);
  return S;
}

void lsb_csr_free(struct csr *A) {
  if (!A)
    return;
  free(A->offs), free(A->cols), free(A->vals);
  free(A);
}

/*
 * The matrix CHOLMOD is given by the reference (src/cholmod-impl.h:5-21): of
 * each row i only the entries with column >= i are kept, as triplets of a
 * symmetric matrix with stype = -1, which CHOLMOD completes by transposition.
 * So S(i,j) = S(j,i) = A(i,j) for j >= i: S = triu(A) + triu(A,1)^T.
 *
 * Built without sorting: row r of S is [ A(i,r) for i < r, i ascending ] ++
 * [ A(r,j) for j >= r ], and scanning the rows of A in ascending order appends
 * the mirrored entries of every row in ascending column order.
 */
struct csr *lsb_csr_symmetrize_upper(const struct csr *A) {
  const unsigned n = A->nrows, base = A->base;
  unsigned *first_upper = lsb_calloc(unsigned, (size_t)n + 1);
  unsigned long long *cnt = lsb_calloc(unsigned long long, (size_t)n + 1);
  for (unsigned i = 0; i < n; i++) {
    unsigned j = A->offs[i];
    const unsigned je = A->offs[i + 1];
    while (j < je && A->cols[j] - base < i) /* strictly lower part: dropped */
      j++;
    first_upper[i] = j;
    for (; j < je; j++) {
      const unsigned c = A->cols[j] - base;
      if (c >= n)
        errx(EXIT_FAILURE, "column %u of row %u is outside the %u x %u matrix",
             c + base, i + base, n, n);
      cnt[i]++;
      if (c != i)
        cnt[c]++;
    }
  }
  unsigned long long nnz = 0;
  for (unsigned i = 0; i < n; i++)
    nnz += cnt[i];
  struct csr *S = csr_alloc(n, nnz);
  /* fill[r] = next free slot of row r; mirrored (lower) entries first */
  unsigned *fill = lsb_calloc(unsigned, (size_t)n + 1);
  unsigned long long acc = 0;
  for (unsigned i = 0; i < n; i++) {
    S->offs[i] = (unsigned)acc, fill[i] = (unsigned)acc;
    acc += cnt[i];
  }
  S->offs[n] = (unsigned)acc;
  for (unsigned i = 0; i < n; i++) {
    for (unsigned j = first_upper[i]; j < A->offs[i + 1]; j++) {
      const unsigned c = A->cols[j] - base;
      if (c != i) {
        S->cols[fill[c]] = i, S->vals[fill[c]] = A->vals[j];
        fill[c]++;
      }
    }
  }
  for (unsigned i = 0; i < n; i++) {
    for (unsigned j = first_upper[i]; j < A->offs[i + 1]; j++) {
      S->cols[fill[i]] = A->cols[j] - base, S->vals[fill[i]] = A->vals[j];
      fill[i]++;
    }
  }
  free(first_upper), free(cnt), free(fill);
  return S;
}

struct csr *lsb_csr_copy_base0(const struct csr *A) {
  const unsigned n = A->nrows, nnz = A->offs[n];
  struct csr *S = csr_alloc(n, nnz);
  memcpy(S->offs, A->offs, ((size_t)n + 1) * sizeof(unsigned));
  memcpy(S->vals, A->vals, (size_t)nnz * sizeof(double));
  for (unsigned j = 0; j < nnz; j++)
    S->cols[j] = A->cols[j] - A->base;
  return S;
}

/* LINE PADDING (include/lsbench_hip.h).  The operator of a 2-D grid with constant coefficients --
 * every row's entries at offsets out of {-nx, -1, 0, +1, +nx}, one value per offset, no +-1 entry
 * across the end of a grid line -- re-numbered so that every grid line starts at a multiple of
 * `slice` rows: row (i, j) -> j nxp + i, nxp = nx rounded up.  The nxp - nx rows a line gains hold
 * the SAME stencil among themselves (a strip of the grid's height, cut off from the real unknowns by
 * the missing +-1 entry at the line's end): the padded operator is block-diagonal, real block =
 * the operator, pad block = a principal submatrix of the same stencil on a larger grid (SPD with
 * it), and with b = 0, x0 = 0 on the pad rows every Krylov vector stays exactly 0 there.  What it
 * buys: the +-nx diagonals become whole-slice offsets, every interior slot of the sliced-ELL
 * copy is constant, and the z-column walk / the two-launch iteration (k_spmv_tmpl_col,
 * k_pcg_col_px) apply to a 2-D grid along y. */
struct csr *lsb_csr_pad_lines(const struct csr *S, unsigned slice, unsigned *nx_out, unsigned *nxp_out, int **map_out) {
  if (!S || S->base != 0 || slice < 2 || S->nrows < 4 * slice)
    return NULL;
  const unsigned n = S->nrows;
  /* the far offset: the first row that reaches forward says nx */
  unsigned nx = 0;
  for (unsigned r = 0; r < n && !nx; r++)
    if (S->offs[r + 1] > S->offs[r] && S->cols[S->offs[r + 1] - 1] > r + 1)
      nx = S->cols[S->offs[r + 1] - 1] - r;
  if (nx < 2 * slice || n % nx || n / nx < 3)
    return NULL;
  const unsigned ny = n / nx, nxp = (nx + slice - 1) / slice * slice;
  if (nxp == nx || (unsigned long long)(nxp - nx) * 16 > nx || (unsigned long long)ny * nxp > 0x7fffffffull)
    return NULL;
  /* offsets out of {-nx, -1, 0, 1, nx}, one value per offset (bit for bit), no +-1 across a line end, a diagonal
   * everywhere */
  double val[5] = {0, 0, 0, 0, 0};
  int have[5] = {0, 0, 0, 0, 0};
  for (unsigned r = 0; r < n; r++) {
    int diag = 0;
    for (unsigned k = S->offs[r]; k < S->offs[r + 1]; k++) {
      const long long d = (long long)S->cols[k] - (long long)r;
      const int w = d == -(long long)nx ? 0 : d == -1 ? 1 : d == 0 ? 2 : d == 1 ? 3 : d == (long long)nx ? 4 : -1;
      if (w < 0 || (w == 1 && r % nx == 0) || (w == 3 && r % nx == nx - 1))
        return NULL;
      if (!have[w])
        val[w] = S->vals[k], have[w] = 1;
      else if (memcmp(&val[w], &S->vals[k], sizeof(double)))
        return NULL;
      diag |= w == 2;
    }
    if (!diag)
      return NULL;
  }
  if (!have[0] || !have[1] || !have[3] || !have[4] || val[2] == 0.0)
    return NULL;
  const unsigned np = ny * nxp;
  /* entries: the operator's + at most five per pad row */
  const unsigned long long cap = (unsigned long long)S->offs[n] + 5ull * (unsigned long long)ny * (nxp - nx);
  if (cap > 0xffffffffull)
    return NULL;
  struct csr *Pd = csr_alloc(np, (unsigned)cap);
  int *map = (int *)malloc((size_t)np * sizeof(int));
  if (!map)
    errx(EXIT_FAILURE, "lsb_csr_pad_lines: out of memory");
  unsigned at = 0;
  for (unsigned j = 0; j < ny; j++)
    for (unsigned i = 0; i < nxp; i++) {
      const unsigned row = j * nxp + i;
      Pd->offs[row] = at;
      if (i < nx) {
        const unsigned r = j * nx + i;
        map[row] = (int)r;
        for (unsigned k = S->offs[r]; k < S->offs[r + 1]; k++) {
          const unsigned c = S->cols[k];
          Pd->cols[at] = c / nx * nxp + c % nx, Pd->vals[at] = S->vals[k], at++;
        }
      } else {
        map[row] = -1;
        if (j > 0)
          Pd->cols[at] = row - nxp, Pd->vals[at] = val[0], at++;
        if (i > nx)
          Pd->cols[at] = row - 1, Pd->vals[at] = val[1], at++;
        Pd->cols[at] = row, Pd->vals[at] = val[2], at++;
        if (i + 1 < nxp)
          Pd->cols[at] = row + 1, Pd->vals[at] = val[3], at++;
        if (j + 1 < ny)
          Pd->cols[at] = row + nxp, Pd->vals[at] = val[4], at++;
      }
    }
  Pd->offs[np] = at;
  if (nx_out)
    *nx_out = nx;
  if (nxp_out)
    *nxp_out = nxp;
  if (map_out)
    *map_out = map;
  else
    free(map);
  return Pd;
}

struct csr *lsb_csr_row_slice(const struct csr *A, unsigned r0, unsigned r1) {
  if (r1 > A->nrows || r0 > r1)
    return NULL;
  const unsigned j0 = A->offs[r0], j1 = A->offs[r1];
  struct csr *S = csr_alloc(r1 - r0, j1 - j0);
  for (unsigned i = r0; i <= r1; i++)
    S->offs[i - r0] = A->offs[i] - j0;
  memcpy(S->vals, A->vals + j0, (size_t)(j1 - j0) * sizeof(double));
  for (unsigned j = j0; j < j1; j++)
    S->cols[j - j0] = A->cols[j] - A->base;
  return S;
}

/* first row whose offset is >= target (offs is non-decreasing) */
static unsigned lower_bound_offs(const unsigned *offs, unsigned n,
                                 unsigned long long target) {
  unsigned lo = 0, hi = n;
  while (lo < hi) {
    unsigned mid = lo + (hi - lo) / 2;
    if (offs[mid] < target)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

/*
 * 1-D row-range partition balanced by non-zeros:
 * bounds[p] = first row whose offset reaches p/nparts of nnz, rounded to an
 * even row so that every shard's vectors stay 16-byte aligned inside the
 * full-length vector (keeps the BLAS-1 kernels on their 16 B/lane path).
 */
int lsb_csr_partition_rows(const struct csr *A, unsigned nparts,
                           unsigned *bounds) {
  if (nparts == 0)
    return 2;
  const unsigned n = A->nrows;
  const unsigned long long nnz = A->offs[n];
  bounds[0] = 0;
  for (unsigned p = 1; p < nparts; p++) {
    unsigned r = lower_bound_offs(A->offs, n, nnz * p / nparts);
    r &= ~1u;
    if (r < bounds[p - 1])
      r = bounds[p - 1];
    bounds[p] = r;
  }
  bounds[nparts] = n;
  return 0;
}

/*
 * Row blocks of the adaptive SpMV (hip_kernels.hip k_spmv_adaptive): greedy
 * packing of consecutive rows into blocks of <= cap non-zeros; a row longer
 * than cap is a block by itself.  Each step is one binary search on offs.
 */
unsigned lsb_csr_row_blocks(const struct csr *A, unsigned cap,
                            unsigned **rowblk_out) {
  const unsigned n = A->nrows;
  size_t capb = (size_t)(A->offs[n] / (cap ? cap : 1)) * 2 + 16, nb = 0;
  unsigned *rb = (unsigned *)malloc(capb * sizeof(unsigned));
  unsigned r = 0;
  rb[nb++] = 0;
  while (r < n) {
    /* last row index e (exclusive) with offs[e] - offs[r] <= cap */
    const unsigned long long lim = (unsigned long long)A->offs[r] + cap;
    unsigned lo = r + 1, hi = n; /* invariant: offs[lo] may exceed; find max e */
    unsigned e = r + 1;          /* at least one row per block */
    if (A->offs[e] <= lim) {
      /* largest e in [r+1, n] with offs[e] <= lim */
      lo = r + 1, hi = n;
      while (lo < hi) {
        unsigned mid = lo + (hi - lo + 1) / 2;
        if (A->offs[mid] <= lim)
          lo = mid;
        else
          hi = mid - 1;
      }
      e = lo;
    }
    if (nb + 1 >= capb) {
      capb *= 2;
      rb = (unsigned *)realloc(rb, capb * sizeof(unsigned));
    }
    rb[nb++] = e;
    r = e;
  }
  *rowblk_out = rb;
  return (unsigned)(nb - 1);
}

/* [lo, hi) = smallest column range covering every entry of A (0-based ids). */
void lsb_csr_col_hull(const struct csr *A, unsigned *lo_out, unsigned *hi_out) {
  unsigned lo = 0xFFFFFFFFu, hi = 0;
  const unsigned nnz = A->offs[A->nrows];
  for (unsigned j = 0; j < nnz; j++) {
    const unsigned c = A->cols[j] - A->base;
    if (c < lo)
      lo = c;
    if (c + 1 > hi)
      hi = c + 1;
  }
  if (nnz == 0)
    lo = hi = 0;
  *lo_out = lo, *hi_out = hi;
}

/*
 * Exchange plan of one shard under 1-D row-range partitioning (SURVEY.md
 * section 8(e)).  hull[4q..4q+3] = {row_begin, nrows, col_lo, col_hi} of shard
 * q (a rank, or a virtual shard), `me` is this shard's id.  Every shard keeps
 * the exchanged vector in GLOBAL index space, so a transfer is just a range
 * [offset, offset+count) of it: shard `me` receives from q the part of q's
 * rows inside its own column hull, and sends the mirror image.  Both sides
 * derive the same ranges from the same table, so sends and receives pair up.
 * recv/send must hold nall entries each.
 */
void lsb_plan_exchange(int me, int nall, const unsigned *hull,
                       struct lsb_xfer *recv, int *nrecv, struct lsb_xfer *send,
                       int *nsend) {
  const unsigned my_b = hull[4 * me], my_e = my_b + hull[4 * me + 1];
  const unsigned my_lo = hull[4 * me + 2], my_hi = hull[4 * me + 3];
  int nr = 0, ns = 0;
  for (int q = 0; q < nall; q++) {
    if (q == me)
      continue;
    const unsigned qb = hull[4 * q], qe = qb + hull[4 * q + 1];
    unsigned lo = my_lo > qb ? my_lo : qb;
    unsigned hi = my_hi < qe ? my_hi : qe;
    if (lo < hi)
      recv[nr].peer = q, recv[nr].offset = lo, recv[nr].count = hi - lo, nr++;
    lo = hull[4 * q + 2] > my_b ? hull[4 * q + 2] : my_b;
    hi = hull[4 * q + 3] < my_e ? hull[4 * q + 3] : my_e;
    if (lo < hi)
      send[ns].peer = q, send[ns].offset = lo, send[ns].count = hi - lo, ns++;
  }
  *nrecv = nr, *nsend = ns;
}

/*
 * Lanes per row for each row block of the adaptive SpMV: the largest power of
 * two that still lets all rows of the block be reduced in one pass of the
 * 256-lane workgroup, raised so that no lane adds more than ~16 products of
 * the block's LONGEST row (a 1500-entry row packed with 29 short ones is then
 * reduced by 64 lanes, not by 8).  1 <= lanes <= 64.
 */
void lsb_csr_block_lanes(const struct csr *A, const unsigned *rowblk,
                         unsigned nblk, unsigned char *lanes) {
  for (unsigned k = 0; k < nblk; k++) {
    const unsigned r0 = rowblk[k], r1 = rowblk[k + 1], nr = r1 - r0;
    unsigned maxlen = 0;
    for (unsigned r = r0; r < r1; r++) {
      const unsigned len = A->offs[r + 1] - A->offs[r];
      if (len > maxlen)
        maxlen = len;
    }
    unsigned L = 1;
    while (L < 64 && (unsigned long long)nr * (L * 2) <= 256)
      L <<= 1;
    unsigned want = 1;
    while (want < 64 && want * 16 < maxlen)
      want <<= 1;
    lanes[k] = (unsigned char)(L > want ? L : want);
  }
}

/* ------------------------------------------------------------------------ */
/* Reordering.  The reference's cuSOLVER     */
/* backend computes a host permutation Q (src/cusparse.c:67-85), applies it    */
/* symmetrically (:87-97), permutes the right-hand side (:177) and un-permutes */
/* x (:204).  Same semantics here, with our own reverse Cuthill-McKee: it      */
/* shrinks the band, which is what the SpMV's x-window-in-L2 design and the    */
/* neighbour-only halo exchange feed on.                                       */
/* ------------------------------------------------------------------------ */

struct deg_node {
  unsigned deg, id;
};

static int deg_cmp(const void *a, const void *b) {
  const struct deg_node *x = (const struct deg_node *)a, *y = (const struct deg_node *)b;
  if (x->deg != y->deg)
    return x->deg < y->deg ? -1 : 1;
  return x->id < y->id ? -1 : (x->id > y->id);
}

/* BFS from `start` over the unvisited part; appends to order[*pos...]; returns
 * the last level's first node and the eccentricity through *depth */
static unsigned bfs_component(const struct csr *S, unsigned start, unsigned char *seen,
                              unsigned *order, unsigned *pos, struct deg_node *tmp,
                              unsigned *depth) {
  unsigned head = *pos, level_end, last_level_first = start, d = 0;
  order[(*pos)++] = start;
  seen[start] = 1;
  level_end = *pos;
  while (head < *pos) {
    if (head == level_end) {
      level_end = *pos;
      last_level_first = order[head];
      d++;
    }
    const unsigned u = order[head++];
    unsigned k = 0;
    for (unsigned j = S->offs[u]; j < S->offs[u + 1]; j++) {
      const unsigned v = S->cols[j] - S->base;
      if (v < S->nrows && !seen[v]) {
        seen[v] = 1;
        tmp[k].id = v, tmp[k].deg = S->offs[v + 1] - S->offs[v], k++;
      }
    }
    if (k > 1)
      qsort(tmp, k, sizeof *tmp, deg_cmp); /* Cuthill-McKee: by increasing degree */
    for (unsigned i = 0; i < k; i++)
      order[(*pos)++] = tmp[i].id;
  }
  *depth = d;
  return last_level_first;
}

/*
 * Reverse Cuthill-McKee of the pattern of S (assumed structurally symmetric,
 * as the operator built by lsb_csr_symmetrize_upper is).  perm[new] = old.
 * Every connected component starts from a pseudo-peripheral node (George-Liu:
 * repeat BFS from the far end while the eccentricity grows).
 */
int lsb_csr_rcm(const struct csr *S, unsigned *perm) {
  const unsigned n = S->nrows;
  unsigned char *seen = (unsigned char *)calloc(n ? n : 1, 1);
  unsigned char *scratch_seen = (unsigned char *)malloc(n ? n : 1);
  unsigned *scratch = (unsigned *)malloc((size_t)(n ? n : 1) * sizeof(unsigned));
  unsigned maxdeg = 1;
  for (unsigned i = 0; i < n; i++)
    if (S->offs[i + 1] - S->offs[i] > maxdeg)
      maxdeg = S->offs[i + 1] - S->offs[i];
  struct deg_node *tmp = (struct deg_node *)malloc((size_t)maxdeg * sizeof *tmp);
  if (!seen || !scratch_seen || !scratch || !tmp)
    return 2;
  unsigned pos = 0;
  for (unsigned root = 0; root < n; root++) {
    if (seen[root])
      continue;
    /* pseudo-peripheral start inside this component */
    unsigned start = root, best_depth = 0;
    for (int trial = 0; trial < 8; trial++) {
      memcpy(scratch_seen, seen, n);
      unsigned p = 0, depth = 0;
      const unsigned far = bfs_component(S, start, scratch_seen, scratch, &p, tmp, &depth);
      if (trial > 0 && depth <= best_depth)
        break;
      best_depth = depth;
      /* among the last level prefer the smallest degree: `far` is the first
       * node of that level, which the degree sort put first among its peers */
      start = far;
    }
    unsigned depth;
    bfs_component(S, start, seen, perm, &pos, tmp, &depth);
  }
  for (unsigned i = 0; i < n / 2; i++) { /* reverse */
    const unsigned t = perm[i];
    perm[i] = perm[n - 1 - i], perm[n - 1 - i] = t;
  }
  free(seen), free(scratch_seen), free(scratch), free(tmp);
  return pos == n ? 0 : 2;
}

/* B = P S P^T with perm[new] = old: row new of B is row perm[new] of S with
 * columns renumbered by the inverse permutation and re-sorted.  0-based out. */
struct csr *lsb_csr_permute_sym(const struct csr *S, const unsigned *perm) {
  const unsigned n = S->nrows;
  unsigned *inv = (unsigned *)malloc((size_t)(n ? n : 1) * sizeof(unsigned));
  for (unsigned i = 0; i < n; i++)
    inv[perm[i]] = i;
  struct csr *B = csr_alloc(n, S->offs[n]);
  unsigned long long acc = 0;
  for (unsigned i = 0; i < n; i++) {
    B->offs[i] = (unsigned)acc;
    acc += S->offs[perm[i] + 1] - S->offs[perm[i]];
  }
  B->offs[n] = (unsigned)acc;
  unsigned maxlen = 1;
  for (unsigned i = 0; i < n; i++)
    if (B->offs[i + 1] - B->offs[i] > maxlen)
      maxlen = B->offs[i + 1] - B->offs[i];
#pragma omp parallel
  {
    struct deg_node *row = (struct deg_node *)malloc((size_t)maxlen * sizeof *row);
#pragma omp for schedule(static)
    for (long long ii = 0; ii < (long long)n; ii++) {
      const unsigned i = (unsigned)ii, src = perm[i], j0 = S->offs[src];
      const unsigned len = S->offs[src + 1] - j0, d0 = B->offs[i];
      for (unsigned k = 0; k < len; k++)
        row[k].deg = inv[S->cols[j0 + k] - S->base], row[k].id = k; /* (new col, slot) */
      qsort(row, len, sizeof *row, deg_cmp);
      for (unsigned k = 0; k < len; k++)
        B->cols[d0 + k] = row[k].deg, B->vals[d0 + k] = S->vals[j0 + row[k].id];
    }
    free(row);
  }
  free(inv);
  return B;
}

/* max |i - j| over the stored entries */
unsigned lsb_csr_bandwidth(const struct csr *S) {
  unsigned bw = 0;
  for (unsigned i = 0; i < S->nrows; i++)
    for (unsigned j = S->offs[i]; j < S->offs[i + 1]; j++) {
      const unsigned c = S->cols[j] - S->base, d = c > i ? c - i : i - c;
      if (d > bw)
        bw = d;
    }
  return bw;
}

/* ------------------------------------------------------------------------ */
/* Column-panel form of a CSR for operators whose rows scatter over the whole */
/* column range (power-law / unordered matrices).  The gather x[col] of such  */
/* a row touches one cache line per non-zero; with x far larger than an XCD's */
/* 4 MiB L2 every one of them comes from beyond L2.  Cutting the columns into */
/* panels of `width` (x panel = 8*width bytes, L2-resident) and sweeping the   */
/* matrix panel by panel turns them into L2 hits at the price of re-visiting   */
/* y once per (row, panel) pair.  Layout: one CSR whose "rows" are those       */
/* pairs, panel-major, rows ascending inside a panel:                          */
/*   pair_row[npairs]   original row of each pair                              */
/*   offs[npairs+1]     into cols/vals (a permuted copy of the operator)       */
/*   pair_begin[np+1]   first pair of each panel                               */
/* ------------------------------------------------------------------------ */
struct lsb_panel_csr *lsb_csr_panelize(const struct csr *A, unsigned width) {
  if (!A || width == 0)
    return NULL;
  const unsigned n = A->nrows, base = A->base;
  unsigned lo, hi;
  lsb_csr_col_hull(A, &lo, &hi);
  const unsigned np = hi ? (hi - 1) / width + 1 : 1;
  unsigned long long *pairs = lsb_calloc(unsigned long long, (size_t)np + 1);
  unsigned long long *nnzp = lsb_calloc(unsigned long long, (size_t)np + 1);
  for (unsigned i = 0; i < n; i++) {
    unsigned last = 0xFFFFFFFFu;
    for (unsigned j = A->offs[i]; j < A->offs[i + 1]; j++) {
      const unsigned p = (A->cols[j] - base) / width;
      nnzp[p + 1]++;
      if (p != last)
        pairs[p + 1]++, last = p;
    }
  }
  for (unsigned p = 0; p < np; p++)
    pairs[p + 1] += pairs[p], nnzp[p + 1] += nnzp[p];
  const unsigned long long npairs = pairs[np], nnz = nnzp[np];
  if (npairs > 0x7FFFFFFEull || nnz > 0x7FFFFFFEull)
    errx(EXIT_FAILURE, "panel CSR too large for 32-bit indices");
  struct lsb_panel_csr *P = lsb_calloc(struct lsb_panel_csr, 1);
  P->npanels = np, P->width = width, P->npairs = (unsigned)npairs, P->nrows = n;
  P->pair_begin = lsb_calloc(unsigned, (size_t)np + 1);
  P->pair_row = (unsigned *)malloc((size_t)(npairs ? npairs : 1) * sizeof(unsigned));
  P->offs = (unsigned *)malloc(((size_t)npairs + 1) * sizeof(unsigned));
  P->cols = (unsigned *)malloc((size_t)(nnz ? nnz : 1) * sizeof(unsigned));
  P->vals = (double *)malloc((size_t)(nnz ? nnz : 1) * sizeof(double));
  if (!P->pair_row || !P->offs || !P->cols || !P->vals)
    errx(EXIT_FAILURE, "out of host memory for the panel CSR");
  unsigned *pc = (unsigned *)malloc((size_t)np * sizeof(unsigned)); /* pair cursor */
  unsigned *zc = (unsigned *)malloc((size_t)np * sizeof(unsigned)); /* nnz cursor  */
  for (unsigned p = 0; p < np; p++)
    P->pair_begin[p] = pc[p] = (unsigned)pairs[p], zc[p] = (unsigned)nnzp[p];
  P->pair_begin[np] = (unsigned)npairs;
  for (unsigned i = 0; i < n; i++) {
    unsigned last = 0xFFFFFFFFu;
    for (unsigned j = A->offs[i]; j < A->offs[i + 1]; j++) {
      const unsigned c = A->cols[j] - base, p = c / width;
      if (p != last) {
        P->pair_row[pc[p]] = i;
        P->offs[pc[p]] = zc[p];
        pc[p]++, last = p;
      }
      P->cols[zc[p]] = c, P->vals[zc[p]] = A->vals[j];
      zc[p]++;
    }
  }
  P->offs[npairs] = (unsigned)nnz;
  free(pairs), free(nnzp), free(pc), free(zc);
  return P;
}

void lsb_panel_csr_free(struct lsb_panel_csr *P) {
  if (!P)
    return;
  free(P->pair_begin), free(P->pair_row), free(P->offs), free(P->cols), free(P->vals);
  free(P);
}

/* ------------------------------------------------------------------------ */
/* Binned form (LSB_SPMV_BINNED): entries bin-major (bin = col / width), rows   */
/* ascending inside a bin, column order of a row kept; chunks = whole (row,     */
/* bin) runs, at most LSB_BIN_CHUNK entries unless one run alone is longer.     */
/* ------------------------------------------------------------------------ */
struct lsb_binned *lsb_csr_binize(const struct csr *A, unsigned width) {
  if (!A || width == 0 || A->nrows == 0)
    return NULL;
  const unsigned n = A->nrows, base = A->base;
  const unsigned long long nnz = A->offs[n];
  unsigned lo, hi;
  lsb_csr_col_hull(A, &lo, &hi);
  const unsigned nb = hi ? (hi - 1) / width + 1 : 1;
  if (nnz > 0x7FFFFFFEull)
    return NULL;
  unsigned long long *cnt = lsb_calloc(unsigned long long, (size_t)nb + 1);
  for (unsigned long long j = 0; j < nnz; j++)
    cnt[(A->cols[j] - base) / width + 1]++;
  for (unsigned b = 0; b < nb; b++)
    cnt[b + 1] += cnt[b];
  struct lsb_binned *B = lsb_calloc(struct lsb_binned, 1);
  B->nbins = nb, B->width = width, B->nrows = n, B->nnz = nnz;
  const unsigned CAP = LSB_BIN_CHUNK; /* (1024 / 1536 measured no better: DESIGN.md section 4) */
  B->chunk_cap = CAP;
  B->rows = (unsigned *)malloc((size_t)(nnz ? nnz : 1) * sizeof(unsigned));
  B->cols = (unsigned *)malloc((size_t)(nnz ? nnz : 1) * sizeof(unsigned));
  B->vals = (double *)malloc((size_t)(nnz ? nnz : 1) * sizeof(double));
  unsigned *cur = (unsigned *)malloc((size_t)nb * sizeof(unsigned));
  if (!B->rows || !B->cols || !B->vals || !cur)
    errx(EXIT_FAILURE, "out of host memory for the binned operator");
  for (unsigned b = 0; b < nb; b++)
    cur[b] = (unsigned)cnt[b];
  for (unsigned i = 0; i < n; i++) /* row-major sweep: rows ascend inside every bin */
    for (unsigned j = A->offs[i]; j < A->offs[i + 1]; j++) {
      const unsigned c = A->cols[j] - base, e = cur[c / width]++;
      B->rows[e] = i, B->cols[e] = c, B->vals[e] = A->vals[j];
    }
  free(cur);
  /* chunks */
  size_t cap = (size_t)(nnz / CAP) * 2 + 2 * (size_t)nb + 16, nc = 0;
  B->chunk_begin = (unsigned *)malloc((cap + 1) * sizeof(unsigned));
  B->bin_chunk = lsb_calloc(unsigned, (size_t)nb + 1);
  for (unsigned b = 0; b < nb; b++) {
    const unsigned e0 = (unsigned)cnt[b], e1 = (unsigned)cnt[b + 1];
    B->bin_chunk[b] = (unsigned)nc;
    unsigned e = e0;
    while (e < e1) {
      const unsigned start = e;
      while (e < e1) {
        unsigned re = e + 1;
        while (re < e1 && B->rows[re] == B->rows[e])
          re++;
        if (re - start <= CAP || e == start) {
          e = re;
          if (e - start >= CAP)
            break;
        } else
          break;
      }
      if (nc + 2 > cap) {
        cap *= 2;
        B->chunk_begin = (unsigned *)realloc(B->chunk_begin, (cap + 1) * sizeof(unsigned));
        if (!B->chunk_begin)
          errx(EXIT_FAILURE, "out of host memory for the binned operator");
      }
      B->chunk_begin[nc++] = start;
    }
  }
  B->bin_chunk[nb] = (unsigned)nc;
  B->chunk_begin[nc] = (unsigned)nnz;
  B->nchunks = (unsigned)nc;
  free(cnt);
  return B;
}

void lsb_binned_free(struct lsb_binned *B) {
  if (!B)
    return;
  free(B->bin_chunk), free(B->chunk_begin), free(B->rows), free(B->cols), free(B->vals);
  free(B);
}

/* ------------------------------------------------------------------------ */
/* Two-phase form (LSB_SPMV_TWOPHASE): see include/lsbench_hip.h.             */
/* ------------------------------------------------------------------------ */
/* entries of a phase-1 work item (a multiple of 64).  Measured on the
 * 8 M-row power-law operator, phase 1: 131072 -> 1322 us, 32768 -> 1247, 8192 -> 1201 (the
 * window of x is loaded once per item -- 64 KB out of L2 -- but the last workgroups of the
 * launch finish together) */
#define PB_ITEM_DEFAULT 8192u
struct lsb_pb *lsb_csr_pbize2(const struct csr *A, unsigned C, unsigned R) {
  if (!A || A->nrows == 0)
    return NULL;
  C = C ? C : LSB_PB_COLS, R = R ? R : LSB_PB_ROWS;
  const unsigned PB_ITEM = PB_ITEM_DEFAULT;
  if (C > 16384 || R > 4096 || C % 64 || R % 64)
    errx(EXIT_FAILURE, "two-phase operator: %u columns x %u rows per piece is not a usable tiling", C, R);
  const unsigned n = A->nrows, base = A->base;
  const unsigned long long nnz = A->offs[n];
  if (nnz == 0 || nnz > 0x7FFFFFF0ull)
    return NULL;
  unsigned lo, hi;
  lsb_csr_col_hull(A, &lo, &hi);
  lo -= lo % C; /* windows start on a multiple of the chunk width */
  const unsigned nch = (hi - lo + C - 1) / C;
  const unsigned nb = (n + R - 1) / R;
  const unsigned long long nbuck = (unsigned long long)nch * nb;
  if (nbuck > (1ull << 30))
    return NULL;
  /* counting sort by (chunk, bin); the row-major sweep keeps rows, then columns,
   * ascending inside a bucket.  cnt[k]: first entry of bucket k in UNPADDED
   * phase-1 order. */
  unsigned *cnt = lsb_calloc(unsigned, (size_t)nbuck + 1);
  if (!cnt)
    errx(EXIT_FAILURE, "out of host memory for the two-phase operator");
  for (unsigned i = 0; i < n; i++) {
    const unsigned b = i / R;
    for (unsigned j = A->offs[i]; j < A->offs[i + 1]; j++)
      cnt[(size_t)((A->cols[j] - base - lo) / C) * nb + b + 1]++;
  }
  for (unsigned long long k = 0; k < nbuck; k++)
    cnt[k + 1] += cnt[k];
  /* every chunk's entries start on a multiple of 64: pad[c] = what was inserted
   * in front of chunk c */
  unsigned *pad = lsb_calloc(unsigned, (size_t)nch + 1);
  for (unsigned c = 0; c < nch; c++) {
    const unsigned long long len = cnt[(size_t)(c + 1) * nb] - cnt[(size_t)c * nb];
    pad[c + 1] = pad[c] + (unsigned)((64 - len % 64) % 64);
  }
  const unsigned long long nent = nnz + pad[nch];
  if (nent > 0xFFFFFFC0ull)
    return NULL;
  struct lsb_pb *P = lsb_calloc(struct lsb_pb, 1);
  P->nrows = n, P->ncols_lo = lo, P->nchunks = nch, P->nbins = nb, P->nnz = nnz, P->nent = nent;
  P->cols = C, P->rows = R;
  P->vals = (double *)calloc((size_t)nent, sizeof(double));
  P->colw = (unsigned short *)calloc((size_t)nent, sizeof(unsigned short));
  P->grp_first = lsb_calloc(unsigned, (size_t)(nent / 64) + 1);
  P->grp_mask = lsb_calloc(unsigned long long, (size_t)(nent / 64) + 1);
  P->bin_ptr = lsb_calloc(unsigned, (size_t)nb + 1);
  P->roww = (unsigned short *)malloc((size_t)nnz * sizeof(unsigned short));
  unsigned short *rw = (unsigned short *)malloc((size_t)nent * sizeof(unsigned short));
  if (!P->vals || !P->colw || !P->roww || !rw)
    errx(EXIT_FAILURE, "out of host memory for the two-phase operator");
  /* phase-1 work items: slices of a chunk, starting on multiples of 64 */
  {
    size_t cap = (size_t)(nent / PB_ITEM) + nch + 8, ni = 0;
    P->item = (unsigned *)malloc(cap * 3 * sizeof(unsigned));
    for (unsigned c = 0; c < nch; c++) {
      const unsigned e0 = cnt[(size_t)c * nb] + pad[c], e1 = cnt[(size_t)(c + 1) * nb] + pad[c];
      for (unsigned e = e0; e < e1; e += PB_ITEM) {
        P->item[3 * ni] = c, P->item[3 * ni + 1] = e;
        P->item[3 * ni + 2] = e1 - e > PB_ITEM ? e + PB_ITEM : e1;
        ni++;
      }
    }
    P->nitems = (unsigned)ni;
  }
  /* slots: bin-major, chunk-minor.  slot0[k]: first slot of bucket k */
  unsigned *slot0 = (unsigned *)malloc((size_t)nbuck * sizeof(unsigned));
  if (!slot0)
    errx(EXIT_FAILURE, "out of host memory for the two-phase operator");
  {
    unsigned long long tot = 0;
    for (unsigned b = 0; b < nb; b++) {
      P->bin_ptr[b] = (unsigned)tot;
      for (unsigned c = 0; c < nch; c++) {
        const size_t k = (size_t)c * nb + b;
        slot0[k] = (unsigned)tot;
        tot += cnt[k + 1] - cnt[k];
      }
    }
    P->bin_ptr[nb] = (unsigned)tot;
  }
  /* pieces = non-empty buckets in phase-1 order; the group words */
  {
    size_t np = 0;
    for (unsigned long long k = 0; k < nbuck; k++)
      np += cnt[k + 1] > cnt[k];
    P->npieces = (unsigned)np;
    P->delta = lsb_calloc(unsigned, np + 1);
    np = 0;
    for (unsigned c = 0; c < nch; c++)
      for (unsigned b = 0; b < nb; b++) {
        const size_t k = (size_t)c * nb + b;
        const unsigned len = cnt[k + 1] - cnt[k];
        if (!len)
          continue;
        const unsigned e0 = cnt[k] + pad[c];
        P->delta[np] = slot0[k] - e0; /* mod 2^32 */
        if (e0 % 64)
          P->grp_mask[e0 / 64] |= 1ull << (e0 % 64);
        for (unsigned g = (e0 + 63) / 64; g <= (e0 + len - 1) / 64; g++)
          P->grp_first[g] = (unsigned)np;
        np++;
      }
  }
  /* scatter into (chunk, bin, row, col) order; cur[k] runs through bucket k */
  {
    unsigned *cur = (unsigned *)malloc((size_t)nbuck * sizeof(unsigned));
    if (!cur)
      errx(EXIT_FAILURE, "out of host memory for the two-phase operator");
    for (unsigned c = 0; c < nch; c++)
      for (unsigned b = 0; b < nb; b++)
        cur[(size_t)c * nb + b] = cnt[(size_t)c * nb + b] + pad[c];
    for (unsigned i = 0; i < n; i++) {
      const unsigned b = i / R;
      for (unsigned j = A->offs[i]; j < A->offs[i + 1]; j++) {
        const unsigned c = A->cols[j] - base - lo, ch = c / C;
        const unsigned e = cur[(size_t)ch * nb + b]++;
        P->vals[e] = A->vals[j];
        P->colw[e] = (unsigned short)(c % C);
        rw[e] = (unsigned short)(i % R);
      }
    }
    free(cur);
  }
  /* the rows in slot order: a piece is contiguous in both orders */
#pragma omp parallel for schedule(static)
  for (long long cc = 0; cc < (long long)nch; cc++)
    for (unsigned b = 0; b < nb; b++) {
      const size_t k = (size_t)cc * nb + b;
      const unsigned len = cnt[k + 1] - cnt[k];
      if (len)
        memcpy(P->roww + slot0[k], rw + cnt[k] + pad[cc], (size_t)len * sizeof(unsigned short));
    }
  free(slot0), free(rw), free(cnt), free(pad);
  return P;
}

/* Every index the two kernels of the two-phase SpMV form (hip_pb.hip), checked on the host
 * against the arrays as they are ALLOCATED -- the bounds those kernels rely on, as assertions:
 *   phase 1: work items start on a multiple of 64 inside [0, nent), name a chunk < nchunks
 *            and end inside nent (its clamped loads reach e1 - 1); the piece of every group
 *            of 64 entries, grp_first[g] + popcount(grp_mask[g]), stays below npieces; the
 *            stores of a piece, entry index + delta[piece] (mod 2^32), land in [0, nnz);
 *   phase 2: bin_ptr is non-decreasing, ends at nnz, and the PAIR loads of a bin -- from the
 *            even slot at or below its first to the even slot at or below its last, two slots
 *            each -- stay inside prod_len / roww_len slots (the device arrays are allocated two
 *            slots longer than they are used for exactly this).
 * deep != 0 also walks all entries: colw < cols, roww < rows.
 * Returns 0 when everything holds, else a non-zero code and the violated rule in `why`.
 * Round 2's two GPU-side failures were experiments that broke exactly
 * these rules: stores at the PADDED entry index, pair loads past the last slot. */
int lsb_pb_check(const struct lsb_pb *P, unsigned long long prod_len, unsigned long long roww_len, int deep,
                 char *why, size_t whylen) {
#define PB_FAIL(code, ...)                                                                     \
  do {                                                                                         \
    if (why && whylen)                                                                         \
      snprintf(why, whylen, __VA_ARGS__);                                                      \
    return code;                                                                               \
  } while (0)
  if (!P)
    PB_FAIL(1, "no layout");
  if (P->nent % 64 || P->nent < P->nnz || P->nent > 0xFFFFFFC0ull)
    PB_FAIL(2, "nent = %llu is not a multiple of 64 in [nnz, 2^32)", P->nent);
  if (prod_len < P->nnz + 1 || roww_len < P->nnz + 1)
    PB_FAIL(3, "product / row arrays of %llu / %llu slots: phase 2's pair loads need nnz + 1 = %llu", prod_len,
            roww_len, P->nnz + 1);
  unsigned long long covered = 0;
  for (unsigned it = 0; it < P->nitems; it++) {
    const unsigned c = P->item[3 * it], e0 = P->item[3 * it + 1], e1 = P->item[3 * it + 2];
    if (c >= P->nchunks || e0 % 64 || e1 <= e0 || e1 > P->nent)
      PB_FAIL(4, "work item %u = {chunk %u, [%u, %u)} outside %u chunks / %llu entries or not 64-aligned", it, c,
              e0, e1, P->nchunks, P->nent);
    covered += e1 - e0;
  }
  if (covered != P->nnz)
    PB_FAIL(5, "work items cover %llu entries, the operator has %llu", covered, P->nnz);
  const unsigned long long ngrp = P->nent / 64;
  for (unsigned long long g = 0; g < ngrp; g++)
    if ((unsigned long long)P->grp_first[g] + (unsigned)__builtin_popcountll(P->grp_mask[g]) >= P->npieces + (P->npieces == 0))
      PB_FAIL(6, "group %llu names piece %u + %d of %u", g, P->grp_first[g], __builtin_popcountll(P->grp_mask[g]),
              P->npieces);
  /* stores: walk the pieces in phase-1 order through the group words, the way the kernel does */
  for (unsigned it = 0; it < P->nitems; it++) {
    const unsigned e0 = P->item[3 * it + 1], e1 = P->item[3 * it + 2];
    for (unsigned e = e0; e < e1;) { /* e = first entry of a piece (or of the item inside one) */
      const unsigned lane = e % 64;
      const unsigned long long le = lane == 63 ? ~0ull : (2ull << lane) - 1ull;
      const unsigned piece = P->grp_first[e / 64] + (unsigned)__builtin_popcountll(P->grp_mask[e / 64] & le);
      /* the piece runs to the next set bit / next group whose first piece differs / the item's end */
      unsigned f = e + 1;
      while (f < e1) {
        const unsigned fl = f % 64;
        const unsigned long long fle = fl == 63 ? ~0ull : (2ull << fl) - 1ull;
        if (P->grp_first[f / 64] + (unsigned)__builtin_popcountll(P->grp_mask[f / 64] & fle) != piece)
          break;
        f = deep ? f + 1 : (fl == 63 || !(P->grp_mask[f / 64] >> (fl + 1)) ? (f | 63u) + 1 : f + 1);
      }
      if (f > e1)
        f = e1;
      const unsigned first = e + P->delta[piece], last = (f - 1) + P->delta[piece]; /* mod 2^32, like the kernel */
      if (first >= P->nnz || last >= P->nnz || last < first)
        PB_FAIL(7, "piece %u: entries [%u, %u) store to slots [%u, %u] of %llu", piece, e, f, first, last, P->nnz);
      e = f;
    }
  }
  if (P->bin_ptr[P->nbins] != P->nnz)
    PB_FAIL(8, "bin_ptr ends at %u, not at nnz = %llu", P->bin_ptr[P->nbins], P->nnz);
  for (unsigned b = 0; b < P->nbins; b++) {
    const unsigned s0 = P->bin_ptr[b], s1 = P->bin_ptr[b + 1];
    if (s1 < s0)
      PB_FAIL(9, "bin %u: slots [%u, %u)", b, s0, s1);
    if (s1 > s0) { /* pair loads at even indices a0 .. last, two slots each */
      const unsigned long long last = (s1 - 1u) & ~1u;
      if (last + 1 >= prod_len || last + 1 >= roww_len)
        PB_FAIL(10, "bin %u: the pair load at slot %llu reaches past %llu / %llu allocated slots", b, last,
                prod_len, roww_len);
    }
  }
  if (deep) {
    for (unsigned it = 0; it < P->nitems; it++)
      for (unsigned e = P->item[3 * it + 1]; e < P->item[3 * it + 2]; e++)
        if (P->colw[e] >= P->cols)
          PB_FAIL(11, "entry %u: column offset %u in a chunk of %u", e, P->colw[e], P->cols);
    for (unsigned long long k = 0; k < P->nnz; k++)
      if (P->roww[k] >= P->rows)
        PB_FAIL(12, "slot %llu: row offset %u in a bin of %u", k, P->roww[k], P->rows);
  }
  return 0;
#undef PB_FAIL
}

struct lsb_pb *lsb_csr_pbize(const struct csr *A) {
  const char *ec = getenv("LSBENCH_HIP_PB_COLS"), *er = getenv("LSBENCH_HIP_PB_ROWS");
  return lsb_csr_pbize2(A, ec ? (unsigned)atoi(ec) : 0, er ? (unsigned)atoi(er) : 0);
}

void lsb_pb_free(struct lsb_pb *P) {
  if (!P)
    return;
  free(P->vals), free(P->colw), free(P->grp_first), free(P->grp_mask), free(P->delta);
  free(P->item), free(P->bin_ptr), free(P->roww);
  free(P);
}

/* mean |col - row| over a sample of the rows: how far the gather of a row
 * strays from the diagonal (banded: ~bandwidth; scattered: ~n/3) */
double lsb_csr_mean_scatter(const struct csr *A, unsigned row_begin) {
  const unsigned n = A->nrows, step = n > 4096 ? n / 4096 : 1;
  double sum = 0.0;
  unsigned long long cnt = 0;
  for (unsigned i = 0; i < n; i += step)
    for (unsigned j = A->offs[i]; j < A->offs[i + 1]; j++) {
      const double c = (double)(A->cols[j] - A->base), r = (double)i + row_begin;
      sum += c > r ? c - r : r - c;
      cnt++;
    }
  return cnt ? sum / (double)cnt : 0.0;
}

/* ------------------------------------------------------------------------ */
/* FSAI pattern (LSB_PRECOND_FSAI; hip_precond.c, hip_fsai.hip).  No reference */
/* counterpart: the reference's own "expensive set-up once" is CHOLMOD's       */
/* factorisation in csr_init (src/cholmod-impl.h:25-26).                       */
/* ------------------------------------------------------------------------ */
struct lsb_fsai_pattern *lsb_csr_fsai_pattern(const struct csr *S, int power, unsigned cap) {
  if (!S || S->nrows == 0 || power < 1 || power > 3 || cap == 0)
    return NULL;
  const unsigned n = S->nrows, base = S->base;
  struct lsb_fsai_pattern *P = lsb_calloc(struct lsb_fsai_pattern, 1);
  P->n = n, P->cap = cap;
  P->offs = lsb_calloc(unsigned, (size_t)n + 1);
  /* two passes (count, fill) of the same row routine; per thread: a stamp array to make
   * the union, a list of what was reached */
  unsigned *len = lsb_calloc(unsigned, (size_t)n);
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1) {
      unsigned long long acc = 0;
      for (unsigned i = 0; i < n; i++)
        P->offs[i] = (unsigned)acc, acc += len[i];
      if (acc > 0xFFFFFFF0ull)
        errx(EXIT_FAILURE, "FSAI pattern with %llu entries exceeds 32-bit offsets", acc);
      P->offs[n] = (unsigned)acc, P->nnz = acc;
      P->cols = (unsigned *)malloc((size_t)(acc ? acc : 1) * sizeof(unsigned));
      if (!P->cols)
        errx(EXIT_FAILURE, "out of host memory for the FSAI pattern");
    }
    /* 8 n bytes per thread (stamps + list): the team is capped so that they stay below 2 GiB in all
     * (10 M rows on a 256-thread host would otherwise ask for 20 GB) */
    int team = omp_get_max_threads();
    while (team > 1 && (unsigned long long)team * n * 8ull > (2ull << 30))
      team--;
#pragma omp parallel num_threads(team)
    {
      unsigned *stamp = (unsigned *)calloc((size_t)n, sizeof(unsigned)); /* stamp[j] == i + 1: j reached from row i */
      unsigned *list = (unsigned *)malloc((size_t)n * sizeof(unsigned));
      if (!stamp || !list)
        errx(EXIT_FAILURE, "out of host memory for the FSAI pattern's work arrays (%u rows, %d threads)", n, team);
#pragma omp for schedule(dynamic, 256)
      for (long long ii = 0; ii < (long long)n; ii++) {
        const unsigned i = (unsigned)ii;
        unsigned cnt = 0, lo = 0;
        stamp[i] = i + 1, list[cnt++] = i;
        for (int step = 0; step < power; step++) { /* one more hop from everything reached so far */
          const unsigned hi = cnt;
          for (unsigned a = lo; a < hi; a++) {
            const unsigned j = list[a];
            for (unsigned e = S->offs[j]; e < S->offs[j + 1]; e++) {
              const unsigned c = S->cols[e] - base;
              if (c < n && stamp[c] != i + 1)
                stamp[c] = i + 1, list[cnt++] = c;
            }
          }
          lo = hi;
        }
        /* keep j <= i, the `cap` largest of them */
        unsigned m = 0;
        for (unsigned a = 0; a < cnt; a++)
          if (list[a] <= i)
            list[m++] = list[a];
        /* (insertion sort would be quadratic on long rows: qsort-free selection by a counting
         * pass over the small index window is overkill too -- sort ascending with a simple
         * shell sort, rows are at most a few thousand long) */
        for (unsigned gap = m / 2; gap > 0; gap /= 2)
          for (unsigned a = gap; a < m; a++) {
            const unsigned v = list[a];
            unsigned b = a;
            for (; b >= gap && list[b - gap] > v; b -= gap)
              list[b] = list[b - gap];
            list[b] = v;
          }
        const unsigned keep = m > cap ? cap : m;
        if (pass == 0)
          len[i] = keep;
        else
          memcpy(P->cols + P->offs[i], list + (m - keep), (size_t)keep * sizeof(unsigned));
      }
      free(stamp), free(list);
    }
  }
  free(len);
  return P;
}

void lsb_fsai_pattern_free(struct lsb_fsai_pattern *P) {
  if (!P)
    return;
  free(P->offs), free(P->cols), free(P);
}

/* ------------------------------------------------------------------------ */
/* Sliced-ELL copy (LSB_SPMV_SELL).  No reference counterpart: a device      */
/* layout of the same operator.                        */
/* ------------------------------------------------------------------------ */
unsigned long long lsb_csr_sell_stored(const struct csr *A) {
  unsigned long long tot = 0;
  for (unsigned r0 = 0; r0 < A->nrows; r0 += LSB_SELL_ROWS) {
    unsigned len = 0;
    const unsigned r1 = r0 + LSB_SELL_ROWS < A->nrows ? r0 + LSB_SELL_ROWS : A->nrows;
    for (unsigned r = r0; r < r1; r++)
      if (A->offs[r + 1] - A->offs[r] > len)
        len = A->offs[r + 1] - A->offs[r];
    tot += (unsigned long long)len * LSB_SELL_ROWS;
  }
  return tot;
}

struct lsb_sell *lsb_csr_sellize(const struct csr *A) {
  if (!A)
    return NULL;
  const unsigned long long stored = lsb_csr_sell_stored(A);
  if (stored > 0xFFFFFF00ull)
    return NULL;
  struct lsb_sell *S = lsb_calloc(struct lsb_sell, 1);
  const unsigned n = A->nrows, ns = (n + LSB_SELL_ROWS - 1) / LSB_SELL_ROWS;
  S->nrows = n, S->nslice = ns, S->stored = stored;
  S->sptr = lsb_calloc(unsigned, (size_t)ns + 1);
  S->cols = (int *)calloc((size_t)stored + LSB_SELL_ROWS, sizeof(int));
  S->vals = (double *)calloc((size_t)stored + LSB_SELL_ROWS, sizeof(double));
  if (!S->cols || !S->vals)
    errx(EXIT_FAILURE, "lsb_csr_sellize: out of memory");
  for (unsigned sl = 0; sl < ns; sl++) {
    unsigned len = 0;
    for (unsigned r = sl * LSB_SELL_ROWS; r < n && r < (sl + 1) * LSB_SELL_ROWS; r++)
      if (A->offs[r + 1] - A->offs[r] > len)
        len = A->offs[r + 1] - A->offs[r];
    S->sptr[sl + 1] = S->sptr[sl] + len * LSB_SELL_ROWS;
  }
#pragma omp parallel for schedule(static)
  for (unsigned sl = 0; sl < ns; sl++) {
    const unsigned len = (S->sptr[sl + 1] - S->sptr[sl]) / LSB_SELL_ROWS;
    for (unsigned l = 0; l < LSB_SELL_ROWS; l++) {
      const unsigned r = sl * LSB_SELL_ROWS + l;
      const unsigned a = r < n ? A->offs[r] : 0, b = r < n ? A->offs[r + 1] : 0;
      const int padc = b > a ? (int)(A->cols[b - 1] - A->base) : 0;
      for (unsigned j = 0; j < len; j++) {
        const size_t at = (size_t)S->sptr[sl] + (size_t)j * LSB_SELL_ROWS + l;
        if (a + j < b)
          S->cols[at] = (int)(A->cols[a + j] - A->base), S->vals[at] = A->vals[a + j];
        else
          S->cols[at] = padc, S->vals[at] = 0.0;
      }
    }
  }
  return S;
}

/* Slot bases of one slice: every entry (as delta = column - global row) must
 * find, in column order, a slot whose base is within +-32767 of it; a new base
 * is inserted where none is.  Returns the number of slots, 0 on overflow. */
#define SELL16_MAX_SLOTS 255
#define SELL16_REACH 32767L
static unsigned sell16_sweep(const struct csr *A, unsigned row_begin, unsigned r0, unsigned r1,
                             long *B, long reach, unsigned limit) {
  unsigned nb = 0;
  for (unsigned r = r0; r < r1; r++) {
    unsigned j = 0;
    for (unsigned k = A->offs[r]; k < A->offs[r + 1]; k++) {
      const long e = (long)(A->cols[k] - A->base) - ((long)r + (long)row_begin);
      while (j < nb && B[j] < e - reach)
        j++;
      if (!(j < nb && B[j] <= e + reach)) {
        if (nb == limit)
          return 0;
        memmove(B + j + 1, B + j, (size_t)(nb - j) * sizeof(long));
        B[j] = e, nb++;
      }
      j++;
    }
  }
  return nb;
}

/* First choice: one slot per DIAGONAL of the slice (reach 0), as long as that
 * pads the slice by no more than a quarter -- on a structured grid it costs
 * nothing and no slot needs a code array.  Otherwise bands of +-32767.
 * *reach receives the rule the slots were built with (the fill uses it too). */
static unsigned sell16_slots(const struct csr *A, unsigned row_begin, unsigned r0, unsigned r1,
                             long *B, long *reach) {
  unsigned longest = 0;
  for (unsigned r = r0; r < r1; r++)
    if (A->offs[r + 1] - A->offs[r] > longest)
      longest = A->offs[r + 1] - A->offs[r];
  unsigned limit = longest + longest / 4 + 1;
  if (limit > SELL16_MAX_SLOTS)
    limit = SELL16_MAX_SLOTS;
  *reach = 0;
  unsigned nb = sell16_sweep(A, row_begin, r0, r1, B, 0, limit);
  if (nb == 0 && longest) {
    *reach = SELL16_REACH;
    nb = sell16_sweep(A, row_begin, r0, r1, B, SELL16_REACH, SELL16_MAX_SLOTS);
  }
  return nb;
}

struct lsb_sell *lsb_csr_sellize16(const struct csr *A, unsigned row_begin) {
  if (!A)
    return NULL;
  const unsigned n = A->nrows, ns = (n + LSB_SELL_ROWS - 1) / LSB_SELL_ROWS;
  unsigned *sptr = lsb_calloc(unsigned, (size_t)ns + 1);
  unsigned char *nslot = (unsigned char *)calloc(ns ? ns : 1, 1);
  int ok = 1;
#pragma omp parallel for schedule(static) reduction(& : ok)
  for (unsigned sl = 0; sl < ns; sl++) {
    long B[SELL16_MAX_SLOTS + 1], reach;
    const unsigned r0 = sl * LSB_SELL_ROWS, r1 = r0 + LSB_SELL_ROWS < n ? r0 + LSB_SELL_ROWS : n;
    const unsigned nb = sell16_slots(A, row_begin, r0, r1, B, &reach);
    unsigned has = 0;
    for (unsigned r = r0; r < r1 && !has; r++)
      has = A->offs[r + 1] > A->offs[r];
    if (nb == 0 && has)
      ok = 0;
    nslot[sl] = (unsigned char)nb;
  }
  unsigned long long stored = 0;
  for (unsigned sl = 0; sl < ns && ok; sl++) {
    stored += (unsigned long long)nslot[sl] * LSB_SELL_ROWS;
    if (stored > 0xFFFFFF00ull)
      ok = 0;
    sptr[sl + 1] = (unsigned)stored;
  }
  if (!ok) {
    free(sptr), free(nslot);
    return NULL;
  }
  struct lsb_sell *S = lsb_calloc(struct lsb_sell, 1);
  S->nrows = n, S->nslice = ns, S->stored = stored, S->sptr = sptr;
  S->codes = (short *)calloc((size_t)stored + LSB_SELL_ROWS, sizeof(short));
  S->vals = (double *)calloc((size_t)stored + LSB_SELL_ROWS, sizeof(double));
  S->sbase = (int *)calloc(2 * ((size_t)stored / LSB_SELL_ROWS + 1), sizeof(int));
  if (!S->codes || !S->vals || !S->sbase)
    errx(EXIT_FAILURE, "lsb_csr_sellize16: out of memory");
#pragma omp parallel for schedule(static) reduction(& : ok)
  for (unsigned sl = 0; sl < ns; sl++) {
    long B[SELL16_MAX_SLOTS + 1], reach;
    const unsigned r0 = sl * LSB_SELL_ROWS, r1 = r0 + LSB_SELL_ROWS < n ? r0 + LSB_SELL_ROWS : n;
    const unsigned nb = sell16_slots(A, row_begin, r0, r1, B, &reach);
    const size_t at0 = sptr[sl];
    for (unsigned j = 0; j < nb; j++)
      S->sbase[2 * (at0 / LSB_SELL_ROWS + j)] = (int)B[j];
    for (unsigned r = r0; r < r1; r++) {
      unsigned j = 0;
      for (unsigned k = A->offs[r]; k < A->offs[r + 1]; k++) {
        const long e = (long)(A->cols[k] - A->base) - ((long)r + (long)row_begin);
        while (j < nb && B[j] < e - reach)
          j++;
        if (!(j < nb && B[j] <= e + reach)) {
          ok = 0; /* cannot happen: the bases were built from these very rows */
          break;
        }
        const size_t at = at0 + (size_t)j * LSB_SELL_ROWS + (r - r0);
        S->codes[at] = (short)(e - B[j]), S->vals[at] = A->vals[k];
        j++;
      }
    }
  }
  free(nslot);
  if (!ok) {
    lsb_sell_free(S);
    return NULL;
  }
  /* A slot whose live entries (value != 0) all carry the same code needs no
   * code array: the code moves into the base.  On a structured grid that is
   * every slot.  The others keep their 128 codes, packed in slot order. */
  const size_t nslots = (size_t)stored / LSB_SELL_ROWS;
  size_t nc = 0;
  for (size_t q = 0; q < nslots; q++) {
    const short *c = S->codes + q * LSB_SELL_ROWS;
    const double *v = S->vals + q * LSB_SELL_ROWS;
    int have = 0, uniform = 1;
    short c0 = 0;
    for (unsigned l = 0; l < LSB_SELL_ROWS && uniform; l++)
      if (v[l] != 0.0) {
        if (!have)
          c0 = c[l], have = 1;
        else
          uniform = c[l] == c0;
      }
    if (uniform) {
      S->sbase[2 * q] += c0, S->sbase[2 * q + 1] = -1;
    } else {
      S->sbase[2 * q + 1] = (int)nc;
      if (nc != q)
        memmove(S->codes + nc * LSB_SELL_ROWS, c, LSB_SELL_ROWS * sizeof(short));
      nc++;
    }
  }
  S->ncode_slots = (unsigned)nc;
  return S;
}

struct lsb_sell_vc *lsb_sell16_value_slots(const struct lsb_sell *S) {
  if (!S || !S->codes || !S->sbase)
    return NULL;
  const size_t nslots = (size_t)(S->stored / LSB_SELL_ROWS);
  struct lsb_sell_vc *V = lsb_calloc(struct lsb_sell_vc, 1);
  V->nslots = nslots;
  V->slots = lsb_calloc(int, 4 * (nslots + 1));
  V->vconst = lsb_calloc(double, nslots + 1);
  size_t nv = 0;
  for (size_t q = 0; q < nslots; q++) { /* which slots keep their values */
    const double *v = S->vals + q * LSB_SELL_ROWS;
    int same = v[0] != 0.0;
    for (unsigned l = 1; l < LSB_SELL_ROWS && same; l++)
      same = memcmp(v + l, v, sizeof(double)) == 0; /* the same BITS */
    V->slots[4 * q] = S->sbase[2 * q], V->slots[4 * q + 1] = S->sbase[2 * q + 1];
    V->slots[4 * q + 2] = same ? -1 : (int)nv++;
    V->vconst[q] = same ? v[0] : 0.0;
  }
  V->nval_slots = (unsigned)nv;
  V->vals = (double *)calloc((nv + 1) * LSB_SELL_ROWS, sizeof(double));
  if (!V->vals)
    errx(EXIT_FAILURE, "lsb_sell16_value_slots: out of memory");
  for (size_t q = 0; q < nslots; q++)
    if (V->slots[4 * q + 2] >= 0)
      memcpy(V->vals + (size_t)V->slots[4 * q + 2] * LSB_SELL_ROWS, S->vals + q * LSB_SELL_ROWS,
             LSB_SELL_ROWS * sizeof(double));
  return V;
}

/* centre of a pure slice's records: slot c with both neighbours one element away; prefers
 * base 0 (the dot's operand comes for free), then an even base (aligned 16-byte gather) */
static int tmpl_centre(const int *base, int n) {
  int best = -1, score = -1;
  for (int c = 1; c + 1 < n; c++)
    if (base[c - 1] == base[c] - 1 && base[c + 1] == base[c] + 1) {
      const int sc = (base[c] == 0) * 2 + !(base[c] & 1);
      if (sc > score)
        best = c, score = sc;
    }
  return best;
}

struct lsb_sell_tmpls *lsb_sell16_templates(const struct lsb_sell *S, const struct lsb_sell_vc *V) {
  if (!S || !V || !S->sptr || S->ncode_slots || !S->nslice)
    return NULL;
  const unsigned ns = S->nslice;
  struct lsb_sell_tmpls *T = lsb_calloc(struct lsb_sell_tmpls, 1);
  T->nslice = ns;
  T->tid = (unsigned char *)malloc((size_t)ns + 8);
  T->vbase = lsb_calloc(unsigned, 2 * ((size_t)ns + 8));
  T->t = lsb_calloc(struct lsb_sell_tmpl, 254);
  memset(T->tid, 255, (size_t)ns + 8);
  unsigned long long *count = lsb_calloc(unsigned long long, 254);
  size_t mcap = 1024, nmask = 0;
  unsigned long long *mask = (unsigned long long *)malloc(2 * mcap * sizeof *mask);
  for (unsigned s = 0; s < ns; s++) {
    const unsigned q0 = S->sptr[s] / LSB_SELL_ROWS, len = (S->sptr[s + 1] - S->sptr[s]) / LSB_SELL_ROWS;
    if (len == 0 || len > LSB_TMPL_SLOTS)
      continue;
    struct lsb_sell_tmpl t;
    memset(&t, 0, sizeof t);
    t.nslots = (int)len;
    int ok = 1, nkept = 0, nmk = 0, vb = -1;
    unsigned long long mk[2 * LSB_TMPL_SLOTS];
    for (unsigned j = 0; j < len && ok; j++) {
      const int *r = V->slots + 4 * ((size_t)q0 + j);
      ok = r[1] < 0 && (j == 0 || r[0] > t.base[j - 1]); /* code-free, ascending */
      t.base[j] = r[0];
      if (r[2] < 0) {
        t.kidx[j] = -1, t.cst[j] = V->vconst[(size_t)q0 + j];
        continue;
      }
      /* keeps its values: one number or zero?  then a mask will do */
      const double *v = V->vals + (size_t)r[2] * LSB_SELL_ROWS;
      double k = 0.0;
      int one = 1;
      unsigned long long m0 = 0, m1 = 0;
      for (unsigned l = 0; l < LSB_SELL_ROWS && one; l++)
        if (v[l] != 0.0) {
          if (k == 0.0)
            k = v[l];
          one = memcmp(&k, &v[l], sizeof k) == 0;
          if (l < 64)
            m0 |= 1ull << l;
          else
            m1 |= 1ull << (l - 64);
        }
      if (one && k != 0.0) {
        t.kind[j] = 2, t.kidx[j] = nmk, t.cst[j] = k;
        mk[2 * nmk] = m0, mk[2 * nmk + 1] = m1, nmk++;
      } else {
        if (vb < 0)
          vb = r[2];
        /* (kept slots of a slice are consecutive value slots as long as no masked one sits
         * between them -- a masked slot still owns its value slot: index by difference) */
        t.kind[j] = 1, t.kidx[j] = r[2] - vb, nkept++;
      }
    }
    if (!ok)
      continue;
    /* kept / masked slots only as the neighbours c-1 / c+1 of a constant centre, far slots constant */
    const int c = tmpl_centre(t.base, (int)len);
    for (int j = 0; j < (int)len && ok; j++)
      if (t.kind[j] != 0 && !(c >= 1 && (j == c - 1 || j == c + 1)))
        ok = 0;
    if (!ok)
      continue;
    unsigned id = 0;
    while (id < T->ntmpl && memcmp(&T->t[id], &t, sizeof t))
      id++;
    if (id == T->ntmpl) {
      if (T->ntmpl == 254)
        continue; /* table full: the slice goes the per-slot way */
      T->t[T->ntmpl++] = t;
    }
    T->tid[s] = (unsigned char)id, count[id]++, T->covered++;
    T->vbase[2 * s] = vb < 0 ? 0u : (unsigned)vb, T->vbase[2 * s + 1] = (unsigned)nmask;
    if (nmask + (size_t)nmk + 1 > mcap) {
      mcap = 2 * mcap + (size_t)nmk;
      mask = (unsigned long long *)realloc(mask, 2 * mcap * sizeof *mask);
    }
    memcpy(mask + 2 * nmask, mk, 2 * (size_t)nmk * sizeof *mask);
    nmask += (size_t)nmk;
  }
  /* the set's shape: the nfar (<= 2, the same on both sides) most slices have */
  unsigned long long by_nfar[3] = {0, 0, 0};
  for (unsigned id = 0; id < T->ntmpl; id++) {
    const int n = T->t[id].nslots, c = tmpl_centre(T->t[id].base, n);
    if (c >= 1 && c - 1 == n - c - 2 && c - 1 <= 2)
      by_nfar[c - 1] += count[id];
  }
  unsigned nf = 0;
  for (unsigned k = 1; k < 3; k++)
    if (by_nfar[k] > by_nfar[nf])
      nf = k;
  T->nfar = nf;
  for (unsigned id = 0; id < T->ntmpl; id++) {
    const int n = T->t[id].nslots, c = tmpl_centre(T->t[id].base, n);
    if (c >= 1 && (unsigned)(c - 1) == nf && (unsigned)(n - c - 2) == nf)
      T->t[id].shaped = 1, T->shaped += count[id];
  }
  /* a template that is not all-constant has to be shaped (only that path reads values or
   * masks): the others' slices go back to the per-slot way */
  for (unsigned s = 0; s < ns; s++)
    if (T->tid[s] != 255) {
      const struct lsb_sell_tmpl *t = &T->t[T->tid[s]];
      int special = 0;
      for (int j = 0; j < t->nslots; j++)
        special |= t->kind[j] != 0;
      if (special && !t->shaped)
        T->tid[s] = 255, T->covered--;
    }
  /* value slots a launch of the template kernel still reads */
  for (unsigned s = 0; s < ns; s++) {
    const unsigned q0 = S->sptr[s] / LSB_SELL_ROWS, len = (S->sptr[s + 1] - S->sptr[s]) / LSB_SELL_ROWS;
    if (T->tid[s] == 255) {
      for (unsigned j = 0; j < len; j++)
        T->kept_read += V->slots[4 * ((size_t)q0 + j) + 2] >= 0;
    } else {
      const struct lsb_sell_tmpl *t = &T->t[T->tid[s]];
      for (int j = 0; j < t->nslots; j++)
        T->kept_read += t->kind[j] == 1;
    }
  }
  T->nmask = nmask, T->mask = mask;
  free(count);
  if (T->covered * 8 < (unsigned long long)ns * 7 || T->shaped * 4 < (unsigned long long)ns * 3) {
    lsb_sell_tmpls_free(T);
    return NULL;
  }
  return T;
}

/* The bounds k_spmv_sell16's constant-slot path and k_spmv_tmpl rely on, as host assertions (run at
 * every upload by shard_upload; deep under tests/asan_host.c).  Both kernels issue UNGUARDED 16-byte
 * gathers x[g + base .. g + base + 1] for the two rows of every lane wherever a slot is constant --
 * legitimate only because a slot that is constant has 128 real entries, i.e. 128 columns inside the
 * operator.  This function checks exactly that instead of trusting the builder:
 *   every slice (constant-slot layout V): a constant slot (value slot -1) carries no code slot and
 *     0 <= g0 + base, g0 + base + 128 <= xlen with g0 = row_begin + 128 s; a kept slot names a value
 *     slot < nval_slots; (deep) every non-zero kept value's column g0 + l + base lies in [0, xlen);
 *   every slice with a template (T): id < ntmpl; template slots ascending, nslots <= 8; constant
 *     template slots obey the same 128-column rule; a shaped template is [nfar][c-1, c, c+1][nfar]
 *     with constant far slots and centre and base[c-1] + 1 = base[c] = base[c+1] - 1; only c-1 / c+1
 *     may be kept (kind 1: vbase + k < nval_slots) or masked (kind 2: first mask + k < nmask); an
 *     unshaped template is all-constant; the template agrees with the slice's slot records
 *     (bases; constants bit for bit; (deep) masks and kept values reproduce the slice's 128 values);
 *   rows >= nrows of the last slice hold no constant slot.
 * Returns 0 or a rule number with the violated rule in `why`.  Round 3's GPU memory fault
 * (gpurun_out/r3_probe18) was a probe variant that issued a template's far
 * gathers for a slice without one: x[row - nx] for rows < nx, the page below the allocation. */
int lsb_tmpl_check(const struct lsb_sell *S, const struct lsb_sell_vc *V, const struct lsb_sell_tmpls *T,
                   unsigned row_begin, unsigned nrows, unsigned xlen, int deep, char *why, size_t whylen) {
#define TC_FAIL(code, ...)                                                                     \
  do {                                                                                         \
    if (why && whylen)                                                                         \
      snprintf(why, whylen, __VA_ARGS__);                                                      \
    return code;                                                                               \
  } while (0)
  if (!S || !V || !S->sptr)
    TC_FAIL(1, "no layout");
  const unsigned ns = S->nslice;
  if ((unsigned long long)ns * LSB_SELL_ROWS < nrows || V->nslots != S->stored / LSB_SELL_ROWS)
    TC_FAIL(2, "%u slices for %u rows, %llu slot records for %llu slots", ns, nrows, V->nslots,
            S->stored / LSB_SELL_ROWS);
  if (T && T->nslice != ns)
    TC_FAIL(3, "templates of %u slices for a copy of %u", T->nslice, ns);
  if (T && (T->ntmpl > 254 || T->nfar > 2))
    TC_FAIL(4, "%u templates (at most 254), %u far slots per side (at most 2)", T->ntmpl, T->nfar);
  for (unsigned s = 0; s < ns; s++) {
    const unsigned q0 = S->sptr[s] / LSB_SELL_ROWS, len = (S->sptr[s + 1] - S->sptr[s]) / LSB_SELL_ROWS;
    const long long g0 = (long long)row_begin + (long long)s * LSB_SELL_ROWS;
    if (S->sptr[s + 1] < S->sptr[s] || S->sptr[s] % LSB_SELL_ROWS || (unsigned long long)q0 + len > V->nslots)
      TC_FAIL(5, "slice %u: slots [%u, %u + %u) of %llu", s, q0, q0, len, V->nslots);
    const unsigned live = nrows - s * LSB_SELL_ROWS < LSB_SELL_ROWS ? nrows - s * LSB_SELL_ROWS : LSB_SELL_ROWS;
    for (unsigned j = 0; j < len; j++) {
      const int *r = V->slots + 4 * ((size_t)q0 + j);
      if (r[2] < 0) { /* constant: the kernels gather all 128 operands without a look at anything */
        if (r[1] >= 0)
          TC_FAIL(6, "slice %u slot %u: constant but with code slot %d", s, j, r[1]);
        if (g0 + r[0] < 0 || g0 + r[0] + LSB_SELL_ROWS > (long long)xlen)
          TC_FAIL(7, "slice %u slot %u: constant slot gathers x[%lld, %lld) of %u", s, j, g0 + r[0],
                  g0 + r[0] + LSB_SELL_ROWS, xlen);
        if (live < LSB_SELL_ROWS)
          TC_FAIL(8, "slice %u slot %u: constant in a slice that holds only %u rows", s, j, live);
        if (V->vconst[(size_t)q0 + j] == 0.0)
          TC_FAIL(9, "slice %u slot %u: constant 0 (padding is never constant)", s, j);
      } else {
        if ((unsigned)r[2] >= V->nval_slots)
          TC_FAIL(10, "slice %u slot %u: value slot %d of %u", s, j, r[2], V->nval_slots);
        if (r[1] >= 0 && (unsigned)r[1] >= S->ncode_slots)
          TC_FAIL(11, "slice %u slot %u: code slot %d of %u", s, j, r[1], S->ncode_slots);
        if (deep && r[1] < 0) {
          const double *v = V->vals + (size_t)r[2] * LSB_SELL_ROWS;
          for (unsigned l = 0; l < LSB_SELL_ROWS; l++)
            if (v[l] != 0.0 && (l >= live || g0 + l + r[0] < 0 || g0 + l + r[0] >= (long long)xlen))
              TC_FAIL(12, "slice %u slot %u row %u: entry at column %lld of %u (rows in the slice: %u)", s, j, l,
                      g0 + l + r[0], xlen, live);
        }
      }
    }
    if (!T || T->tid[s] == 255)
      continue;
    if (T->tid[s] >= T->ntmpl)
      TC_FAIL(13, "slice %u: template %u of %u", s, T->tid[s], T->ntmpl);
    const struct lsb_sell_tmpl *t = &T->t[T->tid[s]];
    if (t->nslots < 1 || t->nslots > LSB_TMPL_SLOTS || (unsigned)t->nslots != len)
      TC_FAIL(14, "slice %u: template of %d slots for a slice of %u", s, t->nslots, len);
    const int nf = (int)T->nfar, c = nf + 1;
    if (t->shaped && (t->nslots != 2 * nf + 3 || t->base[c - 1] != t->base[c] - 1 || t->base[c + 1] != t->base[c] + 1))
      TC_FAIL(15, "slice %u: shaped template without the [%d][c-1, c, c+1][%d] shape", s, nf, nf);
    const unsigned vb = T->vbase[2 * (size_t)s], mb = T->vbase[2 * (size_t)s + 1];
    for (int j = 0; j < t->nslots; j++) {
      const int *r = V->slots + 4 * ((size_t)q0 + j);
      if (j && t->base[j] <= t->base[j - 1])
        TC_FAIL(16, "slice %u: template bases not ascending at slot %d", s, j);
      if (t->base[j] != r[0] || r[1] >= 0)
        TC_FAIL(17, "slice %u slot %d: template base %d, slot record {%d, %d}", s, j, t->base[j], r[0], r[1]);
      const int side = t->shaped && (j == c - 1 || j == c + 1);
      if (t->kind[j] != 0 && !side)
        TC_FAIL(18, "slice %u slot %d: kind %d outside the slots c-1 / c+1 of a shaped template", s, j, t->kind[j]);
      if (t->kind[j] == 0) {
        if (t->kidx[j] != -1 || r[2] >= 0 || memcmp(&t->cst[j], &V->vconst[(size_t)q0 + j], sizeof(double)))
          TC_FAIL(19, "slice %u slot %d: template constant %g, slot record {value slot %d, constant %g}", s, j,
                  t->cst[j], r[2], V->vconst[(size_t)q0 + j]);
        /* (its 128-column rule was checked on the slot record above: same base) */
      } else if (t->kind[j] == 1) {
        if (t->kidx[j] < 0 || (unsigned long long)vb + (unsigned)t->kidx[j] >= V->nval_slots ||
            r[2] != (int)(vb + (unsigned)t->kidx[j]))
          TC_FAIL(20, "slice %u slot %d: kept slot %u + %d, slot record names %d of %u", s, j, vb, t->kidx[j], r[2],
                  V->nval_slots);
      } else if (t->kind[j] == 2) {
        if (t->kidx[j] < 0 || (unsigned long long)mb + (unsigned)t->kidx[j] >= T->nmask || r[2] < 0 ||
            (unsigned)r[2] >= V->nval_slots || t->cst[j] == 0.0)
          TC_FAIL(21, "slice %u slot %d: mask %u + %d of %llu (value slot %d, number %g)", s, j, mb, t->kidx[j],
                  T->nmask, r[2], t->cst[j]);
        if (deep) { /* the mask and the number reproduce the slot's 128 values bit for bit */
          const unsigned long long *m = T->mask + 2 * ((size_t)mb + (unsigned)t->kidx[j]);
          const double *v = V->vals + (size_t)r[2] * LSB_SELL_ROWS;
          for (unsigned l = 0; l < LSB_SELL_ROWS; l++) {
            const double want = (m[l >> 6] >> (l & 63u)) & 1ull ? t->cst[j] : 0.0;
            if (memcmp(&want, &v[l], sizeof want) && !(want == 0.0 && v[l] == 0.0))
              TC_FAIL(22, "slice %u slot %d row %u: mask gives %g, the slot holds %g", s, j, l, want, v[l]);
          }
        }
      } else
        TC_FAIL(23, "slice %u slot %d: kind %d", s, j, t->kind[j]);
    }
  }
  return 0;
#undef TC_FAIL
}

/* ---- z-columns of the template layout (include/lsbench_hip.h: struct lsb_tmpl_cols) ---- */
#define LSB_COL_XCDS 8 /* = NXCD of the kernels: the items are dealt per XCD */

/* may slice s stand inside a z-column of planes of `period` slices?  A shaped template whose
 * outermost far slots reach exactly one plane down and up, constant or masked slots only */
static int col_member(const struct lsb_sell_tmpls *T, unsigned s, unsigned period) {
  if (T->tid[s] == 255 || T->nfar < 1)
    return 0;
  const struct lsb_sell_tmpl *t = &T->t[T->tid[s]];
  const int c = (int)T->nfar + 1, last = 2 * (int)T->nfar + 2;
  if (!t->shaped || t->nslots != last + 1)
    return 0;
  for (int j = 0; j < t->nslots; j++)
    if (t->kind[j] == 1)
      return 0;
  const long long plane = (long long)LSB_SELL_ROWS * period;
  return (long long)t->base[0] == (long long)t->base[c] - plane && (long long)t->base[last] == (long long)t->base[c] + plane;
}
/* the same template and, bit for bit, the same mask words */
static int col_same(const struct lsb_sell_tmpls *T, unsigned a, unsigned b) {
  if (T->tid[a] != T->tid[b])
    return 0;
  const struct lsb_sell_tmpl *t = &T->t[T->tid[a]];
  size_t nmk = 0;
  for (int j = 0; j < t->nslots; j++)
    nmk += t->kind[j] == 2;
  return !nmk || !memcmp(T->mask + 2 * (size_t)T->vbase[2 * (size_t)a + 1], T->mask + 2 * (size_t)T->vbase[2 * (size_t)b + 1],
                         2 * nmk * sizeof(unsigned long long));
}

struct lsb_tmpl_cols *lsb_sell_tmpl_columns(const struct lsb_sell_tmpls *T, unsigned period, unsigned kmax) {
  return lsb_sell_tmpl_columns_range(T, period, kmax, 0, T ? T->nslice : 0);
}

struct lsb_tmpl_cols *lsb_sell_tmpl_columns_range(const struct lsb_sell_tmpls *T, unsigned period, unsigned kmax,
                                                  unsigned s_lo, unsigned s_hi) {
  if (!T || period < LSB_COL_XCDS || T->nslice < 2 * period || T->nfar < 1 || s_hi > T->nslice || s_lo >= s_hi)
    return NULL;
  if (kmax < 2)
    kmax = 2;
  if (kmax > LSB_TMPL_COL_MAX)
    kmax = LSB_TMPL_COL_MAX;
  const unsigned ns = T->nslice, nplanes = (ns + period - 1) / period;
  struct lsb_tmpl_cols *C = lsb_calloc(struct lsb_tmpl_cols, 1);
  C->kmax = kmax, C->period = period, C->centre0 = 1, C->s_lo = s_lo, C->s_hi = s_hi;
  C->item = lsb_calloc(unsigned, 4 * ((size_t)ns + 1));
  /* z-groups of at most kmax planes, as equal as they come (50 planes, kmax 16: 13, 13, 12, 12).  A CELL is one
   * position of one z-group: its slices become one column, or several items where the run breaks (first / last
   * plane, a slice that keeps values).  Cells in z-group-major order, positions ascending -- the order the chip
   * sweeps them in -- and cut into eight runs of (nearly) equal slice counts, one per XCD: a contiguous band
   * of planes each, whole z-groups but for the two a cut falls into.  (Dealing every XCD the same eighth of
   * every plane instead leaves one XCD 4 of 25 positions where the others have 3 when a "plane" is a grid
   * line of 25 slices: the launch ran 28 % longer than on a grid of 64-slice lines, profiles/r04_pad.txt.) */
  const unsigned ngroups = (nplanes + kmax - 1) / kmax;
  const unsigned long long ncell = (unsigned long long)ngroups * period, total = s_hi - s_lo;
  unsigned ni = 0, xk = 0;
  unsigned long long seen = 0; /* slices of the cells emitted so far */
  C->xbeg[0] = 0;
  for (unsigned long long cell = 0; cell < ncell; cell++) {
    const unsigned zg = (unsigned)(cell / period), p = (unsigned)(cell % period);
    const unsigned z0 = (unsigned)((unsigned long long)nplanes * zg / ngroups),
                   zend = (unsigned)((unsigned long long)nplanes * (zg + 1) / ngroups);
    /* the XCD this cell goes to: by the slices in front of it */
    while (xk + 1 < LSB_COL_XCDS && seen * LSB_COL_XCDS >= total * (xk + 1))
      C->xbeg[++xk] = ni;
    for (unsigned z = z0; z < zend;) {
      const unsigned long long s = (unsigned long long)z * period + p;
      if (s >= s_hi)
        break;
      if (s < s_lo) {
        z++;
        continue;
      }
      unsigned run = 1;
      if (col_member(T, (unsigned)s, period))
        while (z + run < zend && s + (unsigned long long)run * period < s_hi &&
               col_member(T, (unsigned)(s + (unsigned long long)run * period), period) &&
               col_same(T, (unsigned)s, (unsigned)(s + (unsigned long long)run * period)))
          run++;
      unsigned *it = C->item + 4 * (size_t)ni++;
      it[0] = (unsigned)s, it[1] = run, it[2] = T->tid[s], it[3] = T->vbase[2 * (size_t)s + 1];
      if (run >= 2) {
        C->in_cols += run;
        C->centre0 &= T->t[T->tid[s]].base[T->nfar + 1] == 0;
      }
      seen += run;
      z += run;
    }
  }
  while (xk + 1 < LSB_COL_XCDS)
    C->xbeg[++xk] = ni;
  C->xbeg[LSB_COL_XCDS] = C->nitem = ni;
  /* LOCKSTEP groups: the four items a workgroup takes in one turn (items xbeg + 4g .. 4g + 3 of its
   * XCD) are four neighbouring columns of equal length -- bit 31 of their slice count says so, and
   * the four waves then keep step through a barrier per plane, so that a workgroup asks for 4 KB of
   * consecutive addresses at a time */
  for (unsigned k = 0; k < LSB_COL_XCDS; k++)
    for (unsigned i = C->xbeg[k]; i + 4 <= C->xbeg[k + 1]; i += 4) {
      const unsigned *it = C->item + 4 * (size_t)i;
      if (it[1] >= 2 && it[5] == it[1] && it[9] == it[1] && it[13] == it[1])
        for (int w = 0; w < 4; w++)
          C->item[4 * (size_t)(i + w) + 1] |= LSB_TMPL_COL_LOCKSTEP;
    }
  if (C->in_cols * 4 < (unsigned long long)(s_hi - s_lo) * 3) {
    lsb_tmpl_cols_free(C);
    return NULL;
  }
  return C;
}

int lsb_tmpl_cols_check(const struct lsb_sell_tmpls *T, const struct lsb_tmpl_cols *C, char *why, size_t whylen) {
#define CC_FAIL(code, ...)                                                                     \
  do {                                                                                         \
    if (why && whylen)                                                                         \
      snprintf(why, whylen, __VA_ARGS__);                                                      \
    free(seen);                                                                                \
    return code;                                                                               \
  } while (0)
  unsigned char *seen = NULL;
  if (!T || !C || !C->item)
    CC_FAIL(1, "no column plan");
  const unsigned ns = T->nslice, period = C->period;
  if (C->s_hi > ns || C->s_lo >= C->s_hi)
    CC_FAIL(11, "slices [%u, %u) of %u", C->s_lo, C->s_hi, ns);
  if (period < LSB_COL_XCDS || C->kmax < 2 || C->kmax > LSB_TMPL_COL_MAX || C->xbeg[0] != 0 || C->xbeg[LSB_COL_XCDS] != C->nitem)
    CC_FAIL(2, "period %u, columns of up to %u slices, items [%u, %u) of %u", period, C->kmax, C->xbeg[0],
            C->xbeg[LSB_COL_XCDS], C->nitem);
  for (unsigned k = 0; k < LSB_COL_XCDS; k++)
    if (C->xbeg[k] > C->xbeg[k + 1])
      CC_FAIL(3, "item ranges of the XCDs not ascending at %u", k);
  seen = (unsigned char *)calloc((size_t)ns + 1, 1);
  if (!seen)
    CC_FAIL(4, "out of memory");
  for (unsigned i = 0; i < C->nitem; i++) {
    const unsigned *it = C->item + 4 * (size_t)i;
    const unsigned s = it[0], run = it[1] & ~LSB_TMPL_COL_LOCKSTEP;
    if (it[1] & LSB_TMPL_COL_LOCKSTEP) { /* the whole turn of a workgroup: four columns of one length */
      unsigned xk = 0; /* the XCD whose range holds item i */
      while (xk + 1 < LSB_COL_XCDS && C->xbeg[xk + 1] <= i)
        xk++;
      const unsigned g0 = C->xbeg[xk] + (i - C->xbeg[xk]) / 4 * 4;
      if (g0 + 4 > C->xbeg[xk + 1])
        CC_FAIL(12, "item %u: lockstep in a turn of fewer than four items", i);
      for (unsigned w = 0; w < 4; w++)
        if (C->item[4 * (size_t)(g0 + w) + 1] != it[1] || run < 2)
          CC_FAIL(12, "item %u: lockstep, but item %u of its turn has another length", i, g0 + w);
    }
    if (run < 1 || run > C->kmax || s < C->s_lo || (unsigned long long)s + (unsigned long long)(run - 1) * period >= C->s_hi)
      CC_FAIL(5, "item %u: %u slices from slice %u, every %u, of [%u, %u)", i, run, s, period, C->s_lo, C->s_hi);
    for (unsigned k = 0; k < run; k++) {
      const unsigned sk = s + k * period;
      if (seen[sk]++)
        CC_FAIL(6, "item %u: slice %u is in two items", i, sk);
      if (run >= 2 && (!col_member(T, sk, period) || !col_same(T, s, sk)))
        CC_FAIL(7, "item %u: slice %u does not continue the column of slice %u", i, sk, s);
    }
    if (run >= 2 && C->centre0 && T->t[T->tid[s]].base[T->nfar + 1] != 0)
      CC_FAIL(10, "item %u: centre base %d in a plan that says 0", i, T->t[T->tid[s]].base[T->nfar + 1]);
    if (run >= 2 && (it[2] != T->tid[s] || it[3] != T->vbase[2 * (size_t)s + 1]))
      CC_FAIL(8, "item %u: template %u / first mask %u, its first slice has %u / %u", i, it[2], it[3], T->tid[s],
              T->vbase[2 * (size_t)s + 1]);
  }
  for (unsigned s = C->s_lo; s < C->s_hi; s++)
    if (!seen[s])
      CC_FAIL(9, "slice %u is in no item", s);
  free(seen);
  return 0;
#undef CC_FAIL
}

void lsb_tmpl_cols_free(struct lsb_tmpl_cols *C) {
  if (!C)
    return;
  free(C->item), free(C);
}

void lsb_sell_tmpls_free(struct lsb_sell_tmpls *T) {
  if (!T)
    return;
  free(T->tid), free(T->vbase), free(T->mask), free(T->t), free(T);
}

void lsb_sell_vc_free(struct lsb_sell_vc *V) {
  if (!V)
    return;
  free(V->slots), free(V->vconst), free(V->vals), free(V);
}

void lsb_sell_free(struct lsb_sell *S) {
  if (!S)
    return;
  free(S->sptr), free(S->cols), free(S->vals), free(S->codes), free(S->sbase), free(S);
}
