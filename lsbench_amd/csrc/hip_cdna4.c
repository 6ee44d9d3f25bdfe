/*
 * hip_cdna4 -- the MI355X (gfx950) backend of lsbench: host side, plain C.
 *
 * Slots into the reference where every other backend does
 * (src/lsbench-impl.h:42-68): hip_cdna4_init / _finalize / _bench.  The
 * skeleton of _bench is that of the reference's only GPU backend that uploads
 * a raw CSR and writes x back (src/cusparse.c:164-213): untimed csr_init,
 * `trials` warm-up solves, sync, timer, `trials` solves, sync, timer, copy x
 * back, CSV record -- with a Jacobi-preconditioned CG (the Krylov+Jacobi
 * semantics of src/ginkgo.cpp:55-69,91-99: x reset to the initial guess
 * before every trial, reset not timed ... here the reset is fused into the
 * first sweep) made of the hand-written kernels in hip_kernels.hip instead of
 * a vendor library call.
 *
 * The operator is what the reference's CHOLMOD path factorises,
 * S = triu(A)+triu(A,1)^T (src/cholmod-impl.h:5-21), so x matches CHOLMOD's
 * answer.
 *
 * Data layout in HBM, per shard (a shard = the contiguous row range one rank
 * owns; one shard per process, or `nvirt` shards on one device in test mode):
 *   offs[n+1] i32, cols[nnz] i32 (GLOBAL column ids), vals[nnz] f64,
 *   rowblk[nblk+1] i32, dinv[n], r[n], q[n] f64,
 *   pfull[n_global] f64  -- the search direction in GLOBAL index space; rows
 *                           [row_begin,row_begin+n) are owned, the rest is
 *                           filled by the exchange step.  SpMV gathers from it
 *                           with the global column ids, so no index
 *                           translation and no halo packing exists.
 *   partial sums + a 64-byte lsb_pcg_state with rz/rr/iters/status.
 * All scalars of the iteration (alpha, beta, stop test) live on the device;
 * the host only enqueues and, every `check_every` iterations, reads the
 * status word.
 */
#define _GNU_SOURCE
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <string.h>
#include <strings.h>
#include <time.h>

#include "lsb_impl.h"

/* ------------------------------------------------------------------------ */
/* backend globals (reference style: file statics, src/cusparse.c:33-36)     */
/* ------------------------------------------------------------------------ */
static int initialized = 0;
static hipStream_t g_stream = 0, g_comm_stream = 0; /* compute / halo exchange */
static struct lsb_hip_opts g_opts;
static int g_opts_set = 0;
static struct lsb_hip_result g_last;

static double wall_seconds(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void lsb_hip_opts_default(struct lsb_hip_opts *o) {
  memset(o, 0, sizeof *o);
  o->tol = 1e-12;
  o->maxit = 20000;
  o->op_mode = LSB_OP_CHOLMOD_UPPER;
  o->precond = LSB_PRECOND_JACOBI;
  o->spmv_variant = LSB_SPMV_AUTO;
  o->check_every = 0;
  o->use_graph = 1;
  o->sample_spmv = 0;
  o->nvirt = 1;
  o->comm = LSB_COMM_AUTO;
  o->overlap = -1; /* on where the halos are large enough to pay for the split, DESIGN.md section 6 */
  o->spmv_tune = -1;
  o->spmv_grid = 0;
  o->reorder = 0;
  o->krylov = LSB_KRYLOV_PCG;
  o->restart = 30;
  o->verbose = 0;
}

static void opts_from_env(struct lsb_hip_opts *o) {
  const char *e;
  if ((e = getenv("LSBENCH_HIP_TOL")))
    o->tol = atof(e);
  if ((e = getenv("LSBENCH_HIP_MAXIT")))
    o->maxit = (unsigned)strtoul(e, NULL, 10);
  if ((e = getenv("LSBENCH_HIP_OPERATOR")))
    o->op_mode = strcasecmp(e, "raw") == 0 ? LSB_OP_RAW : LSB_OP_CHOLMOD_UPPER;
  if ((e = getenv("LSBENCH_HIP_NVIRT")))
    o->nvirt = atoi(e);
  if ((e = getenv("LSBENCH_HIP_GRAPH")))
    o->use_graph = atoi(e);
  if ((e = getenv("LSBENCH_HIP_SPMV")))
    o->spmv_variant = atoi(e);
  if ((e = getenv("LSBENCH_HIP_COMM")))
    o->comm = !strcmp(e, "rccl") ? LSB_COMM_RCCL : !strcmp(e, "p2p") ? LSB_COMM_P2P : LSB_COMM_AUTO;
  if ((e = getenv("LSBENCH_HIP_OVERLAP")))
    o->overlap = atoi(e);
  if ((e = getenv("LSBENCH_HIP_REORDER")))
    o->reorder = atoi(e);
  if ((e = getenv("LSBENCH_HIP_KRYLOV")))
    o->krylov = strcasecmp(e, "gmres") == 0  ? LSB_KRYLOV_GMRES
                : strcasecmp(e, "cg1") == 0  ? LSB_KRYLOV_PCG1
                : strcasecmp(e, "auto") == 0 ? LSB_KRYLOV_AUTO
                                             : LSB_KRYLOV_PCG;
  if ((e = getenv("LSBENCH_HIP_RESTART")))
    o->restart = atoi(e);
  if ((e = getenv("LSBENCH_HIP_SPMV_TUNE")))
    o->spmv_tune = atoi(e);
  if ((e = getenv("LSBENCH_HIP_SPMV_GRID")))
    o->spmv_grid = atoi(e);
  if ((e = getenv("LSBENCH_HIP_BLAS1_NT")))
    lsb_k_set_blas1_nt(atoi(e));
  if ((e = getenv("LSBENCH_HIP_CHECK_EVERY")))
    o->check_every = atoi(e);
  if ((e = getenv("LSBENCH_HIP_VERBOSE")))
    o->verbose = atoi(e);
}

void lsb_hip_set_opts(const struct lsb_hip_opts *o) {
  g_opts = *o;
  g_opts_set = 1;
}

void lsb_hip_get_opts(struct lsb_hip_opts *o) {
  if (!g_opts_set) {
    lsb_hip_opts_default(&g_opts);
    opts_from_env(&g_opts);
    g_opts_set = 1;
  }
  *o = g_opts;
}

void lsb_hip_last_result(struct lsb_hip_result *res) { *res = g_last; }

int lsb_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

void *lsb_hip_stream(void) { return (void *)g_stream; }
int lsb_hip_is_initialized(void) { return initialized; }

int hip_cdna4_init(void) {
  if (initialized)
    return 1;
  /* lsbench_init calls every backend's init whatever --solver says
   * (src/lsbench.c:143-147): no device => stay uninitialised, quietly. */
  if (lsb_hip_device_count() < 1)
    return 1;
  const char *e = getenv("LSBENCH_HIP_DEVICE");
  if (e)
    LSB_CHK_HIP(hipSetDevice(atoi(e)));
  LSB_CHK_HIP(hipStreamCreate(&g_stream)); /* cf. src/cusparse.c:142 */
  LSB_CHK_HIP(hipStreamCreate(&g_comm_stream));
  struct lsb_hip_opts o;
  lsb_hip_get_opts(&o);
  initialized = 1;
  return 0;
}

int hip_cdna4_finalize(void) {
  if (!initialized)
    return 1;
  LSB_CHK_HIP(hipStreamSynchronize(g_stream));
  LSB_CHK_HIP(hipStreamDestroy(g_stream));
  LSB_CHK_HIP(hipStreamDestroy(g_comm_stream));
  g_stream = 0, g_comm_stream = 0;
  initialized = 0;
  return 0;
}

/* ------------------------------------------------------------------------ */
/* device helpers for C callers                                              */
/* ------------------------------------------------------------------------ */
void *lsb_hip_malloc(size_t bytes) {
  void *p = NULL;
  LSB_CHK_HIP(hipMalloc(&p, bytes ? bytes : 8));
  return p;
}
void lsb_hip_free(void *p) {
  if (p)
    LSB_CHK_HIP(hipFree(p));
}
int lsb_hip_memcpy_h2d(void *d, const void *s, size_t bytes) {
  LSB_CHK_HIP(hipMemcpy(d, s, bytes, hipMemcpyHostToDevice));
  return 0;
}
int lsb_hip_memcpy_d2h(void *d, const void *s, size_t bytes) {
  LSB_CHK_HIP(hipMemcpy(d, s, bytes, hipMemcpyDeviceToHost));
  return 0;
}
int lsb_hip_sync(void) {
  if (!initialized)
    return 1;
  LSB_CHK_HIP(hipStreamSynchronize(g_stream));
  return 0;
}

static void *dev_upload(const void *h, size_t bytes) {
  void *d = lsb_hip_malloc(bytes);
  if (bytes)
    LSB_CHK_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, g_stream));
  return d;
}

/* ------------------------------------------------------------------------ */
/* solver object                                                             */
/* ------------------------------------------------------------------------ */
#define SCAL_STRIDE 8 /* doubles per shard in the scalar slab */
#define MAX_SAMPLES 64

struct shard {
  unsigned row_begin, n;
  unsigned long long nnz;
  int *d_offs, *d_cols, *d_rowblk;
  unsigned char *d_blklanes;
  unsigned sp_flags, sp_grid; /* adaptive-SpMV flavour, picked by tune_spmv() */
  double *d_vals, *d_dinv, *d_r, *d_q, *d_pfull;
  double *d_p1, *d_s1; /* single-reduction CG: p and s = S p (pfull then holds u) */
  unsigned npq, np2;   /* partial counts of the SpMV / sweep launches */
  const double *ar2_parts; /* sweep partials the next all-reduce folds in */
  unsigned ar2_n, ar2_width;
  /* rows that reference other shards' columns sit in row blocks [0,ov_b1) and
   * [ov_b2,nblk); the blocks in between need no halo (0,0 = not separable) */
  unsigned ov_b1, ov_b2;
  int ov_ok;
  /* sliced-ELL copy (LSB_SPMV_SELL), built when padding stays under 1/8; the
   * same prefix/interior/suffix split in slices */
  unsigned *d_sptr;
  int *d_scols;
  double *d_svals;
  unsigned nslice, ov_s1, ov_s2;
  int ov_sok;
  /* ... and its 16-bit-code form (LSB_SP_C16 in sp_flags), own slice offsets */
  unsigned *d_sptr16;
  short *d_scodes;
  int *d_sbase;
  double *d_svals16;
  double *d_parts_pq, *d_parts2;
  double *d_scal; /* [0] p.q   [1] r.z'  [2] r.r   (multi-shard path) */
  struct lsb_pcg_state *d_st;
  unsigned nblk, lanes;
  int variant;
  unsigned col_lo, col_hi; /* column hull referenced by the shard's rows */
  /* column-panel form (LSB_SPMV_PANEL), built for scattered operators only */
  unsigned pn;       /* panels, 0 = not built */
  unsigned *h_pblk;  /* pn+1: first row block of each panel */
  int *pd_offs, *pd_cols, *pd_rowmap, *pd_rowblk;
  unsigned char *pd_blklanes;
  double *pd_vals;
  struct lsb_xfer *recv, *send;
  int nrecv, nsend;
};

struct lsb_hip_solver {
  unsigned n_glob;   /* rows of the whole operator                         */
  unsigned n_here;   /* rows held by this process (sum over its shards)     */
  unsigned row_first; /* first row held by this process                     */
  int nshard;        /* shards in this process (1, or nvirt)                */
  int dist;          /* 1: shards of other processes exist (RCCL)           */
  int multi;         /* nshard > 1 || dist: scalars go through all-reduce   */
  struct shard *sh;
  double *d_scal_all; /* nshard * SCAL_STRIDE doubles                        */
  struct lsb_hip_opts o;
  struct lsb_pcg_state *h_st; /* pinned, 2 slots */
  struct {
    hipGraphExec_t exec;
    int iters;
    double *x;
  } gcache[2];
  int gnext;
  unsigned hint_iters; /* iterations of the previous solve, 0 = none yet */
  unsigned agree_nnz, agree_n; /* distributed: largest shard, identical on all ranks */
  unsigned agree_halo;         /* largest halo (doubles) any shard receives from one peer */
  /* reordering: d_perm[new] = old; b and x are permuted through d_bp / d_xp */
  int *d_perm;
  double *d_bp, *d_xp;
  /* GMRES workspace (allocated on first use) */
  struct gm_work { /* per shard */
    double *V, *parts, *ax;
    struct lsb_gmres_state *st;
    size_t ld;
  } *gm;
  double *gm_red; /* nshard x GM_RED doubles: [0] a norm, [8..) h, [48..) h2 -- all-reduced */
  struct lsb_gmres_state *gm_hst;
  int gm_m;
  hipEvent_t ev_poll[2], ev_vec, ev_halo;
  hipEvent_t ev[4 * MAX_SAMPLES], ev_t0, ev_t1; /* per sample: e0 SpMV e1 e2 e3 */
  int have_events;
  double *d_tmp; /* n_here doubles: scratch for spmv_dev / jacobi sweep */
  /* direct xGMI path (hip_p2p.hip), one context per shard; p2p_on: used for
   * the all-reduces, p2p_halo: also for the halo exchange */
  struct lsb_p2p **p2p;
  int p2p_on, p2p_halo;
  double p2p_us, rccl_us; /* self-test: one exchange + all-reduce, each way */
};

static unsigned pow2_ceil(unsigned v) {
  unsigned p = 1;
  while (p < v)
    p <<= 1;
  return p;
}

/* SpMV kernel choice: rows of a few dozen non-zeros at most stream through
 * LDS (adaptive); long-row matrices go wavefront-per-row. */
static void choose_spmv(struct shard *s, const struct lsb_hip_opts *o) {
  const unsigned mean = s->n ? (unsigned)((s->nnz + s->n - 1) / s->n) : 1;
  int v = o->spmv_variant;
  /* A matrix of a few hundred thousand non-zeros is launch-latency bound: the
   * sub-wavefront kernel has a shorter dependent-load chain (offs -> cols ->
   * x) than the row-blocked one (rowblk -> offs -> cols -> x -> LDS -> offs)
   * and wins 3.2 vs 5.7 us per launch on tests/xn3b_A_18.txt. */
  if (v == LSB_SPMV_AUTO)
    v = s->nnz <= 500000ull ? LSB_SPMV_SUBWAVE : LSB_SPMV_ADAPTIVE;
  if (v == LSB_SPMV_PANEL && !s->pn)
    v = LSB_SPMV_ADAPTIVE; /* the operator did not qualify for panels */
  if (v == LSB_SPMV_SELL && !s->d_sptr)
    v = LSB_SPMV_ADAPTIVE; /* no sliced-ELL copy (32-bit offsets exceeded) */
  s->variant = v;
  unsigned L = pow2_ceil(mean ? mean : 1);
  if (L < 2)
    L = 2;
  if (L > 64)
    L = 64;
  s->lanes = L;
}

static void *dev_upload(const void *h, size_t bytes);

/* Column-panel form of the shard (lsb_csr_panelize) + its row blocks, one run
 * of blocks per panel so that a launch never crosses a panel. */
static void shard_build_panels(struct shard *s, const struct csr *view, unsigned width) {
  struct lsb_panel_csr *P = lsb_csr_panelize(view, width);
  const unsigned np = P->npanels;
  s->h_pblk = lsb_calloc(unsigned, (size_t)np + 1);
  size_t cap = (size_t)P->offs[P->npairs] / LSB_BLOCK_NNZ * 2 + 4 * (size_t)np + 16, nb = 0;
  unsigned *rball = (unsigned *)malloc((cap + 1) * sizeof(unsigned));
  unsigned char *lanes = (unsigned char *)malloc(cap + 1);
  for (unsigned p = 0; p < np; p++) {
    const unsigned b0 = P->pair_begin[p], cnt = P->pair_begin[p + 1] - b0;
    s->h_pblk[p] = (unsigned)nb;
    if (cnt == 0)
      continue;
    struct csr sub = {cnt, 0, P->offs + b0, NULL, NULL};
    unsigned *rb = NULL;
    const unsigned k = lsb_csr_row_blocks(&sub, LSB_BLOCK_NNZ, &rb);
    if (nb + k + 1 > cap)
      errx(EXIT_FAILURE, "hip_cdna4: panel row-block estimate too small");
    lsb_csr_block_lanes(&sub, rb, k, lanes + nb);
    for (unsigned i = 0; i <= k; i++)
      rball[nb + i] = rb[i] + b0; /* the last entry is the next panel's first */
    nb += k;
    free(rb);
  }
  s->h_pblk[np] = (unsigned)nb;
  rball[nb] = P->npairs;
  s->pn = np;
  s->pd_offs = (int *)dev_upload(P->offs, ((size_t)P->npairs + 1) * sizeof(int));
  s->pd_cols = (int *)dev_upload(P->cols, (size_t)P->offs[P->npairs] * sizeof(int));
  s->pd_vals = (double *)dev_upload(P->vals, (size_t)P->offs[P->npairs] * sizeof(double));
  s->pd_rowmap = (int *)dev_upload(P->pair_row, (size_t)P->npairs * sizeof(int));
  s->pd_rowblk = (int *)dev_upload(rball, (nb + 1) * sizeof(int));
  s->pd_blklanes = (unsigned char *)dev_upload(lanes, nb ? nb : 1);
  LSB_CHK_HIP(hipStreamSynchronize(g_stream));
  free(rball), free(lanes);
  lsb_panel_csr_free(P);
}

/* Upload rows [r0,r1) of the 0-based operator `S` (global column ids) as one
 * shard.  When `S` holds only the shard's rows, pass local=1. */
static void shard_upload(struct shard *s, const struct csr *S, unsigned r0,
                         unsigned r1, int local, unsigned row_begin,
                         unsigned n_glob, const struct lsb_hip_opts *o) {
  const unsigned a = local ? 0 : r0, b = local ? S->nrows : r1;
  const unsigned n = b - a, j0 = S->offs[a], j1 = S->offs[b];
  const unsigned base = S->base;
  s->row_begin = row_begin, s->n = n, s->nnz = j1 - j0;
  int *offs = (int *)malloc(((size_t)n + 1) * sizeof(int));
  int *cols = (int *)malloc(((size_t)s->nnz + 1) * sizeof(int));
  unsigned lo = 0xFFFFFFFFu, hi = 0;
  for (unsigned i = 0; i <= n; i++)
    offs[i] = (int)(S->offs[a + i] - j0);
  for (unsigned j = j0; j < j1; j++) {
    const unsigned c = S->cols[j] - base;
    if (c >= n_glob)
      errx(EXIT_FAILURE, "column %u outside the %u-column operator", c, n_glob);
    cols[j - j0] = (int)c;
    if (c < lo)
      lo = c;
    if (c + 1 > hi)
      hi = c + 1;
  }
  if (s->nnz == 0)
    lo = hi = row_begin;
  s->col_lo = lo, s->col_hi = hi;
  if ((unsigned long long)s->nnz > 0x7FFFFFFFull || n_glob > 0x7FFFFFFFu)
    errx(EXIT_FAILURE, "shard too large for int32 device indices");
  s->d_offs = (int *)dev_upload(offs, ((size_t)n + 1) * sizeof(int));
  s->d_cols = (int *)dev_upload(cols, (size_t)s->nnz * sizeof(int));
  s->d_vals = (double *)dev_upload(S->vals + j0, (size_t)s->nnz * sizeof(double));
  /* row blocks of the adaptive kernel, on the local offsets */
  struct csr view = {n, 0, (unsigned *)offs, NULL, NULL};
  unsigned *rb = NULL;
  s->nblk = lsb_csr_row_blocks(&view, LSB_BLOCK_NNZ, &rb);
  s->d_rowblk = (int *)dev_upload(rb, ((size_t)s->nblk + 1) * sizeof(int));
  unsigned char *lanes = (unsigned char *)malloc((size_t)s->nblk + 1);
  lsb_csr_block_lanes(&view, rb, s->nblk, lanes);
  s->d_blklanes = (unsigned char *)dev_upload(lanes, (size_t)s->nblk);
  /* Which row blocks touch columns owned by other shards?  Under row-range
   * partitioning of a banded operator they are a prefix and a suffix; the
   * blocks in between can start before the halo has arrived. */
  {
    const int row_end = (int)(row_begin + n);
    unsigned b1 = 0, b2 = s->nblk;
    int ok = 1;
    unsigned char *ext = (unsigned char *)calloc(s->nblk ? s->nblk : 1, 1);
    for (unsigned k = 0; k < s->nblk; k++)
      for (unsigned r = rb[k]; r < rb[k + 1] && !ext[k]; r++)
        if (offs[r + 1] > offs[r] &&
            (cols[offs[r]] < (int)row_begin || cols[offs[r + 1] - 1] >= row_end))
          ext[k] = 1;
    while (b1 < s->nblk && ext[b1])
      b1++;
    while (b2 > b1 && ext[b2 - 1])
      b2--;
    for (unsigned k = b1; k < b2; k++)
      ok &= !ext[k];
    free(ext);
    s->ov_ok = ok && b2 > b1, s->ov_b1 = b1, s->ov_b2 = b2;
  }
  /* Scattered rows (mean |col-row| in the millions, x far beyond L2): also
   * build the column-panel form; tune_spmv() keeps whichever is faster. */
  {
    struct csr gview = {n, 0, (unsigned *)offs, (unsigned *)cols, (double *)(S->vals + j0)};
    const char *e = getenv("LSBENCH_HIP_PANEL_COLS");
    const unsigned width = e ? (unsigned)strtoul(e, NULL, 10) : 262144u; /* 2 MiB of x */
    const int forced = o->spmv_variant == LSB_SPMV_PANEL;
    const int scattered = s->nnz > 4000000ull && (double)(hi - lo) * 8.0 > 16.0e6 &&
                          lsb_csr_mean_scatter(&gview, row_begin) > 1.0e6;
    if (width && (forced || (o->spmv_variant == LSB_SPMV_AUTO && scattered)))
      shard_build_panels(s, &gview, width);
  }
  /* Near-uniform row lengths (stencils, meshes): also keep a sliced-ELL copy;
   * tune_spmv() keeps whichever kernel is faster on this shard. */
  {
    struct csr gview = {n, 0, (unsigned *)offs, (unsigned *)cols, (double *)(S->vals + j0)};
    const int forced = o->spmv_variant == LSB_SPMV_SELL;
    const unsigned long long stored = (forced || s->nnz >= 4000000ull) ? lsb_csr_sell_stored(&gview) : 0;
    struct lsb_sell *E = NULL;
    if (stored && (forced || (o->spmv_variant == LSB_SPMV_AUTO && stored <= s->nnz + s->nnz / 8)))
      E = lsb_csr_sellize(&gview);
    if (E) {
      s->nslice = E->nslice;
      s->d_sptr = (unsigned *)dev_upload(E->sptr, ((size_t)E->nslice + 1) * sizeof(unsigned));
      s->d_scols = (int *)dev_upload(E->cols, ((size_t)E->stored + LSB_SELL_ROWS) * sizeof(int));
      s->d_svals = (double *)dev_upload(E->vals, ((size_t)E->stored + LSB_SELL_ROWS) * sizeof(double));
      const int row_end = (int)(row_begin + n);
      unsigned s1 = 0, s2 = E->nslice;
      int ok = 1;
      unsigned char *ext = (unsigned char *)calloc(E->nslice ? E->nslice : 1, 1);
      for (unsigned k = 0; k < E->nslice; k++)
        for (unsigned r = k * LSB_SELL_ROWS; r < n && r < (k + 1) * LSB_SELL_ROWS && !ext[k]; r++)
          if (offs[r + 1] > offs[r] &&
              (cols[offs[r]] < (int)row_begin || cols[offs[r + 1] - 1] >= row_end))
            ext[k] = 1;
      while (s1 < E->nslice && ext[s1])
        s1++;
      while (s2 > s1 && ext[s2 - 1])
        s2--;
      for (unsigned k = s1; k < s2; k++)
        ok &= !ext[k];
      free(ext);
      s->ov_sok = ok && s2 > s1, s->ov_s1 = s1, s->ov_s2 = s2;
      LSB_CHK_HIP(hipStreamSynchronize(g_stream));
      lsb_sell_free(E);
      /* 10 instead of 12 bytes per entry where every slot of every slice is
       * one diagonal band (stencils, banded meshes) */
      struct lsb_sell *H = getenv("LSBENCH_HIP_NO_C16") ? NULL : lsb_csr_sellize16(&gview, row_begin);
      if (H && H->stored > s->nnz + s->nnz / 8) {
        lsb_sell_free(H);
        H = NULL;
      }
      if (H) {
        s->d_sptr16 = (unsigned *)dev_upload(H->sptr, ((size_t)H->nslice + 1) * sizeof(unsigned));
        s->d_scodes = (short *)dev_upload(H->codes, ((size_t)H->stored + LSB_SELL_ROWS) * sizeof(short));
        s->d_sbase = (int *)dev_upload(H->sbase, ((size_t)H->stored / LSB_SELL_ROWS + 1) * sizeof(int));
        s->d_svals16 = (double *)dev_upload(H->vals, ((size_t)H->stored + LSB_SELL_ROWS) * sizeof(double));
        LSB_CHK_HIP(hipStreamSynchronize(g_stream));
        lsb_sell_free(H);
      }
    }
  }
  LSB_CHK_HIP(hipStreamSynchronize(g_stream)); /* host staging is freed next */
  free(rb), free(offs), free(cols), free(lanes);

  s->d_dinv = (double *)lsb_hip_malloc((size_t)n * sizeof(double));
  s->d_r = (double *)lsb_hip_malloc((size_t)n * sizeof(double));
  s->d_q = (double *)lsb_hip_malloc((size_t)n * sizeof(double));
  s->d_pfull = (double *)lsb_hip_malloc((size_t)n_glob * sizeof(double));
  LSB_CHK_HIP(hipMemsetAsync(s->d_pfull, 0, (size_t)n_glob * sizeof(double), g_stream));
  s->d_parts_pq = (double *)lsb_hip_malloc(3 * LSB_MAX_PARTIALS * sizeof(double));
  /* two buffers: k_cg1_update reads the previous launch's partials while
   * writing its own */
  s->d_parts2 = (double *)lsb_hip_malloc(4 * LSB_MAX_PARTIALS * sizeof(double));
  s->d_st = (struct lsb_pcg_state *)lsb_hip_malloc(sizeof(struct lsb_pcg_state));
  LSB_CHK_HIP(hipMemsetAsync(s->d_st, 0, sizeof(struct lsb_pcg_state), g_stream));
  choose_spmv(s, o);

  if (o->precond == LSB_PRECOND_JACOBI) {
    int *d_nz = (int *)lsb_hip_malloc(sizeof(int)), nz = 0;
    LSB_CHK_HIP(hipMemsetAsync(d_nz, 0, sizeof(int), g_stream));
    lsb_k_jacobi_setup(n, row_begin, s->d_offs, s->d_cols, s->d_vals, s->d_dinv,
                       d_nz, g_stream);
    LSB_CHK_HIP(hipMemcpyAsync(&nz, d_nz, sizeof(int), hipMemcpyDeviceToHost, g_stream));
    LSB_CHK_HIP(hipStreamSynchronize(g_stream));
    lsb_hip_free(d_nz);
    if (nz)
      errx(EXIT_FAILURE, "hip_cdna4: %d rows have no non-zero diagonal entry; "
                         "Jacobi preconditioning needs one (cf. the stored-diagonal "
                         "assumption of src/cholmod-impl.h:13)", nz);
  } else {
    /* dinv = 1: unpreconditioned CG through the same kernels */
    double *ones = (double *)malloc((size_t)(n ? n : 1) * sizeof(double));
    for (unsigned i = 0; i < n; i++)
      ones[i] = 1.0;
    LSB_CHK_HIP(hipMemcpy(s->d_dinv, ones, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    free(ones);
  }
}

static void shard_free(struct shard *s) {
  lsb_hip_free(s->d_offs), lsb_hip_free(s->d_cols), lsb_hip_free(s->d_vals);
  lsb_hip_free(s->d_rowblk), lsb_hip_free(s->d_blklanes);
  lsb_hip_free(s->d_dinv), lsb_hip_free(s->d_r);
  lsb_hip_free(s->d_q), lsb_hip_free(s->d_pfull), lsb_hip_free(s->d_parts_pq);
  lsb_hip_free(s->d_p1), lsb_hip_free(s->d_s1);
  lsb_hip_free(s->d_parts2), lsb_hip_free(s->d_st);
  lsb_hip_free(s->pd_offs), lsb_hip_free(s->pd_cols), lsb_hip_free(s->pd_vals);
  lsb_hip_free(s->pd_rowmap), lsb_hip_free(s->pd_rowblk), lsb_hip_free(s->pd_blklanes);
  lsb_hip_free(s->d_sptr), lsb_hip_free(s->d_scols), lsb_hip_free(s->d_svals);
  lsb_hip_free(s->d_sptr16), lsb_hip_free(s->d_scodes), lsb_hip_free(s->d_sbase);
  lsb_hip_free(s->d_svals16);
  free(s->h_pblk);
  free(s->recv), free(s->send);
}

static void plan_exchange(struct shard *s, int me, int nall, const unsigned *hull) {
  s->recv = lsb_calloc(struct lsb_xfer, nall);
  s->send = lsb_calloc(struct lsb_xfer, nall);
  lsb_plan_exchange(me, nall, hull, s->recv, &s->nrecv, s->send, &s->nsend);
}

static void tune_spmv(lsb_hip_solver *sv, struct shard *s);
static void p2p_setup(lsb_hip_solver *sv);
static void exchange_p(lsb_hip_solver *sv);
static void allreduce_scal(lsb_hip_solver *sv, unsigned off, unsigned cnt);
static void drop_graphs(lsb_hip_solver *sv);

static lsb_hip_solver *solver_alloc(int nshard, const struct lsb_hip_opts *o) {
  lsb_hip_solver *sv = lsb_calloc(lsb_hip_solver, 1);
  sv->nshard = nshard;
  sv->sh = lsb_calloc(struct shard, nshard);
  sv->o = *o;
  LSB_CHK_HIP(hipHostMalloc((void **)&sv->h_st, 2 * sizeof(struct lsb_pcg_state), 0));
  return sv;
}

static void solver_finish_setup(lsb_hip_solver *sv) {
  sv->d_scal_all = (double *)lsb_hip_malloc((size_t)sv->nshard * SCAL_STRIDE * sizeof(double));
  LSB_CHK_HIP(hipMemsetAsync(sv->d_scal_all, 0,
                             (size_t)sv->nshard * SCAL_STRIDE * sizeof(double), g_stream));
  for (int i = 0; i < sv->nshard; i++)
    sv->sh[i].d_scal = sv->d_scal_all + (size_t)i * SCAL_STRIDE;
  sv->d_tmp = (double *)lsb_hip_malloc((size_t)sv->n_here * sizeof(double));
  for (int i = 0; i < 4 * MAX_SAMPLES; i++)
    LSB_CHK_HIP(hipEventCreate(&sv->ev[i]));
  LSB_CHK_HIP(hipEventCreate(&sv->ev_t0));
  LSB_CHK_HIP(hipEventCreate(&sv->ev_t1));
  LSB_CHK_HIP(hipEventCreateWithFlags(&sv->ev_poll[0], hipEventDisableTiming));
  LSB_CHK_HIP(hipEventCreateWithFlags(&sv->ev_poll[1], hipEventDisableTiming));
  LSB_CHK_HIP(hipEventCreateWithFlags(&sv->ev_vec, hipEventDisableTiming));
  LSB_CHK_HIP(hipEventCreateWithFlags(&sv->ev_halo, hipEventDisableTiming));
  sv->have_events = 1;
  LSB_CHK_HIP(hipStreamSynchronize(g_stream));
  for (int i = 0; i < sv->nshard; i++)
    tune_spmv(sv, &sv->sh[i]);
  LSB_CHK_HIP(hipStreamSynchronize(g_stream));
  p2p_setup(sv);
}

/*
 * Direct xGMI path: build it, prove it, time it, and only then use it.
 *   dist:    collective.  Every rank runs P2P_ROUNDS rounds of {pattern ->
 *            exchange -> check the halos bit by bit -> all-reduce of three
 *            known values -> check the sums}, then times 100 x {exchange +
 *            all-reduce} on this path and on RCCL.  The path is kept when
 *            EVERY rank saw zero mismatches, no time-out, and (comm = auto) a
 *            faster loop; the decision is taken on all-gathered numbers, so
 *            all ranks take the same one.
 *   virtual: only on request (comm = p2p), for the one-GPU tests of the
 *            kernels; device copies remain the default there.
 */
#define P2P_ROUNDS 24
#define P2P_TIMED 100
static void p2p_rounds(lsb_hip_solver *sv, int rounds, int check, unsigned *d_bad) {
  for (int t = 0; t < rounds; t++) {
    for (int i = 0; i < sv->nshard && check; i++)
      lsb_p2p_test_pattern(sv->sh[i].d_pfull, sv->sh[i].row_begin, sv->sh[i].n, (unsigned)t,
                           g_stream);
    exchange_p(sv);
    for (int i = 0; i < sv->nshard && check; i++) {
      struct shard *s = &sv->sh[i];
      for (int k = 0; k < s->nrecv && sv->p2p_halo; k++)
        lsb_p2p_test_check_range(s->d_pfull, s->recv[k].offset, s->recv[k].count, (unsigned)t,
                                 d_bad, g_stream);
      lsb_p2p_test_setvals(s->d_scal, sv->dist ? lsb_hip_comm_rank() : i, (unsigned)t, g_stream);
    }
    allreduce_scal(sv, 0, 3);
    for (int i = 0; i < sv->nshard && check; i++)
      lsb_p2p_test_checkvals(sv->sh[i].d_scal, sv->dist ? lsb_hip_comm_size() : sv->nshard,
                             (unsigned)t, d_bad, g_stream);
  }
}

static float timed_rounds(lsb_hip_solver *sv) {
  float ms = 0;
  if (sv->dist)
    lsb_hip_comm_barrier();
  p2p_rounds(sv, 8, 0, NULL);
  LSB_CHK_HIP(hipEventRecord(sv->ev_t0, g_stream));
  p2p_rounds(sv, P2P_TIMED, 0, NULL);
  LSB_CHK_HIP(hipEventRecord(sv->ev_t1, g_stream));
  LSB_CHK_HIP(hipEventSynchronize(sv->ev_t1));
  LSB_CHK_HIP(hipEventElapsedTime(&ms, sv->ev_t0, sv->ev_t1));
  return ms * 1e3f / P2P_TIMED;
}

static void p2p_setup(lsb_hip_solver *sv) {
  if (!sv->multi || sv->o.comm == LSB_COMM_RCCL || (!sv->dist && sv->o.comm != LSB_COMM_P2P))
    return;
  const int P = sv->dist ? lsb_hip_comm_size() : 1;
  sv->p2p = lsb_calloc(struct lsb_p2p *, sv->nshard);
  int ok;
  if (sv->dist) {
    struct shard *s = &sv->sh[0];
    sv->p2p[0] = lsb_p2p_create_dist(s->recv, s->nrecv, s->send, s->nsend);
    ok = sv->p2p[0] != NULL;
  } else {
    struct lsb_xfer **rv = lsb_calloc(struct lsb_xfer *, sv->nshard),
                    **sd = lsb_calloc(struct lsb_xfer *, sv->nshard);
    int *nr = lsb_calloc(int, sv->nshard), *ns = lsb_calloc(int, sv->nshard);
    for (int i = 0; i < sv->nshard; i++)
      rv[i] = sv->sh[i].recv, sd[i] = sv->sh[i].send, nr[i] = sv->sh[i].nrecv,
      ns[i] = sv->sh[i].nsend;
    ok = lsb_p2p_create_virtual(sv->p2p, sv->nshard, rv, nr, sd, ns) == 0;
    free(rv), free(sd), free(nr), free(ns);
  }
  unsigned mine[3] = {(unsigned)ok, 0, 0}, *all = lsb_calloc(unsigned, 3 * (size_t)P);
#define AGREE() (sv->dist ? (void)lsb_hip_comm_allgather_u32(mine, 3, all) : (void)memcpy(all, mine, sizeof mine))
  AGREE();
  for (int q = 0; q < P; q++)
    ok &= all[3 * q] != 0;
  if (ok) {
    unsigned *d_bad = (unsigned *)lsb_hip_malloc(sizeof(unsigned)), bad = 0;
    LSB_CHK_HIP(hipMemsetAsync(d_bad, 0, sizeof(unsigned), g_stream));
    for (int i = 0; i < sv->nshard; i++)
      LSB_CHK_HIP(hipMemsetAsync(sv->sh[i].d_st, 0, sizeof(struct lsb_pcg_state), g_stream));
    sv->p2p_on = 1, sv->p2p_halo = lsb_p2p_has_halo(sv->p2p[0]);
    if (sv->dist)
      lsb_hip_comm_barrier();
    p2p_rounds(sv, P2P_ROUNDS, 1, d_bad);
    LSB_CHK_HIP(hipStreamSynchronize(g_stream));
    lsb_hip_memcpy_d2h(&bad, d_bad, sizeof(unsigned));
    for (int i = 0; i < sv->nshard; i++) {
      struct lsb_pcg_state hst;
      lsb_hip_memcpy_d2h(&hst, sv->sh[i].d_st, sizeof hst);
      bad += hst.status != 0;
    }
    lsb_hip_free(d_bad);
    mine[0] = bad == 0;
    AGREE(); /* nobody times a path somebody saw fail */
    for (int q = 0; q < P; q++)
      ok &= all[3 * q] != 0;
    for (int i = 0; i < sv->nshard; i++)
      LSB_CHK_HIP(hipMemsetAsync(sv->sh[i].d_st, 0, sizeof(struct lsb_pcg_state), g_stream));
    if (ok) {
      sv->p2p_us = timed_rounds(sv);
      const int halo = sv->p2p_halo;
      sv->p2p_on = sv->p2p_halo = 0;
      sv->rccl_us = timed_rounds(sv);
      sv->p2p_on = 1, sv->p2p_halo = halo;
    }
    mine[0] = (unsigned)ok, mine[1] = (unsigned)(sv->p2p_us * 1e3),
    mine[2] = (unsigned)(sv->rccl_us * 1e3);
    AGREE();
    unsigned tp = 0, tr = 0;
    for (int q = 0; q < P; q++) {
      tp = all[3 * q + 1] > tp ? all[3 * q + 1] : tp;
      tr = all[3 * q + 2] > tr ? all[3 * q + 2] : tr;
    }
    if (sv->o.comm == LSB_COMM_AUTO && tp >= tr)
      ok = 0;
    if (sv->o.verbose)
      fprintf(stderr, "hip_cdna4: direct xGMI path %s: %u mismatches here, exchange+all-reduce "
                      "%.1f us vs %.1f us over RCCL -> %s\n",
              lsb_p2p_has_halo(sv->p2p[0]) ? "(halos + all-reduce)" : "(all-reduce only)", bad,
              tp * 1e-3, tr * 1e-3, ok ? "used" : "not used");
  }
#undef AGREE
  free(all);
  if (!ok) {
    if (sv->o.comm == LSB_COMM_P2P)
      errx(EXIT_FAILURE, "hip_cdna4: comm = p2p requested, but the direct xGMI path is not "
                         "available or failed its self-test");
    sv->p2p_on = sv->p2p_halo = 0;
    for (int i = 0; i < sv->nshard; i++)
      lsb_p2p_destroy(sv->p2p[i]);
    free(sv->p2p), sv->p2p = NULL;
  }
  for (int i = 0; i < sv->nshard; i++)
    LSB_CHK_HIP(hipMemsetAsync(sv->sh[i].d_st, 0, sizeof(struct lsb_pcg_state), g_stream));
  LSB_CHK_HIP(hipStreamSynchronize(g_stream));
}

lsb_hip_solver *lsb_hip_solver_create(const struct csr *A,
                                      const struct lsb_hip_opts *o_in) {
  if (!initialized || !A || A->nrows == 0)
    return NULL;
  struct lsb_hip_opts o;
  if (o_in)
    o = *o_in;
  else
    lsb_hip_get_opts(&o);
  /* the operator, 0-based, both triangles */
  struct csr *S = o.op_mode == LSB_OP_CHOLMOD_UPPER ? lsb_csr_symmetrize_upper(A)
                                                    : lsb_csr_copy_base0(A);
  int P = o.nvirt > 1 ? o.nvirt : 1;
  if ((unsigned)P > S->nrows / 2)
    P = 1;
  lsb_hip_solver *sv = solver_alloc(P, &o);
  if (o.reorder) {
    /* Q = RCM(S); S <- Q S Q^T (src/cusparse.c:67-97) */
    unsigned *perm = (unsigned *)malloc((size_t)S->nrows * sizeof(unsigned));
    if (!perm || lsb_csr_rcm(S, perm))
      errx(EXIT_FAILURE, "hip_cdna4: out of memory computing the RCM ordering");
    struct csr *Sp = lsb_csr_permute_sym(S, perm);
    if (o.verbose)
      fprintf(stderr, "hip_cdna4: RCM bandwidth %u -> %u\n", lsb_csr_bandwidth(S),
              lsb_csr_bandwidth(Sp));
    lsbench_matrix_free(S);
    S = Sp;
    sv->d_perm = (int *)dev_upload(perm, (size_t)S->nrows * sizeof(int));
    LSB_CHK_HIP(hipStreamSynchronize(g_stream));
    free(perm);
    sv->d_bp = (double *)lsb_hip_malloc((size_t)S->nrows * sizeof(double));
    sv->d_xp = (double *)lsb_hip_malloc((size_t)S->nrows * sizeof(double));
  }
  sv->n_glob = sv->n_here = S->nrows, sv->row_first = 0;
  sv->dist = 0, sv->multi = P > 1;
  unsigned *bounds = lsb_calloc(unsigned, (size_t)P + 1);
  lsb_csr_partition_rows(S, (unsigned)P, bounds);
  unsigned *hull = lsb_calloc(unsigned, 4 * (size_t)P);
  for (int q = 0; q < P; q++) {
    shard_upload(&sv->sh[q], S, bounds[q], bounds[q + 1], 0, bounds[q], S->nrows, &o);
    hull[4 * q] = bounds[q], hull[4 * q + 1] = bounds[q + 1] - bounds[q];
    hull[4 * q + 2] = sv->sh[q].col_lo, hull[4 * q + 3] = sv->sh[q].col_hi;
  }
  for (int q = 0; q < P; q++) {
    plan_exchange(&sv->sh[q], q, P, hull);
    for (int k = 0; k < sv->sh[q].nrecv; k++)
      if (sv->sh[q].recv[k].count > sv->agree_halo)
        sv->agree_halo = (unsigned)sv->sh[q].recv[k].count;
  }
  free(hull), free(bounds);
  lsbench_matrix_free(S);
  solver_finish_setup(sv);
  return sv;
}

lsb_hip_solver *lsb_hip_solver_create_dist(const struct csr *A_rows,
                                           unsigned row_begin,
                                           unsigned n_global,
                                           const struct lsb_hip_opts *o_in) {
  if (!initialized || !A_rows)
    return NULL;
  struct lsb_hip_opts o;
  if (o_in)
    o = *o_in;
  else
    lsb_hip_get_opts(&o);
  const int P = lsb_hip_comm_size(), me = lsb_hip_comm_rank();
  lsb_hip_solver *sv = solver_alloc(1, &o);
  sv->n_glob = n_global, sv->n_here = A_rows->nrows, sv->row_first = row_begin;
  sv->dist = P > 1, sv->multi = P > 1;
  shard_upload(&sv->sh[0], A_rows, 0, 0, 1, row_begin, n_global, &o);
  unsigned mine[4] = {row_begin, A_rows->nrows, sv->sh[0].col_lo, sv->sh[0].col_hi};
  unsigned *hull = lsb_calloc(unsigned, 4 * (size_t)P);
  lsb_hip_comm_allgather_u32(mine, 4, hull);
  /* every rank must enqueue the SAME number of iterations between polls (the
   * collectives inside have to pair up), so the chunk size is derived from
   * numbers all ranks agree on: the largest shard */
  {
    unsigned nz = (unsigned)sv->sh[0].nnz, *allnz = lsb_calloc(unsigned, (size_t)P);
    lsb_hip_comm_allgather_u32(&nz, 1, allnz);
    for (int q = 0; q < P; q++) {
      if (allnz[q] > sv->agree_nnz)
        sv->agree_nnz = allnz[q];
      if (hull[4 * q + 1] > sv->agree_n)
        sv->agree_n = hull[4 * q + 1];
    }
    free(allnz);
  }
  /* sanity: the shards must tile [0, n_global) in rank order */
  unsigned expect = 0;
  int full = 1, equal = 1;
  for (int q = 0; q < P; q++) {
    if (hull[4 * q] != expect)
      errx(EXIT_FAILURE, "hip_cdna4: rank %d owns rows from %u, expected %u "
                         "(row ranges must tile the operator in rank order)",
           q, hull[4 * q], expect);
    expect += hull[4 * q + 1];
    full &= hull[4 * q + 2] == 0 && hull[4 * q + 3] == n_global;
    equal &= hull[4 * q + 1] == hull[1];
  }
  if (expect != n_global)
    errx(EXIT_FAILURE, "hip_cdna4: shards cover %u rows, operator has %u", expect, n_global);
  plan_exchange(&sv->sh[0], me, P, hull);
  if (P > 1 && full && equal) {
    /* every shard references every row: the north-star all-gather of x */
    sv->sh[0].nsend = 1, sv->sh[0].send[0].peer = -1;
    sv->sh[0].send[0].offset = row_begin, sv->sh[0].send[0].count = A_rows->nrows;
    sv->sh[0].nrecv = 0;
  }
  free(hull);
  {
    unsigned h = 0, *allh = lsb_calloc(unsigned, (size_t)P);
    for (int k = 0; k < sv->sh[0].nrecv; k++)
      if (sv->sh[0].recv[k].count > h)
        h = (unsigned)sv->sh[0].recv[k].count;
    lsb_hip_comm_allgather_u32(&h, 1, allh);
    for (int q = 0; q < P; q++)
      if (allh[q] > sv->agree_halo)
        sv->agree_halo = allh[q];
    free(allh);
  }
  solver_finish_setup(sv);
  return sv;
}

void lsb_hip_solver_destroy(lsb_hip_solver *sv) {
  if (!sv)
    return;
  LSB_CHK_HIP(hipStreamSynchronize(g_stream));
  drop_graphs(sv);
  if (sv->p2p) {
    if (sv->dist) /* no peer may still be storing into a mailbox that goes away */
      lsb_hip_comm_barrier();
    for (int i = 0; i < sv->nshard; i++)
      lsb_p2p_destroy(sv->p2p[i]);
    free(sv->p2p);
  }
  for (int i = 0; i < sv->nshard; i++)
    shard_free(&sv->sh[i]);
  if (sv->have_events) {
    for (int i = 0; i < 4 * MAX_SAMPLES; i++)
      LSB_CHK_HIP(hipEventDestroy(sv->ev[i]));
    LSB_CHK_HIP(hipEventDestroy(sv->ev_t0));
    LSB_CHK_HIP(hipEventDestroy(sv->ev_t1));
    LSB_CHK_HIP(hipEventDestroy(sv->ev_poll[0]));
    LSB_CHK_HIP(hipEventDestroy(sv->ev_poll[1]));
    LSB_CHK_HIP(hipEventDestroy(sv->ev_vec));
    LSB_CHK_HIP(hipEventDestroy(sv->ev_halo));
  }
  lsb_hip_free(sv->d_scal_all), lsb_hip_free(sv->d_tmp);
  lsb_hip_free(sv->d_perm), lsb_hip_free(sv->d_bp), lsb_hip_free(sv->d_xp);
  for (int i = 0; sv->gm && i < sv->nshard; i++) {
    lsb_hip_free(sv->gm[i].V), lsb_hip_free(sv->gm[i].parts);
    lsb_hip_free(sv->gm[i].ax), lsb_hip_free(sv->gm[i].st);
  }
  free(sv->gm);
  lsb_hip_free(sv->gm_red);
  if (sv->gm_hst)
    LSB_CHK_HIP(hipHostFree(sv->gm_hst));
  LSB_CHK_HIP(hipHostFree(sv->h_st));
  free(sv->sh), free(sv);
}

unsigned lsb_hip_solver_nrows_local(const lsb_hip_solver *s) { return s->n_here; }
unsigned lsb_hip_solver_nrows_global(const lsb_hip_solver *s) { return s->n_glob; }
unsigned long long lsb_hip_solver_nnz_local(const lsb_hip_solver *s) {
  unsigned long long z = 0;
  for (int i = 0; i < s->nshard; i++)
    z += s->sh[i].nnz;
  return z;
}
unsigned lsb_hip_solver_nblocks(const lsb_hip_solver *s) { return s->sh[0].nblk; }
int lsb_hip_solver_spmv_variant(const lsb_hip_solver *s) { return s->sh[0].variant; }
unsigned lsb_hip_solver_spmv_flags(const lsb_hip_solver *s) { return s->sh[0].sp_flags; }
unsigned lsb_hip_solver_spmv_grid(const lsb_hip_solver *s) { return s->sh[0].sp_grid; }
static int can_overlap(const lsb_hip_solver *sv);
int lsb_hip_solver_overlaps(const lsb_hip_solver *s) { return can_overlap(s); }
int lsb_hip_solver_comm(const lsb_hip_solver *s, double *p2p_us, double *rccl_us) {
  if (p2p_us)
    *p2p_us = s->p2p_us;
  if (rccl_us)
    *rccl_us = s->rccl_us;
  return !s->multi ? 0 : !s->p2p_on ? 1 : s->p2p_halo ? 3 : 2;
}

/* ------------------------------------------------------------------------ */
/* communication steps: RCCL between processes, device copies between the     */
/* virtual shards of one process                                              */
/* ------------------------------------------------------------------------ */
static void exchange_on(lsb_hip_solver *sv, hipStream_t stream) {
  if (sv->dist) {
    struct shard *s = &sv->sh[0];
    lsb_hip_comm_exchange(s->d_pfull, s->send, s->nsend, s->recv, s->nrecv, stream);
    return;
  }
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    for (int k = 0; k < s->nrecv; k++) {
      const struct lsb_xfer *x = &s->recv[k];
      LSB_CHK_HIP(hipMemcpyAsync(s->d_pfull + x->offset,
                                 sv->sh[x->peer].d_pfull + x->offset,
                                 x->count * sizeof(double), hipMemcpyDeviceToDevice,
                                 stream));
    }
  }
}

/* 1: communicate although no solve is running (the device state's status is
 * whatever the last solve left there) */
static int g_ar_nostate;
static void exchange_p(lsb_hip_solver *sv) {
  if (sv->p2p_halo && sv->dist) { /* peers are other GPUs: both roles in one launch */
    lsb_p2p_sendrecv(sv->p2p[0], sv->sh[0].d_pfull, g_ar_nostate ? NULL : sv->sh[0].d_st,
                     g_stream);
    return;
  }
  if (sv->p2p_halo) { /* all sends before any wait: virtual shards share a stream */
    for (int i = 0; i < sv->nshard; i++)
      lsb_p2p_send(sv->p2p[i], sv->sh[i].d_pfull, g_ar_nostate ? NULL : sv->sh[i].d_st, g_stream);
    for (int i = 0; i < sv->nshard; i++)
      lsb_p2p_recv(sv->p2p[i], sv->sh[i].d_pfull, g_ar_nostate ? NULL : sv->sh[i].d_st, g_stream);
    return;
  }
  exchange_on(sv, g_stream);
}

/* d_scal[off .. off+cnt) <- sum over shards.  With the direct path the
 * shard's own partial sums are folded into the same launch: the first `width`
 * values come from the SpMV's dot partials, the next s->ar2_width from the
 * array the sweep kernel left in s->ar2_parts, and only the rest must already
 * sit, reduced, in d_scal. */
static void allreduce_parts(lsb_hip_solver *sv, unsigned off, unsigned cnt, unsigned width,
                            int with2) {
  for (int ph = 1; ph <= 2; ph++)
    for (int i = 0; i < sv->nshard; i++) {
      struct shard *s = &sv->sh[i];
      const double *parts = width ? s->d_parts_pq : NULL;
      const unsigned w2 = with2 ? s->ar2_width : 0;
      struct lsb_pcg_state *st = g_ar_nostate ? NULL : s->d_st;
      const int phases = sv->nshard == 1 ? 3 : ph;
      if (sv->nshard == 1 && ph == 2)
        continue;
      lsb_p2p_allreduce(sv->p2p[i], parts, s->npq, width, s->ar2_parts, s->ar2_n, w2,
                        s->d_scal + off + width + w2, cnt - width - w2, s->d_scal + off, st,
                        phases, g_stream);
    }
}

static void allreduce_scal(lsb_hip_solver *sv, unsigned off, unsigned cnt) {
  if (sv->p2p_on) {
    allreduce_parts(sv, off, cnt, 0, 0);
    return;
  }
  if (sv->dist)
    lsb_hip_comm_allreduce_stream(sv->sh[0].d_scal + off, (int)cnt, g_stream);
  else if (sv->nshard > 1)
    lsb_k_vreduce(sv->d_scal_all, SCAL_STRIDE, (unsigned)sv->nshard, off, cnt, g_stream);
}

/* d_scal[0] <- all-reduced sum of the SpMV's dot partials; d_scal[1..cnt) are
 * all-reduced along with it */
static void allreduce_pq(lsb_hip_solver *sv, unsigned cnt, int with2) {
  if (sv->p2p_on) {
    allreduce_parts(sv, 0, cnt, 1, with2);
    return;
  }
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    lsb_k_reduce_final(s->d_parts_pq, s->npq, 1, s->d_scal + 0, 0, s->d_st, g_stream);
  }
  allreduce_scal(sv, 0, cnt);
}

static void sell_launch(struct shard *s, unsigned s0, unsigned ns, const double *xfull, double *y,
                        const double *xdot, double *partials, unsigned *np,
                        const struct lsb_pcg_state *st) {
  if ((s->sp_flags & LSB_SP_C16) && s->d_scodes)
    lsb_k_spmv_sell(s->sp_flags, s->sp_grid, s->d_sptr16, s0, ns, s->n, s->row_begin, s->d_scodes,
                    s->d_sbase, s->d_svals16, xfull, y, xdot, partials, np, st, g_stream);
  else
    lsb_k_spmv_sell(s->sp_flags & ~LSB_SP_C16, s->sp_grid, s->d_sptr, s0, ns, s->n, s->row_begin,
                    s->d_scols, NULL, s->d_svals, xfull, y, xdot, partials, np, st, g_stream);
}

static void spmv_shard(struct shard *s, const double *xfull, double *y,
                       const double *xdot, double *partials, unsigned *np,
                       const struct lsb_pcg_state *st) {
  if (s->variant == LSB_SPMV_PANEL) {
    /* y = 0, then one launch per column panel accumulates into it: inside a
     * launch every XCD gathers from the same 2 MiB slice of x, out of its L2 */
    LSB_CHK_HIP(hipMemsetAsync(y, 0, (size_t)s->n * sizeof(double), g_stream));
    for (unsigned p = 0; p < s->pn; p++) {
      const unsigned b0 = s->h_pblk[p], nb = s->h_pblk[p + 1] - b0;
      if (nb)
        lsb_k_spmv(LSB_SPMV_ADAPTIVE, s->n, s->pd_offs, s->pd_cols, s->pd_vals,
                   s->pd_rowblk + b0, s->pd_blklanes + b0, nb, s->lanes, s->sp_flags,
                   s->sp_grid, xfull, y, NULL, NULL, NULL, st, s->pd_rowmap, g_stream);
    }
    if (partials)
      lsb_k_dot(s->n, y, xdot, partials, np, g_stream);
    return;
  }
  if (s->variant == LSB_SPMV_SELL) {
    sell_launch(s, 0, s->nslice, xfull, y, xdot, partials, np, st);
    return;
  }
  lsb_k_spmv(s->variant, s->n, s->d_offs, s->d_cols, s->d_vals, s->d_rowblk, s->d_blklanes,
             s->nblk, s->lanes, s->sp_flags, s->sp_grid, xfull, y, xdot, partials, np, st,
             NULL, g_stream);
}

/*
 * Exchange + SpMV of one iteration with the halo transfer hidden behind the
 * rows that do not need it: the exchange runs on its
 * own stream as soon as the vector is final, the interior row blocks start at
 * once on the compute stream, the boundary blocks wait for the halo.  Three
 * launches of the same kernel on sub-ranges of the row blocks; their partial
 * sums land in consecutive regions of the shard's partial buffer.
 */
static int can_overlap(const lsb_hip_solver *sv) {
  if (!sv->multi || !sv->o.overlap)
    return 0;
  for (int i = 0; i < sv->nshard; i++) {
    const struct shard *s = &sv->sh[i];
    if (!(s->variant == LSB_SPMV_ADAPTIVE && s->ov_ok) && !(s->variant == LSB_SPMV_SELL && s->ov_sok))
      return 0;
  }
  /* auto: the split SpMV costs 2 launches (direct path) or 2 launches and two
   * cross-stream events (RCCL), 6-20 us; a halo of >= 64 Ki doubles takes
   * longer than that on one xGMI link.  agree_halo is the largest halo of ANY
   * rank, so every rank takes the same branch. */
  if (sv->o.overlap < 0)
    return sv->agree_halo >= 65536u;
  return 1;
}

/* part 0: the rows that need no halo; 1 / 2: the ones before / after them */
static void spmv_range(struct shard *s, int part, double *y, double *partials, unsigned *np,
                       const struct lsb_pcg_state *st) {
  *np = 0;
  if (s->variant == LSB_SPMV_SELL) {
    const unsigned b0 = part == 0 ? s->ov_s1 : part == 1 ? 0 : s->ov_s2;
    const unsigned b1 = part == 0 ? s->ov_s2 : part == 1 ? s->ov_s1 : s->nslice;
    if (b1 > b0)
      sell_launch(s, b0, b1 - b0, s->d_pfull, y, s->d_pfull + s->row_begin, partials, np, st);
    return;
  }
  const unsigned b0 = part == 0 ? s->ov_b1 : part == 1 ? 0 : s->ov_b2;
  const unsigned b1 = part == 0 ? s->ov_b2 : part == 1 ? s->ov_b1 : s->nblk;
  if (b1 > b0)
    lsb_k_spmv(LSB_SPMV_ADAPTIVE, s->n, s->d_offs, s->d_cols, s->d_vals, s->d_rowblk + b0,
               s->d_blklanes + b0, b1 - b0, s->lanes, s->sp_flags, s->sp_grid, s->d_pfull, y,
               s->d_pfull + s->row_begin, partials, np, st, NULL, g_stream);
}

static void exchange_and_spmv(lsb_hip_solver *sv, int sample) {
  if (!can_overlap(sv)) {
    exchange_p(sv);
    for (int i = 0; i < sv->nshard; i++) {
      struct shard *s = &sv->sh[i];
      if (i == 0 && sample >= 0)
        LSB_CHK_HIP(hipEventRecord(sv->ev[4 * sample], g_stream));
      spmv_shard(s, s->d_pfull, s->d_q, s->d_pfull + s->row_begin, s->d_parts_pq, &s->npq, s->d_st);
      if (i == 0 && sample >= 0) {
        LSB_CHK_HIP(hipEventRecord(sv->ev[4 * sample + 1], g_stream));
        LSB_CHK_HIP(hipEventRecord(sv->ev[4 * sample + 2], g_stream));
        LSB_CHK_HIP(hipEventRecord(sv->ev[4 * sample + 3], g_stream));
      }
    }
    return;
  }
  if (sv->p2p_halo) { /* direct stores to the peers: no second stream needed */
    for (int i = 0; i < sv->nshard; i++)
      lsb_p2p_send(sv->p2p[i], sv->sh[i].d_pfull, sv->sh[i].d_st, g_stream);
  } else {
    LSB_CHK_HIP(hipEventRecord(sv->ev_vec, g_stream));          /* the vector is final   */
    LSB_CHK_HIP(hipStreamWaitEvent(g_comm_stream, sv->ev_vec, 0));
    exchange_on(sv, g_comm_stream);
    LSB_CHK_HIP(hipEventRecord(sv->ev_halo, g_comm_stream));    /* the halo has landed   */
  }
  if (sample >= 0)
    LSB_CHK_HIP(hipEventRecord(sv->ev[4 * sample], g_stream));
  unsigned na, nb, nc;
  for (int i = 0; i < sv->nshard; i++) {                         /* interior: no halo     */
    struct shard *s = &sv->sh[i];
    spmv_range(s, 0, s->d_q, s->d_parts_pq, &na, s->d_st);
    s->npq = na;
  }
  if (sv->p2p_halo) {
    for (int i = 0; i < sv->nshard; i++)
      lsb_p2p_recv(sv->p2p[i], sv->sh[i].d_pfull, sv->sh[i].d_st, g_stream);
  } else
    LSB_CHK_HIP(hipStreamWaitEvent(g_stream, sv->ev_halo, 0));
  for (int i = 0; i < sv->nshard; i++) {                         /* boundary rows         */
    struct shard *s = &sv->sh[i];
    spmv_range(s, 1, s->d_q, s->d_parts_pq + s->npq, &nb, s->d_st);
    spmv_range(s, 2, s->d_q, s->d_parts_pq + s->npq + nb, &nc, s->d_st);
    s->npq += nb + nc;
  }
  if (sample >= 0) {
    LSB_CHK_HIP(hipEventRecord(sv->ev[4 * sample + 1], g_stream));
    LSB_CHK_HIP(hipEventRecord(sv->ev[4 * sample + 2], g_stream));
    LSB_CHK_HIP(hipEventRecord(sv->ev[4 * sample + 3], g_stream));
  }
}

/*
 * Pick the adaptive SpMV's flavour for this operator by timing it (setup is
 * untimed, like the reference's csr_init): {plain, prefetch, nontemporal,
 * both}, 3 launches each after one warm-up, on the
 * shard's own matrix with the dot product fused as in the solve.  Which one
 * wins depends on how much of x's gather window survives in L2 next to the
 * matrix stream: on the 10M-row 5-point operator nontemporal stream loads win
 * by 15%, on the 7-point 256^3 one prefetch without nontemporal does.
 */
static void tune_spmv(lsb_hip_solver *sv, struct shard *s) {
  const struct lsb_hip_opts *o = &sv->o;
  s->sp_flags = LSB_SP_PREFETCH | LSB_SP_NT;
  s->sp_grid = o->spmv_grid > 0 ? (unsigned)o->spmv_grid : LSB_MAX_PARTIALS;
  if (o->spmv_tune >= 0) {
    s->sp_flags = (unsigned)o->spmv_tune & 7u; /* bit 2: 16-bit codes, where that copy exists */
    return;
  }
  if (s->variant == LSB_SPMV_SELL && !s->d_sptr)
    s->variant = LSB_SPMV_ADAPTIVE; /* the operator did not qualify for the copy */
  if ((s->variant != LSB_SPMV_ADAPTIVE && s->variant != LSB_SPMV_PANEL &&
       s->variant != LSB_SPMV_SELL) ||
      s->nnz < 4000000ull)
    return; /* small operators are launch-latency bound: nothing to tune */
  float best = 1e30f;
  unsigned bf = s->sp_flags, np;
  int bv = s->variant;
  const unsigned grid0 = s->sp_grid;
  unsigned bg = grid0;
  /* candidates {form, flags, grid}: the form asked for, or (auto) every form
   * this shard has; the sliced-ELL kernels have no prefetch flavour, they try
   * 6 instead of 8 resident workgroups per CU instead */
  struct {
    int v;
    unsigned f, g;
  } cand[16];
  int ncand = 0;
  const int any = o->spmv_variant == LSB_SPMV_AUTO;
  if (any || s->variant == LSB_SPMV_ADAPTIVE)
    for (unsigned f = 0; f < 4; f++)
      cand[ncand].v = LSB_SPMV_ADAPTIVE, cand[ncand].f = f, cand[ncand++].g = grid0;
  if (s->pn && (any || s->variant == LSB_SPMV_PANEL))
    for (unsigned f = 0; f < 4; f++)
      cand[ncand].v = LSB_SPMV_PANEL, cand[ncand].f = f, cand[ncand++].g = grid0;
  if (s->d_sptr && (any || s->variant == LSB_SPMV_SELL))
    for (unsigned c16 = 0; c16 <= (s->d_scodes ? LSB_SP_C16 : 0u); c16 += LSB_SP_C16) {
      cand[ncand].v = LSB_SPMV_SELL, cand[ncand].f = c16 | LSB_SP_NT, cand[ncand++].g = grid0;
      cand[ncand].v = LSB_SPMV_SELL, cand[ncand].f = c16, cand[ncand++].g = grid0;
      if (o->spmv_grid <= 0)
        cand[ncand].v = LSB_SPMV_SELL, cand[ncand].f = c16 | LSB_SP_NT, cand[ncand++].g = 1536;
    }
  for (int ci = 0; ci < ncand; ci++) {
    s->variant = cand[ci].v, s->sp_flags = cand[ci].f, s->sp_grid = cand[ci].g;
    spmv_shard(s, s->d_pfull, s->d_q, s->d_pfull + s->row_begin, s->d_parts_pq, &np, NULL);
    LSB_CHK_HIP(hipEventRecord(sv->ev_t0, g_stream));
    for (int r = 0; r < 3; r++)
      spmv_shard(s, s->d_pfull, s->d_q, s->d_pfull + s->row_begin, s->d_parts_pq, &np, NULL);
    LSB_CHK_HIP(hipEventRecord(sv->ev_t1, g_stream));
    LSB_CHK_HIP(hipEventSynchronize(sv->ev_t1));
    float ms = 0.f;
    LSB_CHK_HIP(hipEventElapsedTime(&ms, sv->ev_t0, sv->ev_t1));
    if (o->verbose > 1)
      fprintf(stderr, "hip_cdna4: spmv tune form=%d flags=%u grid=%u: %.1f us\n", s->variant,
              s->sp_flags, s->sp_grid, ms * 1e3f / 3);
    if (ms < best)
      best = ms, bf = s->sp_flags, bv = s->variant, bg = s->sp_grid;
  }
  s->sp_grid = bg;
  s->variant = bv;
  s->sp_flags = bf;
  /* the copies that lost are not kept */
  if (!(bv == LSB_SPMV_SELL && (bf & LSB_SP_C16))) {
    lsb_hip_free(s->d_sptr16), lsb_hip_free(s->d_scodes), lsb_hip_free(s->d_sbase);
    lsb_hip_free(s->d_svals16);
    s->d_sptr16 = NULL, s->d_scodes = NULL, s->d_sbase = NULL, s->d_svals16 = NULL;
  }
  if (any && !(bv == LSB_SPMV_SELL && !(bf & LSB_SP_C16))) {
    lsb_hip_free(s->d_scols), lsb_hip_free(s->d_svals);
    s->d_scols = NULL, s->d_svals = NULL;
    if (bv != LSB_SPMV_SELL)
      lsb_hip_free(s->d_sptr), s->d_sptr = NULL;
  }
}

/* ------------------------------------------------------------------------ */
/* PCG                                                                       */
/* ------------------------------------------------------------------------ */
static int use_cg1(const lsb_hip_solver *sv);
static void cg1_enqueue_init(lsb_hip_solver *sv, const double *d_b, double *d_x);
static void cg1_enqueue_iter(lsb_hip_solver *sv, double *d_x, int parity, int sample);

static void pcg_enqueue_init(lsb_hip_solver *sv, const double *d_b, double *d_x) {
  if (use_cg1(sv)) {
    cg1_enqueue_init(sv, d_b, d_x);
    return;
  }
  unsigned np2 = 0;
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    const size_t o = s->row_begin - sv->row_first;
    lsb_k_pcg_init(s->n, d_b + o, s->d_dinv, d_x + o, s->d_r, s->d_pfull + s->row_begin,
                   s->d_parts2, &np2, g_stream);
    if (sv->multi)
      lsb_k_reduce_final(s->d_parts2, np2, 2, s->d_scal + 1, 0, NULL, g_stream);
  }
  if (sv->multi) {
    g_ar_nostate = 1; /* the device state still holds the previous solve's status */
    allreduce_scal(sv, 1, 2);
    g_ar_nostate = 0;
  }
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    if (sv->multi)
      lsb_k_pcg_init_state(s->d_st, s->d_scal + 1, 1, sv->o.tol, (int)sv->o.maxit, g_stream);
    else
      lsb_k_pcg_init_state(s->d_st, s->d_parts2, np2, sv->o.tol, (int)sv->o.maxit, g_stream);
  }
}

/* One PCG iteration, enqueued.  sample >= 0: bracket the SpMV of shard 0 with
 * events 4*sample .. 4*sample+3. */
static void pcg_enqueue_iter(lsb_hip_solver *sv, double *d_x, int parity, int sample) {
  if (use_cg1(sv)) {
    cg1_enqueue_iter(sv, d_x, parity, sample);
    return;
  }
  unsigned npq = 0, np2 = 0;
  if (sv->multi) {
    exchange_and_spmv(sv, sample);
    allreduce_pq(sv, 1, 0);
  }
  for (int i = 0; i < sv->nshard && !sv->multi; i++) {
    struct shard *s = &sv->sh[i];
    if (i == 0 && sample >= 0)
      LSB_CHK_HIP(hipEventRecord(sv->ev[4 * sample], g_stream));
    spmv_shard(s, s->d_pfull, s->d_q, s->d_pfull + s->row_begin, s->d_parts_pq, &npq, s->d_st);
    if (i == 0 && sample >= 0)
    {
      /* e1 closes the SpMV interval; e2,e3 bracket NOTHING: their distance is
       * what one event marker costs in this very spot of the stream, and is
       * subtracted from e0->e1 (an event pair around a kernel otherwise reads
       * ~9 us longer than the kernel's duration in a rocprofv3 trace). */
      LSB_CHK_HIP(hipEventRecord(sv->ev[4 * sample + 1], g_stream));
      LSB_CHK_HIP(hipEventRecord(sv->ev[4 * sample + 2], g_stream));
      LSB_CHK_HIP(hipEventRecord(sv->ev[4 * sample + 3], g_stream));
    }
  }
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    const size_t o = s->row_begin - sv->row_first;
    lsb_k_pcg_update_xr(s->n, s->d_pfull + s->row_begin, s->d_q, s->d_dinv, d_x + o, s->d_r,
                        s->d_st, parity, sv->multi ? s->d_scal : s->d_parts_pq,
                        sv->multi ? 1u : npq, s->d_parts2, &np2, g_stream);
    if (sv->multi)
      lsb_k_reduce_final(s->d_parts2, np2, 2, s->d_scal + 1, 0, s->d_st, g_stream);
  }
  if (sv->multi)
    allreduce_scal(sv, 1, 2);
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    lsb_k_pcg_update_p(s->n, s->d_r, s->d_dinv, s->d_pfull + s->row_begin, s->d_st, parity,
                       sv->multi ? s->d_scal + 1 : s->d_parts2, sv->multi ? 1u : np2,
                       g_stream);
  }
}

/* ---- single-reduction CG (LSB_KRYLOV_PCG1): see k_cg1_update -------------- */
static int use_cg1(const lsb_hip_solver *sv) {
  if (sv->o.krylov == LSB_KRYLOV_PCG1)
    return 1;
  if (sv->o.krylov != LSB_KRYLOV_AUTO)
    return 0;
  /* measured on one GPU: no gain for small operators (tests/xn3b_A_18.txt: 390 vs
   * 400 solves/s, the fused sweep is as long as the two it replaces) and +6 % time
   * on the 10M-row operator (96 n vs 88 n bytes); what it saves is a collective */
  return sv->multi;
}

static void cg1_enqueue_init(lsb_hip_solver *sv, const double *d_b, double *d_x) {
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    const size_t o = s->row_begin - sv->row_first, bytes = (size_t)s->n * sizeof(double);
    if (!s->d_p1) {
      s->d_p1 = (double *)lsb_hip_malloc(bytes);
      s->d_s1 = (double *)lsb_hip_malloc(bytes);
    }
    /* x = 0, r = b, u = D^-1 b (into the gather vector), partials (r.u, b.b) */
    lsb_k_pcg_init(s->n, d_b + o, s->d_dinv, d_x + o, s->d_r, s->d_pfull + s->row_begin,
                   s->d_parts2, &s->np2, g_stream);
    LSB_CHK_HIP(hipMemsetAsync(s->d_p1, 0, bytes, g_stream));
    LSB_CHK_HIP(hipMemsetAsync(s->d_s1, 0, bytes, g_stream));
    if (sv->multi)
      lsb_k_reduce_final(s->d_parts2, s->np2, 2, s->d_scal + 1, 0, NULL, g_stream);
  }
  if (sv->multi) {
    g_ar_nostate = 1; /* the device state still holds the previous solve's status */
    allreduce_scal(sv, 1, 2);
    g_ar_nostate = 0;
  }
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    if (sv->multi)
      lsb_k_pcg_init_state(s->d_st, s->d_scal + 1, 1, sv->o.tol, (int)sv->o.maxit, g_stream);
    else
      lsb_k_pcg_init_state(s->d_st, s->d_parts2, s->np2, sv->o.tol, (int)sv->o.maxit, g_stream);
  }
  if (sv->multi)
    exchange_and_spmv(sv, -1); /* w = S u, partials w.u */
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    if (!sv->multi)
      spmv_shard(s, s->d_pfull, s->d_q, s->d_pfull + s->row_begin, s->d_parts_pq, &s->npq, s->d_st);
  }
  if (sv->multi)
    allreduce_pq(sv, 1, 0);
}

static void cg1_enqueue_iter(lsb_hip_solver *sv, double *d_x, int parity, int sample) {
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    const size_t o = s->row_begin - sv->row_first;
    double *gr_in = s->d_parts2 + (size_t)parity * 2 * LSB_MAX_PARTIALS;
    double *gr_out = s->d_parts2 + (size_t)(parity ^ 1) * 2 * LSB_MAX_PARTIALS;
    unsigned np2 = 0;
    lsb_k_cg1_update(s->n, s->d_pfull + s->row_begin, s->d_q, s->d_dinv, s->d_p1, s->d_s1,
                     d_x + o, s->d_r, s->d_st, parity, sv->multi ? s->d_scal + 1 : gr_in,
                     sv->multi ? 1u : s->np2, sv->multi ? s->d_scal : s->d_parts_pq,
                     sv->multi ? 1u : s->npq, gr_out, &np2, g_stream);
    s->ar2_parts = gr_out, s->ar2_n = np2, s->ar2_width = 2;
    if (sv->multi && !sv->p2p_on) /* the direct all-reduce reduces these itself */
      lsb_k_reduce_final(gr_out, np2, 2, s->d_scal + 1, 0, s->d_st, g_stream);
  }
  if (sv->multi)
    exchange_and_spmv(sv, sample);
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    if (!sv->multi) {
      if (sample >= 0)
        LSB_CHK_HIP(hipEventRecord(sv->ev[4 * sample], g_stream));
      spmv_shard(s, s->d_pfull, s->d_q, s->d_pfull + s->row_begin, s->d_parts_pq, &s->npq,
                 s->d_st);
      if (sample >= 0) {
        LSB_CHK_HIP(hipEventRecord(sv->ev[4 * sample + 1], g_stream));
        LSB_CHK_HIP(hipEventRecord(sv->ev[4 * sample + 2], g_stream));
        LSB_CHK_HIP(hipEventRecord(sv->ev[4 * sample + 3], g_stream));
      }
    }
  }
  if (sv->multi)
    allreduce_pq(sv, 3, 1); /* w.u, r.u, r.r in ONE collective */
}

static int auto_chunk(const lsb_hip_solver *sv) {
  /* aim at ~0.3 ms of device work per chunk (at least 8 iterations): the poll
   * is pipelined one chunk ahead, so small chunks cost nothing while running
   * and bound the no-op tail enqueued past convergence */
  const struct shard *s = &sv->sh[0];
  double bytes = 12.0 * (double)s->nnz + 108.0 * (double)s->n;
  if (sv->dist) /* must not depend on this rank's own shard size */
    bytes = 12.0 * (double)sv->agree_nnz + 108.0 * (double)sv->agree_n;
  double us = bytes / 4.0e6; /* 4 TB/s => bytes per microsecond */
  if (us < 6.0)
    us = 6.0;
  int c = (int)(300.0 / us);
  if (c < 8)
    c = 8;
  if (c > 256)
    c = 256;
  return c & ~1;
}

/* hipGraph of `iters` PCG iterations writing to d_x; two cached entries (the
 * hinted whole-solve graph and the small continuation chunk). */
static hipGraphExec_t get_graph(lsb_hip_solver *sv, int iters, double *d_x) {
  for (int i = 0; i < 2; i++)
    if (sv->gcache[i].exec && sv->gcache[i].iters == iters && sv->gcache[i].x == d_x)
      return sv->gcache[i].exec;
  const int slot = sv->gnext;
  sv->gnext ^= 1;
  if (sv->gcache[slot].exec)
    LSB_CHK_HIP(hipGraphExecDestroy(sv->gcache[slot].exec));
  hipGraph_t g;
  LSB_CHK_HIP(hipStreamBeginCapture(g_stream, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < iters; i++)
    pcg_enqueue_iter(sv, d_x, i & 1, -1);
  LSB_CHK_HIP(hipStreamEndCapture(g_stream, &g));
  LSB_CHK_HIP(hipGraphInstantiate(&sv->gcache[slot].exec, g, NULL, NULL, 0));
  LSB_CHK_HIP(hipGraphDestroy(g));
  sv->gcache[slot].iters = iters, sv->gcache[slot].x = d_x;
  return sv->gcache[slot].exec;
}

static void drop_graphs(lsb_hip_solver *sv) {
  for (int i = 0; i < 2; i++)
    if (sv->gcache[i].exec) {
      LSB_CHK_HIP(hipGraphExecDestroy(sv->gcache[i].exec));
      sv->gcache[i].exec = NULL;
    }
}

/*
 * Host side of one solve.  The device decides when to stop (lsb_pcg_state);
 * the host only has to enqueue enough iterations and look at the 64-byte state
 * now and then:
 *   - a solver that has solved before enqueues exactly the iteration count of
 *     its previous solve in one go (the benchmark protocol repeats the same
 *     solve `trials` times, src/cholmod-impl.h:44-63) and polls once;
 *   - otherwise, and for whatever is left, chunks of `check_every` iterations
 *     are enqueued one AHEAD of the poll, so the device never waits for the
 *     host; iterations enqueued past convergence are no-op launches.
 */
/*
 * Restarted GMRES(m), right Jacobi preconditioning, x0 = 0 (SURVEY.md section 8
 * a2-6; kernels in hip_gmres.hip).  One restart cycle = up to m inner steps of
 *   z = D^-1 v_j ; w = Op z ; h = V^T w ; w -= V h ; (again: CGS2) ; Givens
 * enqueued in one go; the device closes the cycle early when the residual
 * estimate |g_{j+1}| <= tol ||b||; the host polls the state once per cycle.
 */
#define GM_RED 96
/* sum gm_red[q][off .. off+cnt) over the shards (of all ranks), result in every
 * shard's copy: one collective however many values -- the Gram-Schmidt
 * coefficients of a step travel together */
static void gm_allreduce(lsb_hip_solver *sv, unsigned off, unsigned cnt) {
  if (!sv->multi)
    return;
  if (sv->dist)
    lsb_hip_comm_allreduce_stream(sv->gm_red + off, (int)cnt, g_stream);
  else
    lsb_k_vreduce(sv->gm_red, GM_RED, (unsigned)sv->nshard, off, cnt, g_stream);
}

static int gmres_solve_dev(lsb_hip_solver *sv, const double *d_b, double *d_x,
                           struct lsb_hip_result *res) {
  int m = sv->o.restart;
  if (m < 1)
    m = 1;
  if (m > LSB_GMRES_MAX_RESTART)
    m = LSB_GMRES_MAX_RESTART;
  const int P = sv->nshard;
  if (!sv->gm) {
    sv->gm = lsb_calloc(struct gm_work, P);
    sv->gm_red = (double *)lsb_hip_malloc((size_t)P * GM_RED * sizeof(double));
    LSB_CHK_HIP(hipHostMalloc((void **)&sv->gm_hst, sizeof(struct lsb_gmres_state), 0));
    for (int i = 0; i < P; i++) {
      struct gm_work *w = &sv->gm[i];
      w->parts = (double *)lsb_hip_malloc((size_t)LSB_GMRES_PARTIALS *
                                          (LSB_GMRES_MAX_RESTART + 1) * sizeof(double));
      w->ax = (double *)lsb_hip_malloc((size_t)sv->sh[i].n * sizeof(double));
      w->st = (struct lsb_gmres_state *)lsb_hip_malloc(sizeof(struct lsb_gmres_state));
    }
  }
  if (sv->gm_m != m) {
    for (int i = 0; i < P; i++) {
      struct gm_work *w = &sv->gm[i];
      lsb_hip_free(w->V);
      w->ld = ((size_t)sv->sh[i].n + 1) & ~(size_t)1;
      w->V = (double *)lsb_hip_malloc((size_t)(m + 1) * w->ld * sizeof(double));
    }
    sv->gm_m = m;
  }
  const double t0 = wall_seconds();
  /* every shard carries its own copy of the (identical) small state: residual
   * norm, Hessenberg column, rotations and the stop decision are computed by
   * each from the same all-reduced numbers */
#define EACH(i, s, w)                                                          \
  for (int i = 0; i < P; i++)                                                  \
    for (struct shard *s = &sv->sh[i]; s; s = NULL)                            \
      for (struct gm_work *w = &sv->gm[i]; w; w = NULL)
  EACH(i, s, w) {
    const size_t o = s->row_begin - sv->row_first;
    LSB_CHK_HIP(hipMemsetAsync(w->st, 0, sizeof *w->st, g_stream));
    LSB_CHK_HIP(hipMemsetAsync(d_x + o, 0, (size_t)s->n * sizeof(double), g_stream));
    LSB_CHK_HIP(hipMemsetAsync(w->ax, 0, (size_t)s->n * sizeof(double), g_stream));
  }
  g_ar_nostate = 1; /* exchanges here are not tied to a PCG state */
  for (int cycle = 0;; cycle++) {
    if (cycle > 0) { /* ax = Op x for the restart residual */
      EACH(i, s, w) {
        (void)w;
        LSB_CHK_HIP(hipMemcpyAsync(s->d_pfull + s->row_begin, d_x + (s->row_begin - sv->row_first),
                                   (size_t)s->n * sizeof(double), hipMemcpyDeviceToDevice,
                                   g_stream));
      }
      if (sv->multi)
        exchange_p(sv);
      EACH(i, s, w) spmv_shard(s, s->d_pfull, w->ax, NULL, NULL, NULL, NULL);
    }
    EACH(i, s, w) {
      double *red = sv->gm_red + (size_t)i * GM_RED;
      lsb_k_gm_resid(s->n, d_b + (s->row_begin - sv->row_first), w->ax, w->V, w->parts, w->st,
                     g_stream);
      if (sv->multi)
        lsb_k_reduce_final(w->parts, lsb_k_gm_grid(s->n), 1, red, 0, NULL, g_stream);
    }
    gm_allreduce(sv, 0, 1);
    EACH(i, s, w) {
      double *red = sv->gm_red + (size_t)i * GM_RED;
      lsb_k_gm_begin(w->st, sv->multi ? red : w->parts, sv->multi ? 1u : lsb_k_gm_grid(s->n),
                     sv->o.tol, (int)sv->o.maxit, m, cycle == 0, g_stream);
    }
    for (int j = 0; j < m; j++) {
      /* v_j = (r or w) / norm ; z = D^-1 v_j ; w = Op z */
      EACH(i, s, w) {
        double *vj = w->V + (size_t)j * w->ld;
        lsb_k_gm_scale_prec(s->n, vj, vj, s->d_dinv, s->d_pfull + s->row_begin, w->st, g_stream);
      }
      if (sv->multi)
        exchange_p(sv);
      EACH(i, s, w) spmv_shard(s, s->d_pfull, w->V + (size_t)(j + 1) * w->ld, NULL, NULL, NULL, NULL);
      /* classical Gram-Schmidt, twice (CGS2): h = V^T w ; w -= V h ; h2 likewise */
      for (int pass = 0; pass < 2; pass++) {
        const unsigned off = pass ? 48u : 8u;
        EACH(i, s, w)
          lsb_k_gm_multidot(s->n, w->V, w->ld, j + 1, w->V + (size_t)(j + 1) * w->ld, w->parts,
                            sv->gm_red + (size_t)i * GM_RED + off, 0, w->st, g_stream);
        gm_allreduce(sv, off, (unsigned)j + 1);
        EACH(i, s, w)
          lsb_k_gm_update_w(s->n, w->V, w->ld, j + 1, sv->gm_red + (size_t)i * GM_RED + off,
                            w->V + (size_t)(j + 1) * w->ld, w->parts, w->st, g_stream);
      }
      EACH(i, s, w) /* ||w||^2 partials of the second update */
        if (sv->multi)
          lsb_k_reduce_final(w->parts, lsb_k_gm_grid(s->n), 1, sv->gm_red + (size_t)i * GM_RED, 0,
                             NULL, g_stream);
      gm_allreduce(sv, 0, 1);
      EACH(i, s, w) {
        double *red = sv->gm_red + (size_t)i * GM_RED;
        lsb_k_gm_hess(w->st, j, red + 8, red + 48, sv->multi ? red : w->parts,
                      sv->multi ? 1u : lsb_k_gm_grid(s->n), g_stream);
      }
    }
    EACH(i, s, w)
      lsb_k_gm_finish_cycle(s->n, w->V, w->ld, s->d_dinv, d_x + (s->row_begin - sv->row_first),
                            w->st, g_stream);
    LSB_CHK_HIP(hipMemcpyAsync(sv->gm_hst, sv->gm[0].st, sizeof *sv->gm_hst,
                               hipMemcpyDeviceToHost, g_stream));
    LSB_CHK_HIP(hipStreamSynchronize(g_stream));
    if (sv->gm_hst->status != LSB_STATUS_RUNNING)
      break;
    if ((unsigned)cycle > sv->o.maxit + 2u)
      errx(EXIT_FAILURE, "hip_cdna4: GMRES ran past maxit without a status");
  }
#undef EACH
  g_ar_nostate = 0;
  struct lsb_hip_result r;
  memset(&r, 0, sizeof r);
  r.iters = (unsigned)sv->gm_hst->iters;
  r.status = sv->gm_hst->status;
  r.relres = sv->gm_hst->bnorm > 0.0 ? sv->gm_hst->resid / sv->gm_hst->bnorm : 0.0;
  r.seconds = wall_seconds() - t0;
  if (res)
    *res = r;
  g_last = r;
  return 0;
}

static int solve_core(lsb_hip_solver *sv, const double *d_b, double *d_x,
                      struct lsb_hip_result *res);

/* ||b - S x|| / ||b|| of a finished multi-shard solve, communicating WITHOUT
 * the direct xGMI path; overwrites the search-direction and q vectors. */
static double true_relres(lsb_hip_solver *sv, const double *d_b, const double *d_x) {
  static const double minus_one = -1.0;
  const int on = sv->p2p_on, halo = sv->p2p_halo;
  double rr = 0.0;
  sv->p2p_on = sv->p2p_halo = 0;
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    const size_t o = s->row_begin - sv->row_first;
    LSB_CHK_HIP(hipMemcpyAsync(s->d_pfull + s->row_begin, d_x + o, (size_t)s->n * sizeof(double),
                               hipMemcpyDeviceToDevice, g_stream));
    LSB_CHK_HIP(hipMemcpyAsync(s->d_scal + 5, &minus_one, sizeof(double), hipMemcpyHostToDevice,
                               g_stream));
  }
  exchange_p(sv);
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    const size_t o = s->row_begin - sv->row_first;
    unsigned np = 0;
    spmv_shard(s, s->d_pfull, s->d_q, s->d_pfull + s->row_begin, s->d_parts_pq, &s->npq, NULL);
    lsb_k_axpy(s->n, s->d_scal + 5, d_b + o, s->d_q, g_stream); /* q = S x - b */
    lsb_k_dot(s->n, s->d_q, s->d_q, s->d_parts_pq, &np, g_stream);
    lsb_k_reduce_final(s->d_parts_pq, np, 1, s->d_scal + 4, 0, NULL, g_stream);
  }
  allreduce_scal(sv, 4, 1);
  LSB_CHK_HIP(hipMemcpyAsync(&rr, sv->sh[0].d_scal + 4, sizeof rr, hipMemcpyDeviceToHost,
                             g_stream));
  LSB_CHK_HIP(hipStreamSynchronize(g_stream));
  sv->p2p_on = on, sv->p2p_halo = halo;
  return sv->h_st->bb > 0.0 ? sqrt(rr / sv->h_st->bb) : 0.0;
}

int lsb_hip_solver_solve_dev(lsb_hip_solver *sv, const double *d_b, double *d_x,
                             struct lsb_hip_result *res) {
  if (!initialized)
    return 1;
  if (!sv || !d_b || !d_x)
    return 2;
  if (!sv->d_perm)
    return solve_core(sv, d_b, d_x, res);
  /* b' = Q b ; solve Q S Q^T x' = b' ; x = Q^T x'   (src/cusparse.c:177,204) */
  lsb_k_perm_gather(sv->n_here, sv->d_perm, d_b, sv->d_bp, g_stream);
  const int rc = solve_core(sv, sv->d_bp, sv->d_xp, res);
  lsb_k_perm_scatter(sv->n_here, sv->d_perm, sv->d_xp, d_x, g_stream);
  LSB_CHK_HIP(hipStreamSynchronize(g_stream));
  return rc;
}

static int solve_core(lsb_hip_solver *sv, const double *d_b, double *d_x,
                      struct lsb_hip_result *res) {
  if (sv->o.krylov == LSB_KRYLOV_GMRES)
    return gmres_solve_dev(sv, d_b, d_x, res);
  const int chunk = sv->o.check_every > 0 ? (sv->o.check_every + 1) & ~1 : auto_chunk(sv);
  const int sampling = sv->o.sample_spmv > 0;
  const int use_graph = sv->o.use_graph && !sv->multi && !sampling;
  int nsamp = 0;
  unsigned done_iters = 0;
  struct lsb_pcg_state *hst = sv->h_st; /* two pinned slots */
  double t0 = wall_seconds();

#define ENQUEUE_ITERS(count)                                                   \
  do {                                                                         \
    const int cnt_ = (count);                                                  \
    if (use_graph) {                                                           \
      LSB_CHK_HIP(hipGraphLaunch(get_graph(sv, cnt_, d_x), g_stream));         \
    } else {                                                                   \
      for (int i_ = 0; i_ < cnt_; i_++) {                                      \
        int smp_ = -1;                                                         \
        if (sampling && nsamp < MAX_SAMPLES &&                                 \
            ((done_iters + (unsigned)i_) % (unsigned)sv->o.sample_spmv) == 0)  \
          smp_ = nsamp++;                                                      \
        pcg_enqueue_iter(sv, d_x, i_ & 1, smp_);                               \
      }                                                                        \
    }                                                                          \
    done_iters += (unsigned)cnt_;                                              \
  } while (0)
#define ENQUEUE_POLL(slot)                                                     \
  do {                                                                         \
    LSB_CHK_HIP(hipMemcpyAsync(&hst[slot], sv->sh[0].d_st,                     \
                               sizeof(struct lsb_pcg_state),                   \
                               hipMemcpyDeviceToHost, g_stream));              \
    LSB_CHK_HIP(hipEventRecord(sv->ev_poll[slot], g_stream));                  \
  } while (0)

  pcg_enqueue_init(sv, d_b, d_x);
  int fin = -1; /* slot holding the final state */
  if (sv->hint_iters > 0) {
    /* graphs beyond ~1k iterations cost more to build than they save */
    int first = (int)((sv->hint_iters + 1) & ~1u);
    while (use_graph && first > 1024)
      first = ((first / 2) + 1) & ~1;
    int left = (int)((sv->hint_iters + 1) & ~1u);
    while (left > 0) {
      const int c = left < first ? ((left + 1) & ~1) : first;
      ENQUEUE_ITERS(c);
      left -= c;
    }
    ENQUEUE_POLL(0);
    LSB_CHK_HIP(hipEventSynchronize(sv->ev_poll[0]));
    if (hst[0].status != LSB_STATUS_RUNNING)
      fin = 0;
  }
  if (fin < 0) {
    int cur = 0;
    ENQUEUE_ITERS(chunk);
    ENQUEUE_POLL(0);
    for (;;) {
      ENQUEUE_ITERS(chunk); /* one chunk ahead of the poll */
      ENQUEUE_POLL(cur ^ 1);
      LSB_CHK_HIP(hipEventSynchronize(sv->ev_poll[cur]));
      if (hst[cur].status != LSB_STATUS_RUNNING) {
        fin = cur;
        break;
      }
      cur ^= 1;
      if (done_iters > sv->o.maxit + 3u * (unsigned)chunk) /* cannot happen */
        errx(EXIT_FAILURE, "hip_cdna4: PCG ran past maxit without a status");
    }
    LSB_CHK_HIP(hipStreamSynchronize(g_stream)); /* drain the speculative chunk */
  }
#undef ENQUEUE_ITERS
#undef ENQUEUE_POLL
  if (fin != 0)
    hst[0] = hst[fin];
  if (hst[0].status == LSB_STATUS_COMM)
    errx(EXIT_FAILURE, "hip_cdna4: a peer did not arrive within the time-out of the direct "
                       "xGMI path (LSBENCH_HIP_P2P_TIMEOUT_MS); iteration %d", hst[0].iters);
  sv->hint_iters = (unsigned)hst[0].iters;
  if (use_cg1(sv) && hst[0].status == LSB_STATUS_MAXIT && hst[0].iters > 0) {
    /* The single-reduction form learns r.r of an update one launch later, and
     * the launch after the maxit-th update is a no-op: fetch it from that
     * update's partial sums so that relres (and "converged exactly at maxit")
     * are reported like the classic form does. */
    double rr = 0.0;
    for (int i = 0; i < sv->nshard; i++) {
      struct shard *s = &sv->sh[i];
      double *last = s->d_parts2 + (size_t)(hst[0].iters & 1) * 2 * LSB_MAX_PARTIALS;
      lsb_k_reduce_final(last, s->np2, 2, s->d_scal + 1, 0, NULL, g_stream);
    }
    g_ar_nostate = 1;
    if (sv->multi)
      allreduce_scal(sv, 1, 2);
    g_ar_nostate = 0;
    LSB_CHK_HIP(hipMemcpyAsync(&rr, sv->sh[0].d_scal + 2, sizeof rr, hipMemcpyDeviceToHost,
                               g_stream));
    LSB_CHK_HIP(hipStreamSynchronize(g_stream));
    hst[0].rr = rr;
    if (rr <= hst[0].thresh2)
      hst[0].status = LSB_STATUS_CONVERGED;
  }
  double t1 = wall_seconds();
  struct lsb_hip_result r;
  memset(&r, 0, sizeof r);
  r.iters = (unsigned)sv->h_st->iters;
  r.status = sv->h_st->status;
  r.relres = sv->h_st->bb > 0.0 ? sqrt(sv->h_st->rr / sv->h_st->bb) : 0.0;
  r.seconds = t1 - t0;
  if (nsamp > 0) {
    double tot = 0.0;
    int used = 0;
    for (int k = 0; k < nsamp; k++) {
      float ms = 0.f;
      /* samples enqueued after convergence time a no-op launch: skip them */
      if ((unsigned)k * (unsigned)sv->o.sample_spmv >= r.iters)
        break;
      float pair = 0.f;
      LSB_CHK_HIP(hipEventElapsedTime(&ms, sv->ev[4 * k], sv->ev[4 * k + 1]));
      LSB_CHK_HIP(hipEventElapsedTime(&pair, sv->ev[4 * k + 2], sv->ev[4 * k + 3]));
      tot += ms - pair, used++;
    }
    r.spmv_ms = used ? tot / used : 0.0;
    r.spmv_samples = (unsigned)used;
  }
  if (sv->p2p_on) {
    /* The direct path passed its self-test, but a solve is only reported if
     * the residual b - S x, recomputed with the exchange and the all-reduce
     * going through RCCL (device copies between virtual shards), agrees with
     * the recurrence; otherwise: say so, drop the path, solve again. */
    const double tr = true_relres(sv, d_b, d_x);
    if (!(tr <= 100.0 * fmax(r.relres, sv->o.tol) + 1e-9)) {
      fprintf(stderr, "hip_cdna4: WARNING: true residual %.3e after a solve over the direct xGMI "
                      "path (recurrence: %.3e); falling back to RCCL and solving again\n",
              tr, r.relres);
      sv->p2p_on = sv->p2p_halo = 0, sv->hint_iters = 0;
      return solve_core(sv, d_b, d_x, res);
    }
  }
  if (res)
    *res = r;
  g_last = r;
  return 0;
}

int lsb_hip_solver_solve(lsb_hip_solver *sv, const double *b, double *x,
                         struct lsb_hip_result *res) {
  if (!initialized)
    return 1;
  if (!sv || !b || !x)
    return 2;
  const size_t bytes = (size_t)sv->n_here * sizeof(double);
  double *d_b = (double *)lsb_hip_malloc(bytes), *d_x = (double *)lsb_hip_malloc(bytes);
  LSB_CHK_HIP(hipMemcpy(d_b, b, bytes, hipMemcpyHostToDevice));
  int rc = lsb_hip_solver_solve_dev(sv, d_b, d_x, res);
  LSB_CHK_HIP(hipMemcpy(x, d_x, bytes, hipMemcpyDeviceToHost));
  /* cached graphs must not outlive the buffers they were captured with */
  drop_graphs(sv);
  lsb_hip_free(d_b), lsb_hip_free(d_x);
  return rc;
}

/* y = Op x for the rows of this process */
int lsb_hip_solver_spmv_dev(lsb_hip_solver *sv, const double *d_x, double *d_y) {
  if (!initialized)
    return 1;
  if (!sv || !d_x || !d_y)
    return 2;
  double *d_yout = NULL;
  if (sv->d_perm) { /* y = Q^T (Q S Q^T) Q x */
    lsb_k_perm_gather(sv->n_here, sv->d_perm, d_x, sv->d_bp, g_stream);
    d_x = sv->d_bp, d_yout = d_y, d_y = sv->d_xp;
  }
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    LSB_CHK_HIP(hipMemcpyAsync(s->d_pfull + s->row_begin, d_x + (s->row_begin - sv->row_first),
                               (size_t)s->n * sizeof(double), hipMemcpyDeviceToDevice,
                               g_stream));
  }
  g_ar_nostate = 1;
  if (sv->multi)
    exchange_p(sv);
  g_ar_nostate = 0;
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    spmv_shard(s, s->d_pfull, d_y + (s->row_begin - sv->row_first), NULL, NULL, NULL, NULL);
  }
  if (d_yout)
    lsb_k_perm_scatter(sv->n_here, sv->d_perm, d_y, d_yout, g_stream);
  LSB_CHK_HIP(hipStreamSynchronize(g_stream));
  return 0;
}

int lsb_hip_solver_time_spmv(lsb_hip_solver *sv, int warm, int reps, double *ms_avg) {
  if (!initialized)
    return 1;
  if (!sv || reps < 1 || !ms_avg)
    return 2;
  struct shard *s = &sv->sh[0];
  unsigned np;
  for (int i = 0; i < warm; i++)
    spmv_shard(s, s->d_pfull, s->d_q, s->d_pfull + s->row_begin, s->d_parts_pq, &np, NULL);
  LSB_CHK_HIP(hipEventRecord(sv->ev_t0, g_stream));
  for (int i = 0; i < reps; i++)
    spmv_shard(s, s->d_pfull, s->d_q, s->d_pfull + s->row_begin, s->d_parts_pq, &np, NULL);
  LSB_CHK_HIP(hipEventRecord(sv->ev_t1, g_stream));
  LSB_CHK_HIP(hipEventSynchronize(sv->ev_t1));
  float ms = 0.f;
  LSB_CHK_HIP(hipEventElapsedTime(&ms, sv->ev_t0, sv->ev_t1));
  *ms_avg = (double)ms / reps;
  return 0;
}

int lsb_hip_solver_jacobi_sweep_dev(lsb_hip_solver *sv, double w, const double *d_b,
                                    double *d_x) {
  if (!initialized)
    return 1;
  if (!sv || !d_b || !d_x)
    return 2;
  int rc = lsb_hip_solver_spmv_dev(sv, d_x, sv->d_tmp);
  if (rc)
    return rc;
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    const size_t o = s->row_begin - sv->row_first;
    lsb_k_jacobi_sweep(s->n, w, s->d_dinv, d_b + o, sv->d_tmp + o, d_x + o, g_stream);
  }
  LSB_CHK_HIP(hipStreamSynchronize(g_stream));
  return 0;
}

/* ------------------------------------------------------------------------ */
/* kernel-level entry points                                                 */
/* ------------------------------------------------------------------------ */
unsigned lsb_hip_partials_capacity(void) { return 2 * LSB_MAX_PARTIALS; }

int lsb_hip_spmv_csr_f64(int variant, unsigned n, const int *d_offs,
                         const int *d_cols, const double *d_vals,
                         const int *d_rowblk, const unsigned char *d_blklanes,
                         unsigned nblk, unsigned mean_row_len, unsigned flags,
                         const double *d_x, double *d_y,
                         const double *d_xdot, double *d_dot, double *d_work,
                         void *stream) {
  if (!initialized)
    return 1;
  if (variant == LSB_SPMV_AUTO)
    variant = d_rowblk ? LSB_SPMV_ADAPTIVE : LSB_SPMV_SUBWAVE;
  if (variant == LSB_SPMV_ADAPTIVE && (!d_rowblk || nblk == 0))
    return 2;
  if ((variant < LSB_SPMV_ADAPTIVE || variant > LSB_SPMV_SCALAR) && variant != LSB_SPMV_SELL)
    return 2;
  if (n == 0 || (variant == LSB_SPMV_SELL && nblk != (n + LSB_SELL_ROWS - 1) / LSB_SELL_ROWS))
    return 2;
  if (d_dot && (!d_work || !d_xdot))
    return 2;
  unsigned L = pow2_ceil(mean_row_len ? mean_row_len : 1);
  L = L < 2 ? 2 : (L > 64 ? 64 : L);

  unsigned np = 0;
  if (variant == LSB_SPMV_SELL) {
    if ((flags & LSB_SP_C16) && !d_rowblk)
      return 2; /* the 16-bit form needs its slot bases */
    lsb_k_spmv_sell(flags, 0, (const unsigned *)d_offs, 0, nblk, n, 0, d_cols, d_rowblk, d_vals,
                    d_x, d_y, d_dot ? d_xdot : NULL, d_dot ? d_work : NULL, &np, NULL, stream);
  } else
  lsb_k_spmv(variant, n, d_offs, d_cols, d_vals, d_rowblk, d_blklanes, nblk, L, flags, 0,
             d_x, d_y, d_dot ? d_xdot : NULL, d_dot ? d_work : NULL, &np, NULL, NULL, stream);
  if (d_dot)
    lsb_k_reduce_final(d_work, np, 1, d_dot, 0, NULL, stream);
  return 0;
}

int lsb_hip_dot_f64(unsigned n, const double *d_a, const double *d_b,
                    double *d_out, double *d_work, void *stream) {
  if (!initialized)
    return 1;
  unsigned np = 0;
  lsb_k_dot(n, d_a, d_b, d_work, &np, stream);
  lsb_k_reduce_final(d_work, np, 1, d_out, 0, NULL, stream);
  return 0;
}

int lsb_hip_nrm2_f64(unsigned n, const double *d_a, double *d_out,
                     double *d_work, void *stream) {
  if (!initialized)
    return 1;
  unsigned np = 0;
  lsb_k_dot(n, d_a, d_a, d_work, &np, stream);
  lsb_k_reduce_final(d_work, np, 1, d_out, 1, NULL, stream);
  return 0;
}

int lsb_hip_axpy_f64(unsigned n, const double *d_alpha, const double *d_x,
                     double *d_y, void *stream) {
  if (!initialized)
    return 1;
  lsb_k_axpy(n, d_alpha, d_x, d_y, stream);
  return 0;
}

int lsb_hip_xpay_f64(unsigned n, const double *d_beta, const double *d_x,
                     double *d_y, void *stream) {
  if (!initialized)
    return 1;
  lsb_k_xpay(n, d_beta, d_x, d_y, stream);
  return 0;
}

int lsb_hip_jacobi_setup_f64(unsigned n, unsigned row_begin, const int *d_offs,
                             const int *d_cols, const double *d_vals,
                             double *d_dinv, int *d_nzero, void *stream) {
  if (!initialized)
    return 1;
  lsb_k_jacobi_setup(n, row_begin, d_offs, d_cols, d_vals, d_dinv, d_nzero, stream);
  return 0;
}

int lsb_hip_jacobi_apply_f64(unsigned n, const double *d_dinv,
                             const double *d_r, double *d_z, void *stream) {
  if (!initialized)
    return 1;
  lsb_k_jacobi_apply(n, d_dinv, d_r, d_z, stream);
  return 0;
}

/* ------------------------------------------------------------------------ */
/* the drop-in entry point                                                   */
/* ------------------------------------------------------------------------ */
int hip_cdna4_bench(double *x, struct csr *A, const double *r,
                    const struct lsbench *cb) {
  if (!initialized)
    return 1;
  struct lsb_hip_opts o;
  lsb_hip_get_opts(&o);
  const unsigned m = A->nrows, nnz = A->offs[m];
  const size_t bytes = (size_t)m * sizeof(double);

  /* untimed setup: operator build, upload, Jacobi, row blocks
   * (counterpart of csr_init, src/cusparse.c:47-125) */
  lsb_hip_solver *sv = lsb_hip_solver_create(A, &o);
  if (!sv)
    errx(EXIT_FAILURE, "hip_cdna4: cannot set up the solver");
  double *d_r = (double *)lsb_hip_malloc(bytes), *d_x = (double *)lsb_hip_malloc(bytes);
  LSB_CHK_HIP(hipMemcpy(d_r, r, bytes, hipMemcpyHostToDevice));

  struct lsb_hip_result res;
  memset(&res, 0, sizeof res);
  /* warm-up (src/cholmod-impl.h:44-55, src/cusparse.c:182-186) */
  for (unsigned i = 0; i < cb->trials; i++)
    lsb_hip_solver_solve_dev(sv, d_r, d_x, &res);

  /* timed (src/cusparse.c:189-197).  Wall clock, not clock(): clock() is
   * process CPU time and would not see the device. */
  LSB_CHK_HIP(hipDeviceSynchronize());
  const double t0 = wall_seconds();
  for (unsigned i = 0; i < cb->trials; i++)
    lsb_hip_solver_solve_dev(sv, d_r, d_x, &res);
  LSB_CHK_HIP(hipDeviceSynchronize());
  const double elapsed = wall_seconds() - t0;

  LSB_CHK_HIP(hipMemcpy(x, d_x, bytes, hipMemcpyDeviceToHost)); /* :199 */

  /* the reference's record, verbatim (src/cholmod-impl.h:68-70) ... */
  printf("===matrix,n,nnz,trials,solver,ordering,elapsed===\n");
  printf("%s,%u,%u,%u,%u,%d,%.15lf\n", cb->matrix, m, nnz, cb->trials, cb->solver,
         cb->ordering, elapsed);
  /* ... plus what an iterative backend owes its reader */
  printf("===hip_cdna4:iterations,relres,status,tol,solves_per_sec,nshards===\n");
  printf("%u,%.6e,%d,%.3e,%.6f,%d\n", res.iters, res.relres, res.status, o.tol,
         elapsed > 0 ? cb->trials / elapsed : 0.0, sv->nshard);
  fflush(stdout);

  lsb_hip_free(d_r), lsb_hip_free(d_x);
  lsb_hip_solver_destroy(sv);
  return 0;
}
