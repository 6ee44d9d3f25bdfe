/*
 * Host side of the preconditioners that are not a diagonal scaling (kernels:
 * hip_precond_k.hip): set-up at solver creation
 * (untimed, like the reference's csr_init / CHOLMOD's factorisation,
 * src/cholmod-impl.h:25-26) and the launches of one application z = M^-1 r.
 */
#define _GNU_SOURCE
#include "hip_solver.h"

#define DINV(s) ((s)->dinv_uniform ? NULL : (s)->d_dinv), (s)->dinv_const

int generic_precond(const lsb_hip_solver *sv) {
  return sv->o.precond == LSB_PRECOND_CHEBYSHEV || sv->o.precond == LSB_PRECOND_BLOCKJACOBI ||
         sv->o.precond == LSB_PRECOND_FSAI;
}

/* ---- FSAI: G on the pattern of tril(S^k), rows by batched dense solves on the device ------- */
static void fsai_upload_csr(struct fsai_csr *c, unsigned n, const unsigned *offs, const unsigned *cols,
                            const double *vals) {
  const unsigned long long nnz = offs[n];
  int *o32 = (int *)malloc(((size_t)n + 1) * sizeof(int));
  for (unsigned i = 0; i <= n; i++)
    o32[i] = (int)offs[i];
  c->nnz = nnz;
  c->offs = (int *)dev_upload(o32, ((size_t)n + 1) * sizeof(int));
  c->cols = (int *)dev_upload(cols, (size_t)(nnz ? nnz : 1) * sizeof(int)); /* < 2^31: same bits */
  c->vals = (double *)dev_upload(vals, (size_t)(nnz ? nnz : 1) * sizeof(double));
  /* launch-bound sizes: the sub-wavefront kernel; else the row-blocked one */
  struct csr view = {n, 0, (unsigned *)offs, NULL, NULL};
  const unsigned mean = n ? (unsigned)((nnz + n - 1) / n) : 1;
  unsigned L = pow2_ceil(mean ? mean : 1);
  c->lanes = L < 2 ? 2 : (L > 64 ? 64 : L);
  c->variant = nnz <= 2000000ull ? LSB_SPMV_SUBWAVE : LSB_SPMV_ADAPTIVE;
  unsigned *rb = NULL;
  c->nblk = lsb_csr_row_blocks(&view, LSB_BLOCK_NNZ, &rb);
  c->rowblk = (int *)dev_upload(rb, ((size_t)c->nblk + 1) * sizeof(int));
  unsigned char *lanes = (unsigned char *)malloc((size_t)c->nblk + 1);
  lsb_csr_block_lanes(&view, rb, c->nblk, lanes);
  c->blklanes = (unsigned char *)dev_upload(lanes, (size_t)c->nblk);
  LSB_CHK_HIP(hipStreamSynchronize(g_stream));
  free(o32), free(rb), free(lanes);
}

static void fsai_free_csr(struct fsai_csr *c) {
  lsb_hip_free(c->offs), lsb_hip_free(c->cols), lsb_hip_free(c->vals);
  lsb_hip_free(c->rowblk), lsb_hip_free(c->blklanes);
  memset(c, 0, sizeof *c);
}

static void fsai_spmv(const struct shard *s, const struct fsai_csr *c, const double *x, double *y,
                      const struct lsb_pcg_state *st) {
  lsb_k_spmv(c->variant, s->n, c->offs, c->cols, c->vals, c->rowblk, c->blklanes, c->nblk, c->lanes,
             LSB_SP_PREFETCH | LSB_SP_NT, 0, x, y, NULL, NULL, NULL, st, NULL, NULL, g_stream);
}

static void precond_shard_fsai(struct shard *s, const int *offs, const int *cols, const double *vals,
                               const struct lsb_hip_opts *o) {
  if (s->row_begin != 0 || s->n != s->n_glob)
    errx(EXIT_FAILURE, "hip_cdna4: --precond fsai runs on one shard (rows of G reach into other shards' "
                       "columns); use it without --ngpus / --nvirt");
  const unsigned n = s->n;
  struct csr view = {n, 0, (unsigned *)offs, (unsigned *)cols, (double *)vals};
  const int power = o->fsai_power < 1 ? 1 : (o->fsai_power > 3 ? 3 : o->fsai_power);
  struct lsb_fsai_pattern *P = lsb_csr_fsai_pattern(&view, power, LSB_FSAI_CAP);
  if (!P)
    errx(EXIT_FAILURE, "hip_cdna4: cannot build the FSAI pattern");
  if (P->nnz > 0x7fffffffull) /* G and G^T go through the int-offset CSR kernels */
    errx(EXIT_FAILURE, "hip_cdna4: the FSAI pattern has %llu entries, more than 2^31 - 1; choose a smaller "
                       "--fsai-power", P->nnz);
  /* rows by size class: a wavefront per row up to 32 entries, a workgroup beyond */
  unsigned *small = (unsigned *)malloc((size_t)n * sizeof(unsigned)), *big = (unsigned *)malloc((size_t)n * sizeof(unsigned));
  unsigned nsmall = 0, nbig = 0, maxrow = 0;
  for (unsigned i = 0; i < n; i++) {
    const unsigned m = P->offs[i + 1] - P->offs[i];
    if (m == 0 || P->cols[P->offs[i + 1] - 1] != i)
      errx(EXIT_FAILURE, "hip_cdna4: FSAI pattern row %u does not end in its diagonal", i);
    if (m <= 32)
      small[nsmall++] = i;
    else
      big[nbig++] = i;
    if (m > maxrow)
      maxrow = m;
  }
  s->fs_maxrow = maxrow;
  unsigned *d_poffs = (unsigned *)dev_upload(P->offs, ((size_t)n + 1) * sizeof(unsigned));
  unsigned *d_pcols = (unsigned *)dev_upload(P->cols, (size_t)(P->nnz ? P->nnz : 1) * sizeof(unsigned));
  unsigned *d_small = (unsigned *)dev_upload(small, (size_t)(nsmall ? nsmall : 1) * sizeof(unsigned));
  unsigned *d_big = (unsigned *)dev_upload(big, (size_t)(nbig ? nbig : 1) * sizeof(unsigned));
  double *d_g = (double *)lsb_hip_malloc((size_t)(P->nnz ? P->nnz : 1) * sizeof(double));
  int *d_bad = (int *)lsb_hip_malloc(sizeof(int)), bad = 0;
  LSB_CHK_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), g_stream));
  lsb_k_fsai_rows(d_small, nsmall, 32, s->d_offs, s->d_cols, s->d_vals, s->row_begin, d_poffs, d_pcols, d_g, d_bad,
                  g_stream);
  lsb_k_fsai_rows(d_big, nbig, LSB_FSAI_CAP, s->d_offs, s->d_cols, s->d_vals, s->row_begin, d_poffs, d_pcols, d_g,
                  d_bad, g_stream);
  double *g = (double *)malloc((size_t)(P->nnz ? P->nnz : 1) * sizeof(double));
  LSB_CHK_HIP(hipMemcpyAsync(g, d_g, (size_t)P->nnz * sizeof(double), hipMemcpyDeviceToHost, g_stream));
  LSB_CHK_HIP(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, g_stream));
  LSB_CHK_HIP(hipStreamSynchronize(g_stream));
  if (bad)
    errx(EXIT_FAILURE, "hip_cdna4: --precond fsai needs a symmetric positive definite operator (a local "
                       "system S[J, J] had a pivot <= 0)");
  /* G as it stands; G^T by a counting sort (rows of G^T come out with ascending columns) */
  fsai_upload_csr(&s->fs_g, n, P->offs, P->cols, g);
  unsigned *toffs = lsb_calloc(unsigned, (size_t)n + 2);
  for (unsigned long long e = 0; e < P->nnz; e++)
    toffs[P->cols[e] + 2]++;
  for (unsigned i = 0; i < n; i++)
    toffs[i + 2] += toffs[i + 1];
  unsigned *tcols = (unsigned *)malloc((size_t)(P->nnz ? P->nnz : 1) * sizeof(unsigned));
  double *tvals = (double *)malloc((size_t)(P->nnz ? P->nnz : 1) * sizeof(double));
  for (unsigned i = 0; i < n; i++)
    for (unsigned e = P->offs[i]; e < P->offs[i + 1]; e++) {
      const unsigned at = toffs[P->cols[e] + 1]++;
      tcols[at] = i, tvals[at] = g[e];
    }
  fsai_upload_csr(&s->fs_gt, n, toffs, tcols, tvals);
  s->d_fst = shard_vec(s, n);
  if (o->verbose)
    fprintf(stderr, "hip_cdna4: FSAI on the pattern of tril(S^%d): %llu entries (%.1f per row, longest %u%s), "
                    "%u rows by wavefronts, %u by workgroups\n", power, P->nnz, (double)P->nnz / n, maxrow,
            maxrow == LSB_FSAI_CAP ? " = the cap" : "", nsmall, nbig);
  lsb_hip_free(d_poffs), lsb_hip_free(d_pcols), lsb_hip_free(d_small), lsb_hip_free(d_big);
  lsb_hip_free(d_g), lsb_hip_free(d_bad);
  free(small), free(big), free(g), free(toffs), free(tcols), free(tvals);
  lsb_fsai_pattern_free(P);
}

/* ---- block-Jacobi: dense diagonal blocks out of the shard's rows ------------ */
/* offs/cols: the shard's local CSR with GLOBAL column ids; blocks are runs of bs
 * consecutive rows of the shard (a block never reaches into another shard). */
void precond_shard_blocks(struct shard *s, const int *offs, const int *cols, const double *vals,
                          const struct lsb_hip_opts *o) {
  if (o->precond == LSB_PRECOND_FSAI)
    precond_shard_fsai(s, offs, cols, vals, o);
  if (o->precond != LSB_PRECOND_BLOCKJACOBI)
    return;
  unsigned bs = o->block_size < 1 ? 1u : (unsigned)o->block_size;
  if (bs > s->n)
    bs = s->n;
  if ((unsigned long long)s->n * bs > (1ull << 31))
    errx(EXIT_FAILURE, "hip_cdna4: block-Jacobi with %u-row blocks on %u rows needs %.1f GB; "
                       "choose a smaller --block-size", bs, s->n, (double)s->n * bs * 8e-9);
  const size_t total = (size_t)s->n * bs;
  double *blk = (double *)calloc(total ? total : 1, sizeof(double));
  if (!blk)
    errx(EXIT_FAILURE, "hip_cdna4: out of host memory for the block-Jacobi blocks");
  for (unsigned i = 0; i < s->n; i++) {
    const unsigned k = i / bs, il = i % bs, m = s->n - k * bs < bs ? s->n - k * bs : bs;
    const long long c0 = (long long)s->row_begin + (long long)k * bs;
    for (int j = offs[i]; j < offs[i + 1]; j++) {
      const long long c = (long long)cols[j] - c0;
      if (c >= 0 && c < (long long)m)
        blk[(size_t)k * bs * bs + (size_t)il * m + (size_t)c] = vals[j];
    }
  }
  for (unsigned i = 0; i < s->n; i++) { /* an empty diagonal would make a block singular */
    const unsigned k = i / bs, il = i % bs, m = s->n - k * bs < bs ? s->n - k * bs : bs;
    if (blk[(size_t)k * bs * bs + (size_t)il * m + il] == 0.0)
      errx(EXIT_FAILURE, "hip_cdna4: row %u has no non-zero diagonal entry; block-Jacobi needs one",
           s->row_begin + i);
  }
  s->bj_bs = bs;
  s->d_binv = (double *)dev_upload(blk, total * sizeof(double));
  double *scratch = (double *)lsb_hip_malloc(2 * (size_t)s->n * sizeof(double));
  lsb_k_bj_invert(s->n, bs, s->d_binv, scratch, g_stream);
  LSB_CHK_HIP(hipStreamSynchronize(g_stream));
  lsb_hip_free(scratch);
  free(blk);
  const unsigned nch = lsb_k_bj_chunks(bs);
  if (nch)
    s->d_bjpart = (double *)lsb_hip_malloc((size_t)nch * s->n * sizeof(double));
}

/* sum over all shards (of all ranks) of a_i . b_i, on the host; set-up only */
static double global_dot(lsb_hip_solver *sv, double *const *a, double *const *b) {
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    unsigned np = 0;
    lsb_k_dot(s->n, a[i], b[i], s->d_parts_pq, &np, g_stream);
    lsb_k_reduce_final(s->d_parts_pq, np, 1, s->d_scal + 4, 0, NULL, g_stream);
  }
  const int on = sv->p2p_on, halo = sv->p2p_halo;
  sv->p2p_on = sv->p2p_halo = 0; /* set-up talks over RCCL / device copies */
  allreduce_scal(sv, 4, 1, 0);
  sv->p2p_on = on, sv->p2p_halo = halo;
  double v = 0.0;
  LSB_CHK_HIP(hipMemcpyAsync(&v, sv->sh[0].d_scal + 4, sizeof v, hipMemcpyDeviceToHost, g_stream));
  drain_stream(sv, "preconditioner set-up (all-reduced dot product)");
  return v;
}

/* exchange + SpMV on the gather vector `full` of every shard (own rows at
 * full + row_begin): y_i = (S v)_i */
static void op_apply(lsb_hip_solver *sv, int which_z, double *const *y, int gated) {
  /* the exchange routines work on shard.d_pfull: lend them the other vector */
  for (int i = 0; i < sv->nshard && which_z; i++) {
    double *t = sv->sh[i].d_pfull;
    sv->sh[i].d_pfull = sv->sh[i].d_zfull, sv->sh[i].d_zfull = t;
  }
  if (sv->multi) {
    const int on = sv->p2p_on, halo = sv->p2p_halo;
    if (!gated)
      sv->p2p_on = sv->p2p_halo = 0;
    exchange_p(sv, gated);
    sv->p2p_on = on, sv->p2p_halo = halo;
  }
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    spmv_shard(s, s->d_pfull, y[i], NULL, NULL, NULL, gated ? s->d_st : NULL);
  }
  for (int i = 0; i < sv->nshard && which_z; i++) {
    double *t = sv->sh[i].d_pfull;
    sv->sh[i].d_pfull = sv->sh[i].d_zfull, sv->sh[i].d_zfull = t;
  }
}

#define CHEB_POWER_ITS 20
#define CHEB_SAFETY 1.1
#define CHEB_RATIO 30.0

/* FSAI on a launch-bound operator (everything sub-wavefront kernels, one shard): the iteration
 * runs in three launches (hip_fsai.hip, hip_pcg.c) */
int fsai_three_launches(const lsb_hip_solver *sv) {
  const struct shard *s = &sv->sh[0];
  return sv->o.precond == LSB_PRECOND_FSAI && !sv->multi && !s->mixed && s->variant == LSB_SPMV_SUBWAVE &&
         s->fs_g.variant == LSB_SPMV_SUBWAVE && s->fs_gt.variant == LSB_SPMV_SUBWAVE && sv->o.sample_spmv <= 0 &&
         sv->o.krylov != LSB_KRYLOV_PCG1 && !getenv("LSBENCH_HIP_NO_FSAI_FUSE");
}

void precond_setup(lsb_hip_solver *sv) {
  if (!generic_precond(sv))
    return;
  if (sv->o.precond == LSB_PRECOND_FSAI) { /* buffers of the three-launch iteration */
    struct shard *s = &sv->sh[0];
    if (!s->d_p1)
      s->d_p1 = shard_vec(s, s->n);
    s->d_r1 = shard_vec(s, s->n);
  }
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    /* z lives in a gather vector of its own: Chebyshev multiplies it by S */
    const size_t len = sv->o.precond == LSB_PRECOND_CHEBYSHEV ? (size_t)sv->n_glob : (size_t)s->n;
    s->d_zfull = shard_vec(s, len);
    LSB_CHK_HIP(hipMemsetAsync(s->d_zfull, 0, len * sizeof(double), g_stream));
    s->d_z = sv->o.precond == LSB_PRECOND_CHEBYSHEV ? s->d_zfull + s->row_begin : s->d_zfull;
    if (sv->o.precond == LSB_PRECOND_CHEBYSHEV)
      s->d_chd = shard_vec(s, s->n);
  }
  if (sv->o.precond != LSB_PRECOND_CHEBYSHEV)
    return;
  int m = sv->o.cheb_degree;
  m = m < 1 ? 1 : (m > LSB_CHEB_MAX ? LSB_CHEB_MAX : m);
  sv->cheb_m = m;
  /* lmax of D^-1 S by power iteration: v <- D^-1 S v / ||.||, 20 steps from a
   * fixed start vector, then 10 % on top (an underestimate would make the
   * polynomial grow beyond the interval) */
  double **v = lsb_calloc(double *, sv->nshard), **w = lsb_calloc(double *, sv->nshard);
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    v[i] = s->d_z, w[i] = s->d_q;
    lsb_k_power_start(s->n, s->row_begin, v[i], g_stream);
  }
  double lam = 1.0, vv = global_dot(sv, v, v);
  for (int it = 0; it < CHEB_POWER_ITS; it++) {
    op_apply(sv, 1, w, 0);
    for (int i = 0; i < sv->nshard; i++) /* w <- D^-1 w, in place through the scaling kernel */
      lsb_k_scale_dinv(sv->sh[i].n, 1.0, sv->sh[i].d_dinv, w[i], w[i], g_stream);
    const double ww = global_dot(sv, w, w);
    if (!(ww > 0.0) || !(vv > 0.0))
      break;
    lam = sqrt(ww / vv);
    const double c = 1.0 / sqrt(ww);
    for (int i = 0; i < sv->nshard; i++) { /* v = w / ||w|| */
      struct shard *s = &sv->sh[i];
      LSB_CHK_HIP(hipMemcpyAsync(v[i], w[i], (size_t)s->n * sizeof(double), hipMemcpyDeviceToDevice,
                                 g_stream));
      lsb_k_scale_vec(s->n, c, v[i], g_stream);
    }
    vv = 1.0;
  }
  free(v), free(w);
  /* lmax / lmin of the interval the polynomial is small on: max(30, 16 m^2).  A higher degree
   * resolves a wider interval, and the eigenvalues below it are CG's job; measured on the
   * 10 M-row 5-point operator (solves/s, fixed ratio 30 -> this rule): m = 4: 0.82 -> 0.90,
   * 8: 0.80 -> 1.04, 16: 0.64 -> 1.15 (profiles/r02_chebyshev.txt) */
  const double ratio = 16.0 * m * m > CHEB_RATIO ? 16.0 * m * m : CHEB_RATIO;
  sv->cheb_lmax = CHEB_SAFETY * lam, sv->cheb_lmin = sv->cheb_lmax / ratio;
  const double theta = 0.5 * (sv->cheb_lmax + sv->cheb_lmin), delta = 0.5 * (sv->cheb_lmax - sv->cheb_lmin);
  const double sigma = theta / delta;
  double rho = 1.0 / sigma;
  sv->cheb_c0 = 1.0 / theta;
  for (int k = 0; k < m; k++) {
    const double rho_new = 1.0 / (2.0 * sigma - rho);
    sv->cheb_a[k] = rho_new * rho, sv->cheb_b[k] = 2.0 * rho_new / delta;
    rho = rho_new;
  }
  if (sv->o.verbose)
    fprintf(stderr, "hip_cdna4: Chebyshev preconditioner, degree %d on [%.4g, %.4g] (lmax of D^-1 S by "
                    "%d power iterations: %.6g)\n", m, sv->cheb_lmin, sv->cheb_lmax, CHEB_POWER_ITS, lam);
  for (int i = 0; i < sv->nshard; i++)
    LSB_CHK_HIP(hipMemsetAsync(sv->sh[i].d_zfull, 0, (size_t)sv->n_glob * sizeof(double), g_stream));
  /* Shards in the 16-bit sliced-ELL form: the steps ride in the SpMV's epilogue
   * (k_spmv_sell16<.., CHEB>): S z is never written, z' goes to a second gather vector.
   * Step k reads buffer k & 1 and writes the other; the result is in buffer m & 1.
   * (A rank whose shards do not qualify keeps the launches: the same exchanges, the same
   * bits -- ranks need not agree.) */
  {
    const char *e = getenv("LSBENCH_HIP_CHEB_FUSE");
    sv->cheb_fused = !(e && atoi(e) == 0);
    for (int i = 0; i < sv->nshard; i++) {
      const struct shard *s = &sv->sh[i];
      sv->cheb_fused &= s->variant == LSB_SPMV_SELL && (s->sp_flags & LSB_SP_C16) && s->d_scodes &&
                        !(s->row_begin & 1u);
    }
    for (int i = 0; i < sv->nshard && sv->cheb_fused; i++) {
      struct shard *s = &sv->sh[i];
      s->d_zfull2 = shard_vec(s, sv->n_glob);
      LSB_CHK_HIP(hipMemsetAsync(s->d_zfull2, 0, (size_t)sv->n_glob * sizeof(double), g_stream));
      s->d_z = ((m & 1) ? s->d_zfull2 : s->d_zfull) + s->row_begin;
    }
    if (sv->o.verbose)
      fprintf(stderr, "hip_cdna4: Chebyshev steps %s\n",
              sv->cheb_fused ? "in the SpMV's epilogue" : "as launches of their own");
  }
  LSB_CHK_HIP(hipStreamSynchronize(g_stream));
}

/* z = M^-1 r on every shard (r = shard.d_r, z = shard.d_z); part of a running
 * solve: launches no-op once its state has left RUNNING */
void precond_apply(lsb_hip_solver *sv, int after_update) {
  if (sv->nshard > 64)
    errx(EXIT_FAILURE, "hip_cdna4: more than 64 shards");
  if (sv->o.precond == LSB_PRECOND_FSAI) { /* z = G^T (G r): two SpMVs, nothing else */
    struct shard *s = &sv->sh[0];
    fsai_spmv(s, &s->fs_g, s->d_r, s->d_fst, s->d_st);
    fsai_spmv(s, &s->fs_gt, s->d_fst, s->d_z, s->d_st);
    return;
  }
  if (sv->o.precond == LSB_PRECOND_BLOCKJACOBI) {
    for (int i = 0; i < sv->nshard; i++) {
      struct shard *s = &sv->sh[i];
      /* after_update: k_pcg_update_xr's r.r partial sums are in d_parts2 (s->np2 records) */
      lsb_k_bj_apply(s->n, s->bj_bs, s->d_binv, s->d_r, s->d_z, s->d_bjpart, s->d_st,
                     after_update && !sv->multi ? s->d_parts2 : NULL, s->np2, g_stream);
    }
    return;
  }
  if (sv->cheb_fused) {
    for (int i = 0; i < sv->nshard; i++) {
      struct shard *s = &sv->sh[i];
      lsb_k_cheb_first(s->n, s->d_r, DINV(s), sv->cheb_c0, s->d_chd, s->d_zfull + s->row_begin, s->d_st,
                       g_stream);
    }
    for (int k = 0; k < sv->cheb_m; k++) {
      /* the exchange routines work on shard.d_pfull: lend them this step's z */
      double *keep[64];
      for (int i = 0; i < sv->nshard; i++) {
        struct shard *s = &sv->sh[i];
        keep[i] = s->d_pfull;
        s->d_pfull = (k & 1) ? s->d_zfull2 : s->d_zfull;
      }
      if (sv->multi)
        exchange_p(sv, 1);
      for (int i = 0; i < sv->nshard; i++) {
        struct shard *s = &sv->sh[i];
        s->epi.r = s->d_r, s->epi.dinv = s->dinv_uniform ? NULL : s->d_dinv, s->epi.dc = s->dinv_const;
        s->epi.a = sv->cheb_a[k], s->epi.b = sv->cheb_b[k], s->epi.d = s->d_chd;
        s->epi.zout = ((k & 1) ? s->d_zfull : s->d_zfull2) + s->row_begin;
        sell_launch(s, 0, s->nslice, s->d_pfull, NULL, NULL, NULL, NULL, s->d_st);
        s->epi.zout = NULL;
      }
      for (int i = 0; i < sv->nshard; i++)
        sv->sh[i].d_pfull = keep[i];
      sv->nspmv++;
    }
    return;
  }
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    lsb_k_cheb_first(s->n, s->d_r, DINV(s), sv->cheb_c0, s->d_chd, s->d_z, s->d_st, g_stream);
  }
  double *w[64];
  if (sv->nshard > 64)
    errx(EXIT_FAILURE, "hip_cdna4: more than 64 shards");
  for (int i = 0; i < sv->nshard; i++)
    w[i] = sv->sh[i].d_q;
  for (int k = 0; k < sv->cheb_m; k++) {
    op_apply(sv, 1, w, 1);
    sv->nspmv++;
    for (int i = 0; i < sv->nshard; i++) {
      struct shard *s = &sv->sh[i];
      lsb_k_cheb_step(s->n, s->d_r, s->d_q, DINV(s), sv->cheb_a[k], sv->cheb_b[k], s->d_chd, s->d_z,
                      s->d_st, g_stream);
    }
  }
}

void precond_free_shard(struct shard *s) {
  lsb_hip_free(s->d_binv), lsb_hip_free(s->d_bjpart), shard_vec_free(s, s->d_zfull), shard_vec_free(s, s->d_chd);
  shard_vec_free(s, s->d_zfull2);
  fsai_free_csr(&s->fs_g), fsai_free_csr(&s->fs_gt), shard_vec_free(s, s->d_fst), shard_vec_free(s, s->d_r1);
}
