/* GMRES(m) host driver (kernels: hip_gmres.hip). */
#define _GNU_SOURCE
#include "hip_solver.h"

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

int gmres_solve_dev(lsb_hip_solver *sv, const double *d_b, double *d_x,
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
  for (int cycle = 0;; cycle++) {
    if (cycle > 0) { /* ax = Op x for the restart residual */
      EACH(i, s, w) {
        (void)w;
        LSB_CHK_HIP(hipMemcpyAsync(s->d_pfull + s->row_begin, d_x + (s->row_begin - sv->row_first),
                                   (size_t)s->n * sizeof(double), hipMemcpyDeviceToDevice,
                                   g_stream));
      }
      if (sv->multi)
        exchange_p(sv, 0); /* not tied to a PCG state */
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
        exchange_p(sv, 0); /* not tied to a PCG state */
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
    drain_stream(sv, "GMRES restart cycle");
    if (sv->gm_hst->status != LSB_STATUS_RUNNING)
      break;
    if ((unsigned)cycle > sv->o.maxit + 2u)
      errx(EXIT_FAILURE, "hip_cdna4: GMRES ran past maxit without a status");
  }
#undef EACH
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
