/*
 * What a sharded solve adds to an iteration (DESIGN.md
 * section 6): the exchange of the gather vector's halos, the all-reduce of the
 * dot products, the overlap of the two with the SpMV's interior rows -- over
 * RCCL (hip_comm.c), over direct xGMI stores (hip_p2p.hip), or by device copies
 * between the virtual shards of one process.
 */
#define _GNU_SOURCE
#include "hip_solver.h"

/* ------------------------------------------------------------------------ */
/* communication steps: RCCL between processes, device copies between the     */
/* virtual shards of one process                                              */
/* ------------------------------------------------------------------------ */
void exchange_on(lsb_hip_solver *sv, hipStream_t stream) {
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

/* gated = 1: part of a running solve (a launch no-ops once the device state has
 * left RUNNING); 0: a communication step of its own -- gated on the shard's
 * auxiliary state instead, which only ever leaves RUNNING through a time-out of
 * the direct path (the host reports that: check_aux_status). */
void exchange_p(lsb_hip_solver *sv, int gated) {
  if (sv->p2p_halo && sv->dist) { /* peers are other GPUs: both roles in one launch */
    lsb_p2p_sendrecv(sv->p2p[0], sv->sh[0].d_pfull, gated ? sv->sh[0].d_st : sv->sh[0].d_st_aux,
                     g_stream);
  } else if (sv->p2p_halo) { /* all sends before any wait: virtual shards share a stream */
    for (int i = 0; i < sv->nshard; i++)
      lsb_p2p_send(sv->p2p[i], sv->sh[i].d_pfull, gated ? sv->sh[i].d_st : sv->sh[i].d_st_aux,
                   g_stream);
    for (int i = 0; i < sv->nshard; i++)
      lsb_p2p_recv(sv->p2p[i], sv->sh[i].d_pfull, gated ? sv->sh[i].d_st : sv->sh[i].d_st_aux,
                   g_stream);
  } else {
    exchange_on(sv, g_stream);
    return;
  }
  /* Mailbox regions and flags are re-used by the next exchange; inside an
   * iteration the all-reduce of the dot products lies in between.  An exchange
   * outside one (SpMV entry point called in a row) gets a one-value all-reduce
   * as its closing barrier: no rank rewrites a region a peer is still reading. */
  if (!gated)
    allreduce_scal(sv, 7, 1, 0);
}

/* A time-out recorded by an ungated communication step is fatal, like any
 * failing HIP / RCCL call of this library. */
void check_aux_status(lsb_hip_solver *sv, const char *where) {
  if (!sv->p2p_on)
    return;
  for (int i = 0; i < sv->nshard; i++) {
    struct lsb_pcg_state h;
    LSB_CHK_HIP(hipMemcpy(&h, sv->sh[i].d_st_aux, sizeof h, hipMemcpyDeviceToHost));
    if (h.status == LSB_STATUS_COMM)
      lsb_give_up("hip_cdna4: %s: a peer did not arrive within the time-out of the direct "
                  "xGMI path (LSBENCH_HIP_P2P_TIMEOUT_MS) (rank %d of %d)", where, lsb_hip_comm_rank(),
                  lsb_hip_comm_size());
  }
}

/* d_scal[off .. off+cnt) <- sum over shards.  With the direct path the
 * shard's own partial sums are folded into the same launch: the first `width`
 * values come from the SpMV's dot partials, the next s->ar2_width from the
 * array the sweep kernel left in s->ar2_parts, and only the rest must already
 * sit, reduced, in d_scal. */
static void allreduce_parts(lsb_hip_solver *sv, unsigned off, unsigned cnt, unsigned width,
                            int with2, int gated) {
  for (int ph = 1; ph <= 2; ph++)
    for (int i = 0; i < sv->nshard; i++) {
      struct shard *s = &sv->sh[i];
      const double *parts = width ? s->d_parts_pq : NULL;
      const unsigned w2 = with2 ? s->ar2_width : 0;
      struct lsb_pcg_state *st = gated ? s->d_st : s->d_st_aux;
      const int phases = sv->nshard == 1 ? 3 : ph;
      if (sv->nshard == 1 && ph == 2)
        continue;
      lsb_p2p_allreduce(sv->p2p[i], parts, s->npq, width, s->ar2_parts, s->ar2_n, w2,
                        s->d_scal + off + width + w2, cnt - width - w2, s->d_scal + off, st,
                        phases, g_stream);
    }
}

void allreduce_scal(lsb_hip_solver *sv, unsigned off, unsigned cnt, int gated) {
  if (sv->p2p_on) {
    allreduce_parts(sv, off, cnt, 0, 0, gated);
    return;
  }
  if (sv->dist)
    lsb_hip_comm_allreduce_stream(sv->sh[0].d_scal + off, (int)cnt, g_stream);
  else if (sv->nshard > 1)
    lsb_k_vreduce(sv->d_scal_all, SCAL_STRIDE, (unsigned)sv->nshard, off, cnt, g_stream);
}

/* d_scal[0] <- all-reduced sum of the SpMV's dot partials; d_scal[1..cnt) are
 * all-reduced along with it */
void allreduce_pq(lsb_hip_solver *sv, unsigned cnt, int with2) {
  if (sv->p2p_on) {
    allreduce_parts(sv, 0, cnt, 1, with2, 1);
    return;
  }
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    if (with2) /* the sweep's partial sums ride in the same launch */
      lsb_k_reduce_final2(s->d_parts_pq, s->npq, 1, s->d_scal + 0, s->ar2_parts, s->ar2_n,
                          s->ar2_width, s->d_scal + 1, s->d_st, g_stream);
    else
      lsb_k_reduce_final(s->d_parts_pq, s->npq, 1, s->d_scal + 0, 0, s->d_st, g_stream);
  }
  allreduce_scal(sv, 0, cnt, 1);
}

/*
 * Exchange + SpMV of one iteration with the halo transfer hidden behind the
 * rows that do not need it: the exchange runs on its
 * own stream as soon as the vector is final, the interior row blocks start at
 * once on the compute stream, the boundary blocks wait for the halo.  Three
 * launches of the same kernel on sub-ranges of the row blocks; their partial
 * sums land in consecutive regions of the shard's partial buffer.
 */
int can_overlap(const lsb_hip_solver *sv) {
  if (!sv->multi || !sv->o.overlap)
    return 0;
  for (int i = 0; i < sv->nshard; i++) {
    const struct shard *s = &sv->sh[i];
    if (!(s->variant == LSB_SPMV_ADAPTIVE && s->ov_ok) && !(s->variant == LSB_SPMV_SELL && s->ov_sok))
      return 0;
  }
  /* auto: the split SpMV costs 2 launches (direct path) or 2 launches and two cross-stream events
   * (RCCL) -- 18-25 us per iteration on one device -- and can only win what a halo transfer takes,
   * which no rule of thumb knows: both forms are TIMED on the real communicator at creation
   * (overlap_setup) and every rank takes the decision from the same all-gathered numbers.  Until
   * that pass has run (and where it cannot): no split. */
  if (sv->o.overlap < 0)
    return sv->overlap_on > 0;
  return 1;
}

/* opts.overlap = -1: which form of the sharded SpMV is faster HERE -- the plain one behind the
 * exchange, or interior rows while the halo travels, boundary rows behind it.  24 iterations of the
 * solver's own iteration (exchange, all-reduce, everything) each way on b_i = i, best of two after a
 * warm-up, the slowest rank's time decides (all-gathered: every rank takes the same branch).  Untimed
 * set-up, like the transport's self-test.  Round 3 decided by halo size (>= 64 Ki doubles: split),
 * which every one-device measurement contradicted (config 4 as 8 virtual shards: 185 / 223 us split
 * against 167 / 198 plain, profiles/r03_cfg4_share.txt). */
void overlap_setup(lsb_hip_solver *sv) {
  sv->overlap_on = -1, sv->overlap_us[0] = sv->overlap_us[1] = 0.0;
  if (!sv->multi || sv->o.overlap >= 0 || sv->o.krylov == LSB_KRYLOV_GMRES || getenv("LSBENCH_HIP_NO_OVERLAP_TUNE"))
    return;
  unsigned can = 1; /* the split needs the prefix / interior / suffix shape on every shard of every rank */
  for (int i = 0; i < sv->nshard; i++) {
    const struct shard *s = &sv->sh[i];
    can &= (s->variant == LSB_SPMV_ADAPTIVE && s->ov_ok) || (s->variant == LSB_SPMV_SELL && s->ov_sok);
  }
  const int P = sv->dist ? lsb_hip_comm_size() : 1;
  unsigned *all = lsb_calloc(unsigned, 2 * (size_t)P), mine[2] = {can, 0};
  if (sv->dist)
    lsb_hip_comm_allgather_u32(mine, 1, all);
  else
    all[0] = can;
  for (int q = 0; q < P; q++)
    can &= all[q];
  if (!can) {
    free(all);
    return;
  }
  double *d_b = (double *)lsb_hip_malloc((size_t)sv->n_here * sizeof(double));
  double *d_x = (double *)lsb_hip_malloc((size_t)sv->n_here * sizeof(double));
  lsb_k_fill_index(sv->n_here, sv->row_first + 1u, d_b, g_stream);
  const struct lsb_hip_opts keep = sv->o;
  const int iters = 24;
  sv->o.tol = 0.0, sv->o.maxit = (unsigned)iters, sv->o.verify = 0, sv->o.sample_spmv = 0;
  for (int form = 0; form < 2; form++) {
    sv->overlap_on = form;
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
      struct lsb_hip_result r;
      solve_core(sv, d_b, d_x, &r);
      if (rep && r.seconds < best)
        best = r.seconds;
    }
    mine[form] = (unsigned)(best * 1e9 / iters); /* ns per iteration */
  }
  if (sv->dist)
    lsb_hip_comm_allgather_u32(mine, 2, all);
  else
    all[0] = mine[0], all[1] = mine[1];
  unsigned t[2] = {0, 0};
  for (int q = 0; q < P; q++)
    for (int f = 0; f < 2; f++)
      if (all[2 * q + f] > t[f])
        t[f] = all[2 * q + f];
  free(all);
  sv->o = keep;
  memset(sv->hint_iters, 0, sizeof sv->hint_iters);
  drop_graphs(sv);
  sv->overlap_us[0] = t[0] * 1e-3, sv->overlap_us[1] = t[1] * 1e-3;
  sv->overlap_on = t[1] < t[0];
  if (sv->o.verbose)
    fprintf(stderr, "hip_cdna4: sharded SpMV: %.1f us per iteration behind the exchange, %.1f us with the interior "
                    "rows in front of the halo -> %s\n", sv->overlap_us[0], sv->overlap_us[1],
            sv->overlap_on ? "split" : "plain");
  lsb_hip_free(d_b), lsb_hip_free(d_x);
}

/* part 0: the rows that need no halo; 1 / 2: the ones before / after them */
void spmv_range(struct shard *s, int part, double *y, double *partials, unsigned *np,
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
               s->d_pfull + s->row_begin, partials, np, st, NULL, &s->tail, g_stream);
}

/* rows of the split SpMV's part (see spmv_range) */
static unsigned range_len(const struct shard *s, int part) {
  const unsigned a = s->variant == LSB_SPMV_SELL ? s->ov_s1 : s->ov_b1;
  const unsigned b = s->variant == LSB_SPMV_SELL ? s->ov_s2 : s->ov_b2;
  const unsigned e = s->variant == LSB_SPMV_SELL ? s->nslice : s->nblk;
  return part == 0 ? (b > a ? b - a : 0) : part == 1 ? a : (e > b ? e - b : 0);
}

/* The all-reduce of {w.u; the sweep's r.u, r.r} folded into neighbouring
 * launches (hip_ar.h): the next SpMV launch of shard i is the last one of this
 * iteration's SpMV -- it carries the contribute phase; `before` dot partials of
 * earlier launches precede its own in d_parts_pq. */
static void arm_tail(lsb_hip_solver *sv, int i, unsigned before) {
  struct shard *s = &sv->sh[i];
  if (s->ar2_width != 2)
    errx(EXIT_FAILURE, "hip_cdna4: the folded all-reduce carries {w.u; r.u, r.r} only");
  lsb_p2p_fold_contribute(sv->p2p[i], &s->tail);
  s->tail.parts = s->d_parts_pq, s->tail.nparts_before = before;
  s->tail.parts2 = s->ar2_parts, s->tail.nparts2 = s->ar2_n, s->tail.width2 = s->ar2_width;
}

/* How single-reduction CG over the direct path runs its all-reduce (hip_ar.h):
 *   0  k_p2p_allreduce, one launch: contribute, wait, collect
 *   1  contribute in a launch of its own that waits for nobody; the wait and the
 *      collect at the head of the next k_cg1_update, behind its first loads (default)
 *   2  the contribute phase in the tail of the SpMV's last launch as well: no launch
 *      at all, but 13 us slower per iteration at 1.25 M rows -- the workgroups'
 *      hand-ins are ~1500 atomic adds to one word that all arrive when the launch
 *      ends (profiles/r02_shard_floor.txt); kept for that measurement
 * LSBENCH_HIP_AR_FOLD picks one; ranks need not agree (same stores, same order). */
int can_fold_allreduce(const lsb_hip_solver *sv) {
  const char *e = getenv("LSBENCH_HIP_AR_FOLD");
  int mode = e ? atoi(e) : 1;
  if (!sv->multi || !sv->p2p_on || mode <= 0)
    return 0;
  for (int i = 0; i < sv->nshard && mode == 2; i++)
    if (!lsb_k_spmv_has_tail(sv->sh[i].variant) || sv->sh[i].n == 0)
      mode = 1;
  return mode > 2 ? 1 : mode;
}

/* mode 1: this shard's sums to every rank's mailbox; nobody is waited for */
void allreduce_pq_contribute(lsb_hip_solver *sv) {
  for (int i = 0; i < sv->nshard; i++) {
    struct shard *s = &sv->sh[i];
    lsb_p2p_allreduce(sv->p2p[i], s->d_parts_pq, s->npq, 1, s->ar2_parts, s->ar2_n, s->ar2_width,
                      NULL, 0, s->d_scal, s->d_st, 1, g_stream);
  }
}

void exchange_and_spmv(lsb_hip_solver *sv, int sample) {
  const int fold = sv->fold_next;
  sv->fold_next = 0;
  if (!can_overlap(sv)) {
    exchange_p(sv, 1);
    for (int i = 0; i < sv->nshard; i++) {
      struct shard *s = &sv->sh[i];
      if (i == 0 && sample >= 0)
        LSB_CHK_HIP(hipEventRecord(sv->ev[4 * sample], g_stream));
      if (fold)
        arm_tail(sv, i, 0);
      spmv_shard(s, s->d_pfull, s->d_q, s->d_pfull + s->row_begin, s->d_parts_pq, &s->npq, s->d_st);
      s->tail.counter = NULL;
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
    LSB_CHK_HIP(hipStreamWaitEvent(comm_stream(), sv->ev_vec, 0));
    exchange_on(sv, comm_stream());
    LSB_CHK_HIP(hipEventRecord(sv->ev_halo, comm_stream()));    /* the halo has landed   */
  }
  if (sample >= 0)
    LSB_CHK_HIP(hipEventRecord(sv->ev[4 * sample], g_stream));
  unsigned na, nb, nc;
  for (int i = 0; i < sv->nshard; i++) {                         /* interior: no halo     */
    struct shard *s = &sv->sh[i];
    if (fold && !range_len(s, 1) && !range_len(s, 2))
      arm_tail(sv, i, 0);
    spmv_range(s, 0, s->d_q, s->d_parts_pq, &na, s->d_st);
    s->tail.counter = NULL;
    s->npq = na;
  }
  if (sv->p2p_halo) {
    for (int i = 0; i < sv->nshard; i++)
      lsb_p2p_recv(sv->p2p[i], sv->sh[i].d_pfull, sv->sh[i].d_st, g_stream);
  } else
    LSB_CHK_HIP(hipStreamWaitEvent(g_stream, sv->ev_halo, 0));
  for (int i = 0; i < sv->nshard; i++) {                         /* boundary rows         */
    struct shard *s = &sv->sh[i];
    if (fold && range_len(s, 1) && !range_len(s, 2))
      arm_tail(sv, i, s->npq);
    spmv_range(s, 1, s->d_q, s->d_parts_pq + s->npq, &nb, s->d_st);
    s->tail.counter = NULL;
    if (fold && range_len(s, 2))
      arm_tail(sv, i, s->npq + nb);
    spmv_range(s, 2, s->d_q, s->d_parts_pq + s->npq + nb, &nc, s->d_st);
    s->tail.counter = NULL;
    s->npq += nb + nc;
  }
  if (sample >= 0) {
    LSB_CHK_HIP(hipEventRecord(sv->ev[4 * sample + 1], g_stream));
    LSB_CHK_HIP(hipEventRecord(sv->ev[4 * sample + 2], g_stream));
    LSB_CHK_HIP(hipEventRecord(sv->ev[4 * sample + 3], g_stream));
  }
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
    exchange_p(sv, 1);
    for (int i = 0; i < sv->nshard && check; i++) {
      struct shard *s = &sv->sh[i];
      for (int k = 0; k < s->nrecv && sv->p2p_halo; k++)
        lsb_p2p_test_check_range(s->d_pfull, s->recv[k].offset, s->recv[k].count, (unsigned)t,
                                 d_bad, g_stream);
      lsb_p2p_test_setvals(s->d_scal, sv->dist ? lsb_hip_comm_rank() : i, (unsigned)t, g_stream);
    }
    if (check && sv->p2p_on && (t & 1)) {
      /* odd rounds: the split form single-reduction CG uses -- a contribute launch that
       * waits for nobody, then the collect the way k_cg1_update runs it (64 workgroups, no
       * fence), each comparing with the known sums */
      for (int i = 0; i < sv->nshard; i++)
        lsb_p2p_allreduce(sv->p2p[i], NULL, 0, 0, NULL, 0, 0, sv->sh[i].d_scal, 3, sv->sh[i].d_scal,
                          sv->sh[i].d_st, 1, g_stream);
      for (int i = 0; i < sv->nshard; i++)
        lsb_p2p_test_collect_check(sv->p2p[i], (unsigned)t, d_bad, sv->sh[i].d_st, g_stream);
      continue;
    }
    allreduce_scal(sv, 0, 3, 1);
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
  wait_event(sv, sv->ev_t1, "self-test of the communication paths (timed rounds)");
  LSB_CHK_HIP(hipEventElapsedTime(&ms, sv->ev_t0, sv->ev_t1));
  return ms * 1e3f / P2P_TIMED;
}

void p2p_setup(lsb_hip_solver *sv) {
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
    drain_stream(sv, "self-test of the direct xGMI path");
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
