/*
 * Internals of the solver object shared by the host-side C files of the HIP
 * backend:
 *   hip_cdna4.c      backend entry points (init / finalize / bench), options,
 *                    device memory helpers, kernel-level C-ABI
 *   hip_solver.c     shards: upload, kernel forms, timing pass, create / destroy
 *   hip_dist.c       what sharded solves add: exchange, all-reduce, overlap,
 *                    the direct xGMI path's set-up
 *   hip_pcg.c        PCG / single-reduction PCG iteration and the host loop
 *   hip_gmres_drv.c  GMRES(m) driver
 * Nothing here is part of the C-ABI (include/lsbench_hip.h).
 */
#ifndef HIP_SOLVER_H
#define HIP_SOLVER_H

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <string.h>
#include <strings.h>
#include <time.h>

#include "lsb_impl.h"

#define LSB_INTERNAL __attribute__((visibility("hidden")))

/* backend globals (defined in hip_cdna4.c; reference style, src/cusparse.c:33-36) */
/* The streams, the communicator (hip_comm.c) and the last result are per host
 * THREAD: a rank is a thread -- the caller's own when there is one GPU or one
 * process per GPU, one of hip_multi.c's workers when hip_cdna4_bench drives
 * several GPUs from the one caller process. */
extern LSB_INTERNAL int lsb_initialized;
extern LSB_INTERNAL __thread hipStream_t g_stream; /* compute */
extern LSB_INTERNAL __thread struct lsb_hip_result g_last;
LSB_INTERNAL hipStream_t comm_stream(void); /* halo exchange behind interior rows; made on first use */
LSB_INTERNAL void rank_thread_attach(int device);
LSB_INTERNAL void rank_thread_detach(void);
/* hip_multi.c */
LSB_INTERNAL int bench_multi(double *x, struct csr *A, const double *r, const struct lsbench *cb,
                             const struct lsb_hip_opts *o, int ngpus);

/* ------------------------------------------------------------------------ */
/* solver object                                                             */
/* ------------------------------------------------------------------------ */
#define SCAL_STRIDE 8 /* doubles per shard in the scalar slab */
#define MAX_SAMPLES 64
#define LSB_NGRAPH 4 /* cached hipGraphs: whole-solve + continuation chunk, solve proper + correction */

struct shard {
  /* ONE allocation for the vectors the iteration streams (r, q, the gather vector, the Jacobi
   * diagonal, the single-reduction form's p / s, the preconditioners' z vectors): their placement
   * relative to one another is then the same in every solver of every process -- which of their
   * lines compete for the same sets of the 256 MB Infinity Cache no longer depends on which
   * physical pages a dozen separate hipMallocs happened to get ("Where the
   * vectors land"); used where the vectors are of that cache's scale (<= 160 MB each: larger ones measured
   * slower out of one allocation).  shard_vec() carves 256-byte aligned pieces; what does not fit (or arrives with
   * LSBENCH_HIP_NO_SLAB=1) is a hipMalloc of its own; shard_vec_free() tells the two apart. */
  char *d_slab;
  size_t slab_cap, slab_used;
  unsigned row_begin, n;
  unsigned long long nnz;
  int *d_offs, *d_cols, *d_rowblk;
  unsigned char *d_blklanes;
  unsigned sp_flags, sp_grid; /* adaptive-SpMV flavour, picked by tune_spmv() */
  unsigned col_period;             /* slices per plane of the z-column plan (a period of at least 8 slices) */
  unsigned sp_period, sell_period; /* sliced-ELL: slices per plane the XCD dealing follows (0 =
                                      contiguous eighths); candidate found at upload */
  double *d_vals, *d_dinv, *d_r, *d_q, *d_pfull;
  double *d_p1, *d_s1; /* single-reduction CG: p and s = S p (pfull then holds u) */
  unsigned npq, np2;   /* partial counts of the SpMV / sweep launches */
  const double *ar2_parts; /* sweep partials the next all-reduce folds in */
  unsigned ar2_n, ar2_width;
  struct lsb_cheb_epi epi; /* zout != NULL: the next 16-bit sliced-ELL launch of this shard carries a
                              Chebyshev step in its epilogue (precond_apply arms and clears it) */
  double *d_zfull2;        /* its second gather vector: z' of step k is step k+1's z */
  struct lsb_ar_tail tail; /* counter != NULL: the next SpMV launch of this shard carries the
                              all-reduce's contribute phase (exchange_and_spmv arms and clears it) */
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
  int dinv_uniform;   /* all entries of dinv equal dinv_const */
  double dinv_const;
  unsigned *d_sptr16;
  short *d_scodes;
  int *d_sbase;
  double *d_svals16;
  double *d_svconst;       /* != NULL: the constant-slot layout (lsb_sell16_value_slots): d_sbase holds 4
                              ints per slot, d_svals16 only the sell_vslots slots that keep their values */
  unsigned sell_vslots, sell_slots;
  unsigned long long sell16_bytes, sell32_bytes; /* matrix-side bytes one launch of the form streams */
  /* slice templates of the constant-slot layout (lsb_sell16_templates; LSB_SP_TMPL in sp_flags) */
  unsigned *d_srec; /* per slice {template id (255: none), first kept value slot, first mask, 0} */
  unsigned long long *d_tmask;
  unsigned n_glob; /* columns of the operator = length of the gather vector */
  struct lsb_sell_tmpl *d_tmpl;
  unsigned tmpl_nfar, tmpl_count;
  unsigned long long tmpl_pure, tmpl_shaped, tmpl_bytes;
  /* z-column plan of the template layout (lsb_sell_tmpl_columns; LSB_SP_COL in sp_flags): xbeg[9],
   * padding to 16 unsigneds, 16-byte items */
  unsigned *d_colplan, *d_colplan_in; /* all slices; the interior range [ov_s1, ov_s2) of the split SpMV */
  unsigned col_items, col_items_in, col_kmax;
  int col_centre0;
  unsigned long long col_slices; /* slices inside columns */
  unsigned long long col_bytes;  /* matrix-side bytes one launch of the z-column walk streams */
  unsigned sell_ulen;      /* != 0: every slice of the 16-bit copy has this many slots */
  double *d_parts_pq, *d_parts2;
  double *d_scal; /* [0] p.q   [1] r.z'  [2] r.r   (multi-shard path) */
  struct lsb_pcg_state *d_st;
  /* status word for communication steps that belong to no running solve (the
   * SpMV entry point, the first all-reduce of a solve): a time-out of the direct
   * xGMI path is recorded here and reported by the host (check_aux_status) */
  struct lsb_pcg_state *d_st_aux;
  unsigned nblk, lanes;
  int variant;
  unsigned col_lo, col_hi; /* column hull referenced by the shard's rows */
  /* column-panel form (LSB_SPMV_PANEL), built for scattered operators only */
  unsigned pn;       /* panels, 0 = not built */
  unsigned *h_pblk;  /* pn+1: first row block of each panel */
  int *pd_offs, *pd_cols, *pd_rowmap, *pd_rowblk;
  unsigned char *pd_blklanes;
  double *pd_vals;
  /* opts.precision = LSB_PREC_MIXED: the SpMV forms stream fp32 values (the sliced-
   * ELL value arrays then HOLD floats; d_vals32 is the CSR's copy); d_vals stays
   * fp64 for the residual of the refinement.  exact32: rounding changed nothing */
  int mixed, exact32;
  float *d_vals32;
  /* preconditioners that produce z = M^-1 r as a vector (hip_precond.c) */
  double *d_zfull, *d_z; /* z: a gather vector of its own (Chebyshev), or n doubles */
  double *d_chd;         /* Chebyshev: the recurrence's direction vector */
  double *d_binv, *d_bjpart; /* block-Jacobi: inverted diagonal blocks; chunk partial sums */
  unsigned bj_bs;
  /* FSAI (LSB_PRECOND_FSAI): G and G^T as CSR of their own, z = G^T (G r) through the row kernels */
  struct fsai_csr {
    int *offs, *cols, *rowblk;
    unsigned char *blklanes;
    double *vals;
    unsigned nblk, lanes;
    int variant;
    unsigned long long nnz;
  } fs_g, fs_gt;
  double *d_fst;     /* t = G r */
  double *d_r1;      /* the three-launch iteration's second residual buffer */
  unsigned fs_maxrow; /* longest row of the pattern */
  /* binned form (LSB_SPMV_BINNED), built for scattered operators only */
  unsigned bn, bcap;   /* bins (0 = not built), entries per chunk */
  unsigned *h_binchunk; /* bn+1: first chunk of each bin (host) */
  unsigned *bd_chunk, *bd_rows, *bd_cols;
  double *bd_vals;
  /* two-phase form (LSB_SPMV_TWOPHASE), built for scattered operators only */
  unsigned tp_items, tp_bins, tp_col_lo, tp_xlen, tp_cols, tp_rows; /* tp_bins = 0: not built */
  unsigned *tp_item, *tp_binptr, *tp_first, *tp_delta;
  unsigned long long *tp_mask;
  unsigned short *tp_colw, *tp_roww;
  double *tp_vals, *tp_prod, *tp_binparts;
  struct lsb_xfer *recv, *send;
  int nrecv, nsend;
};

struct lsb_hip_solver {
  unsigned n_glob;   /* rows of the whole operator                         */
  unsigned n_here;   /* rows held by this process (sum over its shards)     */
  unsigned n_user;   /* ... as the caller counts them: n_here less the pad rows of a line-padded grid */
  int padded;        /* the operator was line-padded (lsb_csr_pad_lines): d_perm maps internal rows to the
                        caller's, -1 on pad rows */
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
  } gcache[LSB_NGRAPH];
  int gnext;
#define LSB_MAX_CORRECTIONS 6
#define LSB_MIXED_INNER_TOL 1e-5 /* what an inner solve on fp32-rounded values is asked for */
  unsigned hint_iters[LSB_MAX_CORRECTIONS + 1]; /* iterations of the previous solve and of each of
                                                   its correction runs, 0 = none yet */
  double tol_run;    /* tolerance of the CG run being enqueued (opts.tol, or a correction's) */
  double *d_vr, *d_ve; /* opts.verify: right-hand side and solution of a correction run */
  unsigned agree_nnz, agree_n; /* distributed: largest shard, identical on all ranks */
  unsigned agree_halo;         /* largest halo (doubles) any shard receives from one peer */
  /* opts.overlap = -1: the split SpMV (interior rows while the halo travels) against the plain one,
   * timed on the real communicator at creation (overlap_setup): -1 undecided, else the choice; the
   * two timings in us per iteration (max over ranks), 0 where the pass did not run */
  int overlap_on;
  double overlap_us[2];
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
  unsigned char samp_skip[MAX_SAMPLES];         /* the sample brackets nothing (a run's first iteration in the
                                                   two-launch form: a plain SpMV launch, not k_pcg_col_px) */
  int have_events;
  double *d_tmp; /* n_here doubles: scratch for spmv_dev / jacobi sweep */
  /* direct xGMI path (hip_p2p.hip), one context per shard; p2p_on: used for
   * the all-reduces, p2p_halo: also for the halo exchange */
  /* single-reduction PCG without the vector u = D^-1 r: every shard of every
   * rank has the same constant Jacobi diagonal (k_cg1_update<UI>) */
  int cg1_implicit;
  int pcur, rcur; /* launch-bound fused paths: which direction / residual buffer is current */
  int env_no_fuse_p, env_no_fuse_px; /* LSBENCH_HIP_NO_FUSE_P / _NO_FUSE_PX at creation (tests compare the forms) */
  int nt_mask;    /* which operands of the BLAS-1 sweeps are loaded nontemporal (tune_blas1_nt) */
#define LSB_CHEB_MAX 32
  int cheb_m, cheb_fused; /* fused: the steps ride in the SpMV's epilogue (one shard, 16-bit sliced-ELL) */
  double cheb_lmin, cheb_lmax, cheb_c0, cheb_a[LSB_CHEB_MAX], cheb_b[LSB_CHEB_MAX];
  unsigned nspmv; /* SpMV launches (per shard) of the solve being enqueued */
  /* launch-bound operators: the whole solve as one persistent launch (hip_persist.hip) */
  struct {
    int ok, use;          /* qualifies / chosen */
    unsigned G, stride, lanes;
    unsigned *d_wgrow;    /* G+1 row bounds of the workgroups */
    double *d_ug;         /* the shared vector u */
    void *d_shared;       /* barrier counter + partial records */
    double us_persist, us_launches; /* creation-time timing of 40 iterations each way */
  } ps;
  struct lsb_p2p **p2p;
  int p2p_on, p2p_halo;
  /* single-reduction CG over the direct path: the all-reduce's collect phase rides at the
   * head of k_cg1_update (ar_fold = 1), its contribute phase in the SpMV's last launch too
   * (ar_fold = 2) -- hip_ar.h, can_fold_allreduce.  fold_next: the next exchange_and_spmv
   * arms the tails; ar_pending: a contribution is out, the next k_cg1_update collects it */
  int ar_fold, fold_next, ar_pending;
  double p2p_us, rccl_us; /* self-test: one exchange + all-reduce, each way */
};

/* hip_fsai.hip */
void lsb_k_fsai_rows(const unsigned *rows, unsigned nrows, unsigned mcap, const int *offs, const int *cols,
                     const double *vals, unsigned row_begin, const unsigned *poffs, const unsigned *pcols,
                     double *gvals, int *bad, void *stream);
void lsb_k_fsai_xr_gr(unsigned n, const int *goffs, const int *gcols, const double *gvals, unsigned lanes,
                      const double *p, const double *q, double *x, const double *rold, double *rnew, double *t,
                      struct lsb_pcg_state *st, int parity, const double *pq_parts, unsigned npq, void *stream);
void lsb_k_fsai_gt_dots(unsigned n, const int *offs, const int *cols, const double *vals, unsigned lanes,
                        const double *t, double *z, const double *r, double *partials2, unsigned *npartials,
                        const struct lsb_pcg_state *st, void *stream);
LSB_INTERNAL int fsai_three_launches(const lsb_hip_solver *sv);
/* hip_cdna4.c */
LSB_INTERNAL double wall_seconds(void);
/* Leave the process from a state in which a stream of this process may never drain (a hung
 * collective, a peer that never arrived): message, flush, _exit(EXIT_FAILURE).  Never exit():
 * exit() runs the HIP runtime's teardown, which waits for exactly that stream. */
LSB_INTERNAL void lsb_give_up(const char *fmt, ...) __attribute__((noreturn, format(printf, 1, 2)));
/* hipStreamSynchronize(g_stream) that gives up after opts.comm_deadline_s on a sharded solver */
LSB_INTERNAL void drain_stream(lsb_hip_solver *sv, const char *what);
LSB_INTERNAL void wait_event(lsb_hip_solver *sv, hipEvent_t ev, const char *what);
LSB_INTERNAL void *dev_upload(const void *h, size_t bytes);
/* hip_solver.c */
LSB_INTERNAL unsigned pow2_ceil(unsigned v);
LSB_INTERNAL double *shard_vec(struct shard *s, size_t count);
LSB_INTERNAL void shard_vec_free(struct shard *s, void *p);
LSB_INTERNAL void sell_launch(struct shard *s, unsigned s0, unsigned ns, const double *xfull, double *y,
                              const double *xdot, double *partials, unsigned *np,
                              const struct lsb_pcg_state *st);
LSB_INTERNAL void spmv_shard(struct shard *s, const double *xfull, double *y, const double *xdot,
                             double *partials, unsigned *np, const struct lsb_pcg_state *st);
/* the same with the fp64 values whatever opts.precision says (residuals, y = Op x) */
LSB_INTERNAL void spmv_shard_exact(struct shard *s, const double *xfull, double *y, const double *xdot,
                                   double *partials, unsigned *np, const struct lsb_pcg_state *st);
LSB_INTERNAL void tune_spmv(lsb_hip_solver *sv, struct shard *s);
/* hip_dist.c */
LSB_INTERNAL void p2p_setup(lsb_hip_solver *sv);
LSB_INTERNAL void overlap_setup(lsb_hip_solver *sv);
LSB_INTERNAL void exchange_on(lsb_hip_solver *sv, hipStream_t stream);
LSB_INTERNAL void exchange_p(lsb_hip_solver *sv, int gated);
LSB_INTERNAL void check_aux_status(lsb_hip_solver *sv, const char *where);
LSB_INTERNAL void allreduce_scal(lsb_hip_solver *sv, unsigned off, unsigned cnt, int gated);
LSB_INTERNAL void allreduce_pq(lsb_hip_solver *sv, unsigned cnt, int with2);
LSB_INTERNAL int can_overlap(const lsb_hip_solver *sv);
LSB_INTERNAL int can_fold_allreduce(const lsb_hip_solver *sv);
LSB_INTERNAL void allreduce_pq_contribute(lsb_hip_solver *sv);
LSB_INTERNAL void exchange_and_spmv(lsb_hip_solver *sv, int sample);
LSB_INTERNAL double true_resid2(lsb_hip_solver *sv, const double *d_b, const double *d_x);
/* hip_pcg.c */
LSB_INTERNAL int lsb_fuse_p_kind(const lsb_hip_solver *sv);
LSB_INTERNAL void tune_blas1_nt(lsb_hip_solver *sv);
LSB_INTERNAL void drop_graphs(lsb_hip_solver *sv);
LSB_INTERNAL void persist_setup(lsb_hip_solver *sv);
LSB_INTERNAL int solve_core(lsb_hip_solver *sv, const double *d_b, double *d_x,
                            struct lsb_hip_result *res);
/* hip_precond.c */
LSB_INTERNAL int generic_precond(const lsb_hip_solver *sv);
LSB_INTERNAL void precond_shard_blocks(struct shard *s, const int *offs, const int *cols,
                                       const double *vals, const struct lsb_hip_opts *o);
LSB_INTERNAL void precond_setup(lsb_hip_solver *sv);
LSB_INTERNAL void precond_apply(lsb_hip_solver *sv, int after_update);
LSB_INTERNAL void precond_free_shard(struct shard *s);
/* hip_gmres_drv.c */
LSB_INTERNAL int gmres_solve_dev(lsb_hip_solver *sv, const double *d_b, double *d_x,
                                 struct lsb_hip_result *res);

#endif
