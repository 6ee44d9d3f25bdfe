/*
 * Internal header of liblsbench_hip.so: the role src/lsbench-impl.h plays in
 * the reference (struct layouts + backend prototypes), plus the launcher
 * prototypes of the HIP shim (hip_kernels.hip) that hip_cdna4.c calls.
 */
#ifndef LSB_IMPL_H
#define LSB_IMPL_H

#include "lsbench_hip.h"
#include <err.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

#define lsb_calloc(T, n) ((T *)calloc((size_t)(n) > 0 ? (size_t)(n) : 1, sizeof(T)))

/* ---- device-side PCG scalars (one per shard, lives in HBM) -------------- */
struct lsb_pcg_state {
  double rz[2];    /* r.z, double-buffered by iteration parity              */
  double bb;       /* b.b                                                    */
  double thresh2;  /* tol^2 * b.b                                            */
  double rr;       /* r.r after the last completed iteration                 */
  double pq;       /* p.q of the last completed iteration (diagnostic)       */
  double alpha[2]; /* single-reduction CG: step length, parity-buffered      */
  int iters;       /* completed iterations                                   */
  int status;      /* LSB_STATUS_*; != 0 makes every later kernel a no-op    */
  int maxit;
  int pad;       /* single-reduction CG and the column form's two-launch iteration:
                    1 = the maxit-th update has run, the next launch turns it into
                    the final status (never set and tested in the same launch) */
  int xpend;     /* two-launch iteration on a z-column plan (k_pcg_col_px): 1, 2 = x is one
                    update behind -- x += alpha[0] p with p in direction buffer
                    xpend - 1; 3, 4 = two behind -- x += alpha[1] p'' + alpha[0] p, p in
                    buffer xpend - 3, p'' in the other; 0 = x is up to date      */
  int pad2_;
};

/* ---- the direct-xGMI all-reduce folded into neighbouring launches (hip_ar.h) ---
 * lsb_ar_tail: handed BY VALUE to the SpMV launch that writes the last dot
 * partials of an iteration; counter == NULL: no tail.  lsb_ar_collect: handed
 * to k_cg1_update, which then takes w.u, r.u, r.r from the mailbox instead of
 * from reduced scalars; mbox == NULL: off.  Filled by lsb_p2p_fold_*. */
struct lsb_ar_tail {
  unsigned *counter;          /* workgroups of the launch that have handed in */
  const double *parts;        /* ALL dot partials of this SpMV: the earlier launches' of a
                                 split SpMV first, then this launch's (one per workgroup) */
  const double *parts2;       /* the sweep's records: nparts2 of width2 */
  char *const *peer;          /* mailboxes of all ranks (device array) */
  unsigned long long epoch;
  unsigned nparts_before, nparts2, width2;
  int R, me;
};
/* A Chebyshev step riding in the epilogue of the 16-bit sliced-ELL SpMV (one shard):
 * with w = (S z)_row just formed, d = a d + b D^-1 (r - w) and z' = z + d for the lane's
 * own rows; z' goes to ANOTHER gather vector (other rows are still gathering z).
 * zout == NULL: off.  The same expression as k_cheb_step, bit for bit. */
struct lsb_cheb_epi {
  const double *r, *dinv; /* dinv == NULL: the constant dc */
  double *d, *zout;       /* local row indices */
  double dc, a, b;
};
struct lsb_ar_collect {
  char *mbox; /* the own mailbox */
  unsigned long long epoch;
  long long timeout; /* wall_clock64 ticks */
  int R;
};

/* ---- device-side GMRES(m) scalars ----------------------------------------- */
#define LSB_GMRES_MAX_RESTART 32
#define LSB_GMRES_PARTIALS 512
struct lsb_gmres_state {
  double bnorm, thresh, beta, resid, hnorm;
  double g[LSB_GMRES_MAX_RESTART + 1];
  double cs[LSB_GMRES_MAX_RESTART], sn[LSB_GMRES_MAX_RESTART], y[LSB_GMRES_MAX_RESTART];
  double R[LSB_GMRES_MAX_RESTART * LSB_GMRES_MAX_RESTART]; /* R[i*MAX + j] */
  int iters, status, maxit, restart, jlast, cycle_open;
};

/* Upper bound on per-launch partial sums any reduction kernel writes; the
 * consumer kernels re-reduce them in fixed order (deterministic). */
#define LSB_MAX_PARTIALS 2048
/* Workgroups of a launch that only streams vectors: three per CU.  More of them stream SLOWER once
 * the vectors come out of HBM (lsb_k_blas1_grid, profiles/r03_sweep_grid.txt). */
#define LSB_STREAM_GRID_CAP 768
/* Non-zeros staged in LDS per row block of the adaptive SpMV (16 KiB). */
#define LSB_BLOCK_NNZ 2048

/* ---- HIP shim: kernel launchers (hip_kernels.hip) ------------------------
 * All take raw device pointers and a hipStream_t as void*.  `st` may be NULL
 * for the stand-alone (non-PCG) use of a kernel; when given, a non-zero
 * st->status turns the launch into a no-op on the device. */
void lsb_k_spmv(int variant, unsigned n, const int *offs, const int *cols,
                const double *vals, const int *rowblk,
                const unsigned char *blklanes, unsigned nblk,
                unsigned lanes_per_row, unsigned flags, unsigned grid_cap,
                const double *x, double *y, const double *xdot,
                double *partials, unsigned *npartials,
                const struct lsb_pcg_state *st, const int *rowmap,
                const struct lsb_ar_tail *tail, void *stream);
int lsb_k_spmv_has_tail(int variant); /* the variant's kernel can carry an lsb_ar_tail */
unsigned lsb_k_spmv_grid(int variant, unsigned n, unsigned nblk,
                         unsigned lanes_per_row, unsigned grid_cap);
/* flags of the adaptive SpMV (picked by the timing pass at solver creation) */
#define LSB_SP_PREFETCH 1u
#define LSB_SP_NT 2u
#define LSB_SP_C16 4u /* sliced-ELL only: 16-bit column codes */
#define LSB_SP_F32 32u /* the values pointer holds fp32 (opts.precision = LSB_PREC_MIXED) */
void lsb_k_spmv_sell(unsigned flags, unsigned grid_cap, unsigned period, const unsigned *sptr,
                     unsigned s0, unsigned ns, unsigned n, unsigned row_begin, unsigned xlen,
                     const void *cols,
                     const int *sbase, const double *vals, const double *vconst, unsigned ulen,
                     const double *x, double *y, const double *xdot, double *partials,
                     unsigned *npartials,
                     const struct lsb_pcg_state *st, const struct lsb_ar_tail *tail,
                     const struct lsb_cheb_epi *epi, void *stream);
#define LSB_SP_TMPL 64u /* 16-bit sliced-ELL with constant slots: slice templates (k_spmv_tmpl) */
#define LSB_SP_DEFER 128u /* k_spmv_tmpl: a turn's y is parked in LDS and stored by the wave's next turn */
void lsb_k_spmv_tmpl(unsigned flags, unsigned grid_cap, unsigned period, const unsigned *sptr, unsigned s0,
                     unsigned ns, unsigned n, unsigned row_begin, unsigned xlen, const unsigned *srec,
                     const unsigned long long *mask, const struct lsb_sell_tmpl *td,
                     unsigned nfar, const int *sbase, const void *vals,
                     const double *vconst, const double *x, double *y, const double *xdot,
                     double *partials, unsigned *npartials, const struct lsb_pcg_state *st,
                     const struct lsb_ar_tail *tail, const struct lsb_cheb_epi *epi, void *stream);
/* the classic PCG iteration in two launches on a z-column plan (hip_kernels.hip: k_pcg_col_px) */
void lsb_k_pcg_col_px(unsigned grid_cap, unsigned period, const unsigned *plan, unsigned nitem, unsigned n,
                      const unsigned *sptr, const unsigned long long *mask, const struct lsb_sell_tmpl *td,
                      unsigned nfar, const int *sbase, const double *vals, const double *vconst, const double *r,
                      const double *pold, double *pnew, double *x, int xupd, double dc, double *partials,
                      unsigned *npartials, struct lsb_pcg_state *st, int parity, const double *parts2,
                      unsigned nparts2, void *stream);
/* the r half: alpha, r -= alpha S p with S p formed again out of p (q is never stored), partials of (r.z', r.r) */
void lsb_k_pcg_col_r(unsigned grid_cap, unsigned period, const unsigned *plan, unsigned nitem, unsigned n,
                     const unsigned *sptr, const unsigned long long *mask, const struct lsb_sell_tmpl *td, unsigned nfar,
                     const int *sbase, const double *vals, const double *vconst, const double *p, double *r, double dc,
                     struct lsb_pcg_state *st, int parity, int pbuf, int xtwo, const double *pq_parts, unsigned npq,
                     double *partials2, unsigned *npartials, void *stream);
void lsb_k_pcg_xfix(unsigned n, const double *p0, const double *p1, double *x, const struct lsb_pcg_state *st,
                    void *stream);
#define LSB_SP_COL 256u /* k_spmv_tmpl_col: the template layout walked in z-columns (whole launches of a
                           shard that has a column plan; implies LSB_SP_TMPL) */
void lsb_k_spmv_tmpl_col(unsigned flags, unsigned grid_cap, unsigned period, const unsigned *plan, unsigned nitem,
                         int centre0, unsigned n, unsigned row_begin, unsigned xlen, const unsigned *sptr,
                         const unsigned long long *mask, const struct lsb_sell_tmpl *td, unsigned nfar,
                         const int *sbase, const void *vals, const double *vconst, const double *x, double *y,
                         const double *xdot, double *partials, unsigned *npartials,
                         const struct lsb_pcg_state *st, const struct lsb_ar_tail *tail, void *stream);
void lsb_k_spmv_binned(unsigned flags, unsigned chunk_cap, const unsigned *chunk_begin, unsigned c0,
                       unsigned nchunk, const unsigned *rows, const unsigned *cols, const double *vals,
                       const double *x, double *y, const struct lsb_pcg_state *st, void *stream);
void lsb_k_spmv_twophase(unsigned nitems, const unsigned *item, const double *vals,
                         const unsigned short *colw, const unsigned *grp_first,
                         const unsigned long long *grp_mask, const unsigned *delta,
                         const unsigned short *roww, unsigned col_lo, unsigned cols, unsigned rows,
                         unsigned nbins, const unsigned *bin_ptr, double *prod, unsigned n,
                         const double *x, unsigned xlen, double *y, const double *xdot,
                         double *partials, unsigned *npartials, double *binparts,
                         const struct lsb_pcg_state *st, void *stream);
unsigned lsb_k_twophase_groups(unsigned nbins);
void lsb_k_reduce_final(const double *partials, unsigned nparts, unsigned width,
                        double *out, int take_sqrt,
                        const struct lsb_pcg_state *st, void *stream);
void lsb_k_reduce_final2(const double *pa, unsigned na, unsigned wa, double *outa,
                         const double *pb, unsigned nb, unsigned wb, double *outb,
                         const struct lsb_pcg_state *st, void *stream);
void lsb_k_dot(unsigned n, const double *a, const double *b, double *partials,
               unsigned *npartials, void *stream);
void lsb_k_axpy(unsigned n, const double *alpha, const double *x, double *y,
                void *stream);
void lsb_k_xpay(unsigned n, const double *beta, const double *x, double *y,
                void *stream);
void lsb_k_jacobi_setup(unsigned n, unsigned row_begin, const int *offs,
                        const int *cols, const double *vals, double *dinv,
                        int *nzero, void *stream);
void lsb_k_l1_setup(unsigned n, const int *offs, const double *vals, double *dinv, int *nzero,
                    void *stream);
void lsb_k_jacobi_apply(unsigned n, const double *dinv, const double *r,
                        double *z, void *stream);
void lsb_k_jacobi_sweep(unsigned n, double w, const double *dinv,
                        const double *b, const double *ax, double *x,
                        void *stream);
/* PCG fused sweeps */
/* dinv == NULL: the Jacobi diagonal is the constant dc (not read from memory) */
void lsb_k_pcg_init(unsigned n, const double *b, const double *dinv, double dc, double *x,
                    double *r, double *p, double *partials2,
                    unsigned *npartials, void *stream);
void lsb_k_pcg_init_state(struct lsb_pcg_state *st, const double *partials2,
                          unsigned nparts, double tol, int maxit,
                          void *stream);
void lsb_k_pcg_update_xr(unsigned n, const double *p, const double *q,
                         const double *dinv, double dc, double *x, double *r,
                         struct lsb_pcg_state *st, int parity,
                         const double *pq_parts, unsigned npq,
                         double *partials2, unsigned *npartials, void *stream);
void lsb_k_spmv_subwave_p(unsigned n, const int *offs, const int *cols, const double *vals,
                          unsigned lanes_per_row, const double *r, const double *dinv, double dc,
                          const double *pold, double *pnew, double *y, double *partials,
                          unsigned *npartials, struct lsb_pcg_state *st, int parity,
                          const double *parts2, unsigned nparts2, void *stream);
void lsb_k_pcg_update_p(unsigned n, const double *r, const double *dinv, double dc,
                        const double *pin, double *p, struct lsb_pcg_state *st, int parity,
                        const double *parts2, unsigned nparts2, void *stream);
void lsb_k_cg1_update(unsigned n, double *u, const double *w, const double *dinv, double dc,
                      double *p,
                      double *s, double *x, double *r, struct lsb_pcg_state *st, int parity,
                      const double *parts_gr, unsigned ngr, const double *parts_d, unsigned nd,
                      const struct lsb_ar_collect *collect, double *partials2, unsigned *npartials,
                      void *stream);
unsigned lsb_k_blas1_grid(unsigned n);
void lsb_k_set_blas1_nt(int on); /* mask: bit 0 x, 1 p and q, 2 r (k_pcg_update_xr); 3 r, 4 p (k_pcg_update_p);
                                    5 k_cg1_update; 1 = all; per host thread */
int lsb_k_get_blas1_nt(void);
void lsb_k_fill_index(unsigned n, unsigned first, double *v, void *stream);
void lsb_k_perm_gather(unsigned n, const int *perm, const double *src, double *dst,
                       void *stream);
void lsb_k_perm_scatter(unsigned n, const int *perm, const double *src, double *dst,
                        void *stream);
void lsb_k_vreduce(double *base, unsigned stride, unsigned nshard, unsigned off,
                   unsigned cnt, void *stream);

/* GMRES launchers (hip_gmres.hip) */
unsigned lsb_k_gm_grid(unsigned n);
void lsb_k_gm_resid(unsigned n, const double *b, const double *ax, double *v0,
                    double *partials, const struct lsb_gmres_state *st, void *stream);
void lsb_k_gm_begin(struct lsb_gmres_state *st, const double *partials, unsigned nparts,
                    double tol, int maxit, int restart, int first, void *stream);
void lsb_k_gm_scale_prec(unsigned n, const double *w, double *v, const double *dinv, double *z,
                         const struct lsb_gmres_state *st, void *stream);
void lsb_k_gm_multidot(unsigned n, const double *V, size_t ld, int cnt, const double *w,
                       double *partials, double *h, int accumulate,
                       const struct lsb_gmres_state *st, void *stream);
void lsb_k_gm_update_w(unsigned n, const double *V, size_t ld, int cnt, const double *h,
                       double *w, double *partials, const struct lsb_gmres_state *st,
                       void *stream);
void lsb_k_gm_hess(struct lsb_gmres_state *st, int j, const double *h, const double *h2,
                   const double *partials, unsigned nparts, void *stream);
void lsb_k_gm_finish_cycle(unsigned n, const double *V, size_t ld, const double *dinv, double *x,
                           struct lsb_gmres_state *st, void *stream);

/* ---- preconditioners with z as a vector (hip_precond_k.hip) -------------------- */
void lsb_k_dot2(unsigned n, const double *r, const double *z, double *partials2,
                unsigned *npartials, const struct lsb_pcg_state *st, void *stream);
void lsb_k_cheb_first(unsigned n, const double *r, const double *dinv, double dc, double c0,
                      double *d, double *z, const struct lsb_pcg_state *st, void *stream);
void lsb_k_cheb_step(unsigned n, const double *r, const double *w, const double *dinv, double dc,
                     double a, double b, double *d, double *z, const struct lsb_pcg_state *st,
                     void *stream);
void lsb_k_scale_dinv(unsigned n, double c, const double *dinv, const double *w, double *v,
                      void *stream);
void lsb_k_scale_vec(unsigned n, double c, double *v, void *stream);
void lsb_k_power_start(unsigned n, unsigned first, double *v, void *stream);
void lsb_k_bj_invert(unsigned n, unsigned bs, double *binv, double *scratch, void *stream);
void lsb_k_bj_apply(unsigned n, unsigned bs, const double *binv, const double *r, double *z,
                    double *part, const struct lsb_pcg_state *st, const double *skip2,
                    unsigned nskip, void *stream);
unsigned lsb_k_bj_chunks(unsigned bs);

/* ---- one-launch solve of launch-bound operators (hip_persist.hip) ------------ */
size_t lsb_k_persist_shared_bytes(void);
unsigned lsb_k_persist_limits(unsigned *nzmax, unsigned *rmax, unsigned *gmax);
int lsb_k_pcg_persist(unsigned n, unsigned G, unsigned stride, const unsigned *wg_row,
                      const int *offs, const int *cols, const double *vals, const double *dinv,
                      const double *b, double *x, double *ug, void *shared,
                      struct lsb_pcg_state *st, double tol, int maxit, unsigned lanes,
                      long long timeout_ticks, void *stream);

/* ---- direct xGMI path (hip_p2p.hip) ---------------------------------------- */
struct lsb_p2p;
struct lsb_p2p *lsb_p2p_create_dist(const struct lsb_xfer *recv, int nrecv,
                                    const struct lsb_xfer *send, int nsend);
int lsb_p2p_create_virtual(struct lsb_p2p **out, int n, struct lsb_xfer *const *recv,
                           const int *nrecv, struct lsb_xfer *const *send, const int *nsend);
void lsb_p2p_destroy(struct lsb_p2p *p);
int lsb_p2p_has_halo(const struct lsb_p2p *p);
void lsb_p2p_send(struct lsb_p2p *p, const double *d_full, const struct lsb_pcg_state *st,
                  void *stream);
void lsb_p2p_recv(struct lsb_p2p *p, double *d_full, struct lsb_pcg_state *st, void *stream);
void lsb_p2p_sendrecv(struct lsb_p2p *p, double *d_full, struct lsb_pcg_state *st, void *stream);
void lsb_p2p_allreduce(struct lsb_p2p *p, const double *parts, unsigned nparts, unsigned width,
                       const double *parts2, unsigned nparts2, unsigned width2,
                       const double *extra, unsigned nextra, double *out,
                       struct lsb_pcg_state *st, int phases, void *stream);
void lsb_p2p_fold_contribute(struct lsb_p2p *p, struct lsb_ar_tail *t);
void lsb_p2p_fold_collect(const struct lsb_p2p *p, struct lsb_ar_collect *c);
void lsb_p2p_test_collect_check(struct lsb_p2p *p, unsigned round, unsigned *d_bad,
                                struct lsb_pcg_state *st, void *stream);
void lsb_p2p_test_pattern(double *d_full, size_t goff, size_t n, unsigned round, void *stream);
void lsb_p2p_test_check_range(const double *d_full, size_t goff, size_t n, unsigned round,
                              unsigned *d_bad, void *stream);
void lsb_p2p_test_setvals(double *d_v, int me, unsigned round, void *stream);
void lsb_p2p_test_checkvals(const double *d_v, int R, unsigned round, unsigned *d_bad,
                            void *stream);

/* ---- backend internals shared between hip_cdna4.c and hip_comm.c -------- */
void *lsb_hip_stream(void);
int lsb_hip_is_initialized(void);
/* grouped point-to-point exchange of contiguous ranges of a device vector */
int lsb_hip_comm_exchange(double *d_full, const struct lsb_xfer *sends,
                          int nsend, const struct lsb_xfer *recvs, int nrecv,
                          void *stream);
int lsb_hip_comm_allgather_u32(const unsigned *mine, unsigned count,
                               unsigned *all);
int lsb_hip_comm_allreduce_stream(double *d_buf, int count, void *stream);

#define LSB_CHK_HIP(call)                                                      \
  do {                                                                         \
    hipError_t e_ = (call);                                                    \
    if (e_ != hipSuccess)                                                      \
      errx(EXIT_FAILURE, "%s:%d hip error: %s", __FILE__, __LINE__,            \
           hipGetErrorString(e_));                                             \
  } while (0)

#ifdef __cplusplus
}
#endif
#endif
