/*
 * C-ABI of the MI355X (gfx950) sparse-solve backend for lsbench.
 *
 * Everything here is `extern "C"`, plain pointers and sizes.  The library that
 * exports it is liblsbench_hip.so (built by lsbench_amd/csrc/Makefile).  Three
 * layers, outermost first:
 *
 *   1. the backend trio  hip_cdna4_{init,finalize,bench}  -- the drop-in
 *      boundary: the same three-function contract every reference backend
 *      implements (reference: src/lsbench-impl.h:42-68; template
 *      src/cusparse.c:137-213);
 *   2. the solver-handle API  lsb_hip_solver_*  -- what hip_cdna4_bench is made
 *      of, exposed so a caller can keep the operator resident in HBM and solve
 *      repeatedly (bench.py, tests);
 *   3. kernel-level entry points  lsb_hip_<op>_f64  -- one per hand-written HIP
 *      kernel, raw device pointers + a hipStream_t passed as void*.
 *
 * Return convention (reference: src/cusparse.c:139-140,154-155,166-167):
 * 0 = ok, 1 = backend not initialised / already initialised / no device,
 * 2 = bad argument.  A failing HIP or RCCL call is fatal: errx(EXIT_FAILURE,
 * "file:line ...") exactly like the reference's chk_rt macro
 * (src/cusparse.c:24-31).  Nothing in this library falls back to a CPU path.
 */
#ifndef LSBENCH_HIP_H
#define LSBENCH_HIP_H

#include "lsbench.h"
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------ */
/* Data layouts handed across the boundary                                   */
/* ------------------------------------------------------------------------ */

/* Run configuration.  Layout = reference src/lsbench-impl.h:14-20. */
struct lsbench {
  char *matrix;
  lsbench_solver_t solver;
  lsbench_ordering_t ordering;
  lsbench_precision_t precision;
  unsigned verbose, trials;
};

/* CSR container.  Layout = reference src/lsbench-impl.h:22-26.
 * offs[nrows+1] is always 0-based; cols[] keep the file's base (0 or 1); rows
 * are sorted by column with duplicates summed (src/lsbench-csr.c:54-63). */
struct csr {
  unsigned nrows, base;
  unsigned *offs, *cols;
  double *vals;
};

/* ------------------------------------------------------------------------ */
/* 1. Backend trio (drop-in boundary)                                        */
/* ------------------------------------------------------------------------ */

/* Replaces nothing, adds the 7th backend next to cusparse_init
 * (src/lsbench-impl.h:42).  Creates the HIP stream; returns 1 quietly when no
 * device is present, because lsbench_init initialises every backend whatever
 * --solver says (src/lsbench.c:143-147). */
int hip_cdna4_init(void);

/* Counterpart of cusparse_finalize (src/lsbench-impl.h:43). */
int hip_cdna4_finalize(void);

/* Counterpart of cusparse_bench / cholmod_bench (src/lsbench-impl.h:44-45,
 * 57-58).  x: caller-owned, length nrows, zero on entry (initial guess),
 * receives the solution in original row order (src/cusparse.c:203-204).
 * r: right-hand side, length nrows.  Protocol (src/cholmod-impl.h:34-71):
 * untimed setup, cb->trials warm-up solves, cb->trials timed solves, one CSV
 * record on stdout.  The operator solved is the one CHOLMOD is given:
 * S = triu(A) + triu(A,1)^T (src/cholmod-impl.h:5-16). */
int hip_cdna4_bench(double *x, struct csr *A, const double *r,
                    const struct lsbench *cb);

/* Two optional companions of the trio for a host program's command line and loader
 * (integration/hip-flags.patch adds the calls to src/lsbench.c:84-133 and
 * src/lsbench-csr.c:29; with ENABLE_HIP=OFF the stub file answers 1 / NULL):
 * set one option of hip_cdna4_bench by name -- "tol", "maxit", "ngpus", "krylov",
 * "operator", "precond", "cheb-degree", "block-size", "precision", "nvirt", "restart",
 * "reorder", "comm", "overlap", "verify", ... (hip_cdna4.c: OPTS; the same names upper-
 * cased behind LSBENCH_HIP_ are the environment switches) -- 0 = set, 1 = no such option
 * or not a value it takes; and the backend's synthetic operators (configs 3-5) as a
 * `struct csr *` the reference's lsbench_matrix_free can free. */
int hip_cdna4_set_option(const char *name, const char *value);
struct csr *hip_cdna4_matrix_synth(const char *spec);

/* ------------------------------------------------------------------------ */
/* Options / results (additive; the reference has no such knobs)             */
/* ------------------------------------------------------------------------ */

enum { LSB_OP_CHOLMOD_UPPER = 0, /* S = triu(A)+triu(A,1)^T (default)      */
       LSB_OP_RAW = 1 };         /* the CSR exactly as handed in            */
enum { LSB_PRECOND_JACOBI = 0, LSB_PRECOND_NONE = 1,
       LSB_PRECOND_L1JACOBI = 2, /* M = diag(sum_j |S_ij|): SPD for every
                                    symmetric S with non-empty rows, no
                                    stored diagonal needed */
       LSB_PRECOND_CHEBYSHEV = 3,   /* z = p_k(D^-1 S) D^-1 r: Chebyshev polynomial
                                       of degree m = opts.cheb_degree on the interval
                                       [l / max(30, 16 m^2), l], l = 1.1 lmax of D^-1 S
                                       from a power iteration at setup (the smoother the
                                       reference's AMG backends configure,
                                       src/hypre.c:126-158, src/amgx.c:78-85) */
       LSB_PRECOND_BLOCKJACOBI = 4,   /* M = blockdiag(S) with opts.block_size rows
                                       per block, blocks inverted at setup (the
                                       block form of src/ginkgo.cpp:57-58's
                                       Jacobi preconditioner)                 */
       LSB_PRECOND_FSAI = 5 };        /* factorised sparse approximate inverse,
                                       M^-1 = G^T G with G on the pattern of tril(S^k),
                                       k = opts.fsai_power: the expensive part (one small
                                       dense SPD solve per row, on the device) happens
                                       once, outside the timed loop -- the reference's
                                       own protocol for its CPU path, which factorises
                                       in csr_init (src/cholmod-impl.h:25-26) and times
                                       only solves (:59-62); an application is two SpMVs,
                                       no triangular solve, no reduction        */
enum { LSB_KRYLOV_PCG = 0,    /* preconditioned CG (symmetric operators)    */
       LSB_KRYLOV_GMRES = 1,  /* restarted GMRES(m), right-preconditioned,
                                 for LSB_OP_RAW / unsymmetric operators;
                                 single shard only in this round            */
       LSB_KRYLOV_PCG1 = 2,   /* single-reduction CG (Chronopoulos-Gear): the
                                 same iterates with 2 launches and 1 global
                                 reduction per iteration instead of 3 and 2 */
       LSB_KRYLOV_AUTO = 3 }; /* PCG1 when the operator is spread over several
                                 shards (one collective less per iteration),
                                 PCG otherwise                              */
enum { LSB_SPMV_AUTO = 0,     /* pick by mean row length                    */
       LSB_SPMV_ADAPTIVE = 1, /* row-blocked: LDS-streamed short rows +
                                 wavefront-per-row long rows                */
       LSB_SPMV_SUBWAVE = 2,  /* 2..64 lanes per row, shuffle reduction     */
       LSB_SPMV_SCALAR = 3,   /* one lane per row (test/debug baseline)     */
       LSB_SPMV_PANEL = 4,    /* column panels with an L2-resident x window,
                                 for scattered rows (solver handle only)    */
       LSB_SPMV_SELL = 5,     /* sliced-ELL copy of the rows (lsb_csr_sellize):
                                 128-row slices stored column-major, two rows
                                 per lane, no LDS; for near-uniform row lengths */
       LSB_SPMV_TWOPHASE = 7, /* scattered operators: products streamed out by
                                 column chunk (x window in LDS), added up by row
                                 bin (y rows in LDS): lsb_csr_pbize             */
       LSB_SPMV_BINNED = 6 }; /* scattered operators: entries binned by column
                                 panel (an L2-sized window of x) and sorted by
                                 row inside a bin, streamed as (row, col, value)
                                 with a segmented reduction per row -- uniform
                                 work per lane whatever the row lengths
                                 (lsb_csr_binize; solver handle only)         */
enum { LSB_STATUS_RUNNING = 0, LSB_STATUS_CONVERGED = 1,
       LSB_STATUS_BREAKDOWN = 2, LSB_STATUS_MAXIT = 3,
       LSB_STATUS_COMM = 4 /* a peer never arrived (direct xGMI path) */ };
/* how sharded solves communicate */
enum { LSB_COMM_AUTO = 0,  /* direct xGMI stores where they validate and win */
       LSB_COMM_RCCL = 1,  /* RCCL send/recv + all-reduce                    */
       LSB_COMM_P2P = 2 }; /* direct xGMI stores, or fail                    */

struct lsb_hip_opts {
  double tol;        /* stop when ||r||_2 <= tol*||b||_2            [1e-12] */
  unsigned maxit;    /* iteration cap                               [20000] */
  int op_mode;       /* LSB_OP_*                          [CHOLMOD_UPPER]   */
  int precond;       /* LSB_PRECOND_*                             [JACOBI]  */
  int spmv_variant;  /* LSB_SPMV_*                                  [AUTO]  */
  int check_every;   /* iterations enqueued per host poll             [0=auto] */
  int use_graph;     /* replay iterations from a hipGraph (1 rank)     [1]  */
  int sample_spmv;   /* HIP-event-time every Nth SpMV launch (0=off)   [0]  */
  int nvirt;         /* >1: split into that many row-range shards on ONE
                        device, exchanging by device copies (test mode) [1] */
  int comm;          /* LSB_COMM_*                                      [0] */
  int overlap;       /* multi-shard: start the SpMV's interior rows while the
                        halo is in flight; 1 on, 0 off, -1 = on when some
                        halo is >= 64 Ki doubles (the split costs launches) [-1] */
  int spmv_tune;     /* -1: time the SpMV flavours at creation and keep the
                        fastest; >= 0: force flags (bit 0 prefetch, bit 1
                        nontemporal)                                   [-1] */
  int spmv_grid;     /* workgroup cap of the SpMV launch, 0 = tuned     [0] */
  int reorder;       /* 1: solve P S P^T with P = reverse Cuthill-McKee, permute
                        b, un-permute x (single shard)                  [0] */
  int krylov;        /* LSB_KRYLOV_*                                 [AUTO] */
  int restart;       /* GMRES restart length m, 1..32                  [30] */
  int verbose;
  int ngpus;         /* hip_cdna4_bench only: row-partition the operator over
                        this many GPUs of the node, all driven from the ONE
                        caller process (one host thread per GPU); 0 = every
                        visible device.  Also LSBENCH_HIP_NGPUS, --ngpus   [1] */
  int verify;        /* 1: a solve the recurrence calls converged is reported
                        only once ||b - S x||_2 <= tol ||b||_2 holds for the
                        residual RECOMPUTED from x; if it does not, CG restarts
                        on the true residual (correction solve) until it does;
                        lsb_hip_result.true_relres carries the number      [0] */
  int cheb_degree;   /* LSB_PRECOND_CHEBYSHEV: degree of the polynomial, 1..32 [4] */
  int block_size;    /* LSB_PRECOND_BLOCKJACOBI: rows per diagonal block,
                        2, 4, 8, 16 or 32                                  [8] */
  int precision;     /* LSB_PREC_FP64, or LSB_PREC_MIXED: matrix values stored
                        and streamed as fp32, vectors and every accumulation
                        in fp64, fp64 iterative refinement around it -- the
                        same tolerance on the fp64 operator's residual   [FP64] */
  int persistent;    /* launch-bound operators (everything fits one XCD's L2):
                        the whole solve as ONE launch confined to one XCD;
                        1 on, 0 off, -1 = time both forms at solver creation
                        and keep the faster (which form runs then depends on
                        a timing: iterates differ in the last bits between
                        the forms).  Measured 2x slower than the two-launch
                        iteration on tests/xn3b_A_18.txt, hence             [0] */
  double comm_deadline_s; /* sharded solves: the host gives a poll of the device
                        state at most this long before it reports a hung
                        collective and exits non-zero                    [120] */
  int fsai_power;    /* LSB_PRECOND_FSAI: G lives on the pattern of tril(S^k),
                        k = 1..3 (rows of more than 128 pattern entries are cut
                        to the 128 nearest the diagonal)                    [3] */
  int blas1_nt;      /* which operands of the BLAS-1 sweeps are loaded nontemporal:
                        -1 = timed per solver at creation (tune_blas1_nt); else a mask
                        (bit 0 x, 1 p and q, 2 r in k_pcg_update_xr; 3 r, 4 p in
                        k_pcg_update_p; 5 k_cg1_update; 1 = all)              [-1] */
};
enum { LSB_PREC_FP64 = 0, LSB_PREC_MIXED = 1 };

struct lsb_hip_result {
  unsigned iters;        /* PCG iterations performed                        */
  int status;            /* LSB_STATUS_*                                    */
  double relres;         /* sqrt(r.r / b.b) of the recurrence residual      */
  double seconds;        /* wall-clock of this solve (host, after sync)     */
  double spmv_ms;        /* mean duration of the sampled SpMV launches      */
  unsigned spmv_samples; /* how many launches were sampled                  */
  unsigned corrections;  /* opts.verify / mixed precision: correction solves
                            (restarts on the recomputed residual) it took   */
  double true_relres;    /* ||b - S x|| / ||b|| recomputed from x (fp64
                            operator); -1 when not computed                 */
  unsigned spmvs;        /* multiplications by S the solve enqueued up to its
                            last iteration (outer iterations x (1 + Chebyshev
                            degree) + set-up ones)                          */
};

void lsb_hip_opts_default(struct lsb_hip_opts *o);
/* Options used by hip_cdna4_bench (which has no room for them in its
 * signature).  Also read from the environment at hip_cdna4_init:
 * LSBENCH_HIP_TOL, LSBENCH_HIP_MAXIT, LSBENCH_HIP_OPERATOR=raw|upper,
 * LSBENCH_HIP_NVIRT, LSBENCH_HIP_GRAPH, LSBENCH_HIP_SPMV. */
void lsb_hip_set_opts(const struct lsb_hip_opts *o);
void lsb_hip_get_opts(struct lsb_hip_opts *o);
/* Result of the last timed trial of the last hip_cdna4_bench call. */
void lsb_hip_last_result(struct lsb_hip_result *res);
int lsb_hip_device_count(void);

/* ------------------------------------------------------------------------ */
/* Host-side matrix helpers (no GPU needed)                                  */
/* ------------------------------------------------------------------------ */

/* Frees a CSR made by any helper below or by lsbench_matrix_synth (the three
 * arrays and the struct, like lsbench_matrix_free, src/lsbench-csr.c:101-108;
 * the backend library does not export the reference's own symbols). */
void lsb_csr_free(struct csr *A);
/* The operator CHOLMOD factorises, as a full 0-based CSR (both triangles):
 * restates the triplet construction of src/cholmod-impl.h:5-21.  Caller frees
 * with lsb_csr_free (or lsbench_matrix_free: same malloc'd layout). */
struct csr *lsb_csr_symmetrize_upper(const struct csr *A);
/* Deep copy with cols rebased to 0. */
struct csr *lsb_csr_copy_base0(const struct csr *A);
/* Rows [r0, r1) as a new CSR that keeps GLOBAL column ids (base 0 in, base 0
 * out): the shard one rank owns under 1-D row-range partitioning. */
struct csr *lsb_csr_row_slice(const struct csr *A, unsigned r0, unsigned r1);
/* Row-range partition balanced by non-zeros: bounds[0..nparts], with
 * bounds[0]=0, bounds[nparts]=nrows (prefix-sum split of offs). */
int lsb_csr_partition_rows(const struct csr *A, unsigned nparts,
                           unsigned *bounds);
/* Row blocks for the adaptive SpMV: consecutive rows are packed greedily into
 * blocks of at most `cap` non-zeros; a row longer than `cap` gets a block of
 * its own.  Returns the number of blocks and a malloc'd array of nblk+1 row
 * ids in *rowblk (caller frees with free()). */
unsigned lsb_csr_row_blocks(const struct csr *A, unsigned cap,
                            unsigned **rowblk);
/* Lanes per row (1..64, a power of two) the adaptive SpMV uses to add up the
 * rows of each block; lanes[] holds nblk entries. */
void lsb_csr_block_lanes(const struct csr *A, const unsigned *rowblk,
                         unsigned nblk, unsigned char *lanes);
/* Reverse Cuthill-McKee ordering of a structurally symmetric CSR:
 * perm[new] = old (cf. the host permutation Q of src/cusparse.c:67-85).
 * Returns 0, or 2 on allocation failure. */
/* Line padding of a constant-coefficient 2-D grid operator (0-based, sorted rows): grid lines of nx
 * rows re-numbered to start at multiples of `slice` rows, the gained rows holding the same stencil
 * among themselves (block-diagonal: real block = S).  map[new row] = row of S, or -1 for a pad
 * row (malloc'ed; free()).  NULL where S is no such operator (offsets out of {-nx, -1, 0, 1, nx}
 * with one value each, no +-1 entry across a line end), nx is a multiple of `slice` already, or
 * the padding would exceed 1/16 of the rows.  See lsb_operator.c. */
struct csr *lsb_csr_pad_lines(const struct csr *S, unsigned slice, unsigned *nx, unsigned *nxp, int **map);
int lsb_csr_rcm(const struct csr *S, unsigned *perm);
/* P S P^T for perm[new] = old, as a new 0-based CSR (src/cusparse.c:87-97). */
struct csr *lsb_csr_permute_sym(const struct csr *S, const unsigned *perm);
/* max |row - col| over the stored entries */
unsigned lsb_csr_bandwidth(const struct csr *S);
/* Column-panel form of a CSR (for scattered operators): the (row, panel) pairs
 * of panel-major order as the rows of one CSR.  See lsb_csr_panelize. */
struct lsb_panel_csr {
  unsigned npanels, width, npairs, nrows;
  unsigned *pair_begin; /* npanels+1: first pair of each panel            */
  unsigned *pair_row;   /* npairs: original row of each pair              */
  unsigned *offs;       /* npairs+1 into cols/vals                        */
  unsigned *cols;       /* 0-based global column ids, panel-major copy    */
  double *vals;
};
struct lsb_panel_csr *lsb_csr_panelize(const struct csr *A, unsigned width);
void lsb_panel_csr_free(struct lsb_panel_csr *P);
/* Binned form of a CSR for LSB_SPMV_BINNED (scattered operators: the gather of a
 * row strays over far more of x than an XCD's 4 MiB L2 holds, so a row-major
 * sweep fetches a 128-byte line per non-zero).  The entries are binned by column
 * panel -- `width` consecutive columns, an L2-sized window of x -- and sorted by
 * row inside a bin (column order inside a row is kept); a bin is streamed as
 * (row, col, value) triplets and every lane does the same work whatever the row
 * lengths are.  A bin is cut into CHUNKS of whole (row, bin) runs holding at
 * most LSB_BIN_CHUNK entries (a longer run is a chunk of its own), so that
 * exactly one workgroup adds to a given y entry per bin: no atomics, results
 * bit-identical from run to run. */
#define LSB_BIN_CHUNK 2048
struct lsb_binned {
  unsigned nbins, width, nrows, nchunks;
  unsigned chunk_cap;    /* entries a chunk holds at most: LSB_BIN_CHUNK, or 1024 /
                            1536 / 2048 from LSBENCH_HIP_BIN_CHUNK (tuning)   */
  unsigned long long nnz;
  unsigned *bin_chunk;   /* nbins+1: first chunk of each bin                */
  unsigned *chunk_begin; /* nchunks+1: first entry of each chunk            */
  unsigned *rows;        /* nnz: row of each entry (local to A)             */
  unsigned *cols;        /* nnz: 0-based column                             */
  double *vals;
};
/* NULL when A or width is unusable, or the copy does not fit 32-bit offsets. */
struct lsb_binned *lsb_csr_binize(const struct csr *A, unsigned width);
void lsb_binned_free(struct lsb_binned *B);
/* Two-phase form of a CSR for LSB_SPMV_TWOPHASE (scattered operators; "propagation
 * blocking"): no gather ever leaves a compute unit.
 *   phase 1  the entries are cut by COLUMN into chunks of `cols` columns; a
 *            workgroup copies its chunk's window of x into LDS (8 * cols bytes)
 *            and streams its entries -- value (8 B), column inside the window
 *            (2 B) -- writing each product into the product array;
 *   phase 2  the product array is ROW-BIN major (`rows` rows per bin): a WAVEFRONT
 *            owns a bin, streams its slots -- product (8 B) + row inside the bin
 *            (2 B) -- and adds them into its LDS copy of the bin's rows
 *            (ds_add_f64, the wavefront's own 8 * rows bytes: nobody else touches
 *            them, so the order of the additions is the wavefront's program
 *            order), then writes the rows of y, coalesced.
 * Both orders keep the entries of one (chunk, bin) pair -- a PIECE -- together and in
 * the same order, chunk-major for phase 1 and bin-major for phase 2, so the slot of
 * a product is its entry index plus a per-piece constant (`delta`); which piece an
 * entry belongs to is read off two words per group of 64 entries (`grp_first`,
 * `grp_mask`).  The entries of a chunk start on a multiple of 64 (vals = 0 in the
 * gaps, never read).  18.2 B read/written per non-zero in phase 1, 10 B in phase 2,
 * instead of a 128-byte line per gather. */
#define LSB_PB_COLS 8192 /* defaults; LSBENCH_HIP_PB_COLS / _ROWS (<= 16384 / <= 4096) */
#define LSB_PB_ROWS 2048
struct lsb_pb {
  unsigned nrows, ncols_lo, nchunks, nbins, nitems, npieces;
  unsigned cols, rows;     /* chunk width (columns), bin height (rows)          */
  unsigned long long nnz;  /* = slots of the product array                      */
  unsigned long long nent; /* length of the phase-1 arrays (a multiple of 64)   */
  double *vals;            /* nent, in (chunk, bin, row, col) order             */
  unsigned short *colw;    /* nent: column - (ncols_lo + chunk * cols)          */
  unsigned *grp_first;     /* nent / 64: piece of the group's first entry       */
  unsigned long long *grp_mask; /* nent / 64: bit j > 0: entry j starts a piece */
  unsigned *delta;         /* npieces: slot - entry index, mod 2^32             */
  unsigned *item;          /* 3 * nitems: {chunk, first entry, end entry} of
                              the phase-1 work items                            */
  unsigned *bin_ptr;       /* nbins + 1: first slot of each bin                 */
  unsigned short *roww;    /* nnz: row - bin * rows, in slot order              */
};
/* cols, rows: 0 = the defaults */
struct lsb_pb *lsb_csr_pbize2(const struct csr *A, unsigned cols, unsigned rows);
struct lsb_pb *lsb_csr_pbize(const struct csr *A);
void lsb_pb_free(struct lsb_pb *P);
/* Host-side assertions of the bounds the two kernels rely on: every index they form against
 * arrays of prod_len / roww_len slots as the caller allocated them (the backend allocates
 * nnz + 2).  0 = all hold; else a code, and the violated rule in why[whylen].  deep: also
 * every entry's 16-bit offsets.  Run at every upload (shallow) and under ASan (deep). */
int lsb_pb_check(const struct lsb_pb *P, unsigned long long prod_len, unsigned long long roww_len, int deep,
                 char *why, size_t whylen);
/* Pattern of the factorised sparse approximate inverse (LSB_PRECOND_FSAI): row i holds the
 * columns j <= i that row i of S^power reaches (S taken as a graph: its stored pattern, both
 * triangles, diagonal included), cut to the `cap` largest (nearest the diagonal) where there
 * are more; sorted, the diagonal last.  offs[n+1], cols[offs[n]]; pure host logic. */
struct lsb_fsai_pattern {
  unsigned n, cap;
  unsigned long long nnz;
  unsigned *offs, *cols;
};
#define LSB_FSAI_CAP 128
struct lsb_fsai_pattern *lsb_csr_fsai_pattern(const struct csr *S, int power, unsigned cap);
void lsb_fsai_pattern_free(struct lsb_fsai_pattern *P);
/* Sliced-ELL copy of a CSR for LSB_SPMV_SELL: rows in slices of LSB_SELL_ROWS,
 * every slice padded to its longest row and stored column-major (entry j of
 * row 128s+i at sptr[s] + 128j + i), so that a wavefront's lane l reads the
 * j-th entries of rows 2l and 2l+1 as one int2 / double2.  Padding entries
 * carry value 0 and the row's last column (a column the row references anyway).
 * cols are 0-based whatever A->base is. */
#define LSB_SELL_ROWS 128
struct lsb_sell {
  unsigned nrows, nslice;
  unsigned long long stored; /* entries incl. padding = sptr[nslice]        */
  unsigned *sptr;            /* nslice+1                                    */
  int *cols;                 /* stored (+ LSB_SELL_ROWS slack); NULL in the
                                16-bit form                                 */
  double *vals;
  /* 16-bit form (lsb_csr_sellize16): column of the entry in slot j of row r =
   * r + row_begin + base + code with {base, k} = sbase[2q], sbase[2q+1],
   * q = sptr[s]/128 + j; k >= 0: the slot's 128 codes are codes[128k ..);
   * k = -1: all its live entries share one code, which has been folded into
   * the base (every slot of a structured-grid operator).  A slot of a slice holds
   * entries of ONE diagonal band (+-32767 around its base), so a row's entries
   * may be interleaved with padding (value 0: the kernel does not gather for
   * those).  Entries of a row keep their column order. */
  short *codes;              /* 128 * ncode_slots, or NULL in the 32-bit form */
  int *sbase;                /* 2 * stored / LSB_SELL_ROWS                  */
  unsigned ncode_slots;
};
/* entries a sliced-ELL copy of A would store (to decide before building it) */
unsigned long long lsb_csr_sell_stored(const struct csr *A);
/* NULL when the copy would not fit 32-bit offsets */
struct lsb_sell *lsb_csr_sellize(const struct csr *A);
/* The 16-bit form, 10 instead of 12 bytes per entry; row_begin = global index
 * of A's first row (column ids are global).  NULL when some slice would need
 * more than 255 slots or the copy does not fit 32-bit offsets. */
struct lsb_sell *lsb_csr_sellize16(const struct csr *A, unsigned row_begin);
void lsb_sell_free(struct lsb_sell *S);
/* Constant slots of the 16-bit form: a slot whose 128 values are one and the same
 * non-zero number -- a diagonal of a constant-coefficient stencil inside a slice, unit
 * weights of a graph Laplacian -- needs no value array.  slots[4q .. 4q+3] = {base, code
 * slot or -1 (as sbase), VALUE slot or -1, 0}; vconst[q] = the slot's value where the
 * value slot is -1; vals holds the nval_slots remaining slots' 128 values, packed in slot
 * order.  Lossless: the kernel forms the same products in the same order. */
struct lsb_sell_vc {
  unsigned long long nslots; /* = stored / LSB_SELL_ROWS of the copy it was made from */
  unsigned nval_slots;
  int *slots;     /* 4 * (nslots + 1) */
  double *vconst; /* nslots + 1 */
  double *vals;   /* (nval_slots + 1) * LSB_SELL_ROWS */
};
struct lsb_sell_vc *lsb_sell16_value_slots(const struct lsb_sell *S);
void lsb_sell_vc_free(struct lsb_sell_vc *V);
/* Slice TEMPLATES of the constant-slot layout.  A code-free slice is described by its slots'
 * {base, constant or "keeps its values"} records, and on a structured grid a handful of such
 * descriptions cover everything: slices with identical records share ONE template, tid[slice]
 * says which (255 = none: the slice goes the per-slot way) and vbase[slice] where the slice's
 * kept value slots start (they are consecutive: the k-th kept slot of the slice is value slot
 * vbase + k).  The device copy packs {tid, vbase, first mask, 0} into ONE 16-byte record per slice
 * (a single scalar load); the kernel then reads that and the template out of the scalar cache
 * instead of 24 bytes of cold records per slot.  Where a template holds a slot c with
 * base[c-1] = base[c]-1 and base[c+1] = base[c]+1 (the three inner diagonals of a stencil) it
 * is SHAPED: [nfar far slots][c-1, c, c+1][nfar far slots] with the set's one nfar -- lanes
 * take the operands of c-1 and c+1 from the centre's 16-byte pair by a lane shift: three
 * gathers instead of five on a 5-point row, none misaligned.  In a shaped template the far
 * slots and the centre are constant; c-1 and c+1 may keep their values (a grid line that
 * ends inside the slice: padding zeros there) -- and where those values are ONE number or
 * zero (every such slot of a constant-coefficient stencil), the slot is MASKED: the number
 * in the template, a 128-bit mask per slice and slot instead of 1 KB of values.  Every other
 * template is all-constant.  Lossless: the same products in the same order. */
#define LSB_TMPL_SLOTS 8
struct lsb_sell_tmpl { /* 176 bytes, read by scalar loads */
  int nslots;
  int shaped; /* 1: [nfar][c-1, c, c+1][nfar], nfar = the set's */
  int base[LSB_TMPL_SLOTS];
  int kidx[LSB_TMPL_SLOTS]; /* -1: constant cst[j]; k >= 0: the slice's k-th kept value slot
                               (kind 1) or k-th mask (kind 2) */
  int kind[LSB_TMPL_SLOTS]; /* 0 constant, 1 keeps its 128 values, 2 masked constant cst[j] */
  int pad_[2];
  double cst[LSB_TMPL_SLOTS];
};
struct lsb_sell_tmpls {
  unsigned nslice, ntmpl, nfar;
  unsigned long long covered, shaped; /* slices that have a template / a shaped one */
  unsigned long long nmask;           /* masks (two 64-bit words each: bit r = row r of the slice) */
  unsigned long long kept_read;       /* value slots the template kernel still reads (kind 1 slots
                                         of templated slices + the slices without a template) */
  unsigned char *tid;                 /* nslice + 8 */
  unsigned *vbase;                    /* 2 * (nslice + 8): {first kept value slot, first mask} */
  unsigned long long *mask;           /* 2 * (nmask + 1) */
  struct lsb_sell_tmpl *t;            /* ntmpl <= 254 */
};
/* NULL when fewer than 7/8 of the slices get a template, shaped ones cover less than 3/4,
 * or the copy has code slots (not a structured grid). */
struct lsb_sell_tmpls *lsb_sell16_templates(const struct lsb_sell *S, const struct lsb_sell_vc *V);
void lsb_sell_tmpls_free(struct lsb_sell_tmpls *T);
/* Host-side check of the bounds k_spmv_sell16's constant-slot path and k_spmv_tmpl rely on (their
 * unguarded 16-byte gathers; value-slot, mask and template indices), for the shard whose first
 * global row is row_begin, that holds nrows rows and gathers from a vector of xlen entries.  0 =
 * every rule holds; else a rule number, the rule in `why`.  T may be NULL (no templates).  Run by
 * the backend at every upload; deep != 0 also walks values and masks. */
int lsb_tmpl_check(const struct lsb_sell *S, const struct lsb_sell_vc *V, const struct lsb_sell_tmpls *T,
                   unsigned row_begin, unsigned nrows, unsigned xlen, int deep, char *why, size_t whylen);
/* Z-COLUMNS of the template layout (k_spmv_tmpl_col).  On a 3-D stencil whose planes are whole
 * slices (period = slices per plane) the outermost far slots of a shaped template reach exactly
 * one plane down and up: base[0] = base[c] - 128 period, base[last] = base[c] + 128 period.  The
 * operand pair a lane gathers for slot 0 of slice s + period is then the very pair it gathered for
 * the centre of slice s, and the centre pair of s + period is what slot `last` of s asked for: a
 * wavefront that walks s, s + period, s + 2 period, ... keeps three centre pairs in registers and
 * gathers ONE new plane per step instead of three.  A column is a run of 2..kmax such slices that
 * share one template and one set of mask words (the host checks it: every interior z-column of a
 * structured grid qualifies), so the template and the masks are looked at once per column.
 * Everything else -- first / last plane, slices that keep values, ragged ends -- is an item of
 * one slice and goes slot by slot off the slot records.
 * item[4i..4i+3] = {first slice, slices (1: a single slice; >= 2: a column), template id, first
 * mask}; the items run z-group after z-group (at most kmax planes each, as equal as they come),
 * positions ascending, and XCD k takes [xbeg[k], xbeg[k+1]): a contiguous run with an eighth of the
 * slices -- a band of planes.
 * Lossless: the same operands and products in the same order as k_spmv_tmpl / k_spmv_sell16. */
#define LSB_TMPL_COL_MAX 16
#define LSB_TMPL_COL_LOCKSTEP 0x80000000u /* bit 31 of an item's slice count: the four items of a workgroup's
                                             turn are columns of one length (a barrier per plane keeps them in step) */
struct lsb_tmpl_cols {
  unsigned nitem, kmax, period;
  unsigned s_lo, s_hi;          /* the slices the items cover: [s_lo, s_hi) */
  unsigned xbeg[9];
  unsigned *item;               /* 4 * (nitem + 1) */
  unsigned long long in_cols;   /* slices inside columns (the rest are single items) */
  int centre0;                  /* every column's centre slot has base 0 (the diagonal): where the fused
                                   dot is with the gathered vector itself, the centre pair is its operand */
};
/* NULL where the layout has no such columns (period < 8, no shaped template with plane-reaching
 * far slots) or fewer than 3/4 of the slices fall inside columns. */
struct lsb_tmpl_cols *lsb_sell_tmpl_columns(const struct lsb_sell_tmpls *T, unsigned period, unsigned kmax);
/* the same for the slices [s_lo, s_hi) only (the interior launch of a sharded SpMV that runs while
 * the halo travels): every slice of the range in exactly one item, none outside it */
struct lsb_tmpl_cols *lsb_sell_tmpl_columns_range(const struct lsb_sell_tmpls *T, unsigned period, unsigned kmax,
                                                  unsigned s_lo, unsigned s_hi);
void lsb_tmpl_cols_free(struct lsb_tmpl_cols *C);
/* The rules k_spmv_tmpl_col relies on, as host assertions (run at every upload): every slice in
 * [s_lo, s_hi) in exactly one item, none outside; a column's slices s + k period share the item's template -- shaped,
 * constant or masked slots only, outermost far slots one plane away -- and its mask words bit
 * for bit.  0 or a rule number with the rule in `why`. */
int lsb_tmpl_cols_check(const struct lsb_sell_tmpls *T, const struct lsb_tmpl_cols *C, char *why, size_t whylen);
/* mean |col - (row + row_begin)| over a sample of the rows */
double lsb_csr_mean_scatter(const struct csr *A, unsigned row_begin);
/* [lo,hi) column range referenced by A (0-based). */
void lsb_csr_col_hull(const struct csr *A, unsigned *lo, unsigned *hi);
/* One contiguous range [offset, offset+count) (in doubles) of the exchanged
 * vector, to or from shard/rank `peer`. */
struct lsb_xfer {
  int peer;
  size_t offset, count;
};
/* Exchange plan of shard `me` out of `nall`, from the table
 * hull[4q..4q+3] = {row_begin, nrows, col_lo, col_hi}; recv/send hold nall
 * entries.  Pure host logic (tested on the CPU with gloo ranks). */
void lsb_plan_exchange(int me, int nall, const unsigned *hull,
                       struct lsb_xfer *recv, int *nrecv,
                       struct lsb_xfer *send, int *nsend);
/* Synthetic operators (BASELINE.json configs 3-5), rows [r0,r1) only, global
 * 0-based column ids, generated on the host.  spec:
 *   "lap2d:nx=3162,ny=3162"          5-point Laplacian, diag 4, off -1
 *   "lap3d:nx=400,ny=400,nz=400"     7-point Laplacian, diag 6, off -1
 *   "lap2d:...,coef=K" / "lap3d:...,coef=K" (K != 0)  same pattern, GENERAL values: one
 *                                    weight in [1/2, 3/2) per grid edge from a counter-
 *                                    based hash keyed by K, diagonal = sum of the row's
 *                                    edge weights (Dirichlet) -- SPD, nothing to elide
 *   "powerlaw:n=8000000,avg=32,max=4096,seed=20240607[,spd=1]"
 * r1 = 0 means "to the last row".  *n_global receives the full row count. */
struct csr *lsbench_matrix_synth(const char *spec, unsigned r0, unsigned r1,
                                 unsigned *n_global);

/* ------------------------------------------------------------------------ */
/* 2. Solver handle: operator resident in HBM, repeated solves               */
/* ------------------------------------------------------------------------ */

typedef struct lsb_hip_solver lsb_hip_solver;

/* Whole matrix on this process' device (or on o->nvirt virtual shards of it).
 * Applies o->op_mode.  Untimed setup: counterpart of csr_init
 * (src/cusparse.c:47-125, src/cholmod-impl.h:1-32). */
lsb_hip_solver *lsb_hip_solver_create(const struct csr *A,
                                      const struct lsb_hip_opts *o);
/* One rank's shard of a row-partitioned operator: rows
 * [row_begin, row_begin + A_rows->nrows) with GLOBAL 0-based column ids, used
 * as-is (LSB_OP_RAW).  Collective over the communicator set up by
 * lsb_hip_comm_init_rank; every rank must call it. */
lsb_hip_solver *lsb_hip_solver_create_dist(const struct csr *A_rows,
                                           unsigned row_begin,
                                           unsigned n_global,
                                           const struct lsb_hip_opts *o);
void lsb_hip_solver_destroy(lsb_hip_solver *s);

/* Jacobi-PCG from x0 = 0.  Host buffers, local rows only (length n_local);
 * H2D of b and D2H of x are outside res->seconds. */
int lsb_hip_solver_solve(lsb_hip_solver *s, const double *b, double *x,
                         struct lsb_hip_result *res);
/* Same with device-resident b and x (length n_local each). */
int lsb_hip_solver_solve_dev(lsb_hip_solver *s, const double *d_b, double *d_x,
                             struct lsb_hip_result *res);
/* y_local = Op * x: d_x holds this rank's rows (length n_local); remote
 * entries are exchanged first when the solver is distributed. */
int lsb_hip_solver_spmv_dev(lsb_hip_solver *s, const double *d_x, double *d_y);
/* Time `reps` back-to-back launches of the solver's SpMV kernel with HIP
 * events on the solver's stream (after `warm` untimed ones); *ms_avg = mean
 * milliseconds per launch.  No exchange, local shard 0. */
int lsb_hip_solver_time_spmv(lsb_hip_solver *s, int warm, int reps,
                             double *ms_avg);
/* One weighted-Jacobi sweep x <- x + w D^-1 (b - Op x), device buffers. */
int lsb_hip_solver_jacobi_sweep_dev(lsb_hip_solver *s, double w,
                                    const double *d_b, double *d_x);

unsigned lsb_hip_solver_nrows_local(const lsb_hip_solver *s);
/* rows the solver added inside: 0, or the pad rows of a line-padded 2-D grid (lsb_csr_pad_lines;
 * a constant-coefficient grid of >= 1 M rows whose lines are not whole slices, LSBENCH_HIP_PAD_LINES=0
 * turns it off).  b and x keep the caller's numbering and length. */
int lsb_hip_solver_padded(const lsb_hip_solver *s);
unsigned lsb_hip_solver_nrows_global(const lsb_hip_solver *s);
unsigned long long lsb_hip_solver_nnz_local(const lsb_hip_solver *s);
unsigned lsb_hip_solver_nblocks(const lsb_hip_solver *s);
int lsb_hip_solver_spmv_variant(const lsb_hip_solver *s);
/* What the timing pass at creation picked for shard 0: flags (bit 0 prefetch,
 * bit 1 nontemporal) and the workgroup cap of the SpMV launch. */
unsigned lsb_hip_solver_spmv_flags(const lsb_hip_solver *s);
unsigned lsb_hip_solver_spmv_grid(const lsb_hip_solver *s);
/* sliced-ELL: slices per plane the XCD dealing follows (every XCD the same eighth
 * of every plane of a 3-D stencil), 0 = contiguous eighths of the rows. */
unsigned lsb_hip_solver_spmv_period(const lsb_hip_solver *s);
/* slices the first shard's z-column plan walks in columns (k_spmv_tmpl_col, LSB_SP_COL of the
 * flags: a 3-D stencil whose planes are whole slices); 0 = no plan, the flag changes nothing */
unsigned long long lsb_hip_solver_spmv_col_slices(const lsb_hip_solver *s);
/* 16-bit sliced-ELL form in use: slots that keep their 128 values / all slots of the
 * first shard (lsb_sell16_value_slots); 0 / 0 for every other form */
void lsb_hip_solver_sell_value_slots(const lsb_hip_solver *s, unsigned *kept, unsigned *total);
/* 1 when the halo exchange of this solver runs behind its interior rows. */
int lsb_hip_solver_overlaps(const lsb_hip_solver *s);
/* 0 one shard; 1 RCCL (device copies between virtual shards); 2 direct xGMI
 * stores for the all-reduces; 3 for the halo exchange as well.  p2p_us/rccl_us
 * (may be NULL): what one exchange + all-reduce cost each way in the
 * creation-time self-test, 0 if it did not run. */
int lsb_hip_solver_comm(const lsb_hip_solver *s, double *p2p_us, double *rccl_us);
/* The exchange plan of this process's first shard, as a scaling line has to show it:
 * plan[0] ranks RCCL counts in the communicator (ncclCommCount; 0 = no communicator),
 * plan[1] peers it receives halos from, plan[2] peers it sends to, plan[3] / plan[4] bytes
 * received / sent per exchange (= per SpMV), plan[5] 1 when the plan is the in-place
 * all-gather (every shard needs every row), 0 for point-to-point halos, plan[6] shards
 * in this process, plan[7] 1 when the halo travels behind the interior rows. */
void lsb_hip_solver_comm_plan(const lsb_hip_solver *s, unsigned long long plan[8]);
/* opts.overlap = -1: the creation-time timing of the sharded SpMV's two forms on the real
 * communicator (hip_dist.c overlap_setup): us[0] per iteration with the SpMV behind the exchange,
 * us[1] with the interior rows in front of the halo (slowest rank's, 0 where the pass did not run).
 * Returns the form in use: 1 split, 0 plain. */
int lsb_hip_solver_overlap(const lsb_hip_solver *s, double us[2]);
/* Matrix-side bytes ONE launch of the SpMV form in use must stream (its index, code, slot
 * and value arrays as stored) + x read once + y written once, first shard; 0 for the
 * multi-pass forms (binned, two-phase).  SURVEY 8(d)'s CSR count is 12 nnz + 20 n + 4. */
unsigned long long lsb_hip_solver_spmv_layout_bytes(const lsb_hip_solver *s);
/* Bytes one iteration of the Krylov loop must move on this rank's first shard: the SpMV's layout
 * bytes + 8 B per row and vector pass of the sweeps (classic PCG: 9 passes, + 2 where the inverse
 * diagonal is a vector; single-reduction PCG: 9 or 11 + 1).  0 where the iteration has another
 * shape (GMRES, Chebyshev / block-Jacobi / FSAI, the fused forms of small operators, fp32 values,
 * multi-pass SpMV forms).  bench.py divides it by the measured time per iteration. */
unsigned long long lsb_hip_solver_iteration_bytes(const lsb_hip_solver *s);
/* 1: the PCG iteration runs in its single-reduction form (Chronopoulos & Gear; two launches, one
 * reduction per iteration) -- every sharded solve under krylov = auto, and one-shard solves of
 * large operators whose Jacobi diagonal is one constant; 0: the classic form (or GMRES). */
int lsb_hip_solver_single_reduction(const lsb_hip_solver *s);
/* 0: an iteration's direction update p = D^-1 r + beta p is a launch of its own; 1: it rides in the
 * NEXT iteration's SpMV launch, formed for every gathered operand (the sub-wavefront form of
 * launch-bound operators). */
int lsb_hip_solver_fused_p(const lsb_hip_solver *s);
/* Which operands of the BLAS-1 sweeps this solver loads nontemporal (bit 0 x, 1 p and q, 2 r in
 * k_pcg_update_xr; 3 r, 4 p in k_pcg_update_p; 5 k_cg1_update): timed at creation on the solver's
 * first shard, or LSBENCH_HIP_BLAS1_NT. */
int lsb_hip_solver_blas1_nt(const lsb_hip_solver *s);
/* hipStream_t of the backend (as void*), for callers that time with events. */
void *lsb_hip_stream(void);

/* ------------------------------------------------------------------------ */
/* Multi-GPU: one process per GPU, RCCL over xGMI                            */
/* ------------------------------------------------------------------------ */

#define LSB_HIP_UNIQUE_ID_BYTES 128
/* Rank 0 calls get_unique_id and ships the 128 bytes to the other ranks by
 * whatever side channel the launcher offers (bench.py: torch.distributed
 * broadcast); then every rank calls init_rank. */
int lsb_hip_comm_get_unique_id(void *id128);
int lsb_hip_comm_init_rank(const void *id128, int nranks, int rank);
int lsb_hip_comm_destroy(void);
int lsb_hip_comm_rank(void);
int lsb_hip_comm_size(void);
/* What RCCL itself says (ncclCommCount on this thread's communicator); 0 without one. */
int lsb_hip_comm_count(void);
/* In-place sum of `count` doubles over all ranks, device buffer, on the
 * backend stream; and a barrier built from it. */
int lsb_hip_comm_allreduce_sum_dev(double *d_buf, int count);
int lsb_hip_comm_barrier(void);

/* ------------------------------------------------------------------------ */
/* 3. Kernel-level entry points (device pointers; stream = hipStream_t)      */
/* ------------------------------------------------------------------------ */

/* y = A x for a 0-based int32 CSR.  variant = LSB_SPMV_*; d_rowblk/nblk from
 * lsb_csr_row_blocks (needed by ADAPTIVE, ignored otherwise); d_blklanes from
 * lsb_csr_block_lanes or NULL (lanes then follow the row count alone);
 * flags: bit 0 = prefetch the next block, bit 1 = nontemporal stream loads.
 * If d_dot != NULL the kernel also leaves sum_i xdot[i]*y[i] in d_dot[0]
 * (two-stage, fixed order; d_work must then hold
 * lsb_hip_partials_capacity() doubles). */
/* LSB_SPMV_SELL: pass the device copies of lsb_sell's sptr as d_offs, cols as
 * d_cols, vals as d_vals and nslice as nblk (d_rowblk, d_blklanes unused).
 * With flags & LSB_SPMV_FLAG_C16 (the 16-bit form, row_begin = 0): codes as
 * d_cols and sbase as d_rowblk. */
#define LSB_SPMV_FLAG_PREFETCH 1u
#define LSB_SPMV_FLAG_NT 2u
#define LSB_SPMV_FLAG_C16 4u
int lsb_hip_spmv_csr_f64(int variant, unsigned n, const int *d_offs,
                         const int *d_cols, const double *d_vals,
                         const int *d_rowblk, const unsigned char *d_blklanes,
                         unsigned nblk, unsigned mean_row_len, unsigned flags,
                         const double *d_x, double *d_y,
                         const double *d_xdot, double *d_dot, double *d_work,
                         void *stream);
unsigned lsb_hip_partials_capacity(void);
/* d_out[0] = sum a_i b_i (deterministic two-stage reduction). */
int lsb_hip_dot_f64(unsigned n, const double *d_a, const double *d_b,
                    double *d_out, double *d_work, void *stream);
/* d_out[0] = sqrt(sum a_i^2). */
int lsb_hip_nrm2_f64(unsigned n, const double *d_a, double *d_out,
                     double *d_work, void *stream);
/* y += alpha x, alpha read from device memory (no host sync). */
int lsb_hip_axpy_f64(unsigned n, const double *d_alpha, const double *d_x,
                     double *d_y, void *stream);
/* y = x + beta y, beta read from device memory. */
int lsb_hip_xpay_f64(unsigned n, const double *d_beta, const double *d_x,
                     double *d_y, void *stream);
/* dinv[i] = 1 / A(i, i + row_begin); *d_nzero counts rows without a usable
 * diagonal (their dinv is set to 0). */
int lsb_hip_jacobi_setup_f64(unsigned n, unsigned row_begin, const int *d_offs,
                             const int *d_cols, const double *d_vals,
                             double *d_dinv, int *d_nzero, void *stream);
/* z = dinv .* r */
int lsb_hip_jacobi_apply_f64(unsigned n, const double *d_dinv,
                             const double *d_r, double *d_z, void *stream);

/* Device memory helpers so that a C caller without HIP headers can drive the
 * kernel-level API (tests use torch tensors instead). */
void *lsb_hip_malloc(size_t bytes);
void lsb_hip_free(void *d_ptr);
int lsb_hip_memcpy_h2d(void *d_dst, const void *src, size_t bytes);
int lsb_hip_memcpy_d2h(void *dst, const void *d_src, size_t bytes);
int lsb_hip_sync(void);

#ifdef __cplusplus
}
#endif

#endif /* LSBENCH_HIP_H */
