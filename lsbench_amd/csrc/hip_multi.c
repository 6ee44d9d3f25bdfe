/*
 * hip_cdna4_bench over several GPUs of one node, from the ONE caller process
 * the reference's contract has (src/lsbench-impl.h:42-68: one call, one process,
 * one caller thread; bin/driver.c:5-15).  SURVEY.md section 8(b) "Threading":
 * the backend may start its own per-GPU host threads, invisible to the caller.
 *
 * That is what happens here.  The operator CHOLMOD would factorise,
 * S = triu(A) + triu(A,1)^T (src/cholmod-impl.h:5-21), is built once on the
 * host, cut into `ngpus` contiguous row ranges of equal non-zero count
 * (lsb_csr_partition_rows) and every range is handed to a RANK: a host thread
 * bound to one device (rank 0 is the caller's thread on the device
 * hip_cdna4_init picked).  A rank is exactly what one process of the
 * one-process-per-GPU mode is (bench.py under torch.distributed.run): its
 * stream, its RCCL communicator (ncclCommInitRank, all ranks of the process at
 * once) and its shard, created by lsb_hip_solver_create_dist -- so the solve
 * code, the exchange plan, the RCCL calls and the direct xGMI path are the same
 * code in both modes; only where the mailboxes' addresses come from differs
 * (peer-enabled pointers inside a process, HIP IPC handles between processes,
 * hip_p2p.hip).  Threads rather than one thread driving N devices: an iteration
 * at N = 8 is a few tens of microseconds of kernels per GPU, and one host thread
 * cannot enqueue 8 x 4 launches in that time.
 *
 * Protocol per rank = the reference's (src/cusparse.c:174-209): untimed setup,
 * cb->trials warm-up solves, device sync + barrier, timer (rank 0), cb->trials
 * solves, device sync + barrier, timer, D2H of the rank's rows of x straight
 * into the caller's x (disjoint ranges), teardown.
 */
#define _GNU_SOURCE
#include "hip_solver.h"
#include <pthread.h>

#define MG_MAX 64

struct mg_shared {
  int nranks, dev[MG_MAX];
  const struct csr *S; /* 0-based, both triangles */
  unsigned bounds[MG_MAX + 1];
  const double *r;
  double *x;
  unsigned trials;
  struct lsb_hip_opts o;
  unsigned char id[LSB_HIP_UNIQUE_ID_BYTES];
  pthread_barrier_t bar;
  double t0, t1;
  struct lsb_hip_result res0;
  int comm_mode;
  unsigned long long plan[8]; /* rank 0's exchange plan (lsb_hip_solver_comm_plan) */
};

struct mg_arg {
  struct mg_shared *sh;
  int rank;
};

static void *mg_rank(void *argp) {
  struct mg_arg *a = (struct mg_arg *)argp;
  struct mg_shared *g = a->sh;
  const int rank = a->rank;
  if (rank > 0)
    rank_thread_attach(g->dev[rank]);
  if (lsb_hip_comm_init_rank(g->id, g->nranks, rank) != 0)
    errx(EXIT_FAILURE, "hip_cdna4: rank %d could not join the communicator", rank);

  const unsigned r0 = g->bounds[rank], r1 = g->bounds[rank + 1], nl = r1 - r0;
  struct csr *rows = lsb_csr_row_slice(g->S, r0, r1);
  lsb_hip_solver *sv = lsb_hip_solver_create_dist(rows, r0, g->S->nrows, &g->o);
  lsb_csr_free(rows);
  if (!sv)
    errx(EXIT_FAILURE, "hip_cdna4: rank %d cannot set up its shard", rank);
  const size_t bytes = (size_t)nl * sizeof(double);
  double *d_r = (double *)lsb_hip_malloc(bytes), *d_x = (double *)lsb_hip_malloc(bytes);
  LSB_CHK_HIP(hipMemcpy(d_r, g->r + r0, bytes, hipMemcpyHostToDevice));
  LSB_CHK_HIP(hipMemset(d_x, 0, bytes));

  struct lsb_hip_result res;
  memset(&res, 0, sizeof res);
  for (unsigned i = 0; i < g->trials; i++) /* warm-up, src/cusparse.c:182-186 */
    lsb_hip_solver_solve_dev(sv, d_r, d_x, &res);
  LSB_CHK_HIP(hipDeviceSynchronize());
  pthread_barrier_wait(&g->bar);
  if (rank == 0)
    g->t0 = wall_seconds();
  for (unsigned i = 0; i < g->trials; i++) /* timed, src/cusparse.c:189-197 */
    lsb_hip_solver_solve_dev(sv, d_r, d_x, &res);
  LSB_CHK_HIP(hipDeviceSynchronize());
  pthread_barrier_wait(&g->bar);
  if (rank == 0) {
    g->t1 = wall_seconds();
    g->res0 = res;
    g->comm_mode = lsb_hip_solver_comm(sv, NULL, NULL);
    lsb_hip_solver_comm_plan(sv, g->plan);
  }
  LSB_CHK_HIP(hipMemcpy(g->x + r0, d_x, bytes, hipMemcpyDeviceToHost)); /* :199 */

  lsb_hip_free(d_r), lsb_hip_free(d_x);
  lsb_hip_solver_destroy(sv);
  lsb_hip_comm_destroy();
  if (rank > 0)
    rank_thread_detach();
  return NULL;
}

int bench_multi(double *x, struct csr *A, const double *r, const struct lsbench *cb,
                const struct lsb_hip_opts *o_in, int ngpus) {
  struct mg_shared *g = lsb_calloc(struct mg_shared, 1);
  int ndev = lsb_hip_device_count(), cur = 0;
  LSB_CHK_HIP(hipGetDevice(&cur));
  /* LSBENCH_HIP_SHARE_DEVICE=1: every rank on the caller's device -- the
   * rehearsal of this path on a one-GPU box (RCCL itself refuses two ranks on
   * a device; the tests put a double in front of it) */
  const char *e = getenv("LSBENCH_HIP_SHARE_DEVICE");
  const int share = e && atoi(e) > 0;
  if (ngpus > MG_MAX)
    errx(EXIT_FAILURE, "hip_cdna4: --ngpus %d: at most %d", ngpus, MG_MAX);
  if (!share && ngpus > ndev)
    errx(EXIT_FAILURE, "hip_cdna4: --ngpus %d, but this node shows %d device%s", ngpus, ndev,
         ndev == 1 ? "" : "s");
  g->nranks = ngpus;
  g->dev[0] = cur; /* rank 0 = the caller's thread, on the device it is on */
  for (int q = 1, d = 0; q < ngpus; q++) {
    if (share) {
      g->dev[q] = cur;
      continue;
    }
    if (d == cur)
      d++;
    g->dev[q] = d++;
  }
  g->o = *o_in;
  g->o.nvirt = 1, g->o.ngpus = 1;
  if (g->o.reorder) {
    warnx("hip_cdna4: --reorder applies to one shard; ignored with --ngpus %d", ngpus);
    g->o.reorder = 0;
  }
  /* the operator, once, on the host; the ranks slice it */
  struct csr *S = o_in->op_mode == LSB_OP_CHOLMOD_UPPER ? lsb_csr_symmetrize_upper(A)
                                                        : lsb_csr_copy_base0(A);
  g->o.op_mode = LSB_OP_RAW;
  if ((unsigned)ngpus > S->nrows / 2)
    errx(EXIT_FAILURE, "hip_cdna4: %u rows are too few for %d GPUs", S->nrows, ngpus);
  g->S = S;
  lsb_csr_partition_rows(S, (unsigned)ngpus, g->bounds);
  g->r = r, g->x = x, g->trials = cb->trials;
  lsb_hip_comm_get_unique_id(g->id);
  pthread_barrier_init(&g->bar, NULL, (unsigned)ngpus);

  pthread_t th[MG_MAX];
  struct mg_arg args[MG_MAX];
  for (int q = 0; q < ngpus; q++)
    args[q].sh = g, args[q].rank = q;
  for (int q = 1; q < ngpus; q++)
    if (pthread_create(&th[q], NULL, mg_rank, &args[q]) != 0)
      errx(EXIT_FAILURE, "hip_cdna4: cannot start the host thread of rank %d", q);
  mg_rank(&args[0]);
  for (int q = 1; q < ngpus; q++)
    pthread_join(th[q], NULL);
  pthread_barrier_destroy(&g->bar);

  const unsigned m = A->nrows, nnz = A->offs[m];
  const double elapsed = g->t1 - g->t0;
  g_last = g->res0;
  /* the reference's record (src/cholmod-impl.h:68-70), then this backend's */
  printf("===matrix,n,nnz,trials,solver,ordering,elapsed===\n");
  printf("%s,%u,%u,%u,%u,%d,%.15lf\n", cb->matrix, m, nnz, cb->trials, cb->solver, cb->ordering,
         elapsed);
  printf("===hip_cdna4:iterations,relres,status,tol,solves_per_sec,nshards===\n");
  printf("%u,%.6e,%d,%.3e,%.6f,%d\n", g->res0.iters, g->res0.relres, g->res0.status, o_in->tol,
         elapsed > 0 ? cb->trials / elapsed : 0.0, ngpus);
  /* third record: what ran between the GPUs -- the ranks RCCL itself counts
   * (ncclCommCount) and rank 0's exchange plan, so that a line can be checked for
   * "N ranks, neighbour halos (not an all-gather)" */
  printf("===hip_cdna4:ngpus,comm,rccl_ranks,pattern,recv_peers,send_peers,bytes_recv_per_exchange,"
         "bytes_sent_per_exchange,overlap===\n%d,%s,%llu,%s,%llu,%llu,%llu,%llu,%llu\n", ngpus,
         g->comm_mode == 3   ? "direct-xgmi(halos+allreduce)"
         : g->comm_mode == 2 ? "direct-xgmi(allreduce)+rccl(halos)"
                             : "rccl",
         g->plan[0], g->plan[5] ? "all-gather" : "halos", g->plan[1], g->plan[2], g->plan[3], g->plan[4],
         g->plan[7]);
  fflush(stdout);
  lsb_csr_free(S);
  free(g);
  return 0;
}
