/*
 * TEST DOUBLE for RCCL -- test infrastructure only, never shipped or linked into
 * the product.  The GPU box has ONE MI355X, and RCCL refuses two ranks on one
 * device, so the multi-process path of the library (hip_comm.c: communicator
 * set-up, grouped ncclSend/ncclRecv halo exchange, scalar all-reduces, the
 * identical control flow every rank must follow) could never run before the
 * driver's 8-GPU scaling bench.  LD_PRELOADing this file in front of librccl
 * gives those calls a slow but faithful implementation between PROCESSES that
 * share a GPU: device buffers are staged through a POSIX shared-memory segment
 * with sense-reversing barriers.  Every wait is bounded (FAKE_TIMEOUT_S) and
 * aborts instead of hanging the box.
 *
 * Implements exactly what hip_comm.c calls: ncclGetUniqueId, ncclCommInitRank,
 * ncclCommDestroy, ncclAllReduce (sum), ncclAllGather, ncclGroupStart/End,
 * ncclSend, ncclRecv, ncclGetErrorString.
 */
#define _GNU_SOURCE
#define __HIP_PLATFORM_AMD__ 1
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define FAKE_TIMEOUT_S 60.0
#define MAXR 8
#define BOX_BYTES (4u << 20) /* one mailbox per ordered pair of ranks */
#define SLOT_BYTES (1u << 20) /* per-rank slot for reductions / gathers */

struct shm_hdr {
  volatile int arrived, sense, ready;
  int nranks;
};

struct fake_comm {
  int rank, nranks;
  char name[64];
  size_t bytes;
  struct shm_hdr *h;
  unsigned char *slots, *boxes;
  int local_sense;
};

static void die(const char *what);
struct pending {
  int is_send, peer;
  void *buf;
  size_t bytes;
  hipStream_t stream;
};
/* per thread: a rank may be a host thread of one process (hip_multi.c) */
static __thread struct pending g_ops[4 * MAXR];
static __thread int g_nops = 0, g_group = 0;
static __thread struct fake_comm *g_group_comm = NULL;

/* device <-> shared memory through the caller's stream (no null-stream copies:
 * those would wait for every other rank-thread's stream of the same device) */
static void d2h(void *dst, const void *src, size_t bytes, hipStream_t s) {
  if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess)
    die("D2H");
}
static void h2d(void *dst, const void *src, size_t bytes, hipStream_t s) {
  if (hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess)
    die("H2D");
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static void die(const char *what) {
  fprintf(stderr, "fake_rccl: %s\n", what);
  abort();
}

static void barrier(struct fake_comm *c) {
  struct shm_hdr *h = c->h;
  const int my = c->local_sense ^= 1;
  if (__atomic_add_fetch(&h->arrived, 1, __ATOMIC_ACQ_REL) == c->nranks) {
    __atomic_store_n(&h->arrived, 0, __ATOMIC_RELEASE);
    __atomic_store_n(&h->sense, my, __ATOMIC_RELEASE);
  } else {
    const double t0 = now_s();
    while (__atomic_load_n(&h->sense, __ATOMIC_ACQUIRE) != my) {
      sched_yield();
      if (now_s() - t0 > FAKE_TIMEOUT_S)
        die("barrier timed out: the ranks disagree on the sequence of collectives");
    }
  }
}

static size_t type_bytes(ncclDataType_t t) {
  switch (t) {
  case ncclDouble:
    return 8;
  case ncclUint32:
  case ncclInt32:
  case ncclFloat:
    return 4;
  default:
    die("unsupported data type");
  }
  return 0;
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "fake_rccl error"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  memset(id, 0, sizeof *id);
  static int serial = 0;
  snprintf(id->internal, sizeof id->internal, "/lsbfake_%d_%ld_%d", (int)getpid(), (long)time(NULL),
           __atomic_add_fetch(&serial, 1, __ATOMIC_RELAXED));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > MAXR)
    die("nranks out of range");
  struct fake_comm *c = (struct fake_comm *)calloc(1, sizeof *c);
  c->rank = rank, c->nranks = nranks;
  snprintf(c->name, sizeof c->name, "%.60s", id.internal);
  c->bytes = 4096 + (size_t)nranks * SLOT_BYTES + (size_t)nranks * nranks * BOX_BYTES;
  int fd = -1;
  const double t0 = now_s();
  if (rank == 0) {
    fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0)
      die("shm_open/ftruncate failed");
  } else {
    while ((fd = shm_open(c->name, O_RDWR, 0600)) < 0) {
      sched_yield();
      if (now_s() - t0 > FAKE_TIMEOUT_S)
        die("rank 0 never created the segment");
    }
    struct stat sb; /* mapping it before rank 0 has sized it would SIGBUS on first touch */
    for (;;) {
      if (fstat(fd, &sb) == 0 && (size_t)sb.st_size >= c->bytes)
        break;
      sched_yield();
      if (now_s() - t0 > FAKE_TIMEOUT_S)
        die("rank 0 never sized the segment");
    }
  }
  void *m = MAP_FAILED;
  while (m == MAP_FAILED) { /* rank 0 may not have sized it yet */
    m = mmap(NULL, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED && now_s() - t0 > FAKE_TIMEOUT_S)
      die("mmap failed");
  }
  close(fd);
  c->h = (struct shm_hdr *)m;
  c->slots = (unsigned char *)m + 4096;
  c->boxes = c->slots + (size_t)nranks * SLOT_BYTES;
  if (rank == 0) {
    c->h->nranks = nranks;
    __atomic_store_n(&c->h->ready, 1, __ATOMIC_RELEASE);
  } else {
    while (!__atomic_load_n(&c->h->ready, __ATOMIC_ACQUIRE)) {
      sched_yield();
      if (now_s() - t0 > FAKE_TIMEOUT_S)
        die("segment never became ready");
    }
  }
  barrier(c);
  *out = (ncclComm_t)c;
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) {
  *count = ((const struct fake_comm *)comm)->nranks;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  struct fake_comm *c = (struct fake_comm *)comm;
  barrier(c);
  munmap((void *)c->h, c->bytes);
  if (c->rank == 0)
    shm_unlink(c->name);
  free(c);
  return ncclSuccess;
}

/* FAKE_RCCL_STALL_RANK=r FAKE_RCCL_STALL_AFTER=k: rank r's all-reduces beyond the
 * k-th behave like a collective whose peers never arrive -- the call returns at
 * once (RCCL calls are asynchronous) and the STREAM stops making progress; so do
 * all of that rank's later collectives (a real hung stream never gets to them).
 * The product's host-side deadline has to notice -- and has to leave the process
 * although this stream will not drain for another 45 s (a real hung collective
 * never drains): the host function keeps sleeping whatever the process does, so
 * an exit() that runs the HIP runtime's teardown would sit here; the product's
 * give-up path is _exit() (lsb_give_up), and the test checks the elapsed time. */
static void stall_fn(void *arg) {
  (void)arg;
  for (int i = 0; i < 450; i++)
    usleep(100000);
}
static __thread int g_allreduces = 0, g_stalled = 0;

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t,
                           ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
  struct fake_comm *c = (struct fake_comm *)comm;
  const char *sr = getenv("FAKE_RCCL_STALL_RANK"), *sa = getenv("FAKE_RCCL_STALL_AFTER");
  if (sr && atoi(sr) == c->rank && ++g_allreduces > (sa ? atoi(sa) : 0)) {
    if (!g_stalled && hipLaunchHostFunc(stream, stall_fn, NULL) != hipSuccess)
      die("hipLaunchHostFunc");
    g_stalled = 1;
    return ncclSuccess;
  }
  if (t != ncclDouble || op != ncclSum || count * 8 > SLOT_BYTES)
    die("all-reduce: only sums of a few doubles");
  if (hipStreamSynchronize(stream) != hipSuccess)
    die("stream sync");
  d2h(c->slots + (size_t)c->rank * SLOT_BYTES, send, count * 8, stream);
  barrier(c);
  double acc[64];
  if (count > 64)
    die("all-reduce too long for the test double");
  for (size_t k = 0; k < count; k++) {
    acc[k] = 0.0;
    for (int q = 0; q < c->nranks; q++) /* rank order: identical bits everywhere */
      acc[k] += ((double *)(c->slots + (size_t)q * SLOT_BYTES))[k];
  }
  barrier(c);
  h2d(recv, acc, count * 8, stream);
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t,
                           ncclComm_t comm, hipStream_t stream) {
  struct fake_comm *c = (struct fake_comm *)comm;
  const size_t b = count * type_bytes(t);
  if (g_stalled)
    return ncclSuccess;
  if (b > SLOT_BYTES)
    die("all-gather larger than the test double's slots");
  if (hipStreamSynchronize(stream) != hipSuccess)
    die("stream sync");
  d2h(c->slots + (size_t)c->rank * SLOT_BYTES, send, b, stream);
  barrier(c);
  for (int q = 0; q < c->nranks; q++)
    h2d((unsigned char *)recv + (size_t)q * b, c->slots + (size_t)q * SLOT_BYTES, b, stream);
  barrier(c);
  return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) {
  g_group++;
  return ncclSuccess;
}

static void push(struct fake_comm *c, int is_send, void *buf, size_t bytes, int peer, hipStream_t s) {
  if (!g_group)
    die("send/recv outside a group is not supported by the test double");
  if (bytes > BOX_BYTES)
    die("message larger than a mailbox");
  if (g_nops >= (int)(sizeof g_ops / sizeof g_ops[0]))
    die("too many grouped operations");
  g_group_comm = c;
  g_ops[g_nops].is_send = is_send, g_ops[g_nops].peer = peer, g_ops[g_nops].buf = buf;
  g_ops[g_nops].bytes = bytes, g_ops[g_nops].stream = s;
  g_nops++;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm,
                      hipStream_t s) {
  push((struct fake_comm *)comm, 1, (void *)buf, count * type_bytes(t), peer, s);
  return ncclSuccess;
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm,
                      hipStream_t s) {
  push((struct fake_comm *)comm, 0, buf, count * type_bytes(t), peer, s);
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd(void) {
  if (--g_group > 0)
    return ncclSuccess;
  struct fake_comm *c = g_group_comm;
  if (g_stalled) { /* behind a hung collective on this rank's stream: never runs */
    g_nops = 0, g_group_comm = NULL;
    return ncclSuccess;
  }
  if (!c) { /* an empty group still has to keep the ranks in step?  No: ranks with
               nothing to exchange make no call at all in hip_comm.c */
    g_nops = 0;
    return ncclSuccess;
  }
  /* NOTE: a rank with no sends/recvs never gets here with c != NULL, so the
   * barriers below require every rank to take part in every exchange -- true
   * for a row-partitioned banded operator with >= 2 ranks, which is what the
   * tests run. */
  if (g_nops && hipStreamSynchronize(g_ops[0].stream) != hipSuccess)
    die("stream sync");
  for (int i = 0; i < g_nops; i++)
    if (g_ops[i].is_send) {
      unsigned char *box = c->boxes + ((size_t)c->rank * c->nranks + g_ops[i].peer) * BOX_BYTES;
      memcpy(box, &g_ops[i].bytes, sizeof(size_t));
      d2h(box + 64, g_ops[i].buf, g_ops[i].bytes, g_ops[i].stream);
    }
  barrier(c);
  for (int i = 0; i < g_nops; i++)
    if (!g_ops[i].is_send) {
      unsigned char *box = c->boxes + ((size_t)g_ops[i].peer * c->nranks + c->rank) * BOX_BYTES;
      size_t got;
      memcpy(&got, box, sizeof got);
      if (got != g_ops[i].bytes)
        die("send/recv size mismatch between two ranks");
      h2d(g_ops[i].buf, box + 64, g_ops[i].bytes, g_ops[i].stream);
    }
  barrier(c);
  g_nops = 0, g_group_comm = NULL;
  return ncclSuccess;
}
