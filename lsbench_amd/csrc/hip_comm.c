/*
 * RCCL side of the HIP backend: one process per GPU, one communicator per
 * process.  The reference issues no collective
 * of its own, so nothing here replaces reference code;
 * the call pattern is designed for MI355X's point-to-point xGMI mesh:
 *   - vector exchange before an SpMV: grouped ncclSend/ncclRecv of exactly the
 *     contiguous row ranges each peer's shard references (for a banded
 *     operator: one halo per neighbour, each crossing its own xGMI link), or
 *     one in-place ncclAllGather when every shard needs everything;
 *   - dot products: ncclAllReduce of 1-2 doubles, in place on device scalars.
 * The unique id is created by rank 0 and shipped by the launcher (bench.py
 * broadcasts it with torch.distributed); this library never opens a socket of
 * its own for that.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <string.h>

#include "lsb_impl.h"

#define LSB_CHK_NCCL(call)                                                     \
  do {                                                                         \
    ncclResult_t r_ = (call);                                                  \
    if (r_ != ncclSuccess)                                                     \
      errx(EXIT_FAILURE, "%s:%d rccl error: %s", __FILE__, __LINE__,           \
           ncclGetErrorString(r_));                                            \
  } while (0)

/* per host thread: a rank is a thread (the only one of a one-process-per-GPU
 * job, or one of the workers hip_multi.c starts inside one caller process) */
static __thread ncclComm_t g_comm;
static __thread int g_have_comm = 0, g_nranks = 1, g_rank = 0;

int lsb_hip_comm_get_unique_id(void *id128) {
  ncclUniqueId id;
  if (sizeof id != LSB_HIP_UNIQUE_ID_BYTES)
    errx(EXIT_FAILURE, "ncclUniqueId is %zu bytes, header says %d", sizeof id,
         LSB_HIP_UNIQUE_ID_BYTES);
  LSB_CHK_NCCL(ncclGetUniqueId(&id));
  memcpy(id128, &id, sizeof id);
  return 0;
}

int lsb_hip_comm_init_rank(const void *id128, int nranks, int rank) {
  if (!lsb_hip_is_initialized())
    return 1;
  if (g_have_comm)
    return 1;
  if (nranks < 1 || rank < 0 || rank >= nranks)
    return 2;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  LSB_CHK_NCCL(ncclCommInitRank(&g_comm, nranks, id, rank));
  g_have_comm = 1, g_nranks = nranks, g_rank = rank;
  return 0;
}

int lsb_hip_comm_destroy(void) {
  if (!g_have_comm)
    return 1;
  LSB_CHK_NCCL(ncclCommDestroy(g_comm));
  g_have_comm = 0, g_nranks = 1, g_rank = 0;
  return 0;
}

/* ranks RCCL itself counts in this thread's communicator (ncclCommCount), 0 without one --
 * what a scaling line quotes as proof that the collectives ran over N ranks */
int lsb_hip_comm_count(void) {
  int n = 0;
  if (!g_have_comm)
    return 0;
  LSB_CHK_NCCL(ncclCommCount(g_comm, &n));
  return n;
}

int lsb_hip_comm_rank(void) { return g_rank; }
int lsb_hip_comm_size(void) { return g_nranks; }

int lsb_hip_comm_allreduce_stream(double *d_buf, int count, void *stream) {
  if (!g_have_comm || g_nranks == 1)
    return 0;
  LSB_CHK_NCCL(ncclAllReduce(d_buf, d_buf, (size_t)count, ncclDouble, ncclSum,
                             g_comm, (hipStream_t)stream));
  return 0;
}

int lsb_hip_comm_allreduce_sum_dev(double *d_buf, int count) {
  return lsb_hip_comm_allreduce_stream(d_buf, count, lsb_hip_stream());
}

int lsb_hip_comm_barrier(void) {
  if (!g_have_comm || g_nranks == 1)
    return lsb_hip_sync();
  double *d;
  LSB_CHK_HIP(hipMalloc((void **)&d, sizeof(double)));
  LSB_CHK_HIP(hipMemsetAsync(d, 0, sizeof(double), (hipStream_t)lsb_hip_stream()));
  lsb_hip_comm_allreduce_stream(d, 1, lsb_hip_stream());
  LSB_CHK_HIP(hipStreamSynchronize((hipStream_t)lsb_hip_stream()));
  LSB_CHK_HIP(hipFree(d));
  return 0;
}

/* all[r*count .. r*count+count) = rank r's `mine` (host in, host out) */
int lsb_hip_comm_allgather_u32(const unsigned *mine, unsigned count,
                               unsigned *all) {
  if (!g_have_comm || g_nranks == 1) {
    memcpy(all, mine, (size_t)count * sizeof(unsigned));
    return 0;
  }
  hipStream_t s = (hipStream_t)lsb_hip_stream();
  unsigned *d;
  const size_t bytes = (size_t)count * sizeof(unsigned);
  LSB_CHK_HIP(hipMalloc((void **)&d, bytes * g_nranks));
  LSB_CHK_HIP(hipMemcpyAsync(d + (size_t)g_rank * count, mine, bytes,
                             hipMemcpyHostToDevice, s));
  LSB_CHK_NCCL(ncclAllGather(d + (size_t)g_rank * count, d, count, ncclUint32,
                             g_comm, s));
  LSB_CHK_HIP(hipMemcpyAsync(all, d, bytes * g_nranks, hipMemcpyDeviceToHost, s));
  LSB_CHK_HIP(hipStreamSynchronize(s));
  LSB_CHK_HIP(hipFree(d));
  return 0;
}

/*
 * One exchange step.  Every range is [offset, offset+count) of the
 * full-length vector, which each rank stores at the same global offsets, so a
 * send reads and the matching receive writes the same index range.
 * peer == -1 in sends[0] means "in-place all-gather of `count` doubles per
 * rank" (every shard needs every row and shards are equal-sized).
 */
int lsb_hip_comm_exchange(double *d_full, const struct lsb_xfer *sends,
                          int nsend, const struct lsb_xfer *recvs, int nrecv,
                          void *stream) {
  if (!g_have_comm || g_nranks == 1)
    return 0;
  hipStream_t s = (hipStream_t)stream;
  if (nsend == 1 && sends[0].peer == -1) {
    LSB_CHK_NCCL(ncclAllGather(d_full + sends[0].offset, d_full, sends[0].count,
                               ncclDouble, g_comm, s));
    return 0;
  }
  LSB_CHK_NCCL(ncclGroupStart());
  for (int i = 0; i < nrecv; i++)
    LSB_CHK_NCCL(ncclRecv(d_full + recvs[i].offset, recvs[i].count, ncclDouble,
                          recvs[i].peer, g_comm, s));
  for (int i = 0; i < nsend; i++)
    LSB_CHK_NCCL(ncclSend(d_full + sends[i].offset, sends[i].count, ncclDouble,
                          sends[i].peer, g_comm, s));
  LSB_CHK_NCCL(ncclGroupEnd());
  return 0;
}
