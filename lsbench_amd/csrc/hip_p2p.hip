// Direct xGMI path of the HIP backend: the two
// communication steps of a sharded Krylov iteration -- the halo exchange in
// front of the SpMV and the all-reduce of 1-3 dot products -- done by plain
// stores into the peers' memory instead of RCCL calls.
//
// Why: with the operator split over 8 GPUs an iteration is ~30 us of kernels,
// and an RCCL collective of 8-24 bytes costs about as much again (launch,
// ring/tree protocol, two hops).  MI355X's xGMI is a full point-to-point mesh
// and every GPU can store into every other GPU's HBM, so a few bytes travel
// in ONE hop: each rank owns a MAILBOX (fine-grained device memory, opened by
// all peers through HIP IPC handles) and
//   send       copies the rows a peer's shard references into that peer's
//              mailbox, then sets the peer's flag for this rank to the epoch;
//   recv       waits for the flags of the ranks it receives from, copies the
//              halos out of the mailbox into its full-length vector;
//   allreduce  reduces this rank's per-workgroup partial sums, stores the 1-3
//              results into slot [epoch&1][rank] of EVERY mailbox, waits for
//              all R slots of its own mailbox and adds them in rank order (the
//              same order on every rank: all ranks get identical bits, which
//              the convergence test needs).
// Flags are monotonically increasing epochs, so nothing is ever reset; slot
// reuse is safe because an all-reduce separates two successive exchanges and
// all-reduce k+2 cannot start anywhere before every rank has left k.  Inside a
// Krylov iteration that all-reduce is the algorithm's own; an exchange issued
// outside one (lsb_hip_solver_spmv_dev, the Jacobi sweep) is closed by a
// one-value all-reduce for exactly this purpose (exchange_p, hip_dist.c) -- a
// fast rank could otherwise overwrite a halo region its peer is still copying.
// Every wait is bounded (wall_clock64): a peer that never arrives turns into
// st->status = LSB_STATUS_COMM -- later kernels no-op, the host reports it --
// never into a hung wave.
//
// Memory model: payload stores, system-scope release fence, flag store
// (system scope) on the producer; system-scope flag loads, acquire fence and
// system-scope payload loads on the consumer -- the HSA rules for fine-grained
// memory shared between agents.  hip_cdna4.c does not trust this blindly: at
// solver creation it runs patterns through both this path and RCCL and keeps
// this one only if it was bit-exact AND faster on every rank.
//
// The reference has no counterpart (it is single-process, SURVEY.md section
// 2.2); the RCCL path in hip_comm.c stays the default wherever this one is not
// available (no IPC, no peer access) or not faster.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "hip_ar.h"
#include "lsb_impl.h"

#define P2P_MAX_RANKS AR_MAX_RANKS
#define P2P_AR_BYTES 4096u   // 2 parities x 64 ranks x 32 B
#define P2P_FLAG_BYTES 4096u // 64 ranks x one 64-B line
#define P2P_HEADER (P2P_AR_BYTES + P2P_FLAG_BYTES)
#define P2P_NONE 0xFFFFFFFFu
#define P2P_WG AR_WG

typedef unsigned long long u64;

struct p2p_send_ent {
  double *dst;   // region in the RECEIVER's mailbox
  u64 *flag;     // the receiver's flag for this rank
  size_t src_off, count;
  unsigned wgs, first_wg;
};
struct p2p_recv_ent {
  const double *src; // region in MY mailbox
  const u64 *flag;   // my flag for the sending rank
  size_t dst_off, count;
  unsigned wgs, first_wg;
};

struct lsb_p2p {
  int R, me, virt, halo;
  u64 epoch_x, epoch_r;
  long long timeout_ticks;
  char *mbox;
  size_t mbox_bytes;
  unsigned region_off[P2P_MAX_RANKS]; // doubles from the halo base, per source
  char *peer[P2P_MAX_RANKS];
  int opened[P2P_MAX_RANKS];
  char **d_peer;
  p2p_send_ent *d_send;
  p2p_recv_ent *d_recv;
  unsigned *d_counters;
  unsigned *d_tail; // hand-in counter of an all-reduce folded into an SpMV launch (hip_ar.h)
  int nsend, nrecv;
  unsigned send_grid, recv_grid;
};

// --------------------------------------------------------------------------
// (flag waits, slots and the two all-reduce phases: hip_ar.h, shared with the
// kernels the phases can be folded into)
__device__ __forceinline__ double ld_sys(const double *p) { return ar_ld_sys(p); }
__device__ __forceinline__ bool wait_flag(const u64 *flag, u64 epoch, long long timeout) {
  return ar_wait_flag(flag, epoch, timeout);
}

__device__ __forceinline__ void p2p_send_body(const p2p_send_ent *__restrict__ ents, int nent,
                                              unsigned blk, const double *__restrict__ full,
                                              u64 epoch, unsigned *__restrict__ counters) {
  int e = 0;
  while (e + 1 < nent && blk >= ents[e + 1].first_wg)
    e++;
  const p2p_send_ent en = ents[e];
  const unsigned part = blk - en.first_wg;
  const size_t per = (en.count + en.wgs - 1) / en.wgs;
  const size_t lo = (size_t)part * per, hi = lo + per < en.count ? lo + per : en.count;
  const double *src = full + en.src_off;
  for (size_t i = lo + threadIdx.x; i < hi; i += P2P_WG)
    en.dst[i] = src[i];
  __threadfence_system(); // this thread's stores have reached the peer
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old =
        __hip_atomic_fetch_add(&counters[e], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if ((old + 1) % en.wgs == 0) // the last workgroup of this entry raises the flag
      __hip_atomic_store(en.flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__device__ __forceinline__ void p2p_recv_body(const p2p_recv_ent *__restrict__ ents, int nent,
                                              unsigned blk, double *__restrict__ full, u64 epoch,
                                              lsb_pcg_state *__restrict__ st, long long timeout) {
  __shared__ int ok;
  int e = 0;
  while (e + 1 < nent && blk >= ents[e + 1].first_wg)
    e++;
  const p2p_recv_ent en = ents[e];
  if (threadIdx.x == 0)
    ok = wait_flag(en.flag, epoch, timeout) ? 1 : 0;
  __syncthreads();
  if (!ok) {
    if (threadIdx.x == 0 && st)
      __hip_atomic_store(&st->status, (int)LSB_STATUS_COMM, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  const unsigned part = blk - en.first_wg;
  const size_t per = (en.count + en.wgs - 1) / en.wgs;
  const size_t lo = (size_t)part * per, hi = lo + per < en.count ? lo + per : en.count;
  double *dst = full + en.dst_off;
  for (size_t i = lo + threadIdx.x; i < hi; i += P2P_WG)
    dst[i] = ld_sys(en.src + i);
}

__global__ __launch_bounds__(P2P_WG) void k_p2p_send(const p2p_send_ent *__restrict__ ents,
                                                     int nent, const double *__restrict__ full,
                                                     u64 epoch, unsigned *__restrict__ counters,
                                                     const lsb_pcg_state *__restrict__ st) {
  if (st && st->status)
    return;
  p2p_send_body(ents, nent, blockIdx.x, full, epoch, counters);
}

__global__ __launch_bounds__(P2P_WG) void k_p2p_recv(const p2p_recv_ent *__restrict__ ents,
                                                     int nent, double *__restrict__ full,
                                                     u64 epoch, lsb_pcg_state *__restrict__ st,
                                                     long long timeout) {
  if (st && st->status)
    return;
  p2p_recv_body(ents, nent, blockIdx.x, full, epoch, st, timeout);
}

// One process per GPU: both roles in ONE launch (the workgroups that wait and
// the ones that send are all resident at once; what a waiter waits for comes
// from another GPU).  Sends read rows this rank owns, receives write rows it
// does not, so the two roles never touch the same element of `full`.
__global__ __launch_bounds__(P2P_WG) void k_p2p_sendrecv(
    const p2p_send_ent *__restrict__ sends, int nsend, unsigned send_grid,
    const p2p_recv_ent *__restrict__ recvs, int nrecv, double *full, u64 epoch,
    unsigned *__restrict__ counters, lsb_pcg_state *__restrict__ st, long long timeout) {
  if (st && st->status)
    return;
  if (blockIdx.x < send_grid)
    p2p_send_body(sends, nsend, blockIdx.x, full, epoch, counters);
  else
    p2p_recv_body(recvs, nrecv, blockIdx.x - send_grid, full, epoch, st, timeout);
}

__global__ __launch_bounds__(P2P_WG) void k_p2p_allreduce(
    const double *__restrict__ parts, unsigned nparts, unsigned width,
    const double *__restrict__ parts2, unsigned nparts2, unsigned width2,
    const double *extra, unsigned nextra, double *out, // extra and out may alias
    char *const *__restrict__ peer, int R, int me, u64 epoch, lsb_pcg_state *__restrict__ st,
    int phases, long long timeout) {
  if (st && st->status)
    return;
  __shared__ double sred[3 * (P2P_WG / 64)];
  __shared__ double sval[3];
  const unsigned nvals = width + width2 + nextra;
  if (phases & 1)
    ar_contribute<false, false>(parts, nparts, width, parts2, nparts2, width2, extra, nextra, peer, R, me,
                         epoch, sred, sval);
  if ((phases & 2) && threadIdx.x < 64) {
    double v[3];
    if (!ar_collect<true>(peer[me], R, epoch, timeout, nvals, v)) {
      if (threadIdx.x == 0 && st)
        __hip_atomic_store(&st->status, (int)LSB_STATUS_COMM, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (threadIdx.x == 0)
      for (unsigned k = 0; k < nvals; k++)
        out[k] = v[k];
  }
}

// ---- self-test helpers ------------------------------------------------------
__device__ __forceinline__ double pattern(u64 g, unsigned round) {
  return (double)((g * 1315423911ull + (u64)round * 2654435761ull + 12345ull) & 0xFFFFFFFFFFFFull);
}
__global__ void k_p2p_pattern(double *v, size_t n, size_t goff, unsigned round) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    v[goff + i] = pattern(goff + i, round);
}
__global__ void k_p2p_check_range(const double *full, size_t goff, size_t n, unsigned round,
                                  unsigned *bad) {
  unsigned mine = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    mine += full[goff + i] != pattern(goff + i, round);
  if (mine)
    atomicAdd(bad, mine);
}
__global__ void k_p2p_setvals(double *v, int me, unsigned round) {
  v[0] = (double)((me + 1) * (round + 1));
  v[1] = (double)((me + 1) * (round + 7)) * 0.5;
  v[2] = (double)(me + 1);
}
__global__ void k_p2p_checkvals(const double *v, int R, unsigned round, unsigned *bad) {
  const double t = 0.5 * R * (R + 1);
  if (v[0] != t * (round + 1) || v[1] != t * (round + 7) * 0.5 || v[2] != t)
    atomicAdd(bad, 1u);
}

// The collect phase the way k_cg1_update runs it (hip_ar.h: every workgroup's first
// wavefront, no acquire fence between the flag and the values), checked against the
// self-test's known sums by every one of them.
__global__ __launch_bounds__(P2P_WG) void k_p2p_collect_check(char *mbox, int R, u64 epoch,
                                                              long long timeout, unsigned round,
                                                              unsigned *bad,
                                                              lsb_pcg_state *__restrict__ st) {
  if (st && st->status)
    return;
  if (threadIdx.x >= 64)
    return;
  double v[3];
  if (!ar_collect<false>(mbox, R, epoch, timeout, 3, v)) {
    if (threadIdx.x == 0 && st)
      __hip_atomic_store(&st->status, (int)LSB_STATUS_COMM, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const double t = 0.5 * R * (R + 1);
  if (threadIdx.x == 0 &&
      (v[0] != t * (round + 1) || v[1] != t * (round + 7) * 0.5 || v[2] != t))
    atomicAdd(bad, 1u);
}

// ---- host side --------------------------------------------------------------
static unsigned wgs_for(size_t count) {
  size_t w = (count + 4095) / 4096;
  return (unsigned)(w < 1 ? 1 : (w > 16 ? 16 : w));
}

static struct lsb_p2p *p2p_alloc(int R, int me, int virt, const struct lsb_xfer *recv, int nrecv,
                                 const struct lsb_xfer *send, int nsend) {
  if (R > P2P_MAX_RANKS)
    return NULL;
  struct lsb_p2p *p = (struct lsb_p2p *)calloc(1, sizeof *p);
  p->R = R, p->me = me, p->virt = virt;
  const char *e = getenv("LSBENCH_HIP_P2P_TIMEOUT_MS");
  int dev = 0, khz = 100000;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0)
    khz = 100000, (void)hipGetLastError();
  p->timeout_ticks = (e ? atoll(e) : 10000) * (long long)khz; // wall_clock64 ticks per ms
  for (int q = 0; q < R; q++)
    p->region_off[q] = P2P_NONE;
  size_t doubles = 0;
  p->halo = !(nsend == 1 && send[0].peer == -1); // not the all-gather plan
  for (int i = 0; i < nrecv && p->halo; i++) {
    p->region_off[recv[i].peer] = (unsigned)doubles;
    doubles += (recv[i].count + 31) & ~(size_t)31;
    if (doubles > (size_t)8 << 20) // 64 MiB of halos: leave that to RCCL
      p->halo = 0;
  }
  if (!p->halo) {
    doubles = 0;
    for (int q = 0; q < R; q++)
      p->region_off[q] = P2P_NONE;
  }
  p->mbox_bytes = P2P_HEADER + doubles * sizeof(double) + 256;
  if (hipExtMallocWithFlags((void **)&p->mbox, p->mbox_bytes, hipDeviceMallocFinegrained) !=
      hipSuccess) {
    (void)hipGetLastError();
    free(p);
    return NULL;
  }
  if (hipMemset(p->mbox, 0, p->mbox_bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(p->mbox);
    free(p);
    return NULL;
  }
  p->peer[me] = p->mbox;
  return p;
}

// peer[] and tab[q*R + src] (region offsets of rank q) known: build the tables
static int p2p_connect(struct lsb_p2p *p, const unsigned *tab, const unsigned *halo_ok,
                       const struct lsb_xfer *recv, int nrecv, const struct lsb_xfer *send,
                       int nsend) {
  const int R = p->R, me = p->me;
  for (int q = 0; q < R; q++)
    p->halo &= halo_ok[q] != 0;
  if (hipMalloc((void **)&p->d_peer, sizeof(char *) * R) != hipSuccess ||
      hipMemcpy(p->d_peer, p->peer, sizeof(char *) * R, hipMemcpyHostToDevice) != hipSuccess)
    return 1;
  if (hipMalloc((void **)&p->d_tail, sizeof(unsigned)) != hipSuccess ||
      hipMemset(p->d_tail, 0, sizeof(unsigned)) != hipSuccess)
    return 1;
  if (!p->halo)
    return 0;
  p2p_send_ent hs[P2P_MAX_RANKS];
  p2p_recv_ent hr[P2P_MAX_RANKS];
  unsigned g = 0;
  for (int i = 0; i < nsend; i++) {
    const int q = send[i].peer;
    const unsigned off = tab[(size_t)q * R + me];
    if (off == P2P_NONE)
      return 2; // the receiver does not expect what the plan sends
    hs[i].dst = (double *)(p->peer[q] + P2P_HEADER) + off;
    hs[i].flag = (u64 *)(p->peer[q] + P2P_AR_BYTES + 64 * (size_t)me);
    hs[i].src_off = send[i].offset, hs[i].count = send[i].count;
    hs[i].wgs = wgs_for(send[i].count), hs[i].first_wg = g;
    g += hs[i].wgs;
  }
  p->send_grid = g, p->nsend = nsend;
  g = 0;
  for (int i = 0; i < nrecv; i++) {
    const int q = recv[i].peer;
    hr[i].src = (const double *)(p->mbox + P2P_HEADER) + p->region_off[q];
    hr[i].flag = (const u64 *)(p->mbox + P2P_AR_BYTES + 64 * (size_t)q);
    hr[i].dst_off = recv[i].offset, hr[i].count = recv[i].count;
    hr[i].wgs = wgs_for(recv[i].count), hr[i].first_wg = g;
    g += hr[i].wgs;
  }
  p->recv_grid = g, p->nrecv = nrecv;
  if (hipMalloc((void **)&p->d_send, sizeof(p2p_send_ent) * (nsend ? nsend : 1)) != hipSuccess ||
      hipMalloc((void **)&p->d_recv, sizeof(p2p_recv_ent) * (nrecv ? nrecv : 1)) != hipSuccess ||
      hipMalloc((void **)&p->d_counters, sizeof(unsigned) * P2P_MAX_RANKS) != hipSuccess)
    return 1;
  if (hipMemcpy(p->d_send, hs, sizeof(p2p_send_ent) * nsend, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(p->d_recv, hr, sizeof(p2p_recv_ent) * nrecv, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemset(p->d_counters, 0, sizeof(unsigned) * P2P_MAX_RANKS) != hipSuccess)
    return 1;
  return 0;
}

extern "C" void lsb_p2p_destroy(struct lsb_p2p *p) {
  if (!p)
    return;
  for (int q = 0; q < p->R; q++)
    if (p->opened[q])
      (void)hipIpcCloseMemHandle(p->peer[q]);
  (void)hipFree(p->d_peer), (void)hipFree(p->d_send), (void)hipFree(p->d_recv);
  (void)hipFree(p->d_counters), (void)hipFree(p->d_tail), (void)hipFree(p->mbox);
  (void)hipGetLastError();
  free(p);
}

/* One rank per GPU: a process each (bench.py under torch.distributed.run) or a
 * host thread each inside one process (hip_multi.c).  Collective over the RCCL
 * communicator: every rank's record -- IPC handle of its mailbox, process id,
 * device, the mailbox's address, region table -- travels by
 * lsb_hip_comm_allgather_u32.  A peer in ANOTHER process is reached through its
 * IPC handle; a peer in THIS process through the address itself, after
 * hipDeviceEnablePeerAccess (an IPC handle cannot be opened by the process that
 * made it).  Returns NULL when the path is unavailable HERE; the caller still
 * has to agree with its peers. */
#include <unistd.h>
extern "C" struct lsb_p2p *lsb_p2p_create_dist(const struct lsb_xfer *recv, int nrecv,
                                               const struct lsb_xfer *send, int nsend) {
  const int R = lsb_hip_comm_size(), me = lsb_hip_comm_rank();
  struct lsb_p2p *p = p2p_alloc(R, me, 0, recv, nrecv, send, nsend);
  /* every rank takes part in the all-gather, whatever happened locally */
  enum { R_OK_ = 16, R_HALO, R_IPC, R_PID, R_DEV, R_PTR_LO, R_PTR_HI, R_TAB };
  const unsigned W = R_TAB + P2P_MAX_RANKS;
  unsigned mine[R_TAB + P2P_MAX_RANKS];
  memset(mine, 0, sizeof mine);
  hipIpcMemHandle_t h;
  int ok = p != NULL, mydev = 0;
  if (ok && hipGetDevice(&mydev) != hipSuccess)
    ok = 0, (void)hipGetLastError();
  if (ok) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle is 64 bytes");
    if (hipIpcGetMemHandle(&h, p->mbox) == hipSuccess)
      memcpy(mine, &h, 64), mine[R_IPC] = 1;
    else
      (void)hipGetLastError(); /* peers of this process do not need it */
    const unsigned long long addr = (unsigned long long)(uintptr_t)p->mbox;
    mine[R_OK_] = 1, mine[R_HALO] = (unsigned)p->halo;
    mine[R_PID] = (unsigned)getpid(), mine[R_DEV] = (unsigned)mydev;
    mine[R_PTR_LO] = (unsigned)(addr & 0xFFFFFFFFull), mine[R_PTR_HI] = (unsigned)(addr >> 32);
    for (int q = 0; q < R; q++)
      mine[R_TAB + q] = p->region_off[q];
  }
  unsigned *all = (unsigned *)calloc((size_t)W * R, sizeof(unsigned));
  lsb_hip_comm_allgather_u32(mine, W, all);
  for (int q = 0; q < R; q++)
    ok &= all[(size_t)q * W + R_OK_] != 0;
  for (int q = 0; q < R && ok; q++) {
    if (q == me)
      continue;
    const unsigned *rec = all + (size_t)q * W;
    if (rec[R_PID] == (unsigned)getpid()) { /* a thread of this process */
      const int qdev = (int)rec[R_DEV];
      if (qdev != mydev) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, mydev, qdev) != hipSuccess || !can) {
          (void)hipGetLastError();
          ok = 0;
          break;
        }
        const hipError_t pe = hipDeviceEnablePeerAccess(qdev, 0);
        if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) {
          (void)hipGetLastError();
          ok = 0;
          break;
        }
        (void)hipGetLastError();
      }
      p->peer[q] = (char *)(uintptr_t)(((unsigned long long)rec[R_PTR_HI] << 32) | rec[R_PTR_LO]);
      continue;
    }
    if (!rec[R_IPC]) {
      ok = 0;
      break;
    }
    hipIpcMemHandle_t hq;
    memcpy(&hq, rec, 64);
    void *ptr = NULL;
    if (hipIpcOpenMemHandle(&ptr, hq, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
      (void)hipGetLastError();
      ok = 0;
      break;
    }
    p->peer[q] = (char *)ptr, p->opened[q] = 1;
  }
  if (ok) {
    unsigned tab[P2P_MAX_RANKS * P2P_MAX_RANKS], halo_ok[P2P_MAX_RANKS];
    for (int q = 0; q < R; q++) {
      halo_ok[q] = all[(size_t)q * W + R_HALO];
      for (int s = 0; s < R; s++)
        tab[(size_t)q * R + s] = all[(size_t)q * W + R_TAB + s];
    }
    ok = p2p_connect(p, tab, halo_ok, recv, nrecv, send, nsend) == 0;
  }
  free(all);
  if (!ok) {
    lsb_p2p_destroy(p);
    return NULL;
  }
  return p;
}

/* n shards on ONE device (test mode): mailboxes are ordinary local pointers. */
extern "C" int lsb_p2p_create_virtual(struct lsb_p2p **out, int n, struct lsb_xfer *const *recv,
                                      const int *nrecv, struct lsb_xfer *const *send,
                                      const int *nsend) {
  if (n > P2P_MAX_RANKS)
    return 1;
  unsigned tab[P2P_MAX_RANKS * P2P_MAX_RANKS], halo_ok[P2P_MAX_RANKS];
  for (int s = 0; s < n; s++) {
    out[s] = p2p_alloc(n, s, 1, recv[s], nrecv[s], send[s], nsend[s]);
    if (!out[s])
      return 1;
    halo_ok[s] = (unsigned)out[s]->halo;
    for (int q = 0; q < n; q++)
      tab[(size_t)s * n + q] = out[s]->region_off[q];
  }
  for (int s = 0; s < n; s++) {
    for (int q = 0; q < n; q++)
      out[s]->peer[q] = out[q]->mbox;
    if (p2p_connect(out[s], tab, halo_ok, recv[s], nrecv[s], send[s], nsend[s]))
      return 1;
  }
  return 0;
}

extern "C" int lsb_p2p_has_halo(const struct lsb_p2p *p) { return p && p->halo; }

extern "C" void lsb_p2p_send(struct lsb_p2p *p, const double *d_full,
                             const struct lsb_pcg_state *st, void *stream) {
  p->epoch_x++;
  if (p->nsend)
    k_p2p_send<<<p->send_grid, P2P_WG, 0, (hipStream_t)stream>>>(p->d_send, p->nsend, d_full,
                                                                 p->epoch_x, p->d_counters, st);
}

extern "C" void lsb_p2p_recv(struct lsb_p2p *p, double *d_full, struct lsb_pcg_state *st,
                             void *stream) {
  if (p->nrecv)
    k_p2p_recv<<<p->recv_grid, P2P_WG, 0, (hipStream_t)stream>>>(p->d_recv, p->nrecv, d_full,
                                                                 p->epoch_x, st, p->timeout_ticks);
}

/* send + recv in one launch; only where the peers are other processes */
extern "C" void lsb_p2p_sendrecv(struct lsb_p2p *p, double *d_full, struct lsb_pcg_state *st,
                                 void *stream) {
  if (p->virt)
    errx(EXIT_FAILURE, "lsb_p2p_sendrecv between shards of one stream would wait on itself");
  p->epoch_x++;
  if (p->send_grid + p->recv_grid)
    k_p2p_sendrecv<<<p->send_grid + p->recv_grid, P2P_WG, 0, (hipStream_t)stream>>>(
        p->d_send, p->nsend, p->send_grid, p->d_recv, p->nrecv, d_full, p->epoch_x,
        p->d_counters, st, p->timeout_ticks);
}

/* out[0..width+width2+nextra) = sum over ranks of {column sums of
 * parts[nparts][width], column sums of parts2[nparts2][width2], extra[0..nextra)};
 * phases: 1 = contribute, 2 = collect, 3 = both (one rank per process); out may
 * alias extra. */
extern "C" void lsb_p2p_allreduce(struct lsb_p2p *p, const double *parts, unsigned nparts,
                                  unsigned width, const double *parts2, unsigned nparts2,
                                  unsigned width2, const double *extra, unsigned nextra,
                                  double *out, struct lsb_pcg_state *st, int phases,
                                  void *stream) {
  if (width + width2 + nextra > 3)
    errx(EXIT_FAILURE, "lsb_p2p_allreduce: at most 3 values");
  if (phases & 1)
    p->epoch_r++;
  k_p2p_allreduce<<<1, P2P_WG, 0, (hipStream_t)stream>>>(
      parts, nparts, width, parts2, nparts2, width2, extra, nextra, out, p->d_peer, p->R, p->me,
      p->epoch_r, st, phases, p->timeout_ticks);
}

/* The same all-reduce with its two phases folded into neighbouring launches
 * (hip_ar.h): lsb_p2p_fold_contribute opens epoch k and fills what the SpMV
 * launcher needs for the tail (the caller adds the partial-sum arrays);
 * lsb_p2p_fold_collect describes the matching collect for k_cg1_update. */
extern "C" void lsb_p2p_fold_contribute(struct lsb_p2p *p, struct lsb_ar_tail *t) {
  p->epoch_r++;
  t->counter = p->d_tail;
  t->peer = p->d_peer, t->R = p->R, t->me = p->me, t->epoch = p->epoch_r;
}
extern "C" void lsb_p2p_fold_collect(const struct lsb_p2p *p, struct lsb_ar_collect *c) {
  c->mbox = p->mbox, c->R = p->R, c->epoch = p->epoch_r, c->timeout = p->timeout_ticks;
}

/* self-test: collect what the last contribute (phases = 1) sent, the way the sweep does,
 * in 64 workgroups at once, and compare with the round's known sums */
extern "C" void lsb_p2p_test_collect_check(struct lsb_p2p *p, unsigned round, unsigned *d_bad,
                                           struct lsb_pcg_state *st, void *stream) {
  k_p2p_collect_check<<<64, P2P_WG, 0, (hipStream_t)stream>>>(p->mbox, p->R, p->epoch_r,
                                                              p->timeout_ticks, round, d_bad, st);
}

/* self-test pieces (hip_cdna4.c drives them) */
extern "C" void lsb_p2p_test_pattern(double *d_full, size_t goff, size_t n, unsigned round,
                                     void *stream) {
  if (n)
    k_p2p_pattern<<<64, 256, 0, (hipStream_t)stream>>>(d_full, n, goff, round);
}
extern "C" void lsb_p2p_test_check_range(const double *d_full, size_t goff, size_t n,
                                         unsigned round, unsigned *d_bad, void *stream) {
  if (n)
    k_p2p_check_range<<<64, 256, 0, (hipStream_t)stream>>>(d_full, goff, n, round, d_bad);
}
extern "C" void lsb_p2p_test_setvals(double *d_v, int me, unsigned round, void *stream) {
  k_p2p_setvals<<<1, 1, 0, (hipStream_t)stream>>>(d_v, me, round);
}
extern "C" void lsb_p2p_test_checkvals(const double *d_v, int R, unsigned round, unsigned *d_bad,
                                       void *stream) {
  k_p2p_checkvals<<<1, 1, 0, (hipStream_t)stream>>>(d_v, R, round, d_bad);
}
