// Small-message all-reduce for the SyncBN statistics exchanges of one node (SURVEY §5.8 / §8b `comm_allreduce_small`).
//
// What it replaces: per BatchNorm call the reference's master / slave pipes move (sum, ssum, count) to a master copy and
// (mean, inv_std) back (modeling/sync_batchnorm/batchnorm.py:95-108, comm.py:56-129); the builder's plan merges the vectors of one
// dependency level into one arena (plan.Graph._bind_late), which round 2-4 sent through a stock RCCL all_reduce: 313 collectives of
// 0.3-5 KB per step on the critical path, each a ring / tree launch priced for megabytes.
//
// Protocol (push, flags, rank-ordered sum — one single-workgroup launch per exchange, capturable in a hipGraph: the sequence number lives
// in device memory):
//   * every rank owns a MAILBOX in its own HBM (fine-grained allocation, exported once as a hipIpc handle and mapped by every peer):
//       flags[2][W]           one 64-byte line per (set, sending rank)
//       data [2][W][slot]     the sending rank's vector of exchange set = seq & 1
//   * exchange seq: a rank WRITES its vector into slot [seq & 1][rank] of EVERY mailbox (posted xGMI writes), waits for their acknowledgement, then
//     writes flags[seq & 1][rank] = seq in every PEER's mailbox; it POLLS only its own mailbox (local HBM) until the W - 1 peer flags carry seq, and adds the
//     W slots in RANK ORDER — every rank adds the same numbers in the same order: results are bit-identical across ranks, as the reference's
//     master copy makes them.
//   * two sets suffice: a peer can publish exchange seq + 1 (other set) while this rank still reads seq, but seq + 2 only after its own
//     wait for seq + 1 has seen this rank's flag, which this rank writes after it has finished reading seq.
//   * polls are BOUNDED by the 100 MHz real-time counter (default 20 s, ADDK_COMM_TIMEOUT_MS: ranks of a node drift by launch jitter, a one-time code-object load, a host stall — never by that much): a flag that never arrives sets the error word
//     (sequence number and the missing peer) and the kernel returns; addk_comm_status hands it to the host, which raises.  Never an
//     unbounded spin (a hung wave can take the node down).
// Latency over xGMI is UNMEASURED (the builder's box has one GPU): the two-process rehearsal of tests/test_gpu_comm.py runs both ranks on
// one device.  torch.distributed's all_reduce stays the fallback and the checker (parallel.SmallComm self-test at start-up).
#include <stdlib.h>
#include <string.h>
#include "common.h"

namespace {

constexpr int COMM_MAXW = 16;
constexpr int COMM_LINE = 64;

struct CommDev { unsigned long long seq; unsigned long long err; };      // err: 0, or (1 << 63) | (seq << 8) | missing peer

struct Comm {
  int rank, world;
  int64_t slot;                    // bytes per (set, rank) data slot
  char* box[COMM_MAXW];            // mailbox base of every rank as mapped in THIS process (box[rank] = the own allocation)
  CommDev* dev;
  unsigned long long timeout_ticks;
};

struct CommK {
  char* box[COMM_MAXW];
  CommDev* dev;
  void* buf; long n;               // elements of 8 bytes (fp64, or pairs of fp32)
  int rank, world; long slot;
  unsigned long long timeout_ticks;
};

__host__ __device__ inline long comm_flag_off(int set, int r) { return ((long)set * COMM_MAXW + r) * COMM_LINE; }
__host__ __device__ inline long comm_data_off(int set, int r, int world, long slot) { return 2L * COMM_MAXW * COMM_LINE + ((long)set * world + r) * slot; }

typedef unsigned long long u64;
__device__ __forceinline__ void st_sys(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ u64 ld_sys(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// F64 = true: the 8-byte units are doubles; false: pairs of floats (the sum is per component)
template <bool F64>
__global__ void __launch_bounds__(256) comm_allreduce_kernel(const CommK k) {
  const int t = threadIdx.x;
  const u64 seq = ld_sys(&k.dev->seq) + 1;
  const int set = (int)(seq & 1);
  u64* buf = reinterpret_cast<u64*>(k.buf);
  // 1. push this rank's vector into every mailbox (own included), then the flags
  for (long i = t; i < k.n; i += 256) {
    const u64 v = buf[i];
    for (int r = 0; r < k.world; ++r)
      if (r != k.rank) st_sys(reinterpret_cast<u64*>(k.box[r] + comm_data_off(set, k.rank, k.world, k.slot)) + i, v);      // (the own share is taken from `buf` below)
  }
  // every access of the protocol is a system-scope (sc0 sc1) access that bypasses the caches, so ordering needs no cache maintenance: the pushed
  // vector has LEFT this GPU once its stores have been acknowledged (vmcnt 0), and only then the flags follow.  (__threadfence_system() here cost a
  // write-back AND invalidate of the whole L2 — buffer_wbl2 / buffer_inv — twice per exchange: +3.8 us per exchange, and cold caches for what follows.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t < k.world && t != k.rank) st_sys(reinterpret_cast<u64*>(k.box[t] + comm_flag_off(set, k.rank)), seq);
  // 2. wait (bounded) until every rank's flag of this set carries seq: thread r watches rank r, in this rank's own memory
  __shared__ unsigned int bad;
  if (t == 0) bad = 0;
  __syncthreads();
  const bool dead = ld_sys(&k.dev->err) != 0;      // an earlier exchange timed out: the communicator is broken, later exchanges do not wait again (the host raises)
  if (t < k.world && t != k.rank && !dead) {
    const u64* f = reinterpret_cast<const u64*>(k.box[k.rank] + comm_flag_off(set, t));
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    bool ok = false;
    while (true) {
      if (ld_sys(f) >= seq) { ok = true; break; }
      if (__builtin_amdgcn_s_memrealtime() - t0 > k.timeout_ticks) break;
      __builtin_amdgcn_s_sleep(8);
    }
    if (!ok) { atomicOr(&bad, 1u); st_sys(&k.dev->err, (1ull << 63) | (seq << 8) | (u64)t); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the polls have returned; the slots are read with cache-bypassing loads below
  __syncthreads();
  // 3. rank-ordered sum from the own mailbox (a timed-out exchange still writes something: the host raises on the error word)
  const u64* base = reinterpret_cast<const u64*>(k.box[k.rank] + comm_data_off(set, 0, k.world, k.slot));
  const long stride = k.slot / 8;
  for (long i = t; i < k.n; i += 256) {
    const u64 own = buf[i];
    u64 acc = k.rank == 0 ? own : ld_sys(base + i);
    for (int r = 1; r < k.world; ++r) {
      const u64 v = r == k.rank ? own : ld_sys(base + r * stride + i);
      if (F64) acc = __builtin_bit_cast(u64, __builtin_bit_cast(double, acc) + __builtin_bit_cast(double, v));
      else {
        const float a0 = __uint_as_float((unsigned)acc) + __uint_as_float((unsigned)v);
        const float a1 = __uint_as_float((unsigned)(acc >> 32)) + __uint_as_float((unsigned)(v >> 32));
        acc = (u64)__float_as_uint(a0) | ((u64)__float_as_uint(a1) << 32);
      }
    }
    buf[i] = acc;
  }
  __syncthreads();
  if (t == 0) st_sys(&k.dev->seq, seq);
}

}  // namespace

// bytes of one rank's mailbox for exchanges of at most max_bytes each
extern "C" int64_t addk_comm_mailbox_bytes(int32_t world, int64_t max_bytes) {
  if (world < 1 || world > COMM_MAXW || max_bytes < 8) return 0;
  const int64_t slot = (max_bytes + 63) / 64 * 64;
  return 2LL * COMM_MAXW * COMM_LINE + 2LL * world * slot;
}

// Allocates this rank's mailbox (fine-grained device memory, zeroed) and exports it: `handle64` receives the 64-byte hipIpc handle the
// peers open.  One-time control-plane allocation owned by the library (a hipIpc handle needs a whole allocation, not a slice of a caller's pool).
extern "C" int addk_comm_alloc(int32_t world, int64_t max_bytes, void** mailbox, void* handle64) {
  ADDK_REQUIRE(mailbox && handle64, "comm_alloc: null pointer");
  const int64_t bytes = addk_comm_mailbox_bytes(world, max_bytes);
  ADDK_REQUIRE(bytes > 0, "comm_alloc: world must be 1..%d and max_bytes >= 8", COMM_MAXW);
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) { (void)hipGetLastError(); addk_set_error("comm_alloc: hipExtMallocWithFlags(fine-grained, %lld B): %s", (long long)bytes, hipGetErrorString(e)); return ADDK_ERR_HIP; }
  e = hipMemset(p, 0, (size_t)bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  hipIpcMemHandle_t h;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpc handle size");
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); addk_set_error("comm_alloc: %s", hipGetErrorString(e)); return ADDK_ERR_HIP; }
  memcpy(handle64, &h, 64);
  *mailbox = p;
  return ADDK_OK;
}

// Maps every peer's mailbox (handles: world x 64 bytes, rank order; the own entry is not opened) and creates the communicator.
extern "C" int addk_comm_open(int32_t rank, int32_t world, int64_t max_bytes, void* my_mailbox, const void* handles, void** comm_out) {
  ADDK_REQUIRE(my_mailbox && comm_out && (handles || world == 1), "comm_open: null pointer");
  ADDK_REQUIRE(world >= 1 && world <= COMM_MAXW && rank >= 0 && rank < world, "comm_open: rank %d of %d", rank, world);
  ADDK_REQUIRE(addk_comm_mailbox_bytes(world, max_bytes) > 0, "comm_open: max_bytes");
  Comm* c = new Comm();
  c->rank = rank; c->world = world; c->slot = (max_bytes + 63) / 64 * 64;
  c->timeout_ticks = (unsigned long long)addk_env("ADDK_COMM_TIMEOUT_MS", 20000) * 100000ull;      // 100 MHz ticks
  for (int r = 0; r < COMM_MAXW; ++r) c->box[r] = nullptr;
  c->box[rank] = reinterpret_cast<char*>(my_mailbox);
  for (int r = 0; r < world; ++r) {
    if (r == rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, reinterpret_cast<const char*>(handles) + 64 * r, 64);
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      for (int q = 0; q < r; ++q) if (q != rank && c->box[q]) (void)hipIpcCloseMemHandle(c->box[q]);
      delete c;
      addk_set_error("comm_open: hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(e));
      return ADDK_ERR_HIP;
    }
    c->box[r] = reinterpret_cast<char*>(p);
  }
  hipError_t e = hipMalloc(&c->dev, sizeof(CommDev));
  if (e == hipSuccess) e = hipMemset(c->dev, 0, sizeof(CommDev));
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) { (void)hipGetLastError(); delete c; addk_set_error("comm_open: %s", hipGetErrorString(e)); return ADDK_ERR_HIP; }
  *comm_out = c;
  return ADDK_OK;
}

// In-place sum over the ranks of `count` elements (dtype 0: fp32, 1: fp64) at `buf`, on `stream`.  Every rank must issue the same sequence of
// calls (the protocol's sequence number is implicit).  count * element size <= the max_bytes the communicator was opened with; buf 8-byte aligned.
extern "C" int addk_comm_allreduce(void* comm, void* buf, int64_t count, int32_t dtype, void* stream) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  ADDK_REQUIRE(c && buf && count > 0 && (dtype == 0 || dtype == 1), "comm_allreduce: bad args");
  const int64_t bytes = count * (dtype ? 8 : 4);
  ADDK_REQUIRE(bytes <= c->slot, "comm_allreduce: %lld bytes exceed the communicator's %lld-byte slots", (long long)bytes, (long long)c->slot);
  ADDK_REQUIRE((((uintptr_t)buf) & 7) == 0 && (dtype == 1 || count % 2 == 0), "comm_allreduce: buffer must be 8-byte aligned and hold whole 8-byte units");
  CommK k;
  for (int r = 0; r < COMM_MAXW; ++r) k.box[r] = c->box[r];
  k.dev = c->dev; k.buf = buf; k.n = bytes / 8; k.rank = c->rank; k.world = c->world; k.slot = c->slot; k.timeout_ticks = c->timeout_ticks;
  if (dtype) hipLaunchKernelGGL(comm_allreduce_kernel<true>, dim3(1), dim3(256), 0, (hipStream_t)stream, k);
  else hipLaunchKernelGGL(comm_allreduce_kernel<false>, dim3(1), dim3(256), 0, (hipStream_t)stream, k);
  return addk_check_launch("comm_allreduce");
}

// Synchronous read of the device words: exchanges completed and the error word (0 = none; else bit 63 | seq << 8 | the peer whose flag never came).
extern "C" int addk_comm_status(void* comm, int64_t* seq, int64_t* err) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  ADDK_REQUIRE(c && seq && err, "comm_status: null pointer");
  CommDev h;
  const hipError_t e = hipMemcpy(&h, c->dev, sizeof h, hipMemcpyDeviceToHost);
  if (e != hipSuccess) { (void)hipGetLastError(); addk_set_error("comm_status: %s", hipGetErrorString(e)); return ADDK_ERR_HIP; }
  *seq = (int64_t)h.seq; *err = (int64_t)h.err;
  return ADDK_OK;
}

// Unmaps the peers and frees the own mailbox (pass NULL to keep it).  Call after the device has drained and every captured graph that holds
// comm launches has been released.
extern "C" int addk_comm_close(void* comm, void* my_mailbox) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (c) {
    for (int r = 0; r < c->world; ++r) if (r != c->rank && c->box[r]) (void)hipIpcCloseMemHandle(c->box[r]);
    if (c->dev) (void)hipFree(c->dev);
    delete c;
  }
  if (my_mailbox) (void)hipFree(my_mailbox);
  (void)hipGetLastError();
  return ADDK_OK;
}
