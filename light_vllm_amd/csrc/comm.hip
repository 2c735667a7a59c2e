// Direct small-message all-reduce between the GPUs (or processes) of one node, declared in
// include/lv_comm.h. It serves the decode-sized sum all-reduces of the tensor-parallel layers
// (reference: dist.all_reduce at lightvllm/layers/linear.py:388-389 and layers/embed_head.py:112).
//
// Every rank owns one allocation of uncached device memory that all peers map through HIP IPC:
//
//     [ flags: start[block][rank] ]  [ staging, parity 0 ]  [ staging, parity 1 ]
//
// One kernel per rank and call ("one shot"). Workgroup b of every rank owns slice b of the message
// and region b of every staging buffer (1/64 of it, whatever the message length: a workgroup's
// staging bytes never lie in another workgroup's region, in this call or any other):
//   1. copies its slice of the input into region b of this rank's staging buffer (parity =
//      sequence number & 1),
//   2. makes the copy visible (system-scope release), then one lane per peer writes the sequence
//      number into start[b][this rank] of THAT peer's flags (a remote 4-byte store over xGMI),
//   3. one lane per peer polls this rank's own start[b][peer] until it has reached the sequence
//      number (bounded by a wall-clock timeout), workgroup barrier, system-scope acquire,
//   4. reads slice b of every rank's staging buffer (remote 16-byte loads, all in flight together),
//      adds them in fp32 in rank order and writes the bf16 sum over the input.
// There is no second barrier: the two staging parities make it unnecessary. Workgroup b can
// overwrite parity p of its region only two of ITS calls later, and it can pass the wait of the
// call in between only after every peer's workgroup b has entered that call, i.e. the peer has
// finished the kernel that read parity p of region b (kernels of one rank run in stream order).
// A peer can be at most one call ahead, hence the ">=" in the poll.
// The sequence numbers are per workgroup and live in device memory, the kernel increments them
// itself: a captured hipGraph replays correctly, no workgroup ever waits for another workgroup of
// its own rank, and calls of different lengths (which use different numbers of workgroups) mix
// freely because a workgroup's parity only ever guards its own fixed region.
//
// "Two shot" for larger messages on many ranks (one shot reads (n-1) x the message per rank):
// after the same publish + wait, rank r sums only sub-slice r of slice b over all ranks and writes
// the bf16 result into sub-slice r of EVERY rank's staging buffer - in place: rank r is the only
// reader of that sub-slice anywhere, and it has finished reading before it writes - then a second
// flag exchange (done[b][rank]) and every rank copies slice b out of its own staging buffer.
// (n-1)/n of the message read and as much written remotely per rank; same sums, same bits.
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include <new>

#include "lv_comm.h"
#include "lv_common.h"

namespace lv {

constexpr int kCommMaxBlocks = 64;
constexpr size_t kCommBlockBytes = 8192;     // message bytes per workgroup before the grid stops growing
constexpr size_t kCommFlagBytes = 4096;      // start[64][8] words = 2 KiB, padded
constexpr size_t kCommAllocBytes = kCommFlagBytes + 2 * static_cast<size_t>(LV_COMM_MAX_BYTES);
constexpr size_t kCommRegionVecs = LV_COMM_MAX_BYTES / 16 / kCommMaxBlocks;   // staging region of one workgroup, 16-byte units

struct CommFlags {
  uint32_t start[kCommMaxBlocks][LV_COMM_MAX_RANKS];   // "rank r has published slice b of call e"
  uint32_t done[kCommMaxBlocks][LV_COMM_MAX_RANKS];    // two shot: "rank r has written its sums of slice b"
};
static_assert(sizeof(CommFlags) <= kCommFlagBytes, "flag words must fit their region");

struct CommParams {
  uint16_t* data;
  size_t nvec;                              // message length in 16-byte units
  unsigned char* base[LV_COMM_MAX_RANKS];   // every rank's allocation as mapped here (own: local pointer)
  uint32_t* epoch;                          // [kCommMaxBlocks] this rank's per-workgroup sequence numbers
  int* dev_failed;                          // sticky: set once a wait has timed out
  int* host_failed;                         // the same word in host-mapped memory
  long long timeout_ticks;                  // of the 100 MHz wall clock
  int rank, world;
};

__device__ __forceinline__ void add_bf16x8(float* acc, u32x4 v) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    acc[2 * j] += bf16_lo(v[j]);
    acc[2 * j + 1] += bf16_hi(v[j]);
  }
}

// One lane per rank: store `epoch` into word [b][p.rank] of every rank's flag array at byte offset
// `which` (start or done), then poll this rank's own word [b][lane's rank] until it has got there.
__device__ __forceinline__ void flag_exchange(const CommParams& p, int world, int b, int tid, size_t which,
                                              uint32_t epoch, int failed) {
  if (tid < world) {
    uint32_t* theirs = reinterpret_cast<uint32_t*>(p.base[tid] + which) + b * LV_COMM_MAX_RANKS + p.rank;
    __hip_atomic_store(theirs, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (!failed) {
      const uint32_t* mine = reinterpret_cast<const uint32_t*>(p.base[p.rank] + which) + b * LV_COMM_MAX_RANKS + tid;
      const long long t0 = wall_clock64();
      // relaxed system-scope loads (they bypass the caches); the acquire is the caller's fence
      while (static_cast<int32_t>(__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
        if (wall_clock64() - t0 > p.timeout_ticks) {   // the exit every waiting lane reaches
          __hip_atomic_store(p.dev_failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(p.host_failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
        __builtin_amdgcn_s_sleep(4);
      }
    }
  }
}

// WORLD > 0: number of ranks known at compile time (loads of all ranks unrolled); 0: p.world.
template <int WORLD, bool TWO_SHOT>
__global__ __launch_bounds__(256) void allreduce_kernel(const CommParams p) {
  const int world = WORLD > 0 ? WORLD : p.world;
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  __shared__ uint32_t s_epoch;
  __shared__ int s_failed;
  if (tid == 0) {
    const uint32_t e = p.epoch[b] + 1;
    p.epoch[b] = e;
    s_epoch = e;
    s_failed = *p.dev_failed;
  }
  __syncthreads();
  const uint32_t epoch = s_epoch;
  const size_t parity_off = kCommFlagBytes + static_cast<size_t>(epoch & 1u) * LV_COMM_MAX_BYTES;

  const size_t per = (p.nvec + gridDim.x - 1) / gridDim.x;   // <= kCommRegionVecs (host-checked)
  const size_t lo = static_cast<size_t>(b) * per;
  const size_t hi = lo + per < p.nvec ? lo + per : p.nvec;
  // message element i of this slice is staged at reg + i of every rank's buffer
  const ptrdiff_t reg = static_cast<ptrdiff_t>(b * kCommRegionVecs) - static_cast<ptrdiff_t>(lo);

  // 1. publish this rank's slice
  {
    u32x4* mine = reinterpret_cast<u32x4*>(p.base[p.rank] + parity_off) + reg;
    const u32x4* src = reinterpret_cast<const u32x4*>(p.data);
    for (size_t i = lo + tid; i < hi; i += 256) mine[i] = src[i];
  }
  __threadfence_system();   // every lane: its staging stores are visible to the node before the flag
  __syncthreads();

  // 2. + 3. tell every rank, then wait for every rank (one lane each)
  flag_exchange(p, world, b, tid, offsetof(CommFlags, start), epoch, s_failed);
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // system scope: nothing stale of the peers' slices

  // 4. sum over all ranks, rank order, fp32. One shot: the whole slice, straight into the output.
  //    Two shot: this rank's sub-slice only, into every rank's staging buffer.
  size_t slo = lo, shi = hi;
  if constexpr (TWO_SHOT) {
    const size_t per2 = (hi - lo + world - 1) / world;
    slo = lo + p.rank * per2 < hi ? lo + p.rank * per2 : hi;
    shi = slo + per2 < hi ? slo + per2 : hi;
  }
  u32x4* dst = reinterpret_cast<u32x4*>(p.data);
  for (size_t i = slo + tid; i < shi; i += 256) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (WORLD > 0) {
      u32x4 v[WORLD];
#pragma unroll
      for (int r = 0; r < WORLD; ++r) v[r] = (reinterpret_cast<const u32x4*>(p.base[r] + parity_off) + reg)[i];
#pragma unroll
      for (int r = 0; r < WORLD; ++r) add_bf16x8(acc, v[r]);
    } else {
      for (int r = 0; r < world; ++r) add_bf16x8(acc, (reinterpret_cast<const u32x4*>(p.base[r] + parity_off) + reg)[i]);
    }
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack_bf16x2(acc[2 * j], acc[2 * j + 1]);
    if constexpr (TWO_SHOT) {
      for (int r = 0; r < world; ++r) (reinterpret_cast<u32x4*>(p.base[r] + parity_off) + reg)[i] = o;
    } else {
      dst[i] = o;
    }
  }
  if constexpr (TWO_SHOT) {
    // 5. every rank's sums of slice b have landed in this rank's staging buffer: copy the slice out
    __threadfence_system();
    __syncthreads();
    flag_exchange(p, world, b, tid, offsetof(CommFlags, done), epoch, s_failed);
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    const u32x4* mine = reinterpret_cast<const u32x4*>(p.base[p.rank] + parity_off) + reg;
    for (size_t i = lo + tid; i < hi; i += 256) dst[i] = mine[i];
  }
}

// ---------------------------------------------------------------------------------------------
// The all-reduce above followed by the step every tensor-parallel layer runs on its result:
// residual add + RMSNorm (reference: dist.all_reduce in RowParallelLinear.forward,
// layers/linear.py:388-389, then RMSNorm.add_rms_forward, layers/layernorm.py:52-80). One kernel
// instead of two and one pass less over [rows, hidden]. Workgroup b owns whole rows; after the
// same publish + wait as the one-shot kernel it forms, per row,
//     y = bf16(sum over ranks, fp32, rank order)         (what the all-reduce alone leaves in x)
//     s = float(y) + float(residual);  residual = bf16(s)
//     x = bf16(s * rsqrt(mean(s^2) + eps)) * weight
// with the arithmetic, the element-to-lane mapping and the reduction tree of add_rms_norm_kernel
// (csrc/fused_ops.hip), so the pair of calls and this kernel give the same bits.
// ---------------------------------------------------------------------------------------------
constexpr int kNormMaxChunks = 4;   // hidden <= 8 * 256 * 4

struct CommNormParams {
  CommParams c;            // c.data = x, c.nvec = rows * chunks
  uint16_t* residual;
  const uint16_t* weight;
  int rows, chunks;        // chunks = hidden / 8 (16-byte units per row)
  float eps, inv_hidden;
};

__device__ __forceinline__ float comm_round_bf16(float x) {
  return __builtin_bit_cast(float, static_cast<u32>(f32_to_bf16_bits(x)) << 16);
}

template <int WORLD>
__global__ __launch_bounds__(256) void allreduce_add_rms_norm_kernel(const CommNormParams q) {
  const CommParams& p = q.c;
  const int world = WORLD > 0 ? WORLD : p.world;
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  __shared__ uint32_t s_epoch;
  __shared__ int s_failed;
  __shared__ float wave_sums[2][4];
  if (tid == 0) {
    const uint32_t e = p.epoch[b] + 1;
    p.epoch[b] = e;
    s_epoch = e;
    s_failed = *p.dev_failed;
  }
  __syncthreads();
  const uint32_t epoch = s_epoch;
  const size_t parity_off = kCommFlagBytes + static_cast<size_t>(epoch & 1u) * LV_COMM_MAX_BYTES;

  const int rows_per = (q.rows + gridDim.x - 1) / gridDim.x;
  const int r_lo = b * rows_per;
  const int r_hi = r_lo + rows_per < q.rows ? r_lo + rows_per : q.rows;
  const size_t lo = static_cast<size_t>(r_lo) * q.chunks;
  const size_t hi = static_cast<size_t>(r_hi > r_lo ? r_hi : r_lo) * q.chunks;
  const ptrdiff_t reg = static_cast<ptrdiff_t>(b * kCommRegionVecs) - static_cast<ptrdiff_t>(lo);

  {
    u32x4* mine = reinterpret_cast<u32x4*>(p.base[p.rank] + parity_off) + reg;
    const u32x4* src = reinterpret_cast<const u32x4*>(p.data);
    for (size_t i = lo + tid; i < hi; i += 256) mine[i] = src[i];
  }
  __threadfence_system();
  __syncthreads();
  flag_exchange(p, world, b, tid, offsetof(CommFlags, start), epoch, s_failed);
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");

  u32x4* xo = reinterpret_cast<u32x4*>(p.data);
  u32x4* res = reinterpret_cast<u32x4*>(q.residual);
  const u32x4* wgt = reinterpret_cast<const u32x4*>(q.weight);
  for (int row = r_lo; row < r_hi; ++row) {
    const size_t row0 = static_cast<size_t>(row) * q.chunks;
    float v[kNormMaxChunks][8];
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < kNormMaxChunks; ++j) {
      const int c = tid + j * 256;
      if (c < q.chunks) {
        const size_t i = row0 + c;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (WORLD > 0) {
          u32x4 pv[WORLD];
#pragma unroll
          for (int r = 0; r < WORLD; ++r) pv[r] = (reinterpret_cast<const u32x4*>(p.base[r] + parity_off) + reg)[i];
#pragma unroll
          for (int r = 0; r < WORLD; ++r) add_bf16x8(acc, pv[r]);
        } else {
          for (int r = 0; r < world; ++r) add_bf16x8(acc, (reinterpret_cast<const u32x4*>(p.base[r] + parity_off) + reg)[i]);
        }
        const u32x4 rv = res[i];
        const u32 rw[4] = {rv.x, rv.y, rv.z, rv.w};
        u32 nw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          // the all-reduce's rounding to bf16, then the residual add in fp32
          v[j][2 * k] = __fadd_rn(comm_round_bf16(acc[2 * k]), bf16_lo(rw[k]));
          v[j][2 * k + 1] = __fadd_rn(comm_round_bf16(acc[2 * k + 1]), bf16_hi(rw[k]));
          nw[k] = pack_bf16x2(v[j][2 * k], v[j][2 * k + 1]);
        }
        u32x4 nr;
        nr.x = nw[0]; nr.y = nw[1]; nr.z = nw[2]; nr.w = nw[3];
        res[i] = nr;
#pragma unroll
        for (int k = 0; k < 8; ++k) sq += v[j][k] * v[j][k];
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off);
    float* ws = wave_sums[(row - r_lo) & 1];     // two sets: the next row's writes cannot overtake this row's reads
    if ((tid & 63) == 0) ws[tid >> 6] = sq;
    __syncthreads();
    const float total = (ws[0] + ws[1]) + (ws[2] + ws[3]);
    const float rn = rsqrtf(__fadd_rn(total * q.inv_hidden, q.eps));
#pragma unroll
    for (int j = 0; j < kNormMaxChunks; ++j) {
      const int c = tid + j * 256;
      if (c < q.chunks) {
        const u32x4 wv = wgt[c];
        const u32 ww[4] = {wv.x, wv.y, wv.z, wv.w};
        u32 ow[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float lo_ = __fmul_rn(comm_round_bf16(__fmul_rn(v[j][2 * k], rn)), bf16_lo(ww[k]));
          const float hi_ = __fmul_rn(comm_round_bf16(__fmul_rn(v[j][2 * k + 1], rn)), bf16_hi(ww[k]));
          ow[k] = pack_bf16x2(lo_, hi_);
        }
        u32x4 ov;
        ov.x = ow[0]; ov.y = ow[1]; ov.z = ow[2]; ov.w = ow[3];
        xo[row0 + c] = ov;
      }
    }
  }
}

}  // namespace lv

using namespace lv;

struct lv_comm {
  int rank = 0, world = 0, device = -1;
  bool connected = false;
  long long timeout_ticks = 0;
  size_t one_shot_inflow = 2u << 20;                   // AUTO: two shot once (world - 1) x bytes passes this
  unsigned char* mine = nullptr;                       // this rank's shared allocation
  unsigned char* base[LV_COMM_MAX_RANKS] = {};         // every rank's allocation as mapped here
  uint32_t* epoch = nullptr;
  int* dev_failed = nullptr;
  int* host_failed = nullptr;                          // host pointer of the mapped word
  int* host_failed_dev = nullptr;                      // its device pointer
};

#define LV_HIP_OK(call, what)                                                                   \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess) return fail(LV_ERR_COMM, "%s: %s", what, hipGetErrorString(e_));      \
  } while (0)

extern "C" {

int lv_comm_create(int rank, int world_size, int timeout_ms, lv_comm_t** comm) {
  if (!comm) return fail(LV_ERR_INVALID_ARGUMENT, "comm_create: null output pointer");
  *comm = nullptr;
  if (world_size < 2 || world_size > LV_COMM_MAX_RANKS || rank < 0 || rank >= world_size)
    return fail(LV_ERR_INVALID_ARGUMENT, "comm_create: rank %d of %d (2..%d ranks)", rank, world_size,
                LV_COMM_MAX_RANKS);
  if (timeout_ms < 0) return fail(LV_ERR_INVALID_ARGUMENT, "comm_create: negative timeout");
  lv_comm* c = new (std::nothrow) lv_comm();
  if (!c) return fail(LV_ERR_COMM, "comm_create: out of host memory");
  c->rank = rank;
  c->world = world_size;
  c->timeout_ticks = static_cast<long long>(timeout_ms > 0 ? timeout_ms : 10000) * 100000;   // 100 MHz
  if (const char* env = getenv("LV_COMM_ONE_SHOT_INFLOW_BYTES")) {
    char* end = nullptr;
    const unsigned long long v = strtoull(env, &end, 10);
    if (end != env) c->one_shot_inflow = static_cast<size_t>(v);
  }
  int rc = LV_OK;
  do {
    hipError_t e = hipGetDevice(&c->device);
    if (e == hipSuccess)
      e = hipExtMallocWithFlags(reinterpret_cast<void**>(&c->mine), kCommAllocBytes, hipDeviceMallocUncached);
    if (e == hipSuccess) e = hipMemset(c->mine, 0, kCommAllocBytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->epoch), kCommMaxBlocks * sizeof(uint32_t) + 256);
    if (e == hipSuccess) e = hipMemset(c->epoch, 0, kCommMaxBlocks * sizeof(uint32_t) + 256);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->host_failed), 64, hipHostMallocMapped);
    if (e == hipSuccess) {
      *c->host_failed = 0;
      e = hipHostGetDevicePointer(reinterpret_cast<void**>(&c->host_failed_dev), c->host_failed, 0);
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
      rc = fail(LV_ERR_COMM, "comm_create: %s", hipGetErrorString(e));
      break;
    }
    c->dev_failed = reinterpret_cast<int*>(c->epoch + kCommMaxBlocks);   // behind the sequence numbers
    c->base[rank] = c->mine;
  } while (0);
  if (rc != LV_OK) {
    if (c->mine) (void)hipFree(c->mine);
    if (c->epoch) (void)hipFree(c->epoch);
    if (c->host_failed) (void)hipHostFree(c->host_failed);
    delete c;
    return rc;
  }
  *comm = c;
  return LV_OK;
}

int lv_comm_get_handle(lv_comm_t* c, void* handle_out) {
  if (!c || !handle_out) return fail(LV_ERR_INVALID_ARGUMENT, "comm_get_handle: null pointer");
  static_assert(sizeof(hipIpcMemHandle_t) == LV_COMM_HANDLE_BYTES, "IPC handle size");
  hipIpcMemHandle_t h;
  LV_HIP_OK(hipIpcGetMemHandle(&h, c->mine), "comm_get_handle: hipIpcGetMemHandle");
  memcpy(handle_out, &h, sizeof(h));
  return LV_OK;
}

int lv_comm_connect(lv_comm_t* c, const void* all_handles) {
  if (!c || !all_handles) return fail(LV_ERR_INVALID_ARGUMENT, "comm_connect: null pointer");
  if (c->connected) return fail(LV_ERR_INVALID_ARGUMENT, "comm_connect: already connected");
  int dev = -1;
  LV_HIP_OK(hipGetDevice(&dev), "comm_connect: hipGetDevice");
  if (dev != c->device) return fail(LV_ERR_INVALID_ARGUMENT, "comm_connect: created on device %d, current is %d", c->device, dev);
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, static_cast<const unsigned char*>(all_handles) + static_cast<size_t>(r) * LV_COMM_HANDLE_BYTES, sizeof(h));
    void* ptr = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess || !ptr) {
      for (int q = 0; q < r; ++q)
        if (q != c->rank && c->base[q]) {
          (void)hipIpcCloseMemHandle(c->base[q]);
          c->base[q] = nullptr;
        }
      return fail(LV_ERR_COMM, "comm_connect: hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(e));
    }
    c->base[r] = static_cast<unsigned char*>(ptr);
  }
  // Touch every mapping from the host before any kernel does: a mapping this device cannot reach
  // then fails here with an error code instead of faulting a kernel.
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank) continue;
    uint32_t probe = 0;
    hipError_t e = hipMemcpy(&probe, c->base[r], sizeof(probe), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
      for (int q = 0; q < c->world; ++q)
        if (q != c->rank && c->base[q]) {
          (void)hipIpcCloseMemHandle(c->base[q]);
          c->base[q] = nullptr;
        }
      return fail(LV_ERR_COMM, "comm_connect: rank %d's buffer is mapped but not readable from device %d: %s", r,
                  c->device, hipGetErrorString(e));
    }
  }
  c->connected = true;
  return LV_OK;
}

int lv_comm_set_one_shot_inflow(lv_comm_t* c, size_t bytes) {
  if (!c) return fail(LV_ERR_INVALID_ARGUMENT, "set_one_shot_inflow: null communicator");
  c->one_shot_inflow = bytes;
  return LV_OK;
}

size_t lv_comm_one_shot_inflow(lv_comm_t* c) { return c ? c->one_shot_inflow : 0; }

int lv_comm_failed(lv_comm_t* c) {
  if (!c) return 1;
  return *reinterpret_cast<volatile int*>(c->host_failed) != 0;
}

int lv_comm_all_reduce(lv_comm_t* c, void* data, size_t count, int algorithm, lv_stream_t stream) {
  if (!c) return fail(LV_ERR_INVALID_ARGUMENT, "all_reduce: null communicator");
  if (!c->connected) return fail(LV_ERR_INVALID_ARGUMENT, "all_reduce: communicator not connected");
  if (lv_comm_failed(c))
    return fail(LV_ERR_COMM, "all_reduce: rank %d gave up waiting for a peer in an earlier call; the communicator is dead", c->rank);
  if (algorithm != LV_COMM_ALGO_AUTO && algorithm != LV_COMM_ALGO_ONE_SHOT && algorithm != LV_COMM_ALGO_TWO_SHOT)
    return fail(LV_ERR_INVALID_ARGUMENT, "all_reduce: algorithm %d", algorithm);
  if (count == 0) return LV_OK;
  if (!data) return fail(LV_ERR_INVALID_ARGUMENT, "all_reduce: null data");
  const size_t bytes = count * 2;
  if (bytes % 16 != 0 || (reinterpret_cast<uintptr_t>(data) & 15u) != 0)
    return fail(LV_ERR_UNSUPPORTED, "all_reduce: %zu bytes at %p: 16-byte multiples and alignment required", bytes, data);
  if (bytes > LV_COMM_MAX_BYTES)
    return fail(LV_ERR_UNSUPPORTED, "all_reduce: %zu bytes exceed the %u-byte limit of the direct path", bytes,
                static_cast<unsigned>(LV_COMM_MAX_BYTES));
  CommParams p;
  p.data = static_cast<uint16_t*>(data);
  p.nvec = bytes / 16;
  for (int r = 0; r < LV_COMM_MAX_RANKS; ++r) p.base[r] = r < c->world ? c->base[r] : nullptr;
  p.epoch = c->epoch;
  p.dev_failed = c->dev_failed;
  p.host_failed = c->host_failed_dev;
  p.timeout_ticks = c->timeout_ticks;
  p.rank = c->rank;
  p.world = c->world;
  size_t blocks = (bytes + kCommBlockBytes - 1) / kCommBlockBytes;
  if (blocks > static_cast<size_t>(kCommMaxBlocks)) blocks = kCommMaxBlocks;
  if ((p.nvec + blocks - 1) / blocks > kCommRegionVecs)   // cannot happen within LV_COMM_MAX_BYTES; guards the region arithmetic
    return fail(LV_ERR_UNSUPPORTED, "all_reduce: slice of %zu bytes per workgroup exceeds its staging region", (p.nvec + blocks - 1) / blocks * 16);
  const dim3 grid(static_cast<unsigned>(blocks)), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // One shot moves (world - 1) x the message into every rank, two shot 2 (world - 1) / world of it in
  // two exchanges: two shot once the one-shot inflow passes 2 MiB (8 ranks: messages over 292 KiB,
  // 4 ranks: over 683 KiB, 2 ranks: never). The 2 MiB default is set from the byte counts alone (no
  // run across two GPUs has tuned it yet): lv_comm_set_one_shot_inflow / LV_COMM_ONE_SHOT_INFLOW_BYTES
  // move it, the same on every rank.
  const bool two = algorithm == LV_COMM_ALGO_TWO_SHOT ||
                   (algorithm == LV_COMM_ALGO_AUTO && static_cast<size_t>(c->world - 1) * bytes > c->one_shot_inflow);
  if (two) {
    switch (c->world) {
      case 2: hipLaunchKernelGGL((allreduce_kernel<2, true>), grid, block, 0, s, p); break;
      case 4: hipLaunchKernelGGL((allreduce_kernel<4, true>), grid, block, 0, s, p); break;
      case 8: hipLaunchKernelGGL((allreduce_kernel<8, true>), grid, block, 0, s, p); break;
      default: hipLaunchKernelGGL((allreduce_kernel<0, true>), grid, block, 0, s, p); break;
    }
  } else {
    switch (c->world) {
      case 2: hipLaunchKernelGGL((allreduce_kernel<2, false>), grid, block, 0, s, p); break;
      case 4: hipLaunchKernelGGL((allreduce_kernel<4, false>), grid, block, 0, s, p); break;
      case 8: hipLaunchKernelGGL((allreduce_kernel<8, false>), grid, block, 0, s, p); break;
      default: hipLaunchKernelGGL((allreduce_kernel<0, false>), grid, block, 0, s, p); break;
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(LV_ERR_LAUNCH, "all_reduce launch: %s", hipGetErrorString(e));
  return LV_OK;
}

int lv_comm_all_reduce_add_rms_norm(lv_comm_t* c, void* x, void* residual, const void* weight, int64_t rows,
                                    int hidden, float eps, lv_stream_t stream) {
  if (!c) return fail(LV_ERR_INVALID_ARGUMENT, "all_reduce_add_rms_norm: null communicator");
  if (!c->connected) return fail(LV_ERR_INVALID_ARGUMENT, "all_reduce_add_rms_norm: communicator not connected");
  if (lv_comm_failed(c))
    return fail(LV_ERR_COMM, "all_reduce_add_rms_norm: rank %d gave up waiting for a peer in an earlier call; the communicator is dead", c->rank);
  if (rows == 0) return LV_OK;
  if (!x || !residual || !weight) return fail(LV_ERR_INVALID_ARGUMENT, "all_reduce_add_rms_norm: null pointer");
  if (rows < 0 || hidden <= 0 || hidden % 8 != 0 || hidden > 8 * 256 * kNormMaxChunks)
    return fail(LV_ERR_UNSUPPORTED, "all_reduce_add_rms_norm: %lld rows of %d (hidden: a multiple of 8 up to %d)",
                static_cast<long long>(rows), hidden, 8 * 256 * kNormMaxChunks);
  if (!(eps > 0.f) || !(eps < 1.f)) return fail(LV_ERR_INVALID_ARGUMENT, "all_reduce_add_rms_norm: eps %g", eps);
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(weight)) & 15u) != 0)
    return fail(LV_ERR_UNSUPPORTED, "all_reduce_add_rms_norm: 16-byte alignment required");
  const size_t bytes = static_cast<size_t>(rows) * hidden * 2;
  if (bytes > LV_COMM_MAX_BYTES)
    return fail(LV_ERR_UNSUPPORTED, "all_reduce_add_rms_norm: %zu bytes exceed the %u-byte limit of the direct path", bytes,
                static_cast<unsigned>(LV_COMM_MAX_BYTES));
  const int chunks = hidden / 8;
  int64_t blocks = rows < kCommMaxBlocks ? rows : kCommMaxBlocks;
  const int64_t rows_per = (rows + blocks - 1) / blocks;
  if (static_cast<size_t>(rows_per) * chunks > kCommRegionVecs)   // cannot happen within LV_COMM_MAX_BYTES
    return fail(LV_ERR_UNSUPPORTED, "all_reduce_add_rms_norm: %lld rows per workgroup exceed its staging region",
                static_cast<long long>(rows_per));
  CommNormParams q;
  q.c.data = static_cast<uint16_t*>(x);
  q.c.nvec = static_cast<size_t>(rows) * chunks;
  for (int r = 0; r < LV_COMM_MAX_RANKS; ++r) q.c.base[r] = r < c->world ? c->base[r] : nullptr;
  q.c.epoch = c->epoch;
  q.c.dev_failed = c->dev_failed;
  q.c.host_failed = c->host_failed_dev;
  q.c.timeout_ticks = c->timeout_ticks;
  q.c.rank = c->rank;
  q.c.world = c->world;
  q.residual = static_cast<uint16_t*>(residual);
  q.weight = static_cast<const uint16_t*>(weight);
  q.rows = static_cast<int>(rows);
  q.chunks = chunks;
  q.eps = eps;
  q.inv_hidden = static_cast<float>(1.0 / hidden);
  const dim3 grid(static_cast<unsigned>(blocks)), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (c->world) {
    case 2: hipLaunchKernelGGL((allreduce_add_rms_norm_kernel<2>), grid, block, 0, s, q); break;
    case 4: hipLaunchKernelGGL((allreduce_add_rms_norm_kernel<4>), grid, block, 0, s, q); break;
    case 8: hipLaunchKernelGGL((allreduce_add_rms_norm_kernel<8>), grid, block, 0, s, q); break;
    default: hipLaunchKernelGGL((allreduce_add_rms_norm_kernel<0>), grid, block, 0, s, q); break;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(LV_ERR_LAUNCH, "all_reduce_add_rms_norm launch: %s", hipGetErrorString(e));
  return LV_OK;
}

int lv_comm_destroy(lv_comm_t* c) {
  if (!c) return LV_OK;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < c->world; ++r)
    if (r != c->rank && c->base[r]) (void)hipIpcCloseMemHandle(c->base[r]);
  if (c->mine) (void)hipFree(c->mine);
  if (c->epoch) (void)hipFree(c->epoch);
  if (c->host_failed) (void)hipHostFree(c->host_failed);
  delete c;
  return LV_OK;
}

}  // extern "C"
