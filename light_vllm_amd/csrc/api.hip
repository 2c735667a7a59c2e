// C-ABI entry points of liblvattn.so (declared in include/lv_attention.h): argument validation
// on the host, then stream-ordered kernel launches. Nothing here synchronises or allocates.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>

#include <mutex>

#include "lv_kernels.h"

namespace lv {

static thread_local char g_last_error[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
  return code;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr float kLog2e = 1.4426950408889634f;
constexpr int kMaxSplits = 512;
constexpr int kSplitUnitTokens = 64;  // split granularity of the decode kernel
constexpr int kFillWorkgroups = 256;      // one workgroup per CU
constexpr int kTargetStreams = 256;       // sequences x splits the HBM channels want to see at once
constexpr int kMinCutTokens = 512;        // ... but no split made for that reason shorter than this
constexpr int kMaxWorkgroups = 4096;
constexpr int kSaturatedUnits = 1024;     // (sequence, kv head) pairs that fill the chip unsplit

// Split count from the launch shape alone (lengths live on the device; `max_blocks_per_seq` is
// the table capacity). Fitted to sweeps under hipGraph replay (tools/kbench.py decode_splits,
// decode_mid, decode_tp: 1-256 sequences, 1-8 kv heads, 1k-32k tokens), two needs:
//   fill     at least one workgroup per CU: ceil(256 / (batch * kv heads));
//   streams  about 256 independent (sequence, split) streams, whatever the head count - what
//            matters is how many places of the cache are read at once: 64 sequences x 4096
//            tokens x 8 kv heads run 9 % faster cut in 4, 64 x 32768 x 4 heads 12 %, while 256
//            sequences never gain from a cut (-3 ... -8 %) - but no such split under 512 tokens.
// The larger of the two, never below the kernel's 64-token split unit. Within 4 % of the best
// count on every shape of the sweeps (the first rule, 1024 workgroups always: up to 30 % off).
static int heuristic_splits(int batch, int num_kv_heads, int max_blocks_per_seq, int block_size) {
  const int64_t units = static_cast<int64_t>(batch) * num_kv_heads;
  if (units <= 0 || max_blocks_per_seq <= 0) return 1;
  if (units >= kSaturatedUnits) return 1;
  const int64_t cap_tokens = static_cast<int64_t>(max_blocks_per_seq) * block_size;
  const int64_t fill = (kFillWorkgroups + units - 1) / units;
  int64_t streams = (kTargetStreams + batch - 1) / batch;
  const int64_t longest = cap_tokens / kMinCutTokens > 1 ? cap_tokens / kMinCutTokens : 1;
  if (streams > longest) streams = longest;
  int64_t n = fill > streams ? fill : streams;
  const int64_t max_n = (cap_tokens + kSplitUnitTokens - 1) / kSplitUnitTokens;
  if (n > max_n) n = max_n;
  if (n > kMaxSplits) n = kMaxSplits;
  if (n * units > kMaxWorkgroups) n = kMaxWorkgroups / units;
  return n < 1 ? 1 : static_cast<int>(n);
}

constexpr int kMaxPrefillSplits = 32;

// KV splits of a prefill launch, from its static shape alone (lengths live on the device): none
// while the query tiles by themselves give more than one workgroup per CU or the context is short;
// otherwise enough runs to reach about two workgroups per CU, each at least eight 64-key tiles
// long. The case it is for: few new tokens against a long cached prefix (a prefix-cache hit, a
// short batch), where one workgroup per query tile would walk the whole context alone. Fitted to
// tools/kbench.py --what prefill_prefix (forced 1/2/4/8 runs on seven shapes): 8 x 100 new tokens
// on 4096 cached, 8q/1kv: 107 -> 39 us; 1 x 100 on 32000: 760 -> 44 us; 256 workgroups (8 x 100 on
// 4096, 32q/8kv): 109 -> 95 us with 2 runs, slower from 8; 144 workgroups of 21 tiles: no gain.
// Whole prompts (no cached prefix; --what prefill_split_full) gain the same way up to 128
// workgroups (1 x 4096, 4q/1kv: 98 -> 47 us; 2 x 4096: 100 -> 64 us) but at 256 workgroups two
// runs pay only on long prompts (1 x 16384, 4q/1kv: 381 -> 286 us; 4 x 4096, 4q/1kv: 104 -> 125 us,
// where the unsplit launch also keeps the faster phase-locked kernel), so between 128 and 256
// workgroups a launch that is not prefix-dominated splits only from 8192 keys.
static int prefill_heuristic_splits(int batch, int max_seqlen_q, int max_seqlen_k, int num_heads) {
  if (batch <= 0 || max_seqlen_q <= 0 || max_seqlen_k <= 0 || num_heads <= 0) return 1;
  const int bm = max_seqlen_q <= 1024 ? 128 : 256;
  const int64_t wgs = static_cast<int64_t>((max_seqlen_q + bm - 1) / bm) * num_heads * batch;
  const int ktiles = (max_seqlen_k + 63) / 64;
  if (wgs > 256) return 1;
  const bool prefix_dominated = max_seqlen_k >= 4 * static_cast<int64_t>(max_seqlen_q);
  if (wgs > 128 && !prefix_dominated && ktiles < 128) return 1;
  int64_t n = (512 + wgs - 1) / wgs;
  if (n > ktiles / 8) n = ktiles / 8;
  if (n > kMaxPrefillSplits) n = kMaxPrefillSplits;
  return n < 1 ? 1 : static_cast<int>(n);
}

}  // namespace lv

// Split-KV scratch of lv_decode_attn_static: a device array of the code object, so that entry
// point never allocates. 17 MiB holds the partials of every split count the library's own
// heuristic picks (batch x kv heads x splits <= 4096 workgroups, <= 8 query heads each:
// 32768 x (128 + 2) floats = 16.25 MiB). Every GPU the library runs on holds its own copy at its
// own address until the process ends. In host code the array's NAME is only a handle for
// hipGetSymbolAddress; kernels and memsets take the resolved per-device pointer.
static __device__ __align__(256) unsigned char scratch[LV_DECODE_STATIC_SCRATCH_BYTES];

namespace lv {

constexpr size_t kStaticScratchBytes = LV_DECODE_STATIC_SCRATCH_BYTES;
constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mutex;
static void* g_scratch_ptr[kMaxDevices] = {};

// Device address of `scratch` on the current device; resolved on the first call per device
// (with that device current) and kept. nullptr + recorded message on failure.
static void* static_scratch_ptr() {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    fail(LV_ERR_LAUNCH, "decode_static: hipGetDevice: %s (device %d)", hipGetErrorString(e), dev);
    return nullptr;
  }
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    void* ptr = nullptr;
    e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) {
      fail(LV_ERR_LAUNCH, "decode_static: hipGetSymbolAddress(scratch) on device %d: %s", dev,
           hipGetErrorString(e));
      return nullptr;
    }
    g_scratch_ptr[dev] = ptr;
  }
  return g_scratch_ptr[dev];
}

}  // namespace lv

using namespace lv;

extern "C" {

int lv_abi_version(void) { return LV_ABI_VERSION; }

const char* lv_error_string(int code) {
  switch (code) {
    case LV_OK: return "ok";
    case LV_ERR_INVALID_ARGUMENT: return "invalid argument";
    case LV_ERR_UNSUPPORTED: return "unsupported configuration";
    case LV_ERR_LAUNCH: return "kernel launch failed";
    case LV_ERR_WORKSPACE: return "workspace too small or misaligned";
    case -5: return "communicator failure";
    default: return "unknown error";
  }
}

const char* lv_last_error(void) { return g_last_error; }

int lv_store_kvcache(const void* key, const void* value, int64_t key_stride0,
                     int64_t value_stride0, void* k_cache, void* v_cache,
                     const int32_t* slot_mapping, int64_t num_tokens, int num_kv_heads,
                     int head_dim, int64_t num_slots, lv_stream_t stream) {
  if (num_tokens < 0 || num_kv_heads <= 0 || head_dim <= 0 || num_slots < 0)
    return fail(LV_ERR_INVALID_ARGUMENT, "store_kvcache: negative size");
  if (num_tokens == 0 || num_slots == 0) return LV_OK;
  if (!key || !value || !k_cache || !v_cache || !slot_mapping)
    return fail(LV_ERR_INVALID_ARGUMENT, "store_kvcache: null pointer");
  const int64_t row = static_cast<int64_t>(num_kv_heads) * head_dim;
  if (row % 8 != 0) return fail(LV_ERR_UNSUPPORTED, "store_kvcache: row of %lld elements is not a multiple of 8", (long long)row);
  if (row > INT32_MAX) return fail(LV_ERR_UNSUPPORTED, "store_kvcache: row too long");
  if (key_stride0 < row || value_stride0 < row)
    return fail(LV_ERR_INVALID_ARGUMENT, "store_kvcache: row stride smaller than num_kv_heads*head_dim");
  if (key_stride0 % 8 != 0 || value_stride0 % 8 != 0)
    return fail(LV_ERR_UNSUPPORTED, "store_kvcache: row strides must be multiples of 8 elements");
  if (!aligned16(key) || !aligned16(value) || !aligned16(k_cache) || !aligned16(v_cache))
    return fail(LV_ERR_UNSUPPORTED, "store_kvcache: pointers must be 16-byte aligned");
  return launch_store_kvcache(key, value, key_stride0, value_stride0, k_cache, v_cache,
                              slot_mapping, num_tokens, static_cast<int>(row), num_slots,
                              static_cast<hipStream_t>(stream));
}

int lv_qk_norm_rope_store(void* q, void* k, const void* v, int64_t q_stride0, int64_t k_stride0,
                          int64_t v_stride0, const int64_t* positions, const void* q_weight,
                          const void* k_weight, float eps, const float* cos_sin_cache,
                          int64_t max_position, void* k_cache, void* v_cache,
                          const int32_t* slot_mapping, int64_t num_slots, int64_t num_tokens,
                          int num_heads, int num_kv_heads, int head_dim, lv_stream_t stream) {
  if (num_tokens < 0 || num_heads <= 0 || num_kv_heads <= 0 || num_slots < 0 || max_position <= 0)
    return fail(LV_ERR_INVALID_ARGUMENT, "qk_norm_rope_store: negative size");
  if (head_dim != kHeadDim)
    return fail(LV_ERR_UNSUPPORTED, "qk_norm_rope_store: head_dim %d != 128", head_dim);
  if (num_tokens == 0) return LV_OK;
  if (!q || !k || !v || !positions || !q_weight || !k_weight || !cos_sin_cache)
    return fail(LV_ERR_INVALID_ARGUMENT, "qk_norm_rope_store: null pointer");
  const bool store = k_cache || v_cache || slot_mapping;
  if (store && (!k_cache || !v_cache || !slot_mapping))
    return fail(LV_ERR_INVALID_ARGUMENT,
                "qk_norm_rope_store: k_cache, v_cache and slot_mapping must be given together");
  const int64_t qrow = static_cast<int64_t>(num_heads) * head_dim;
  const int64_t krow = static_cast<int64_t>(num_kv_heads) * head_dim;
  if (q_stride0 < qrow || k_stride0 < krow || v_stride0 < krow)
    return fail(LV_ERR_INVALID_ARGUMENT, "qk_norm_rope_store: row stride smaller than heads*head_dim");
  if ((q_stride0 | k_stride0 | v_stride0) % 8 != 0)
    return fail(LV_ERR_UNSUPPORTED, "qk_norm_rope_store: row strides must be multiples of 8 elements");
  if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(q_weight) || !aligned16(k_weight) ||
      !aligned16(cos_sin_cache) || !aligned16(k_cache) || !aligned16(v_cache))
    return fail(LV_ERR_UNSUPPORTED, "qk_norm_rope_store: pointers must be 16-byte aligned");
  QkNormRopeParams p;
  p.q = static_cast<uint16_t*>(q);
  p.k = static_cast<uint16_t*>(k);
  p.v = static_cast<const uint16_t*>(v);
  p.q_stride0 = q_stride0;
  p.k_stride0 = k_stride0;
  p.v_stride0 = v_stride0;
  p.positions = positions;
  p.q_weight = static_cast<const uint16_t*>(q_weight);
  p.k_weight = static_cast<const uint16_t*>(k_weight);
  p.cos_sin = cos_sin_cache;
  p.max_position = max_position;
  p.k_cache = store && num_slots > 0 ? static_cast<uint16_t*>(k_cache) : nullptr;
  p.v_cache = store && num_slots > 0 ? static_cast<uint16_t*>(v_cache) : nullptr;
  p.slot_mapping = store && num_slots > 0 ? slot_mapping : nullptr;
  p.num_slots = num_slots;
  p.num_tokens = num_tokens;
  p.num_heads = num_heads;
  p.num_kv_heads = num_kv_heads;
  p.eps = eps;
  return launch_qk_norm_rope_store(p, static_cast<hipStream_t>(stream));
}

int lv_add_rms_norm(const void* x, void* residual, const void* weight, void* out,
                    int64_t x_stride0, int64_t residual_stride0, int64_t out_stride0,
                    int64_t num_tokens, int hidden_size, float eps, lv_stream_t stream) {
  if (num_tokens < 0 || hidden_size <= 0)
    return fail(LV_ERR_INVALID_ARGUMENT, "add_rms_norm: negative size");
  if (hidden_size % 8 != 0)
    return fail(LV_ERR_UNSUPPORTED, "add_rms_norm: hidden_size %d is not a multiple of 8", hidden_size);
  if (num_tokens == 0) return LV_OK;
  if (!x || !weight || !out) return fail(LV_ERR_INVALID_ARGUMENT, "add_rms_norm: null pointer");
  if (x_stride0 < hidden_size || out_stride0 < hidden_size || (residual && residual_stride0 < hidden_size))
    return fail(LV_ERR_INVALID_ARGUMENT, "add_rms_norm: row stride smaller than hidden_size");
  if ((x_stride0 | out_stride0 | (residual ? residual_stride0 : 0)) % 8 != 0)
    return fail(LV_ERR_UNSUPPORTED, "add_rms_norm: row strides must be multiples of 8 elements");
  if (!aligned16(x) || !aligned16(residual) || !aligned16(weight) || !aligned16(out))
    return fail(LV_ERR_UNSUPPORTED, "add_rms_norm: pointers must be 16-byte aligned");
  return launch_add_rms_norm(x, residual, weight, out, x_stride0, residual_stride0, out_stride0,
                             num_tokens, hidden_size, eps, static_cast<hipStream_t>(stream));
}

int lv_silu_and_mul(const void* x, void* out, int64_t x_stride0, int64_t out_stride0,
                    int64_t num_tokens, int intermediate_size, lv_stream_t stream) {
  if (num_tokens < 0 || intermediate_size <= 0)
    return fail(LV_ERR_INVALID_ARGUMENT, "silu_and_mul: negative size");
  if (intermediate_size % 8 != 0)
    return fail(LV_ERR_UNSUPPORTED, "silu_and_mul: intermediate_size %d is not a multiple of 8", intermediate_size);
  if (num_tokens == 0) return LV_OK;
  if (!x || !out) return fail(LV_ERR_INVALID_ARGUMENT, "silu_and_mul: null pointer");
  if (x_stride0 < 2 * static_cast<int64_t>(intermediate_size) || out_stride0 < intermediate_size)
    return fail(LV_ERR_INVALID_ARGUMENT, "silu_and_mul: row stride smaller than the row");
  if ((x_stride0 | out_stride0) % 8 != 0 || !aligned16(x) || !aligned16(out))
    return fail(LV_ERR_UNSUPPORTED, "silu_and_mul: 16-byte aligned pointers and strides %% 8 == 0 required");
  return launch_silu_and_mul(x, out, x_stride0, out_stride0, num_tokens, intermediate_size,
                             static_cast<hipStream_t>(stream));
}

int lv_prefill_attn(const void* q, const void* k, const void* v, void* o, int64_t q_stride0,
                    int64_t k_stride0, int64_t v_stride0, int64_t o_stride0,
                    const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k, int batch,
                    int max_seqlen_q, int max_seqlen_k, const int32_t* block_table,
                    int block_table_stride, int block_size, int num_heads, int num_kv_heads,
                    int head_dim, float scale, int64_t total_q, void* workspace, size_t workspace_bytes,
                    int num_splits, const int32_t* tile_order, int tile_order_len, int tile_order_rows,
                    int tile_order_lanes, int kernel_hint, lv_stream_t stream) {
  if (batch < 0 || max_seqlen_q < 0 || max_seqlen_k < 0 || num_heads <= 0 || num_kv_heads <= 0)
    return fail(LV_ERR_INVALID_ARGUMENT, "prefill: negative size");
  if (head_dim != kHeadDim) return fail(LV_ERR_UNSUPPORTED, "prefill: head_dim %d != 128", head_dim);
  if (num_heads % num_kv_heads != 0)
    return fail(LV_ERR_INVALID_ARGUMENT, "prefill: num_heads %% num_kv_heads != 0");
  if (!isfinite(scale) || scale <= 0.f)
    return fail(LV_ERR_INVALID_ARGUMENT, "prefill: scale %g must be finite and > 0", (double)scale);
  if (batch == 0 || max_seqlen_q == 0) return LV_OK;
  if (!q || !k || !v || !o || !cu_seqlens_q || !cu_seqlens_k)
    return fail(LV_ERR_INVALID_ARGUMENT, "prefill: null pointer");
  const int64_t qrow = static_cast<int64_t>(num_heads) * head_dim;
  const int64_t krow = static_cast<int64_t>(num_kv_heads) * head_dim;
  if (q_stride0 < qrow || o_stride0 < qrow || k_stride0 < krow || v_stride0 < krow)
    return fail(LV_ERR_INVALID_ARGUMENT, "prefill: row stride smaller than heads*head_dim");
  if ((q_stride0 | k_stride0 | v_stride0 | o_stride0) % 8 != 0)
    return fail(LV_ERR_UNSUPPORTED, "prefill: row strides must be multiples of 8 elements");
  if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(o))
    return fail(LV_ERR_UNSUPPORTED, "prefill: pointers must be 16-byte aligned");
  if (block_table) {
    if (block_size <= 0 || block_size % 64 != 0)
      return fail(LV_ERR_UNSUPPORTED, "prefill: paged block_size %d must be a positive multiple of 64", block_size);
    if (block_table_stride <= 0) return fail(LV_ERR_INVALID_ARGUMENT, "prefill: block_table_stride");
    if (k_stride0 != krow || v_stride0 != krow)
      return fail(LV_ERR_INVALID_ARGUMENT, "prefill: paged caches must be contiguous");
  }
  PrefillParams p;
  p.q = static_cast<const uint16_t*>(q);
  p.k = static_cast<const uint16_t*>(k);
  p.v = static_cast<const uint16_t*>(v);
  p.o = static_cast<uint16_t*>(o);
  p.q_stride0 = q_stride0;
  p.k_stride0 = k_stride0;
  p.v_stride0 = v_stride0;
  p.o_stride0 = o_stride0;
  p.cu_q = cu_seqlens_q;
  p.cu_k = cu_seqlens_k;
  p.block_table = block_table;
  p.bt_stride = block_table_stride;
  p.block_size = block_size;
  p.batch = batch;
  p.max_seqlen_q = max_seqlen_q;
  p.num_heads = num_heads;
  p.num_kv_heads = num_kv_heads;
  p.kernel_hint = kernel_hint;
  if (num_splits < 0 || num_splits > kMaxPrefillSplits)
    return fail(LV_ERR_INVALID_ARGUMENT, "prefill: num_splits %d not in 0..%d", num_splits, kMaxPrefillSplits);
  int nsplit = num_splits == 0 ? prefill_heuristic_splits(batch, max_seqlen_q, max_seqlen_k, num_heads) : num_splits;
  // the automatic choice falls back to one pass when the caller brought no workspace for it (or
  // did not say how many rows q has)
  if (num_splits == 0 && nsplit > 1 &&
      (total_q <= 0 || !workspace ||
       workspace_bytes < lv_prefill_workspace_bytes(total_q, num_heads, head_dim, nsplit)))
    nsplit = 1;
  p.nsplit = nsplit;
  p.ws_o = nullptr;
  p.ws_ml = nullptr;
  p.total_q = total_q;
  if (nsplit > 1) {
    if (total_q <= 0) return fail(LV_ERR_INVALID_ARGUMENT, "prefill: total_q must be given with num_splits > 1");
    const size_t need = lv_prefill_workspace_bytes(total_q, num_heads, head_dim, nsplit);
    if (!workspace || workspace_bytes < need || !aligned16(workspace))
      return fail(LV_ERR_WORKSPACE, "prefill: workspace %zu B < %zu B needed for %d splits (or misaligned)",
                  workspace_bytes, need, nsplit);
    p.ws_o = static_cast<float*>(workspace);
    p.ws_ml = p.ws_o + static_cast<size_t>(total_q) * num_heads * nsplit * kHeadDim;
  }
  p.scale_log2e = scale * kLog2e;
  p.tile_order = nullptr;
  p.order_cols = 1;
  p.order_len = 0;
  p.order_rows = 0;
  p.order_lanes = 0;
  if (tile_order) {
    const int cols = prefill_order_columns(num_kv_heads);
    if (tile_order_lanes < 0) return fail(LV_ERR_INVALID_ARGUMENT, "prefill: tile_order_lanes %d", tile_order_lanes);
    if (tile_order_lanes > 0) {
      // an item table for persistent workgroups: 8 ints per entry, [depth][lanes][columns]
      if (tile_order_lanes != prefill_order_lanes(num_heads, num_kv_heads))
        return fail(LV_ERR_INVALID_ARGUMENT, "prefill: tile_order_lanes %d, this launch needs %d (lv_prefill_order_lanes)",
                    tile_order_lanes, prefill_order_lanes(num_heads, num_kv_heads));
      if (tile_order_len <= 0 || tile_order_len % (8 * cols * tile_order_lanes) != 0)
        return fail(LV_ERR_INVALID_ARGUMENT, "prefill: an item table of %d ints is not a multiple of 8 x %d columns x %d lanes",
                    tile_order_len, cols, tile_order_lanes);
      if (nsplit != 1) return fail(LV_ERR_INVALID_ARGUMENT, "prefill: an item table serves one KV pass (num_splits 1)");
      if ((reinterpret_cast<uintptr_t>(tile_order) & 31u) != 0)
        return fail(LV_ERR_UNSUPPORTED, "prefill: an item table must be 32-byte aligned");
    } else if (tile_order_len <= 0 || tile_order_len % cols != 0) {
      return fail(LV_ERR_INVALID_ARGUMENT, "prefill: tile_order_len %d must be a positive multiple of %d "
                  "(lv_prefill_order_columns)", tile_order_len, cols);
    }
    p.tile_order = tile_order;
    p.order_cols = cols;
    p.order_len = tile_order_len;
    p.order_rows = tile_order_rows;
    p.order_lanes = tile_order_lanes;
  }
  return launch_prefill_attn(p, static_cast<hipStream_t>(stream));
}

int lv_prefill_tile_rows(int max_seqlen_q, int num_splits, int kernel_hint) {
  return prefill_tile_rows(max_seqlen_q, num_splits, kernel_hint);
}

int lv_prefill_order_columns(int num_kv_heads) { return prefill_order_columns(num_kv_heads); }

int lv_prefill_order_lanes(int num_heads, int num_kv_heads) {
  if (num_heads <= 0 || num_kv_heads <= 0) return 0;
  return prefill_order_lanes(num_heads, num_kv_heads);
}

int lv_prefill_num_splits(int batch, int max_seqlen_q, int max_seqlen_k, int num_heads) {
  return prefill_heuristic_splits(batch, max_seqlen_q, max_seqlen_k, num_heads);
}

size_t lv_prefill_workspace_bytes(int64_t total_q, int num_heads, int head_dim, int num_splits) {
  if (total_q <= 0 || num_heads <= 0 || head_dim != kHeadDim || num_splits <= 1) return 0;
  return static_cast<size_t>(total_q) * num_heads * num_splits * (kHeadDim + 2) * sizeof(float);
}

size_t lv_prefill_max_workspace_bytes(int64_t max_total_q, int num_heads, int head_dim) {
  if (max_total_q <= 0 || num_heads <= 0 || head_dim != kHeadDim) return 0;
  // prefill_heuristic_splits cuts only launches of at most 256 workgroups of bm = 128 or 256 rows
  // (rows x heads <= workgroups x bm) into n <= 512 / workgroups + 1 runs, so
  // rows x heads x n <= bm x (512 + workgroups) <= 256 x 768; and n <= kMaxPrefillSplits.
  int64_t units = max_total_q * num_heads * kMaxPrefillSplits;
  if (units > 256 * 768) units = 256 * 768;
  return static_cast<size_t>(units) * (kHeadDim + 2) * sizeof(float);
}

int lv_decode_num_splits(int batch, int num_kv_heads, int max_blocks_per_seq, int block_size) {
  return heuristic_splits(batch, num_kv_heads, max_blocks_per_seq, block_size > 0 ? block_size : 256);
}

size_t lv_decode_workspace_bytes(int batch, int num_heads, int head_dim, int num_splits) {
  if (batch <= 0 || num_heads <= 0 || head_dim != kHeadDim || num_splits <= 1) return 0;
  return static_cast<size_t>(batch) * num_heads * num_splits * (kHeadDim + 2) * sizeof(float);
}

}  // extern "C"

// Shared body of lv_decode_attn and lv_decode_attn_static. `static_scratch`: the workspace is the
// library's own device array, resolved (once per device) only when the call needs splits.
static int decode_entry(const void* q, const void* k_cache, const void* v_cache, void* o,
                        int64_t q_stride0, int64_t o_stride0, const int32_t* context_lens,
                        const int32_t* block_table, int block_table_stride, int batch,
                        int max_blocks_per_seq, int block_size, int num_heads, int num_kv_heads,
                        int head_dim, float scale, void* workspace, size_t workspace_bytes,
                        bool static_scratch, int num_splits, const int32_t* seq_order,
                        const int32_t* work_items, int num_work_items, const int32_t* seq_splits,
                        int kernel_hint, lv_stream_t stream) {
  if (batch < 0 || max_blocks_per_seq < 0 || num_heads <= 0 || num_kv_heads <= 0 || num_splits < 0)
    return fail(LV_ERR_INVALID_ARGUMENT, "decode: negative size");
  if (head_dim != kHeadDim) return fail(LV_ERR_UNSUPPORTED, "decode: head_dim %d != 128", head_dim);
  if (num_heads % num_kv_heads != 0)
    return fail(LV_ERR_INVALID_ARGUMENT, "decode: num_heads %% num_kv_heads != 0");
  if (block_size <= 0 || block_size % 32 != 0)
    return fail(LV_ERR_UNSUPPORTED, "decode: block_size %d must be a positive multiple of 32", block_size);
  if (!isfinite(scale) || scale <= 0.f)
    return fail(LV_ERR_INVALID_ARGUMENT, "decode: scale %g must be finite and > 0", (double)scale);
  if (kernel_hint != LV_DECODE_KERNEL_AUTO && kernel_hint != LV_DECODE_KERNEL_VALU &&
      kernel_hint != LV_DECODE_KERNEL_MFMA)
    return fail(LV_ERR_INVALID_ARGUMENT, "decode: kernel_hint %d", kernel_hint);
  {
    const int group = num_heads / num_kv_heads;
    if (kernel_hint == LV_DECODE_KERNEL_VALU && group != 1 && group != 2 && group != 4 && group != 8)
      return fail(LV_ERR_UNSUPPORTED, "decode: the vector-ALU kernel serves group sizes 1, 2, 4, 8, not %d", group);
  }
  if (batch == 0) return LV_OK;
  if (!q || !k_cache || !v_cache || !o || !context_lens || !block_table)
    return fail(LV_ERR_INVALID_ARGUMENT, "decode: null pointer");
  if (block_table_stride < max_blocks_per_seq)
    return fail(LV_ERR_INVALID_ARGUMENT, "decode: block_table_stride < max_blocks_per_seq");
  const int64_t qrow = static_cast<int64_t>(num_heads) * head_dim;
  if (q_stride0 < qrow || o_stride0 < qrow)
    return fail(LV_ERR_INVALID_ARGUMENT, "decode: row stride smaller than num_heads*head_dim");
  if (q_stride0 % 8 != 0 || o_stride0 % 8 != 0 || !aligned16(q) || !aligned16(o) ||
      !aligned16(k_cache) || !aligned16(v_cache))
    return fail(LV_ERR_UNSUPPORTED, "decode: q/o/caches must be 16-byte aligned with strides %% 8 == 0");
  int nsplit = num_splits == 0 ? heuristic_splits(batch, num_kv_heads, max_blocks_per_seq, block_size) : num_splits;
  if (nsplit > kMaxSplits) return fail(LV_ERR_INVALID_ARGUMENT, "decode: num_splits %d > %d", nsplit, kMaxSplits);
  if (work_items) {
    // a work list: num_splits is the split capacity of a workspace row, the grid is the list's capacity
    if (num_splits < 1) return fail(LV_ERR_INVALID_ARGUMENT, "decode: a work list needs max_splits >= 1");
    if (num_work_items < 1) return fail(LV_ERR_INVALID_ARGUMENT, "decode: a work list needs num_work_items >= 1");
    if (!seq_splits) return fail(LV_ERR_INVALID_ARGUMENT, "decode: a work list needs seq_splits");
    if (static_cast<int64_t>(num_work_items) * num_kv_heads > INT32_MAX)
      return fail(LV_ERR_UNSUPPORTED, "decode: grid too large");
  } else if (static_cast<int64_t>(batch) * num_kv_heads * nsplit > INT32_MAX) {
    return fail(LV_ERR_UNSUPPORTED, "decode: grid too large");
  }
  DecodeParams p;
  p.q = static_cast<const uint16_t*>(q);
  p.k_cache = static_cast<const uint16_t*>(k_cache);
  p.v_cache = static_cast<const uint16_t*>(v_cache);
  p.o = static_cast<uint16_t*>(o);
  p.ws_o = nullptr;
  p.ws_ml = nullptr;
  if (nsplit > 1) {
    const size_t need = lv_decode_workspace_bytes(batch, num_heads, head_dim, nsplit);
    if (static_scratch) {
      if (need > kStaticScratchBytes)
        return fail(LV_ERR_WORKSPACE,
                    "decode_static: batch %d x %d heads x %d splits needs %zu B of scratch, the built-in "
                    "array holds %zu B; use lv_decode_attn with a caller workspace",
                    batch, num_heads, nsplit, need, kStaticScratchBytes);
      workspace = static_scratch_ptr();
      if (!workspace) return LV_ERR_LAUNCH;   // message recorded by static_scratch_ptr
      workspace_bytes = kStaticScratchBytes;
    }
    if (!workspace || workspace_bytes < need || !aligned16(workspace))
      return fail(LV_ERR_WORKSPACE, "decode: workspace %zu B < %zu B needed for %d splits (or misaligned)",
                  workspace_bytes, need, nsplit);
    p.ws_o = static_cast<float*>(workspace);
    p.ws_ml = p.ws_o + static_cast<size_t>(batch) * num_heads * nsplit * kHeadDim;
  }
  p.context_lens = context_lens;
  p.block_table = block_table;
  p.seq_order = seq_order;
  p.work_items = work_items;
  p.seq_splits = work_items ? seq_splits : nullptr;
  p.num_work_items = work_items ? num_work_items : 0;
  p.q_stride0 = q_stride0;
  p.o_stride0 = o_stride0;
  p.bt_stride = block_table_stride;
  p.batch = batch;
  p.block_size = block_size;
  p.num_heads = num_heads;
  p.num_kv_heads = num_kv_heads;
  p.nsplit = nsplit;
  p.kernel_hint = kernel_hint;
  p.scale_log2e = scale * kLog2e;
  return launch_decode_attn(p, static_cast<hipStream_t>(stream));
}

extern "C" {

int lv_decode_attn(const void* q, const void* k_cache, const void* v_cache, void* o,
                   int64_t q_stride0, int64_t o_stride0, const int32_t* context_lens,
                   const int32_t* block_table, int block_table_stride, int batch,
                   int max_blocks_per_seq, int block_size, int num_heads, int num_kv_heads,
                   int head_dim, float scale, void* workspace, size_t workspace_bytes,
                   int num_splits, const int32_t* seq_order, int kernel_hint, lv_stream_t stream) {
  return decode_entry(q, k_cache, v_cache, o, q_stride0, o_stride0, context_lens, block_table,
                      block_table_stride, batch, max_blocks_per_seq, block_size, num_heads,
                      num_kv_heads, head_dim, scale, workspace, workspace_bytes, false, num_splits,
                      seq_order, nullptr, 0, nullptr, kernel_hint, stream);
}

int lv_decode_attn_static(const void* q, const void* k_cache, const void* v_cache, void* o,
                          int64_t q_stride0, int64_t o_stride0, const int32_t* context_lens,
                          const int32_t* block_table, int block_table_stride, int batch,
                          int max_blocks_per_seq, int block_size, int num_heads, int num_kv_heads,
                          int head_dim, float scale, int num_splits, const int32_t* seq_order,
                          int kernel_hint, lv_stream_t stream) {
  return decode_entry(q, k_cache, v_cache, o, q_stride0, o_stride0, context_lens, block_table,
                      block_table_stride, batch, max_blocks_per_seq, block_size, num_heads,
                      num_kv_heads, head_dim, scale, nullptr, 0, true, num_splits, seq_order,
                      nullptr, 0, nullptr, kernel_hint, stream);
}

int lv_decode_attn_planned(const void* q, const void* k_cache, const void* v_cache, void* o,
                           int64_t q_stride0, int64_t o_stride0, const int32_t* context_lens,
                           const int32_t* block_table, int block_table_stride, int batch,
                           int max_blocks_per_seq, int block_size, int num_heads, int num_kv_heads,
                           int head_dim, float scale, void* workspace, size_t workspace_bytes,
                           int max_splits, const int32_t* work_items, int num_work_items,
                           const int32_t* seq_splits, int kernel_hint, lv_stream_t stream) {
  if (!work_items) return fail(LV_ERR_INVALID_ARGUMENT, "decode_planned: work_items is null");
  return decode_entry(q, k_cache, v_cache, o, q_stride0, o_stride0, context_lens, block_table,
                      block_table_stride, batch, max_blocks_per_seq, block_size, num_heads,
                      num_kv_heads, head_dim, scale, workspace, workspace_bytes, false, max_splits,
                      nullptr, work_items, num_work_items, seq_splits, kernel_hint, stream);
}

}  // extern "C"
