// Launcher interfaces between the C-ABI layer (api.hip) and the kernel translation units.
#pragma once

#include "lv_common.h"

namespace lv {

struct DecodeParams {
  const uint16_t* q;
  const uint16_t* k_cache;
  const uint16_t* v_cache;
  uint16_t* o;
  float* ws_o;   // [batch, Hq, nsplit, 128] unnormalised partial outputs
  float* ws_ml;  // [batch, Hq, nsplit, 2]   (running max in log2 units, sum)
  const int32_t* context_lens;
  const int32_t* block_table;
  const int32_t* seq_order;  // optional launch order of the sequences (nullptr = identity)
  const int32_t* work_items; // optional work list {used, group size, (sequence, split | splits << 16) ...}
  const int32_t* seq_splits; // with a work list: splits of every sequence [batch]
  int num_work_items;        // capacity of the list in items = workgroups per kv head of the launch
  int64_t q_stride0, o_stride0;
  int bt_stride, batch, block_size, num_heads, num_kv_heads, nsplit;
  int kernel_hint;           // LV_DECODE_KERNEL_*
  float scale_log2e;
};

struct PrefillParams {
  const uint16_t* q;
  const uint16_t* k;
  const uint16_t* v;
  uint16_t* o;
  int64_t q_stride0, k_stride0, v_stride0, o_stride0;
  const int32_t* cu_q;
  const int32_t* cu_k;
  const int32_t* block_table;  // nullptr = packed K/V
  int bt_stride, block_size, batch, max_seqlen_q, num_heads, num_kv_heads;
  int kernel_hint;             // LV_PREFILL_KERNEL_* (| LV_PREFILL_STAGE_DMA)
  int nsplit;                  // KV splits per query tile (1: none); > 1 needs the workspace below
  float* ws_o;                 // [total_q, Hq, nsplit, 128] unnormalised partial outputs
  float* ws_ml;                // [total_q, Hq, nsplit, 2]   (row maximum in log2 units, row sum)
  int64_t total_q;
  float scale_log2e;
  const int32_t* tile_order;   // optional [order_len / order_cols][order_cols] table of b * qtiles + t (< 0: empty)
  int order_cols;              // columns of the table: 8 / num_kv_heads for 1, 2, 4 kv heads, else 1
  int64_t order_len;
  int order_rows;              // query rows per tile the table was built for (must match the kernel's)
  int order_lanes;             // 0: tile_order holds codes b * qtiles + t. > 0: it holds 8-int entries
                               // {q row, valid rows, causal limit of the first row, packed K row or sequence,
                               // keys, 0, 0, 0}, [depth][lanes][order_cols]: the lists persistent workgroups walk
};
// Lanes of an item table: workgroups per (query head, table column) that one launch keeps resident.
int prefill_order_lanes(int num_heads, int num_kv_heads);

// Query rows per workgroup tile of the prefill kernel a launch with these arguments runs.
int prefill_tile_rows(int max_seqlen_q, int num_splits, int kernel_hint);
// Columns of a tile_order table for this many kv heads.
int prefill_order_columns(int num_kv_heads);

struct QkNormRopeParams {
  uint16_t* q;              // [N, Hq, 128], updated in place
  uint16_t* k;              // [N, Hkv, 128], updated in place
  const uint16_t* v;        // [N, Hkv, 128]
  int64_t q_stride0, k_stride0, v_stride0;
  const int64_t* positions; // [N]
  const uint16_t* q_weight; // [128]
  const uint16_t* k_weight; // [128]
  const float* cos_sin;     // [max_position, 128]: cos (64) | sin (64)
  int64_t max_position;
  uint16_t* k_cache;        // nullable together with v_cache / slot_mapping: no store
  uint16_t* v_cache;
  const int32_t* slot_mapping;
  int64_t num_slots, num_tokens;
  int num_heads, num_kv_heads;
  float eps;
};

int launch_store_kvcache(const void* key, const void* value, int64_t key_stride0,
                         int64_t value_stride0, void* k_cache, void* v_cache,
                         const int32_t* slot_mapping, int64_t num_tokens, int row_elems,
                         int64_t num_slots, hipStream_t stream);
int launch_qk_norm_rope_store(const QkNormRopeParams& p, hipStream_t stream);
int launch_add_rms_norm(const void* x, void* residual, const void* weight, void* out,
                        int64_t x_stride0, int64_t res_stride0, int64_t out_stride0,
                        int64_t num_tokens, int hidden, float eps, hipStream_t stream);
int launch_silu_and_mul(const void* x, void* out, int64_t x_stride0, int64_t out_stride0,
                        int64_t num_tokens, int intermediate, hipStream_t stream);
int launch_decode_attn(const DecodeParams& p, hipStream_t stream);
int launch_prefill_attn(const PrefillParams& p, hipStream_t stream);

}  // namespace lv
