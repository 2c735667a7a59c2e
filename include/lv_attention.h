/*
 * lv_attention.h — C ABI of the MI355X (gfx950) paged-attention library, liblvattn.so.
 *
 * The three compute entry points replace the three device calls that light-vllm's
 * Attention.forward makes (reference: lightvllm/layers/attention.py):
 *
 *   lv_store_kvcache   <- store_kvcache_kernel + store_kvcache      attention.py:9-52, 54-108
 *   lv_prefill_attn    <- flash_attn_varlen_func(... causal, block_table)   attention.py:159-165
 *   lv_decode_attn     <- flash_attn_with_kvcache(... cache_seqlens, block_table)   attention.py:172-175
 *
 * Conventions
 *   - Every pointer except `workspace` sizes is a DEVICE pointer on the current HIP device.
 *   - All tensors are bf16 (2 bytes per element) unless stated; index tensors are int32.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - Every call only enqueues work on `stream`: no host synchronisation, no allocation,
 *     safe to capture into a hipGraph.
 *   - Return value: LV_OK (0) or a negative LV_ERR_* code; lv_last_error() returns a
 *     thread-local human-readable message for the last failing call.
 *   - head_dim must be 128 (every Qwen3 size). Other values return LV_ERR_UNSUPPORTED.
 *   - Strides are in ELEMENTS (bf16), as torch's Tensor.stride() reports them.
 */
#ifndef LV_ATTENTION_H
#define LV_ATTENTION_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LV_OK 0
#define LV_ERR_INVALID_ARGUMENT (-1)
#define LV_ERR_UNSUPPORTED (-2)
#define LV_ERR_LAUNCH (-3)
#define LV_ERR_WORKSPACE (-4)

#define LV_ABI_VERSION 9

/* kernel_hint of lv_decode_attn: which of the two decode kernels serves the call. */
#define LV_DECODE_KERNEL_AUTO 0   /* by num_heads / num_kv_heads: vector-ALU for 1-2, matrix-core for 3-8 */
#define LV_DECODE_KERNEL_VALU 1   /* vector-ALU kernel (group sizes 1, 2, 4, 8 only) */
#define LV_DECODE_KERNEL_MFMA 2   /* matrix-core kernel (every group size 1..8) */

/* kernel_hint of lv_prefill_attn: which prefill kernel serves the call. */
#define LV_PREFILL_KERNEL_AUTO 0  /* by max_seqlen_q: W4 up to 1024, W8 below 2048, PP from 2048 */
#define LV_PREFILL_KERNEL_W4 1    /* 4 waves x 32 query rows, two workgroups per CU */
#define LV_PREFILL_KERNEL_W8 2    /* 8 waves x 32 query rows */
#define LV_PREFILL_KERNEL_PP 3    /* 8 waves in two phase-locked groups (long prompts) */
#define LV_PREFILL_KERNEL_W64 5   /* 4 waves x 64 query rows, one wave per SIMD */
#define LV_PREFILL_STAGE_DMA 0x100 /* or-ed to W4 / W8: stage K/V tiles with direct-to-LDS loads */

typedef void* lv_stream_t;

/* ABI version of the loaded library (LV_ABI_VERSION at build time). */
int lv_abi_version(void);

/* Static description of an LV_ERR_* code. Never NULL. */
const char* lv_error_string(int code);

/* Message recorded by the last failing call on this thread ("" if none). */
const char* lv_last_error(void);

/*
 * Scatter the K/V rows of this step's new tokens into the paged cache.
 * Reference: store_kvcache (attention.py:54-108) launching store_kvcache_kernel (attention.py:9-52).
 *
 *   for i in [0, num_tokens):  s = slot_mapping[i]
 *       k_cache[s, :] = key[i, :]      (row = num_kv_heads * head_dim contiguous elements)
 *       v_cache[s, :] = value[i, :]
 *
 *   key, value       [num_tokens, num_kv_heads, head_dim]; last two dims contiguous,
 *                    row strides key_stride0 / value_stride0 free (value is normally a view
 *                    into the fused qkv activation; models/qwen3.py:95). 16-byte aligned rows.
 *   k_cache, v_cache [num_slots, num_kv_heads * head_dim] contiguous
 *                    (= [num_blocks, block_size, num_kv_heads, head_dim]; attention.py:96).
 *   slot_mapping     int32 [num_tokens]. Entries < 0 or >= num_slots are SKIPPED (the reference
 *                    kernel has no such guard; -1 is the padding value hipGraph replay uses).
 * Bit-exact copy; offsets are 64-bit (slot * row exceeds 2^31 elements on a 288 GB cache).
 * Duplicate slots inside one call are last-writer-wins in unspecified order.
 */
int lv_store_kvcache(const void* key, const void* value,
                     int64_t key_stride0, int64_t value_stride0,
                     void* k_cache, void* v_cache,
                     const int32_t* slot_mapping,
                     int64_t num_tokens, int num_kv_heads, int head_dim,
                     int64_t num_slots, lv_stream_t stream);

/*
 * The step in front of the attention call, fused with the cache store: per-head RMSNorm of q and
 * k, neox rotary embedding of both, then the K/V scatter of lv_store_kvcache.
 * Reference: Qwen3Attention.forward, models/qwen3.py:100-113 (q_norm / k_norm :100-106,
 * rotary_emb :109, store inside self.attn :113 -> attention.py:149-150); arithmetic and rounding
 * points of layers/layernorm.py:39-50 and layers/rotary_embedding.py:33-42:
 *
 *   for every head x of q (weight q_weight) and of k (weight k_weight), in fp32:
 *       n = bf16( bf16( x * rsqrt(mean(x^2) + eps) ) * weight )
 *       y[0:64]   = bf16( n[0:64] * cos - n[64:128] * sin )
 *       y[64:128] = bf16( n[64:128] * cos + n[0:64] * sin )      cos, sin = cos_sin_cache[positions[i]]
 *   q and k are overwritten with y; k_cache[slot_mapping[i]] = k[i] (after rotation),
 *   v_cache[slot_mapping[i]] = value[i].
 *
 *   q [num_tokens, num_heads, 128], k, v [num_tokens, num_kv_heads, 128]: row strides free (all
 *   three are normally views into one fused qkv activation), last two dims contiguous.
 *   positions      int64 [num_tokens]; clamped into [0, max_position).
 *   q_weight, k_weight  bf16 [128].
 *   cos_sin_cache  fp32 [max_position, 128]: 64 cosines then 64 sines per position
 *                  (rotary_embedding.py:76-91).
 *   k_cache, v_cache, slot_mapping, num_slots: as lv_store_kvcache; pass all three pointers NULL
 *                  to skip the store (the reference skips it while the caches are unallocated,
 *                  attention.py:149).
 * Products and sums are evaluated unfused, in the reference's order; only the summation order of
 * the mean of squares is the library's own.
 */
int lv_qk_norm_rope_store(void* q, void* k, const void* v,
                          int64_t q_stride0, int64_t k_stride0, int64_t v_stride0,
                          const int64_t* positions, const void* q_weight, const void* k_weight,
                          float eps, const float* cos_sin_cache, int64_t max_position,
                          void* k_cache, void* v_cache, const int32_t* slot_mapping,
                          int64_t num_slots, int64_t num_tokens,
                          int num_heads, int num_kv_heads, int head_dim, lv_stream_t stream);

/*
 * RMSNorm with optional residual add, the normalisation either side of the attention block.
 * Reference: RMSNorm.rms_forward / add_rms_forward, layers/layernorm.py:28-80.
 *
 *   residual == NULL:  out = bf16( bf16( x * rsqrt(mean(x^2) + eps) ) * weight )
 *   residual != NULL:  s = fp32(x) + fp32(residual);  residual = bf16(s)   (updated in place)
 *                      out = bf16( bf16( s * rsqrt(mean(s^2) + eps) ) * weight )
 *
 *   x, out [num_tokens, hidden_size] (out may alias x), residual [num_tokens, hidden_size],
 *   weight bf16 [hidden_size]; hidden_size a multiple of 8, at most 8192.
 */
int lv_add_rms_norm(const void* x, void* residual, const void* weight, void* out,
                    int64_t x_stride0, int64_t residual_stride0, int64_t out_stride0,
                    int64_t num_tokens, int hidden_size, float eps, lv_stream_t stream);

/*
 * out = bf16( bf16(silu(gate)) * up ) with gate = x[:, :intermediate_size], up = x[:, intermediate_size:].
 * Reference: silu_and_mul_native, layers/activation.py:7-29.
 *   x [num_tokens, 2 * intermediate_size], out [num_tokens, intermediate_size].
 */
int lv_silu_and_mul(const void* x, void* out, int64_t x_stride0, int64_t out_stride0,
                    int64_t num_tokens, int intermediate_size, lv_stream_t stream);

/*
 * Variable-length causal self-attention over a packed batch (prefill).
 * Reference call site: flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q,
 * max_seqlen_k, softmax_scale, causal=True, block_table) — attention.py:159-165.
 *
 *   q   [total_q, num_heads, head_dim]      row stride q_stride0
 *   o   [total_q, num_heads, head_dim]      row stride o_stride0
 *   cu_seqlens_q / cu_seqlens_k  int32 [batch + 1] (prefix sums of Lq_b / Lk_b)
 *
 *   Packed mode (block_table == NULL): k, v are [total_k, num_kv_heads, head_dim] with row
 *     strides k_stride0 / v_stride0; sequence b owns rows cu_seqlens_k[b] .. cu_seqlens_k[b+1].
 *   Paged mode (block_table != NULL): k, v are the paged caches [num_blocks, block_size,
 *     num_kv_heads, head_dim] contiguous (k_stride0 == v_stride0 == num_kv_heads * head_dim);
 *     key t of sequence b lives at row block_table[b * block_table_stride + t / block_size]
 *     * block_size + t % block_size. Entries past ceil(Lk_b / block_size) are never read
 *     (the reference pads them with -1; model_runner.py:266-267).
 *
 * Semantics: per sequence b and query head h (kv head h / (num_heads / num_kv_heads)),
 *   O = softmax(scale * Q K^T + causal) V with the causal mask bottom-right aligned:
 *   query row r (0-based in the sequence's Lq_b rows) sees keys t <= r + (Lk_b - Lq_b).
 *   bf16 inputs, fp32 scores / softmax statistics / accumulation, bf16 output.
 *   A query row that sees no key (Lk_b < Lq_b rows) is written as zeros.
 * max_seqlen_q bounds the grid (rows beyond a sequence's Lq_b exit early); max_seqlen_k is
 * accepted for signature parity and only validated. scale must be finite and > 0.
 * kernel_hint: LV_PREFILL_KERNEL_AUTO, or one of the LV_PREFILL_KERNEL_* values to force that
 * kernel (every one of them computes the same function; the choice is a performance matter and
 * what the parity tests iterate over). No counterpart in the reference call.
 * num_splits / workspace / total_q: KV splits of every query tile (flash-decoding for prefill): with
 *   few query rows against a long context (a prefix-cache hit, a short batch) one workgroup per
 *   query tile would walk the whole context alone; n > 1 cuts each tile's keys into n runs, whose
 *   fp32 partial outputs go through `workspace` (>= lv_prefill_workspace_bytes, 16-byte aligned) to a
 *   merge pass. 0 = library heuristic (lv_prefill_num_splits; it stays at one pass when no
 *   sufficient workspace is given), 1 = one pass, up to 32. total_q = rows of q (needed with
 *   num_splits != 1). Results agree with the one-pass kernels to bf16 rounding (different
 *   summation order). The workspace rule of lv_decode_attn applies: one stream per workspace.
 *   Rows of q at or past cu_seqlens_q[batch] (a padded q) are left untouched by every path.
 * tile_order / tile_order_len / tile_order_rows: optional launch order of the query tiles, built by a
 *   caller that has the sequence lengths on the host (NULL = the library's own mapping from the
 *   static shape; results never depend on it). A query tile is tile_order_rows consecutive query
 *   rows of one sequence, tile t of a sequence starting at row t * tile_order_rows;
 *   tile_order_rows must equal lv_prefill_tile_rows(max_seqlen_q, num_splits, kernel_hint). The
 *   table is int32 [tile_order_len / C][C] with C = lv_prefill_order_columns(num_kv_heads); an entry
 *   is b * ceil(max_seqlen_q / tile_order_rows) + t, or negative for an empty place. Every tile of
 *   every sequence must appear exactly once (a tile that is left out is not computed). Column c's
 *   entries run, for kv head h, on XCD (h + num_kv_heads * c) % 8, in table order from row 0: with
 *   1, 2 or 4 kv heads per rank the library's own mapping spreads each sequence's K/V stream over
 *   all XCDs, and a table that keeps a sequence's tiles in one or two columns lets an XCD's L2 fetch
 *   the stream once; rows should go heaviest (most visible keys) first. No counterpart in the
 *   reference call; light_vllm_amd/prefill_order.py builds such a table.
 * tile_order_lanes: 0 for the table above. > 0 (= lv_prefill_order_lanes(num_heads, num_kv_heads)):
 *   tile_order is an ITEM table for the long-prompt kernel (max_seqlen_q >= 2048 or
 *   LV_PREFILL_KERNEL_PP; one KV pass): the launch keeps one workgroup per compute unit resident and
 *   each walks a list of query tiles instead of ending after one, the next tile's Q rows and first
 *   K/V tiles arriving while the current one is computed. Layout int32 [depth][lanes][C][8]
 *   (tile_order_len counts ints; 32-byte aligned); workgroup (query head, column c, lane r) serves
 *   entries [0][r][c], [1][r][c], ... up to the first whose second word is <= 0. An entry is
 *   {first row of the tile in q / o, its valid rows (1 .. tile_order_rows), last key visible to its first
 *   row (row i of the tile sees keys <= that + i: bottom-right aligned causal mask), first packed K/V
 *   row of its sequence (paged: the sequence's index into block_table), keys of its sequence, 0, 0, 0},
 *   i.e. what cu_seqlens_q / cu_seqlens_k say about the tile, which the kernel then does not read:
 *   unlike the code table, a wrong item table gives wrong results (it is trusted like cu_seqlens).
 *   The host balances the lists (prefill_order.build_item_table: longest-processing-time-first).
 */
int lv_prefill_attn(const void* q, const void* k, const void* v, void* o,
                    int64_t q_stride0, int64_t k_stride0, int64_t v_stride0, int64_t o_stride0,
                    const int32_t* cu_seqlens_q, const int32_t* cu_seqlens_k,
                    int batch, int max_seqlen_q, int max_seqlen_k,
                    const int32_t* block_table, int block_table_stride, int block_size,
                    int num_heads, int num_kv_heads, int head_dim, float scale,
                    int64_t total_q, void* workspace, size_t workspace_bytes, int num_splits,
                    const int32_t* tile_order, int tile_order_len, int tile_order_rows,
                    int tile_order_lanes, int kernel_hint, lv_stream_t stream);

/* Query rows per tile of the kernel lv_prefill_attn runs for these arguments (128 or 256; 0 for an
 * unknown kernel_hint). num_splits as resolved (lv_prefill_num_splits for 0). */
int lv_prefill_tile_rows(int max_seqlen_q, int num_splits, int kernel_hint);

/* Columns of a tile_order table: 8 / num_kv_heads for 1, 2 or 4 kv heads, else 1. */
int lv_prefill_order_columns(int num_kv_heads);

/* Lanes of an item table (tile_order_lanes): lists per (query head, column), so that the launch is one
 * workgroup per compute unit: 256 / (num_heads * columns), at least 1. */
int lv_prefill_order_lanes(int num_heads, int num_kv_heads);

/* Split count lv_prefill_attn uses for num_splits == 0 on these static shapes. */
int lv_prefill_num_splits(int batch, int max_seqlen_q, int max_seqlen_k, int num_heads);

/* Bytes of workspace lv_prefill_attn needs for (total_q, num_heads, num_splits); 0 for <= 1 split. */
size_t lv_prefill_workspace_bytes(int64_t total_q, int num_heads, int head_dim, int num_splits);

/* Upper bound of lv_prefill_workspace_bytes over every launch of at most max_total_q rows with the
 * split count lv_prefill_num_splits picks: what a caller reserves once, before it sizes its KV cache. */
size_t lv_prefill_max_workspace_bytes(int64_t max_total_q, int num_heads, int head_dim);

/*
 * Split count lv_decode_attn will use when called with num_splits == 0, for these static
 * shapes (it never depends on device data, so it is stable under hipGraph replay).
 */
int lv_decode_num_splits(int batch, int num_kv_heads, int max_blocks_per_seq, int block_size);

/*
 * Bytes of device workspace lv_decode_attn needs for (batch, num_heads, num_splits).
 * 0 when num_splits <= 1. head_dim must be 128.
 */
size_t lv_decode_workspace_bytes(int batch, int num_heads, int head_dim, int num_splits);

/*
 * Single-query-token attention over the paged KV cache (decode).
 * Reference call site: flash_attn_with_kvcache(q.unsqueeze(1), k_cache, v_cache,
 * cache_seqlens=context_lens, block_table=block_tables, softmax_scale, causal=True)
 * — attention.py:172-175.
 *
 *   q, o          [batch, num_heads, head_dim], row strides q_stride0 / o_stride0;
 *                 num_heads / num_kv_heads must be 1..8 (LV_ERR_UNSUPPORTED otherwise)
 *   k_cache, v_cache  [num_blocks, block_size, num_kv_heads, head_dim] contiguous
 *   context_lens  int32 [batch]; sequence b attends to cache tokens [0, context_lens[b]),
 *                 which already include the token stored by this step's lv_store_kvcache.
 *                 context_lens[b] == 0 (graph padding rows) writes a zero output row.
 *   block_table   int32 [batch, block_table_stride], >= max_blocks_per_seq valid columns;
 *                 entries past ceil(context_lens[b] / block_size) are never read.
 *   block_size    tokens per block; a multiple of 32.
 *   num_splits    KV splits per (sequence, kv head): 0 = library heuristic
 *                 (lv_decode_num_splits), 1 = single pass, n > 1 = split-KV with a merge
 *                 pass through `workspace` (>= lv_decode_workspace_bytes bytes, 16-byte aligned).
 *   seq_order     optional int32 [batch] permutation of 0..batch-1 (NULL = identity): the order
 *                 in which sequences are handed to the GPU. Passing the sequences longest first
 *                 evens out the tail of a ragged batch; results do not depend on it. With an
 *                 order the kernels also take all kv heads of 32 launch positions before the
 *                 next 32 (without one: every sequence of one kv head before the next head,
 *                 which is ~3 % faster when all lengths are equal, so pass NULL for such a
 *                 batch). A negative FIRST entry makes the call behave as with NULL (a captured
 *                 hipGraph, whose pointers are fixed, switches per replay this way). Other entries
 *                 outside [0, batch) are skipped (that sequence's output is then not written).
 *                 No counterpart in the reference call; a caller that has no lengths on the
 *                 host passes NULL.
 *   kernel_hint   LV_DECODE_KERNEL_AUTO, or _VALU / _MFMA to force one of the two kernels (same
 *                 function; a performance matter and what the parity tests iterate over).
 *   scale         finite and > 0.
 *
 * bf16 inputs, fp32 scores / online softmax / accumulation, bf16 output.
 *
 * The workspace is written by the split kernel and read by the merge kernel of the same call.
 * Calls that share one workspace must therefore be ordered on ONE stream (one workspace per
 * device and stream); two streams decoding concurrently need a workspace each.
 */
int lv_decode_attn(const void* q, const void* k_cache, const void* v_cache, void* o,
                   int64_t q_stride0, int64_t o_stride0,
                   const int32_t* context_lens,
                   const int32_t* block_table, int block_table_stride,
                   int batch, int max_blocks_per_seq, int block_size,
                   int num_heads, int num_kv_heads, int head_dim, float scale,
                   void* workspace, size_t workspace_bytes, int num_splits,
                   const int32_t* seq_order, int kernel_hint, lv_stream_t stream);

/*
 * lv_decode_attn with a host-built work list, for batches of unequal lengths. The caller of the
 * reference's flash_attn_with_kvcache (lightvllm/layers/attention.py:172-175) builds context_lens
 * from len(seq) on the host (lightvllm/engine/model_runner.py:383-418), so it can cut long
 * sequences into more pieces than short ones and say in which order the pieces start, where
 * lv_decode_attn must give every sequence the same num_splits (lengths live on the device).
 * Same function, same results up to the fp32 summation order of the split merge.
 *
 *   work_items  device int32 [2 + 2 * num_work_items]: {used, group, then per item: sequence,
 *               split | (splits of that sequence << 16)}. `used` <= num_work_items items are live.
 *               Every split 0 .. n_b - 1 of every sequence with context_len > 0 must appear exactly
 *               once, with the same n_b (<= max_splits) that seq_splits[b] holds. Items run in list
 *               order (put the longest pieces first) in groups of `group` items x all kv heads.
 *               An item that does not fit the launch (sequence >= batch, n_b > max_splits,
 *               split >= n_b) runs nothing; it cannot make the kernels touch memory outside the
 *               arguments, but its output rows are then not written.
 *   num_work_items  capacity of the list; it alone sizes the grid (num_work_items x num_kv_heads
 *               workgroups), so a captured hipGraph replays whatever list the host wrote.
 *   seq_splits  device int32 [batch]: n_b of every sequence (1 = its one item writes `o` itself; a
 *               sequence of length 0 needs one item too, which writes its zero row).
 *   max_splits  split capacity of a workspace row: workspace_bytes >=
 *               lv_decode_workspace_bytes(batch, num_heads, head_dim, max_splits); no workspace
 *               needed when it is 1.
 * light_vllm_amd/decode_plan.py builds the list from python ints.
 */
int lv_decode_attn_planned(const void* q, const void* k_cache, const void* v_cache, void* o,
                           int64_t q_stride0, int64_t o_stride0,
                           const int32_t* context_lens,
                           const int32_t* block_table, int block_table_stride,
                           int batch, int max_blocks_per_seq, int block_size,
                           int num_heads, int num_kv_heads, int head_dim, float scale,
                           void* workspace, size_t workspace_bytes, int max_splits,
                           const int32_t* work_items, int num_work_items,
                           const int32_t* seq_splits, int kernel_hint, lv_stream_t stream);

/*
 * lv_decode_attn for callers that cannot allocate device memory or carry a workspace (code under
 * hipGraph capture, tight serving loops, harnesses that forbid hipMalloc). No counterpart in the
 * reference: it is the same function as lv_decode_attn (same arguments, kernels and results, bit for
 * bit), but the split-KV partials go to a device array built into the library instead of a caller
 * workspace, and the call allocates nothing.
 *
 *   - The array is LV_DECODE_STATIC_SCRATCH_BYTES large on every GPU the library runs on, for the
 *     life of the process. A call whose (batch, num_heads, num_splits) needs more
 *     (lv_decode_workspace_bytes) returns LV_ERR_WORKSPACE and launches nothing; every split count
 *     lv_decode_num_splits returns fits.
 *   - Every call on a GPU shares that GPU's array: calls on one stream are ordered and safe (each
 *     call's merge pass reads only what its own split pass wrote); calls that overlap on one GPU
 *     (two streams, two threads) race. Use lv_decode_attn with separate workspaces for those.
 *   - The array's address is looked up on the first call on each device; make that first call
 *     outside stream capture.
 */
#define LV_DECODE_STATIC_SCRATCH_BYTES (17u << 20)
int lv_decode_attn_static(const void* q, const void* k_cache, const void* v_cache, void* o,
                          int64_t q_stride0, int64_t o_stride0,
                          const int32_t* context_lens,
                          const int32_t* block_table, int block_table_stride,
                          int batch, int max_blocks_per_seq, int block_size,
                          int num_heads, int num_kv_heads, int head_dim, float scale,
                          int num_splits, const int32_t* seq_order, int kernel_hint,
                          lv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LV_ATTENTION_H */
