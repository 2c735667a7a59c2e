// Paged decode attention for gfx950 (MI355X): one query token per sequence attends to its
// block_table-addressed KV cache. Semantics of the reference call
// flash_attn_with_kvcache(q[:,None], k_cache, v_cache, cache_seqlens, block_table, causal)
// at lightvllm/layers/attention.py:172-175.
//
// The op has G = num_heads/num_kv_heads FLOP per KV byte, so it is an HBM streaming problem.
// Design:
//   * one workgroup (4 waves) per (sequence, kv head, kv split); all G query heads of the kv
//     head are served from one pass over K and V, so every cache byte is read exactly once;
//   * K/V rows go HBM -> VGPR directly (no LDS round trip): 16 lanes x 16 B cover one 256-B
//     head row, one wave instruction covers 4 consecutive tokens; U such loads for K and U for V
//     are kept in flight per wave (software pipelined one chunk of 4U tokens ahead);
//   * q.k partial dots use v_dot2c_f32_bf16, the 16-lane reduction uses DPP row ops;
//   * online softmax state (m, l) per 16-lane group, merged wave -> workgroup (LDS) -> splits
//     (fp32 workspace + a small merge kernel) with the usual (m, l) rescaling rule;
//   * the grid depends only on static shapes (batch, kv heads, splits); each workgroup derives
//     its token range from context_lens on device, so the launch is hipGraph-replay safe.
#include "lv_kernels.h"

namespace lv {


// Split boundaries are multiples of this many tokens: a multiple of every chunk size 4*U used
// below and a divisor-compatible unit for block sizes that are multiples of 32 (a chunk of 4*U
// tokens starting at a multiple of 4*U never straddles two blocks).
constexpr int kSplitUnit = 64;

// Diagnostic builds only (tools/decode_abl.sh): pieces of the vector-ALU kernel's arithmetic
// compiled out, results then wrong by design. 1 = dot products, 2 = 16-lane reductions,
// 4 = exponentials, 8 = p.V accumulation. The library is built with 0.
#ifndef LV_DECODE_ABL
#define LV_DECODE_ABL 0
#endif

template <bool NT>
__device__ __forceinline__ u32x4 load16(const uint16_t* p) {
  if constexpr (NT) {
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  } else {
    return *reinterpret_cast<const u32x4*>(p);
  }
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Workgroup index -> (position in the launch order, kv head).
// Without a caller-chosen order: sequence fastest, kv head slowest (the 8 heads of one token row
// are streamed at different times, the fastest pattern for batches of equal lengths: see
// decode_attn_kernel). With one (the caller has sorted a ragged batch longest first): groups of
// kOrderGroup consecutive positions, inside a group sequence fastest and kv head slowest, so that
// all heads of the longest sequences start early and the launch ends on the shortest ones.
// Measured, 256 sequences of 1..1024 / 1..4096 tokens: 68 -> 73 % / 73 -> 78 % of HBM peak with
// 4 query heads per kv head, 73 -> 77 % / 77 -> 81 % with 2; groups of 8 lose it again, and
// batches of equal lengths run 3 % slower grouped, which is why the order is the switch.
// A negative first entry stands for "no order" (identity, ungrouped): a captured hipGraph, whose
// pointer arguments are fixed, can then switch per replay by what it writes into its buffer.
constexpr int kOrderGroup = 32;
__device__ __forceinline__ void work_item(const DecodeParams& p, int bk, int& b, int& kvh) {
  if (p.seq_order == nullptr || p.seq_order[0] < 0) {
    b = bk % p.batch;
    kvh = bk / p.batch;
    return;
  }
  const int per_group = kOrderGroup * p.num_kv_heads;
  const int grp = bk / per_group;
  const int rem = bk - grp * per_group;
  const int size = min(kOrderGroup, p.batch - grp * kOrderGroup);   // the last group may be short
  b = p.seq_order[grp * kOrderGroup + rem % size];
  kvh = rem / size;
}

// What a workgroup serves: split `split` of `nsplit` of sequence b's tokens, kv head kvh.
// Without a work list every sequence is cut into the launch's p.nsplit splits (split fastest in
// the workgroup index). With one (lv_decode_attn_planned: the caller knows the lengths and has cut
// long sequences into more pieces than short ones, pieces in the order they should start) the list
// is {used items, group size, then per item: sequence, split | splits of that sequence << 16}; the
// workgroup index walks groups of `group size` items x all kv heads, items fastest. The list lives
// in device memory, so a captured graph follows whatever the host wrote before the replay; entries
// past `used`, and entries that do not fit the launch (sequence out of range, more splits than the
// workspace has room for), run nothing. p.nsplit is then only the split capacity of a workspace row.
struct DecodeWork { int b, kvh, split, nsplit; };
__device__ __forceinline__ bool decode_work(const DecodeParams& p, DecodeWork& w) {
  const int bid = blockIdx.x;
  if (p.work_items == nullptr) {
    w.nsplit = p.nsplit;
    w.split = bid % p.nsplit;
    work_item(p, bid / p.nsplit, w.b, w.kvh);   // caller-chosen launch order (longest first), if any
    return static_cast<unsigned>(w.b) < static_cast<unsigned>(p.batch);
  }
  const int used = min(p.work_items[0], p.num_work_items);
  const int gs = max(p.work_items[1], 1);
  const int per_group = gs * p.num_kv_heads;
  const int grp = bid / per_group;
  if (grp * gs >= used) return false;
  const int rem = bid - grp * per_group;
  const int size = min(gs, used - grp * gs);                        // the last group may be short
  w.kvh = rem / size;
  if (w.kvh >= p.num_kv_heads) return false;
  const int it = grp * gs + rem % size;
  w.b = p.work_items[2 + 2 * it];
  const unsigned code = static_cast<unsigned>(p.work_items[3 + 2 * it]);
  w.split = static_cast<int>(code & 0xffffu);
  w.nsplit = static_cast<int>(code >> 16);
  return static_cast<unsigned>(w.b) < static_cast<unsigned>(p.batch) && w.nsplit >= 1 &&
         w.nsplit <= p.nsplit && w.split < w.nsplit;
}

template <int G, int U, bool NT>
__global__ __launch_bounds__(256) void decode_attn_kernel(const DecodeParams p) {
  constexpr int TOK = 4 * U;  // tokens per wave chunk
  const int Hkv = p.num_kv_heads;
  // Workgroup order: split fastest, then sequence, kv head slowest, so the 8 heads of one token
  // row (8 adjacent 256-byte pieces) are streamed at different times. Measured on this access
  // pattern with no arithmetic at all (tools/hbm_pattern_bw.hip): 7.0 TB/s this way, 6.5 TB/s when
  // the heads of a sequence are read concurrently (kv head fastest, or 8 sequences x 8 rotated
  // heads in adjacent workgroups).
  DecodeWork dw;
  if (!decode_work(p, dw)) return;
  const int nsplit = dw.nsplit, split = dw.split, b = dw.b, kvh = dw.kvh;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int g = lane >> 4;  // token sub-slot inside one load instruction
  const int i = lane & 15;  // 16-byte chunk of the 256-byte head row

  const int BS = p.block_size;
  const int ctx = p.context_lens[b];
  // splits divide the sequence's actual length evenly, in units of kSplitUnit tokens
  const int nunits = (ctx + kSplitUnit - 1) / kSplitUnit;
  const int ulo = static_cast<int>(static_cast<int64_t>(split) * nunits / nsplit);
  const int uhi = static_cast<int>(static_cast<int64_t>(split + 1) * nunits / nsplit);
  const int t0 = ulo * kSplitUnit;
  const int t1 = min(uhi * kSplitUnit, ctx);
  const int ntok = t1 - t0;
  const int nchunks = ntok > 0 ? (ntok + TOK - 1) / TOK : 0;

  const int64_t row_elems = static_cast<int64_t>(Hkv) * kHeadDim;
  const int32_t* bt_row = p.block_table + static_cast<int64_t>(b) * p.bt_stride;

  // q fragments: 8 dims of each of the G heads, kept packed as bf16 pairs.
  u32x4 qv[G];
#pragma unroll
  for (int h = 0; h < G; ++h) {
    qv[h] = *reinterpret_cast<const u32x4*>(p.q + static_cast<int64_t>(b) * p.q_stride0 +
                                            (kvh * G + h) * kHeadDim + 8 * i);
  }

  float m[G], l[G], acc[G][8];
#pragma unroll
  for (int h = 0; h < G; ++h) {
    m[h] = -INFINITY;
    l[h] = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[h][d] = 0.f;
  }

  // Element offset of (first token of chunk c, this kv head, this lane's 16-byte chunk).
  auto chunk_base = [&](int c) -> int64_t {
    const int t = t0 + c * TOK;
    const int blk = bt_row[t / BS];
    const int64_t row = static_cast<int64_t>(blk) * BS + (t % BS);
    return (row * Hkv + kvh) * kHeadDim + 8 * i;
  };
  // Token offset inside chunk c for load slot u; rows past the split's end are redirected
  // to its last valid row (their scores are masked below), so no out-of-range row is read.
  auto tok_off = [&](int c, int u) -> int {
    const int last = ntok - 1 - c * TOK;
    return min(4 * u + g, last);
  };

  int c = wave;
  if (c < nchunks) {
    u32x4 kb[U], vb[U];
    {
      const int64_t base = chunk_base(c);
#pragma unroll
      for (int u = 0; u < U; ++u) kb[u] = load16<NT>(p.k_cache + base + tok_off(c, u) * row_elems);
#pragma unroll
      for (int u = 0; u < U; ++u) vb[u] = load16<NT>(p.v_cache + base + tok_off(c, u) * row_elems);
    }
    for (; c < nchunks; c += 4) {
      const int cn = c + 4;
      const bool has_next = cn < nchunks;
      int64_t base_n = 0;
      if (has_next) base_n = chunk_base(cn);

      // ---- scores: s[u][h] = log2(e) * scale * <q_h, k_u>
      float s[U][G];
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int h = 0; h < G; ++h) {
          float d = 0.f;
          if constexpr ((LV_DECODE_ABL & 1) == 0) {
            d = dot2_bf16(kb[u][0], qv[h][0], d);
            d = dot2_bf16(kb[u][1], qv[h][1], d);
            d = dot2_bf16(kb[u][2], qv[h][2], d);
            d = dot2_bf16(kb[u][3], qv[h][3], d);
          } else {
            d = bf16_lo((kb[u][0] ^ kb[u][1]) + (kb[u][2] ^ kb[u][3]) + qv[h][0]);
          }
          s[u][h] = d;
        }
      }
      if (has_next) {
#pragma unroll
        for (int u = 0; u < U; ++u)
          kb[u] = load16<NT>(p.k_cache + base_n + tok_off(cn, u) * row_elems);
      }
      const int valid = ntok - c * TOK;  // tokens of this chunk inside the split
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool ok = (4 * u + g) < valid;
#pragma unroll
        for (int h = 0; h < G; ++h) {
          const float r = ((LV_DECODE_ABL & 2) ? s[u][h] : row16_allreduce_sum(s[u][h])) * p.scale_log2e;
          s[u][h] = ok ? r : -INFINITY;
        }
      }
      // ---- online softmax (per 16-lane group)
#pragma unroll
      for (int h = 0; h < G; ++h) {
        float mx = s[0][h];
#pragma unroll
        for (int u = 1; u < U; ++u) mx = fmaxf(mx, s[u][h]);
        const float m_new = fmaxf(m[h], mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = fast_exp2(m[h] - m_use);
        float psum = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float pe = (LV_DECODE_ABL & 4) ? s[u][h] - m_use : fast_exp2(s[u][h] - m_use);
          s[u][h] = pe;
          psum += pe;
        }
        l[h] = l[h] * alpha + psum;
        m[h] = m_new;
#pragma unroll
        for (int d = 0; d < 8; ++d) acc[h][d] *= alpha;
      }
      // ---- acc += p * V
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if constexpr ((LV_DECODE_ABL & 8) == 0) {
          float vf[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            vf[2 * j] = bf16_lo(vb[u][j]);
            vf[2 * j + 1] = bf16_hi(vb[u][j]);
          }
#pragma unroll
          for (int h = 0; h < G; ++h) {
#pragma unroll
            for (int d = 0; d < 8; ++d) acc[h][d] = fmaf(s[u][h], vf[d], acc[h][d]);
          }
        } else {
          const float vx = bf16_lo((vb[u][0] ^ vb[u][1]) + (vb[u][2] ^ vb[u][3]));
#pragma unroll
          for (int h = 0; h < G; ++h) acc[h][u & 7] = fmaf(s[u][h], vx, acc[h][u & 7]);
        }
      }
      if (has_next) {
#pragma unroll
        for (int u = 0; u < U; ++u)
          vb[u] = load16<NT>(p.v_cache + base_n + tok_off(cn, u) * row_elems);
      }
    }
  }

  // ---- merge the 4 token groups of the wave (lanes ^16, ^32)
#pragma unroll
  for (int off = 16; off <= 32; off <<= 1) {
#pragma unroll
    for (int h = 0; h < G; ++h) {
      const float mo = __shfl_xor(m[h], off);
      const float lo = __shfl_xor(l[h], off);
      const float M = fmaxf(m[h], mo);
      const float Mu = (M == -INFINITY) ? 0.f : M;
      const float a = fast_exp2(m[h] - Mu);
      const float bsc = fast_exp2(mo - Mu);
      l[h] = l[h] * a + lo * bsc;
      m[h] = M;
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const float ao = __shfl_xor(acc[h][d], off);
        acc[h][d] = acc[h][d] * a + ao * bsc;
      }
    }
  }

  // ---- merge the 4 waves through LDS
  __shared__ float sm_acc[4][G][kHeadDim];
  __shared__ float sm_ml[4][G][2];
  if (g == 0) {
#pragma unroll
    for (int h = 0; h < G; ++h) {
      float4* dst = reinterpret_cast<float4*>(&sm_acc[wave][h][8 * i]);
      dst[0] = make_float4(acc[h][0], acc[h][1], acc[h][2], acc[h][3]);
      dst[1] = make_float4(acc[h][4], acc[h][5], acc[h][6], acc[h][7]);
      if (i == 0) {
        sm_ml[wave][h][0] = m[h];
        sm_ml[wave][h][1] = l[h];
      }
    }
  }
  __syncthreads();

  for (int e = tid; e < G * kHeadDim; e += 256) {
    const int h = e / kHeadDim;
    const int d = e % kHeadDim;
    float M = sm_ml[0][h][0];
#pragma unroll
    for (int w = 1; w < 4; ++w) M = fmaxf(M, sm_ml[w][h][0]);
    const float Mu = (M == -INFINITY) ? 0.f : M;
    float val = 0.f, L = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float sc = fast_exp2(sm_ml[w][h][0] - Mu);
      val += sc * sm_acc[w][h][d];
      L += sc * sm_ml[w][h][1];
    }
    const int hq = kvh * G + h;
    if (nsplit == 1) {
      const float r = L > 0.f ? val / L : 0.f;
      p.o[static_cast<int64_t>(b) * p.o_stride0 + hq * kHeadDim + d] = f32_to_bf16_bits(r);
    } else {
      const int64_t slot = (static_cast<int64_t>(b) * p.num_heads + hq) * p.nsplit + split;
      p.ws_o[slot * kHeadDim + d] = val;
      if (d == 0) {
        p.ws_ml[slot * 2 + 0] = M;
        p.ws_ml[slot * 2 + 1] = L;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Matrix-core variant for wide query groups (G = 8, optionally 4). With 8 query heads per kv
// head the vector-ALU formulation above spends ~170 instructions per 1 KiB loaded and is
// ALU-bound; here both products run on v_mfma_f32_16x16x32_bf16 with the G heads padded to the
// 16-wide dimension:
//   S^T[key][head] = K[key][:] Q^T      A = K rows straight from global memory (a lane's 16-byte
//                                       load IS its A fragment: row = key, 8 consecutive d),
//                                       B = Q^T fragments held in registers
//   O^T[d][head]  += V^T[d][key] P^T    A = V^T via ds_read_b64_tr_b16 from a wave-private LDS
//                                       image of the V rows, B = the bf16-rounded P registers
// The score tile has the head on the lane and 4 keys in registers, so the softmax statistics are
// in-register plus two row exchanges, and P needs no lane movement to become the B operand.
// Wave chunk = 32 keys (two 16-key score tiles = one 32-deep k-step of the second product).
// A-operand row i of a score tile holds key pi(i) = 8*((i>>2)&1) + 4*(i>>3) + (i&3) of the tile, so
// that the two 16-lane groups of a 32-lane half read V blocks 8 rows apart (bank-conflict free).
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((address_space(3))) unsigned char lds_byte_t;

__device__ __forceinline__ int v_lds_off(int row, int ch) {  // [rows][128 x bf16] image, 16-byte chunks swizzled
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

template <int G>
__global__ __launch_bounds__(256, 2) void decode_mfma_kernel(const DecodeParams p) {
  constexpr int CH = 32;  // keys per wave chunk
  constexpr float kDefer = 6.0f;  // running max moves only when a row outgrows it by 2^6
  // per wave: 8 KiB V image during the sweep; afterwards per wave G x (128 + 2) floats of partials
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * 8192];
  lds_byte_t* const lds3 = (lds_byte_t*)lds;

  const int Hkv = p.num_kv_heads;
  DecodeWork dw;
  if (!decode_work(p, dw)) return;
  const int nsplit = dw.nsplit, split = dw.split, b = dw.b, kvh = dw.kvh;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int c16 = lane & 15;  // MFMA row (A) / column (B, C): head index for B and C
  const int q = lane >> 4;    // 8-element k group of A/B, 4-row group of C

  const int BS = p.block_size;
  const int ctx = p.context_lens[b];
  const int nunits = (ctx + kSplitUnit - 1) / kSplitUnit;
  const int ulo = static_cast<int>(static_cast<int64_t>(split) * nunits / nsplit);
  const int uhi = static_cast<int>(static_cast<int64_t>(split + 1) * nunits / nsplit);
  const int t0 = ulo * kSplitUnit;
  const int t1 = min(uhi * kSplitUnit, ctx);
  const int ntok = t1 - t0;
  const int nchunks = ntok > 0 ? (ntok + CH - 1) / CH : 0;
  const int64_t row_bytes = static_cast<int64_t>(Hkv) * kHeadDim * 2;
  const int32_t* bt_row = p.block_table + static_cast<int64_t>(b) * p.bt_stride;

  // Q^T fragments (B operand), k-step s covers d = 32s .. 32s+31; heads >= G are zero columns
  bf16x8_t qf[4];
  {
    const int hq = kvh * G + min(c16, G - 1);
    const uint16_t* qp = p.q + static_cast<int64_t>(b) * p.q_stride0 + hq * kHeadDim + 8 * q;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4 t = *reinterpret_cast<const u32x4*>(qp + 32 * s);
      if (c16 >= G) t = u32x4{0, 0, 0, 0};
      qf[s] = __builtin_bit_cast(bf16x8_t, t);
    }
  }

  // which key of a 16-key tile this lane's A row holds / this lane's C registers hold
  const int a_key = 8 * ((c16 >> 2) & 1) + 4 * (c16 >> 3) + (c16 & 3);
  const int c_key0 = 8 * (q & 1) + 4 * (q >> 1);  // + r for register r

  // V staging: lane loads chunk (lane & 15) of rows 4u + q (u = 0..7) and stores it swizzled
  lds_byte_t* const vimg = lds3 + wave * 8192;
  const int vw_off = v_lds_off(q, c16);          // row 4u + q: + 1024 * u, chunk index ^ (u & 3)
  // transposed reads: lane 4q''+p'' of a group addresses row r0 + q'', columns 4p'' .. of block dt
  const int tq = c16 >> 2, tp = c16 & 3;
  const int vr_row = c_key0 + tq;                // second read: + 16 rows
  // byte offset for dt = 0; block dt adds chunk 2*dt, which XORs into the swizzled chunk index
  int vr_off[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) vr_off[t] = v_lds_off(vr_row + 16 * t, tp >> 1) + 8 * (tp & 1);

  f32x4_t oacc[8];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) oacc[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const float sc = p.scale_log2e;

  auto chunk_row = [&](int c) -> int64_t {  // cache row of the chunk's first key
    const int t = t0 + c * CH;
    const int blk = bt_row[t / BS];
    return static_cast<int64_t>(blk) * BS + (t % BS);
  };
  u32x4 kreg[2][4], vreg[8];
  auto issue_k = [&](int c) {
    const char* base = reinterpret_cast<const char*>(p.k_cache) + chunk_row(c) * row_bytes + kvh * (kHeadDim * 2);
    const int last = ntok - 1 - c * CH;  // rows past the split's end re-read its last row (masked below)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const char* rp = base + static_cast<int64_t>(min(16 * t + a_key, last)) * row_bytes + 16 * q;
#pragma unroll
      for (int s = 0; s < 4; ++s) kreg[t][s] = load16<true>(reinterpret_cast<const uint16_t*>(rp + 64 * s));
    }
  };
  auto issue_v = [&](int c) {
    const char* base = reinterpret_cast<const char*>(p.v_cache) + chunk_row(c) * row_bytes + kvh * (kHeadDim * 2) + 16 * c16;
    const int last = ntok - 1 - c * CH;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      vreg[u] = load16<true>(reinterpret_cast<const uint16_t*>(base + static_cast<int64_t>(min(4 * u + q, last)) * row_bytes));
  };

  int c = wave;
  if (c < nchunks) {
    issue_k(c);
    issue_v(c);
  }
  for (; c < nchunks; c += 4) {
    const int cn = c + 4;
    // ---- S^T = K Q^T for the two 16-key tiles
    f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kreg[0][s]), qf[s], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kreg[1][s]), qf[s], s1, 0, 0, 0);
    }
    if (cn < nchunks) issue_k(cn);
    // ---- scale, mask keys past the split's end, row maximum over the chunk's 32 keys
    const int valid = ntok - c * CH;
    float sv[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sv[r] = (c_key0 + r < valid) ? s0[r] * sc : -INFINITY;
      sv[4 + r] = (16 + c_key0 + r < valid) ? s1[r] * sc : -INFINITY;
    }
    float mx = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    // deferred running-max update (wave-uniform decision)
    if (__builtin_amdgcn_ballot_w64(mx - m_run > kDefer || (m_run == -INFINITY && mx > -INFINITY)) != 0) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = (m_run == -INFINITY) ? 0.f : fast_exp2(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) oacc[dt][r] *= alpha;
    }
    const float m_use = (m_run == -INFINITY) ? 0.f : m_run;
    float pe[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pe[j] = fast_exp2(sv[j] - m_use);
      l_run += pe[j];
    }
    bf16x8_t pf;  // k slot (q, j): j < 4 -> tile 0 register j, j >= 4 -> tile 1 register j-4
#pragma unroll
    for (int j = 0; j < 8; ++j) pf[j] = static_cast<__bf16>(pe[j]);
    // ---- V rows -> LDS image (wave private: same-wave LDS operations execute in order)
#pragma unroll
    for (int u = 0; u < 8; ++u)
      *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(vimg + ((vw_off ^ (16 * (u & 3))) + 1024 * u)) = vreg[u];
    if (cn < nchunks) issue_v(cn);
    // ---- O^T += V^T P^T, one 16-column block of d per MFMA
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      bf16x8_t va;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x4_t x = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            reinterpret_cast<__attribute__((address_space(3))) bf16x4_t*>(vimg + (vr_off[t] ^ (32 * dt))));
        va[4 * t + 0] = x[0];
        va[4 * t + 1] = x[1];
        va[4 * t + 2] = x[2];
        va[4 * t + 3] = x[3];
      }
      oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pf, oacc[dt], 0, 0, 0);
    }
  }

  // ---- row sums: every 16-lane group holds a share
  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);

  // ---- merge the 4 waves through LDS: per wave [G][128] partial outputs + (m, l)
  __syncthreads();   // all waves are done with their V images
  float* const part = reinterpret_cast<float*>(lds);
  constexpr int kPartStride = G * kHeadDim + 2 * G;  // floats per wave
  if (c16 < G) {
    float* dst = part + wave * kPartStride + c16 * kHeadDim;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
      *reinterpret_cast<float4*>(dst + 16 * dt + 4 * q) = make_float4(oacc[dt][0], oacc[dt][1], oacc[dt][2], oacc[dt][3]);
    if (q == 0) {
      part[wave * kPartStride + G * kHeadDim + 2 * c16 + 0] = m_run;
      part[wave * kPartStride + G * kHeadDim + 2 * c16 + 1] = l_run;
    }
  }
  __syncthreads();
  for (int e = tid; e < G * kHeadDim; e += 256) {
    const int h = e / kHeadDim;
    const int d = e % kHeadDim;
    float M = part[G * kHeadDim + 2 * h];
#pragma unroll
    for (int w = 1; w < 4; ++w) M = fmaxf(M, part[w * kPartStride + G * kHeadDim + 2 * h]);
    const float Mu = (M == -INFINITY) ? 0.f : M;
    float val = 0.f, L = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = fast_exp2(part[w * kPartStride + G * kHeadDim + 2 * h] - Mu);
      val += f * part[w * kPartStride + h * kHeadDim + d];
      L += f * part[w * kPartStride + G * kHeadDim + 2 * h + 1];
    }
    const int hq = kvh * G + h;
    if (nsplit == 1) {
      const float r = L > 0.f ? val / L : 0.f;
      p.o[static_cast<int64_t>(b) * p.o_stride0 + hq * kHeadDim + d] = f32_to_bf16_bits(r);
    } else {
      const int64_t slot = (static_cast<int64_t>(b) * p.num_heads + hq) * p.nsplit + split;
      p.ws_o[slot * kHeadDim + d] = val;
      if (d == 0) {
        p.ws_ml[slot * 2 + 0] = M;
        p.ws_ml[slot * 2 + 1] = L;
      }
    }
  }
}

// Combine the nsplit partial results of one (sequence, query head). DS workgroups share one
// (sequence, head), each merging 128 / DS of its output dims. 256 lanes = groups of 32 / DS lanes;
// a lane owns 4 output dims (one 16-byte load per split) and its group walks splits g, g + NG, ...,
// so many independent loads are in flight; groups are merged through LDS. DS = 4 is for launches
// with few (sequence, head) pairs and many splits: four times the workgroups, each group's splits
// in one round of loads.
template <int DS>
__global__ __launch_bounds__(256) void decode_merge_kernel(const float* __restrict__ ws_o,
                                                           const float* __restrict__ ws_ml,
                                                           uint16_t* __restrict__ o,
                                                           int64_t o_stride0, int num_heads,
                                                           int ws_splits, const int32_t* __restrict__ seq_splits) {
  constexpr int LPG = 32 / DS;          // lanes per group
  constexpr int NG = 256 / LPG;         // groups
  constexpr int DIMS = kHeadDim / DS;   // output dims of this workgroup
  __shared__ float sm_max[4];
  __shared__ float sm_part[NG][DIMS + 4];
  const int bh = blockIdx.x / DS;
  const int dim0 = (blockIdx.x % DS) * DIMS;
  const int b = bh / num_heads;
  const int hq = bh % num_heads;
  const int tid = threadIdx.x;
  const int grp = tid / LPG;
  const int ln = tid % LPG;
  // With per-sequence split counts (a work-list launch) a workspace row still has room for
  // ws_splits partials, of which this sequence filled the first seq_splits[b]; a sequence served in
  // one piece was written to `o` by the attention kernel itself.
  const int nsplit = seq_splits ? min(seq_splits[b], ws_splits) : ws_splits;
  if (nsplit <= 1 && seq_splits) return;
  const int64_t base = static_cast<int64_t>(bh) * ws_splits;

  // The first kPre partials of this group are requested before the maximum is known: the two
  // load latencies (statistics, partial outputs) then overlap instead of following each other.
  constexpr int kPre = 16 / DS < 8 ? 8 : 16 / DS;
  float4 pv[kPre];
  float2 pml[kPre];
#pragma unroll
  for (int j = 0; j < kPre; ++j) {
    const int s = grp + NG * j;
    if (s < nsplit) {
      pml[j] = *reinterpret_cast<const float2*>(ws_ml + (base + s) * 2);
      pv[j] = *reinterpret_cast<const float4*>(ws_o + (base + s) * kHeadDim + dim0 + 4 * ln);
    }
  }

  // global maximum over splits
  float m = -INFINITY;
  for (int s = tid; s < nsplit; s += 256) m = fmaxf(m, ws_ml[(base + s) * 2]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((tid & 63) == 0) sm_max[tid >> 6] = m;
  __syncthreads();
  const float M = fmaxf(fmaxf(sm_max[0], sm_max[1]), fmaxf(sm_max[2], sm_max[3]));
  const float Mu = (M == -INFINITY) ? 0.f : M;

  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float L = 0.f;
#pragma unroll
  for (int j = 0; j < kPre; ++j) {
    if (grp + NG * j < nsplit) {
      const float sc = fast_exp2(pml[j].x - Mu);
      acc.x += sc * pv[j].x;
      acc.y += sc * pv[j].y;
      acc.z += sc * pv[j].z;
      acc.w += sc * pv[j].w;
      L += sc * pml[j].y;
    }
  }
  // the rest: a chain of load latencies (about 0.6 us per batch of loads in flight)
#pragma unroll 8
  for (int s = grp + NG * kPre; s < nsplit; s += NG) {
    const float2 ml = *reinterpret_cast<const float2*>(ws_ml + (base + s) * 2);
    const float4 v = *reinterpret_cast<const float4*>(ws_o + (base + s) * kHeadDim + dim0 + 4 * ln);
    const float sc = fast_exp2(ml.x - Mu);
    acc.x += sc * v.x;
    acc.y += sc * v.y;
    acc.z += sc * v.z;
    acc.w += sc * v.w;
    L += sc * ml.y;
  }
  *reinterpret_cast<float4*>(&sm_part[grp][4 * ln]) = acc;
  if (ln == 0) sm_part[grp][DIMS] = L;
  __syncthreads();
  if (tid < DIMS) {
    float val = 0.f, Lt = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      val += sm_part[g][tid];
      Lt += sm_part[g][DIMS];
    }
    const float r = Lt > 0.f ? val / Lt : 0.f;
    o[static_cast<int64_t>(b) * o_stride0 + hq * kHeadDim + dim0 + tid] = f32_to_bf16_bits(r);
  }
}

static unsigned decode_grid(const DecodeParams& p) {
  if (p.work_items) return static_cast<unsigned>(p.num_work_items) * p.num_kv_heads;
  return static_cast<unsigned>(p.batch) * p.num_kv_heads * p.nsplit;
}

template <int G, int U>
static void launch_g(const DecodeParams& p, hipStream_t stream) {
  hipLaunchKernelGGL((decode_attn_kernel<G, U, true>), dim3(decode_grid(p)), dim3(256), 0, stream, p);
}

template <int G>
static void launch_mfma(const DecodeParams& p, hipStream_t stream) {
  hipLaunchKernelGGL((decode_mfma_kernel<G>), dim3(decode_grid(p)), dim3(256), 0, stream, p);
}

int launch_decode_attn(const DecodeParams& p, hipStream_t stream) {
  const int G = p.num_heads / p.num_kv_heads;
  // Kernel choice: the matrix-core kernel for 4 and 8 query heads per kv head, the vector-ALU
  // kernel for 1 and 2, unless the caller's kernel_hint names one of them.
  const bool mfma = p.kernel_hint == LV_DECODE_KERNEL_MFMA ||
                    (p.kernel_hint == LV_DECODE_KERNEL_AUTO && G >= 4);
  switch (G) {
    case 1: if (mfma) launch_mfma<1>(p, stream); else launch_g<1, 8>(p, stream); break;
    case 2: if (mfma) launch_mfma<2>(p, stream); else launch_g<2, 8>(p, stream); break;
    case 4: if (mfma) launch_mfma<4>(p, stream); else launch_g<4, 4>(p, stream); break;
    case 8: if (mfma) launch_mfma<8>(p, stream); else launch_g<8, 4>(p, stream); break;
    // other group sizes (Qwen3-14B has 40 q / 8 kv heads = 5): the matrix-core kernel only, its
    // head dimension is a 16-column tile of which G columns are live
    case 3: launch_mfma<3>(p, stream); break;
    case 5: launch_mfma<5>(p, stream); break;
    case 6: launch_mfma<6>(p, stream); break;
    case 7: launch_mfma<7>(p, stream); break;
    default:
      return fail(LV_ERR_UNSUPPORTED, "decode: num_heads/num_kv_heads = %d not in 1..8", G);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(LV_ERR_LAUNCH, "decode_attn launch: %s", hipGetErrorString(e));
  if (p.nsplit > 1) {
    const unsigned pairs = static_cast<unsigned>(p.batch) * p.num_heads;
    // Few (sequence, head) pairs cut into many splits (one long sequence on a TP shard): four
    // workgroups per pair. 1 x 32768 tokens, 4q/1kv, 256 splits: 11.8 -> 10.0 us for the launch pair
    // under graph replay; eight per pair measured 10.2; below 128 splits nothing moves either way.
    if (pairs <= 64 && p.nsplit >= 64)
      hipLaunchKernelGGL(decode_merge_kernel<4>, dim3(pairs * 4), dim3(256), 0, stream, p.ws_o, p.ws_ml, p.o,
                         p.o_stride0, p.num_heads, p.nsplit, p.seq_splits);
    else
      hipLaunchKernelGGL(decode_merge_kernel<1>, dim3(pairs), dim3(256), 0, stream, p.ws_o, p.ws_ml, p.o,
                         p.o_stride0, p.num_heads, p.nsplit, p.seq_splits);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(LV_ERR_LAUNCH, "decode_merge launch: %s", hipGetErrorString(e));
  }
  return LV_OK;
}

}  // namespace lv
