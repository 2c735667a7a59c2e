// Varlen causal prefill attention for gfx950 (MI355X) on bf16 MFMA.
// Semantics of the reference call flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, ...,
// causal=True, block_table) at lightvllm/layers/attention.py:159-165: packed K/V, or K/V read from
// the paged cache through block_table when a prefix is cached; grouped-query heads; causal mask
// aligned to the bottom-right corner (query row r sees keys t <= r + Lk - Lq).
//
// Structure (flash-attention forward, one pass over K/V with online softmax):
//   * workgroup = 4 waves = one 128-row query tile of one (sequence, query head); wave w owns rows
//     32w..32w+31. Query tiles are launched heaviest (latest rows) first.
//   * K/V tiles of 64 keys are staged HBM -> VGPR -> LDS by all 256 lanes in whole 256-byte rows
//     (16 B per lane); the next tile's global loads are issued before the current tile's math so
//     their latency hides under it. The LDS image XOR-swizzles the 16-byte chunk index with the row
//     so that both the row reads (K as MFMA A operand) and the transposed reads (V^T as MFMA A
//     operand, ds_read_b64_tr_b16) are bank-conflict free.
//   * S^T = K Q^T with v_mfma_f32_32x32x16_bf16: the accumulator then holds one query row per lane
//     (column) and 16 keys in registers, so row max / row sum are in-register plus one exchange
//     with lane^32, and the bf16-rounded P registers are directly the B operand of
//     O^T += V^T P^T (same lane = same query row), with no LDS round trip for P.
//   * fp32 scores, statistics and accumulation; exp2 with scale*log2(e) folded into one FMA.
#include "lv_kernels.h"

namespace lv {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int BM = 128;  // query rows per workgroup
constexpr int BN = 64;   // keys per staged tile
constexpr int kTileBytes = BN * kHeadDim * 2;  // 16 KiB

// Byte offset of 16-byte chunk `ch` (0..15) of row `row` inside a [rows][128 x bf16] LDS tile.
__device__ __forceinline__ int lds_off(int row, int ch) {
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

template <bool PAGED>
__global__ __launch_bounds__(256, 2) void prefill_attn_kernel(const PrefillParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kTileBytes];
  unsigned char* const lds_k = lds;
  unsigned char* const lds_v = lds + kTileBytes;

  const int b = blockIdx.z;
  const int hq = blockIdx.y;
  const int G = p.num_heads / p.num_kv_heads;
  const int kvh = hq / G;
  const int q_start = p.cu_q[b];
  const int Lq = p.cu_q[b + 1] - q_start;
  const int k_start = p.cu_k[b];
  const int Lk = p.cu_k[b + 1] - k_start;
  const int q0 = (static_cast<int>(gridDim.x) - 1 - static_cast<int>(blockIdx.x)) * BM;
  if (q0 >= Lq) return;
  const int coff = Lk - Lq;  // causal offset: row r sees keys <= r + coff

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int h = lane >> 5;  // which 8-wide half of an MFMA k-step this lane feeds
  const int c = lane & 31;  // MFMA column = query row inside the wave's 32 rows

  const int wrow0 = q0 + wave * 32;
  const int my_row = wrow0 + c;
  const int my_lim = my_row + coff;  // last visible key of this lane's row
  const int row_ld = min(my_row, Lq - 1);

  // Q^T fragments (B operand of S^T = K Q^T): 8 k-steps x 8 bf16.
  bf16x8 qf[8];
  {
    const uint16_t* qp = p.q + static_cast<int64_t>(q_start + row_ld) * p.q_stride0 + hq * kHeadDim + 8 * h;
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
  }

  const int last_row = min(q0 + BM - 1, Lq - 1);
  const int max_key = min(last_row + coff, Lk - 1);
  const int ntiles = max_key >= 0 ? max_key / BN + 1 : 0;

  // staging role of this thread: chunk st_ch of rows st_r + 16*i, i = 0..3
  const int st_ch = tid & 15;
  const int st_r = tid >> 4;
  const int32_t* bt_row = PAGED ? p.block_table + static_cast<int64_t>(b) * p.bt_stride : nullptr;

  u32x4 kreg[4], vreg[4];
  auto stage_load = [&](int tile) {
    const int kt0 = tile * BN;
    int64_t base_row;  // cache/packed row index of key kt0
    if constexpr (PAGED) {
      const int blk = bt_row[kt0 / p.block_size];
      base_row = static_cast<int64_t>(blk) * p.block_size + (kt0 % p.block_size);
    } else {
      base_row = static_cast<int64_t>(k_start) + kt0;
    }
    const int last = Lk - 1 - kt0;  // rows past the sequence end re-read its last row (masked later)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = min(st_r + 16 * i, last);
      const int64_t row = base_row + r;
      kreg[i] = *reinterpret_cast<const u32x4*>(p.k + row * p.k_stride0 + kvh * kHeadDim + 8 * st_ch);
      vreg[i] = *reinterpret_cast<const u32x4*>(p.v + row * p.v_stride0 + kvh * kHeadDim + 8 * st_ch);
    }
  };
  auto stage_write = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int off = lds_off(st_r + 16 * i, st_ch);
      *reinterpret_cast<u32x4*>(lds_k + off) = kreg[i];
      *reinterpret_cast<u32x4*>(lds_v + off) = vreg[i];
    }
  };

  f32x16 oacc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
  float m_run = -INFINITY;  // running row max, raw score units
  float l_run = 0.f;        // this lane's share of the row sum (keys with its h)

  const float cexp = p.scale_log2e;

  // transposed-read lane roles (ds_read_b64_tr_b16 works per 16-lane group)
  const int gi = lane >> 4;        // group: h = gi>>1, d half = gi&1
  const int tq = (lane & 15) >> 2;  // row of the 4x16 block this lane addresses
  const int tp = lane & 3;          // 4-column piece of that row

  if (ntiles > 0) stage_load(0);
  for (int tile = 0; tile < ntiles; ++tile) {
    const int kt0 = tile * BN;
    __syncthreads();  // every wave is done reading the previous tile
    stage_write();
    __syncthreads();
    if (tile + 1 < ntiles) stage_load(tile + 1);

    // waves whose rows all lie above this tile have nothing to add
    if (kt0 > wrow0 + 31 + coff) continue;

    // ---- S^T = K Q^T : 2 sub-tiles of 32 keys
    f32x16 sacc[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[sub][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(lds_k + lds_off(32 * sub + c, 2 * s + h));
        sacc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[s], sacc[sub], 0, 0, 0);
      }
    }
    // ---- causal mask (only tiles that reach past the wave's first row limit)
    if (kt0 + BN - 1 > wrow0 + coff) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt0 + 32 * sub + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (key > my_lim) sacc[sub][r] = -INFINITY;
        }
    }
    // ---- online softmax for this lane's query row
    float mx = sacc[0][0];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[sub][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_use) * cexp);
    const float mc = m_use * cexp;
    m_run = m_new;
    float psum = 0.f;
    bf16x8 pf[2][2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pe = __builtin_amdgcn_exp2f(fmaf(sacc[sub][r], cexp, -mc));
        psum += pe;
        pf[sub][r >> 3][r & 7] = static_cast<__bf16>(pe);
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;

    // ---- O^T += V^T P^T
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          bf16x8 a;
#pragma unroll
          for (int second = 0; second < 2; ++second) {
            const int row = 32 * sub + 16 * s + 8 * second + 4 * (gi >> 1) + tq;
            const int ch = 4 * dt + 2 * (gi & 1) + (tp >> 1);
            const bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                reinterpret_cast<__attribute__((address_space(3))) bf16x4*>(
                    reinterpret_cast<uintptr_t>(lds_v + lds_off(row, ch) + 8 * (tp & 1))));
            a[4 * second + 0] = t[0];
            a[4 * second + 1] = t[1];
            a[4 * second + 2] = t[2];
            a[4 * second + 3] = t[3];
          }
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf[sub][s], oacc[dt], 0, 0, 0);
        }
  }

  // ---- epilogue: normalise and store this lane's row (d = 32*dt + (r&3) + 8*(r>>2) + 4*h)
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
  if (my_row < Lq) {
    uint16_t* op = p.o + static_cast<int64_t>(q_start + my_row) * p.o_stride0 + hq * kHeadDim + 4 * h;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        u32x2 w;
        w[0] = pack_bf16x2(oacc[dt][4 * rg + 0] * inv, oacc[dt][4 * rg + 1] * inv);
        w[1] = pack_bf16x2(oacc[dt][4 * rg + 2] * inv, oacc[dt][4 * rg + 3] * inv);
        *reinterpret_cast<u32x2*>(op + 32 * dt + 8 * rg) = w;
      }
  }
}

int launch_prefill_attn(const PrefillParams& p, hipStream_t stream) {
  const unsigned qtiles = static_cast<unsigned>((p.max_seqlen_q + BM - 1) / BM);
  if (p.batch > 65535 || p.num_heads > 65535)
    return fail(LV_ERR_UNSUPPORTED, "prefill: batch or num_heads > 65535");
  const dim3 grid(qtiles, static_cast<unsigned>(p.num_heads), static_cast<unsigned>(p.batch));
  if (p.block_table) {
    hipLaunchKernelGGL(prefill_attn_kernel<true>, grid, dim3(256), 0, stream, p);
  } else {
    hipLaunchKernelGGL(prefill_attn_kernel<false>, grid, dim3(256), 0, stream, p);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(LV_ERR_LAUNCH, "prefill_attn launch: %s", hipGetErrorString(e));
  return LV_OK;
}

}  // namespace lv
