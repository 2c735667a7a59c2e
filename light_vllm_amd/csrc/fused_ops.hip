// Memory-bound steps either side of the attention call, one launch each instead of a chain of
// elementwise torch kernels. Every kernel keeps the reference's rounding points (fp32 math, one
// rounding to bf16 where the reference's tensors change dtype), so results differ from the
// unfused torch ops only through the summation order of the mean of squares.
//
//   qk_norm_rope_store_kernel  per-head RMSNorm of q and k, neox RoPE, K/V scatter into the paged
//                              cache (reference: models/qwen3.py:100-113, layers/layernorm.py:28-50,
//                              layers/rotary_embedding.py:5-42, layers/attention.py:35-52)
//   add_rms_norm_kernel        residual add + RMSNorm (layers/layernorm.py:28-80)
//   silu_and_mul_kernel        silu(gate) * up (layers/activation.py:7-29)
#include "lv_kernels.h"

namespace lv {

// Products and sums that the reference evaluates as separate torch ops must not be contracted
// into fused multiply-adds: __fmul_rn / __fadd_rn / __fsub_rn are never contracted.
__device__ __forceinline__ float round_bf16(float x) {
  return __builtin_bit_cast(float, static_cast<u32>(f32_to_bf16_bits(x)) << 16);
}

// ------------------------------------------------------------------------------------------
// q/k RMSNorm + RoPE + cache store.
// One head (128 elements) per 16-lane row: lane l holds elements [4l, 4l+4) of the first half
// and the same offsets of the second half, so the rotation partner of every element is in the
// same lane and the mean of squares is one DPP row reduction. Work item = (token, head slot),
// head slots ordered q heads | k heads | v heads; v heads are copied to the cache untouched.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void qk_norm_rope_store_kernel(QkNormRopeParams p) {
  const int lane16 = threadIdx.x & 15;
  const int64_t item = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 4;
  const int slots_per_token = p.num_heads + 2 * p.num_kv_heads;
  if (item >= p.num_tokens * slots_per_token) return;   // whole 16-lane rows leave together
  const int64_t tok = item / slots_per_token;
  const int hs = static_cast<int>(item - tok * slots_per_token);

  int64_t slot = -1;
  if (p.slot_mapping != nullptr) {
    slot = p.slot_mapping[tok];
    if (slot >= p.num_slots) slot = -1;
  }
  const int64_t cache_row = static_cast<int64_t>(p.num_kv_heads) * kHeadDim;

  if (hs >= p.num_heads + p.num_kv_heads) {             // V head: copy to the cache
    if (slot < 0) return;
    const int h = hs - p.num_heads - p.num_kv_heads;
    const uint16_t* src = p.v + tok * p.v_stride0 + h * kHeadDim;
    uint16_t* dst = p.v_cache + slot * cache_row + h * kHeadDim;
    const u32x2 a = *reinterpret_cast<const u32x2*>(src + 4 * lane16);
    const u32x2 b = *reinterpret_cast<const u32x2*>(src + 64 + 4 * lane16);
    *reinterpret_cast<u32x2*>(dst + 4 * lane16) = a;
    *reinterpret_cast<u32x2*>(dst + 64 + 4 * lane16) = b;
    return;
  }

  const bool is_k = hs >= p.num_heads;
  const int h = is_k ? hs - p.num_heads : hs;
  uint16_t* row = is_k ? p.k + tok * p.k_stride0 + h * kHeadDim : p.q + tok * p.q_stride0 + h * kHeadDim;
  const uint16_t* wrow = is_k ? p.k_weight : p.q_weight;

  const u32x2 a = *reinterpret_cast<const u32x2*>(row + 4 * lane16);
  const u32x2 b = *reinterpret_cast<const u32x2*>(row + 64 + 4 * lane16);
  const u32x2 wa = *reinterpret_cast<const u32x2*>(wrow + 4 * lane16);
  const u32x2 wb = *reinterpret_cast<const u32x2*>(wrow + 64 + 4 * lane16);
  int64_t pos = p.positions[tok];
  pos = pos < 0 ? 0 : (pos >= p.max_position ? p.max_position - 1 : pos);
  const float* cs = p.cos_sin + pos * kHeadDim;
  const float4 c = *reinterpret_cast<const float4*>(cs + 4 * lane16);
  const float4 s = *reinterpret_cast<const float4*>(cs + 64 + 4 * lane16);

  float x1[4] = {bf16_lo(a.x), bf16_hi(a.x), bf16_lo(a.y), bf16_hi(a.y)};
  float x2[4] = {bf16_lo(b.x), bf16_hi(b.x), bf16_lo(b.y), bf16_hi(b.y)};
  const float w1[4] = {bf16_lo(wa.x), bf16_hi(wa.x), bf16_lo(wa.y), bf16_hi(wa.y)};
  const float w2[4] = {bf16_lo(wb.x), bf16_hi(wb.x), bf16_lo(wb.y), bf16_hi(wb.y)};
  const float cc[4] = {c.x, c.y, c.z, c.w};
  const float ss[4] = {s.x, s.y, s.z, s.w};

  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) sq += x1[i] * x1[i] + x2[i] * x2[i];
  sq = row16_allreduce_sum(sq);
  const float r = rsqrtf(__fadd_rn(sq * (1.0f / kHeadDim), p.eps));

  float o1[4], o2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // normalise in fp32, round to bf16, scale by the bf16 weight, round again
    const float n1 = round_bf16(__fmul_rn(round_bf16(__fmul_rn(x1[i], r)), w1[i]));
    const float n2 = round_bf16(__fmul_rn(round_bf16(__fmul_rn(x2[i], r)), w2[i]));
    // rotate the (i, i + 64) pair in fp32, round once
    o1[i] = __fsub_rn(__fmul_rn(n1, cc[i]), __fmul_rn(n2, ss[i]));
    o2[i] = __fadd_rn(__fmul_rn(n2, cc[i]), __fmul_rn(n1, ss[i]));
  }
  u32x2 ra, rb;
  ra.x = pack_bf16x2(o1[0], o1[1]);
  ra.y = pack_bf16x2(o1[2], o1[3]);
  rb.x = pack_bf16x2(o2[0], o2[1]);
  rb.y = pack_bf16x2(o2[2], o2[3]);
  *reinterpret_cast<u32x2*>(row + 4 * lane16) = ra;
  *reinterpret_cast<u32x2*>(row + 64 + 4 * lane16) = rb;
  if (is_k && slot >= 0) {
    uint16_t* dst = p.k_cache + slot * cache_row + h * kHeadDim;
    *reinterpret_cast<u32x2*>(dst + 4 * lane16) = ra;
    *reinterpret_cast<u32x2*>(dst + 64 + 4 * lane16) = rb;
  }
}

int launch_qk_norm_rope_store(const QkNormRopeParams& p, hipStream_t stream) {
  const int64_t items = p.num_tokens * (p.num_heads + 2 * p.num_kv_heads);
  const int64_t blocks = (items + 15) / 16;
  if (blocks > INT32_MAX) return fail(LV_ERR_UNSUPPORTED, "qk_norm_rope_store: grid too large");
  hipLaunchKernelGGL(qk_norm_rope_store_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                     stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(LV_ERR_LAUNCH, "qk_norm_rope_store launch: %s", hipGetErrorString(e));
  return LV_OK;
}

// ------------------------------------------------------------------------------------------
// residual add + RMSNorm. One 256-thread workgroup per row; a thread keeps its 16-byte chunks
// (tid, tid + 256, ...) in registers between the sum of squares and the scaling, so the row is
// read once. hidden <= 8 * 256 * kMaxChunks.
// ------------------------------------------------------------------------------------------
constexpr int kMaxChunks = 4;

template <bool HAS_RES>
__global__ __launch_bounds__(256) void add_rms_norm_kernel(
    const u32x4* __restrict__ x, int64_t x_stride_c, u32x4* residual, int64_t res_stride_c,
    const u32x4* __restrict__ weight, u32x4* out, int64_t out_stride_c, int chunks, float eps,
    float inv_hidden) {
  __shared__ float wave_sums[4];
  const int tid = threadIdx.x;
  const int64_t rowi = blockIdx.x;
  const u32x4* xr = x + rowi * x_stride_c;
  u32x4* rr = HAS_RES ? residual + rowi * res_stride_c : nullptr;
  u32x4* orow = out + rowi * out_stride_c;

  float v[kMaxChunks][8];
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < kMaxChunks; ++j) {
    const int c = tid + j * 256;
    if (c < chunks) {
      const u32x4 xv = xr[c];
      const u32 xw[4] = {xv.x, xv.y, xv.z, xv.w};
      if (HAS_RES) {
        const u32x4 rv = rr[c];
        const u32 rw[4] = {rv.x, rv.y, rv.z, rv.w};
        u32 nw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[j][2 * i] = __fadd_rn(bf16_lo(xw[i]), bf16_lo(rw[i]));
          v[j][2 * i + 1] = __fadd_rn(bf16_hi(xw[i]), bf16_hi(rw[i]));
          nw[i] = pack_bf16x2(v[j][2 * i], v[j][2 * i + 1]);
        }
        u32x4 nr;
        nr.x = nw[0]; nr.y = nw[1]; nr.z = nw[2]; nr.w = nw[3];
        rr[c] = nr;                 // new residual: the fp32 sum rounded once
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[j][2 * i] = bf16_lo(xw[i]);
          v[j][2 * i + 1] = bf16_hi(xw[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) sq += v[j][i] * v[j][i];
    }
  }
  // wave reduction, then across the 4 waves through LDS
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off);
  if ((tid & 63) == 0) wave_sums[tid >> 6] = sq;
  __syncthreads();
  const float total = (wave_sums[0] + wave_sums[1]) + (wave_sums[2] + wave_sums[3]);
  const float r = rsqrtf(__fadd_rn(total * inv_hidden, eps));

#pragma unroll
  for (int j = 0; j < kMaxChunks; ++j) {
    const int c = tid + j * 256;
    if (c < chunks) {
      const u32x4 wv = weight[c];
      const u32 ww[4] = {wv.x, wv.y, wv.z, wv.w};
      u32 ow[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float lo = __fmul_rn(round_bf16(__fmul_rn(v[j][2 * i], r)), bf16_lo(ww[i]));
        const float hi = __fmul_rn(round_bf16(__fmul_rn(v[j][2 * i + 1], r)), bf16_hi(ww[i]));
        ow[i] = pack_bf16x2(lo, hi);
      }
      u32x4 ov;
      ov.x = ow[0]; ov.y = ow[1]; ov.z = ow[2]; ov.w = ow[3];
      orow[c] = ov;
    }
  }
}

int launch_add_rms_norm(const void* x, void* residual, const void* weight, void* out,
                        int64_t x_stride0, int64_t res_stride0, int64_t out_stride0,
                        int64_t num_tokens, int hidden, float eps, hipStream_t stream) {
  const int chunks = hidden / 8;
  if (chunks > 256 * kMaxChunks)
    return fail(LV_ERR_UNSUPPORTED, "add_rms_norm: hidden %d > %d", hidden, 8 * 256 * kMaxChunks);
  if (num_tokens > INT32_MAX) return fail(LV_ERR_UNSUPPORTED, "add_rms_norm: too many rows");
  const dim3 grid(static_cast<unsigned>(num_tokens)), block(256);
  const float inv_hidden = static_cast<float>(1.0 / hidden);
  if (residual)
    hipLaunchKernelGGL(add_rms_norm_kernel<true>, grid, block, 0, stream,
                       static_cast<const u32x4*>(x), x_stride0 / 8, static_cast<u32x4*>(residual),
                       res_stride0 / 8, static_cast<const u32x4*>(weight), static_cast<u32x4*>(out),
                       out_stride0 / 8, chunks, eps, inv_hidden);
  else
    hipLaunchKernelGGL(add_rms_norm_kernel<false>, grid, block, 0, stream,
                       static_cast<const u32x4*>(x), x_stride0 / 8, static_cast<u32x4*>(nullptr),
                       int64_t{0}, static_cast<const u32x4*>(weight), static_cast<u32x4*>(out),
                       out_stride0 / 8, chunks, eps, inv_hidden);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(LV_ERR_LAUNCH, "add_rms_norm launch: %s", hipGetErrorString(e));
  return LV_OK;
}

// ------------------------------------------------------------------------------------------
// silu(gate) * up: 16 bytes of gate and of up per lane, grid-stride over rows x chunks.
// silu is evaluated in fp32 and rounded to bf16 before the product, as torch's bf16 silu does
// (to within the hardware exp2/reciprocal's 1 ulp before that rounding).
// ------------------------------------------------------------------------------------------
// x / (1 + exp(-x)) with the hardware exp2 and reciprocal (1 ulp each) instead of the library
// exp and an IEEE division: the result is rounded to bf16 (8 significant bits) right away, and
// the kernel, ~25 vector instructions per element with the exact forms, was bound by them
// rather than by HBM (16384 x 12288: 303 us = 4.0 TB/s exact, 227 us = 5.3 TB/s this way; two
// chunks per trip and nontemporal loads on top measured the same).
__device__ __forceinline__ float silu_f32(float x) {
  const float e = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

__global__ __launch_bounds__(256) void silu_and_mul_kernel(
    const u32x4* __restrict__ x, int64_t x_stride_c, u32x4* __restrict__ out, int64_t out_stride_c,
    int64_t num_tokens, int chunks) {
  const int64_t total = num_tokens * chunks;
  const int64_t step = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
       idx += step) {
    const int64_t tok = idx / chunks;
    const int c = static_cast<int>(idx - tok * chunks);
    const u32x4 g = x[tok * x_stride_c + c];
    const u32x4 u = x[tok * x_stride_c + chunks + c];
    const u32 gw[4] = {g.x, g.y, g.z, g.w};
    const u32 uw[4] = {u.x, u.y, u.z, u.w};
    u32 ow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float g0 = bf16_lo(gw[i]), g1 = bf16_hi(gw[i]);
      const float s0 = round_bf16(silu_f32(g0));
      const float s1 = round_bf16(silu_f32(g1));
      ow[i] = pack_bf16x2(__fmul_rn(s0, bf16_lo(uw[i])), __fmul_rn(s1, bf16_hi(uw[i])));
    }
    u32x4 ov;
    ov.x = ow[0]; ov.y = ow[1]; ov.z = ow[2]; ov.w = ow[3];
    out[tok * out_stride_c + c] = ov;
  }
}

int launch_silu_and_mul(const void* x, void* out, int64_t x_stride0, int64_t out_stride0,
                        int64_t num_tokens, int intermediate, hipStream_t stream) {
  const int chunks = intermediate / 8;
  const int64_t total = num_tokens * chunks;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(silu_and_mul_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream,
                     static_cast<const u32x4*>(x), x_stride0 / 8, static_cast<u32x4*>(out),
                     out_stride0 / 8, num_tokens, chunks);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(LV_ERR_LAUNCH, "silu_and_mul launch: %s", hipGetErrorString(e));
  return LV_OK;
}

}  // namespace lv
