// Shared device/host helpers for the gfx950 paged-attention kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lv_attention.h"

namespace lv {

constexpr int kHeadDim = 128;   // every Qwen3 size
constexpr int kWave = 64;       // gfx950 wavefront

typedef uint32_t u32;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// Two bf16 packed in a dword -> their two f32 values (exact).
__device__ __forceinline__ float bf16_lo(u32 w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(u32 w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// f32 -> bf16 bits, round-to-nearest-even, NaN preserved (hardware conversion on gfx950).
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float x) {
  __bf16 b = static_cast<__bf16>(x);
  return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ u32 pack_bf16x2(float lo, float hi) {
  return static_cast<u32>(f32_to_bf16_bits(lo)) | (static_cast<u32>(f32_to_bf16_bits(hi)) << 16);
}

// acc + a.lo*b.lo + a.hi*b.hi with f32 accumulation (v_dot2c_f32_bf16).
__device__ __forceinline__ float dot2_bf16(u32 a, u32 b, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a),
                                         __builtin_bit_cast(bf16x2_t, b), acc, false);
}

// DPP lane exchange inside a row of 16 lanes (folds into the consuming VALU op).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}
constexpr int kDppXor1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // row_half_mirror: lane i <-> 7-i inside 8 lanes
constexpr int kDppMirror = 0x140;      // row_mirror: lane i <-> 15-i inside 16 lanes

// Sum over the 16 lanes of a DPP row; every lane of the row ends with the total.
__device__ __forceinline__ float row16_allreduce_sum(float x) {
  x += dpp_f32<kDppXor1>(x);
  x += dpp_f32<kDppXor2>(x);
  x += dpp_f32<kDppHalfMirror>(x);
  x += dpp_f32<kDppMirror>(x);
  return x;
}

// Host-side error recording (defined in api.hip).
int fail(int code, const char* fmt, ...);

}  // namespace lv
