// store_kvcache for gfx950: scatter this step's K/V rows into the paged cache by slot_mapping.
//
// Semantics follow the reference's Triton kernel (lightvllm/layers/attention.py:35-52):
// one cache row of num_kv_heads*head_dim elements per token, destination row = slot_mapping[i].
// This kernel moves 16 bytes per lane; a row of C 16-byte chunks is covered by C consecutive
// lanes, so reads and writes are coalesced in whole-row (>= 256 B) runs. Offsets are 64-bit.
// Slots < 0 or >= num_slots are skipped.
#include "lv_kernels.h"

namespace lv {

__global__ __launch_bounds__(256) void store_kvcache_kernel(
    const u32x4* __restrict__ key, const u32x4* __restrict__ value,
    int64_t key_stride_c, int64_t value_stride_c,  // row strides in 16-byte chunks
    u32x4* __restrict__ k_cache, u32x4* __restrict__ v_cache,
    const int32_t* __restrict__ slot_mapping,
    int64_t num_tokens, int chunks_per_row, int64_t num_slots) {
  const int64_t total = num_tokens * chunks_per_row;
  const int64_t step = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
       idx += step) {
    const int64_t tok = idx / chunks_per_row;
    const int c = static_cast<int>(idx - tok * chunks_per_row);
    const int64_t slot = slot_mapping[tok];
    if (slot < 0 || slot >= num_slots) continue;
    const u32x4 kx = key[tok * key_stride_c + c];
    const u32x4 vx = value[tok * value_stride_c + c];
    const int64_t dst = slot * chunks_per_row + c;
    k_cache[dst] = kx;
    v_cache[dst] = vx;
  }
}

int launch_store_kvcache(const void* key, const void* value, int64_t key_stride0,
                         int64_t value_stride0, void* k_cache, void* v_cache,
                         const int32_t* slot_mapping, int64_t num_tokens, int row_elems,
                         int64_t num_slots, hipStream_t stream) {
  const int chunks = row_elems / 8;
  const int64_t total = num_tokens * chunks;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;  // grid-stride beyond 32 waves/CU x 256 CUs
  hipLaunchKernelGGL(store_kvcache_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                     stream, static_cast<const u32x4*>(key), static_cast<const u32x4*>(value),
                     key_stride0 / 8, value_stride0 / 8, static_cast<u32x4*>(k_cache),
                     static_cast<u32x4*>(v_cache), slot_mapping, num_tokens, chunks, num_slots);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(LV_ERR_LAUNCH, "store_kvcache launch: %s", hipGetErrorString(e));
  return LV_OK;
}

}  // namespace lv
