// Microbenchmark: the sustained dense bf16 MFMA rate by instruction SHAPE (32x32x16 against 16x16x32).
// Same structure as mfma_peak.hip: Every SIMD of every CU issues
// v_mfma_f32_32x32x16_bf16 back to back from registers (no memory traffic at all) for a few
// milliseconds per launch; reports TFLOP/s from the wall time of the launch and the shader clock
// the waves saw (s_memtime ticks per 100 MHz s_memrealtime tick). The spec peak (2.5 PFLOP/s) is
// 1024 SIMDs x 1024 FLOP per cycle x 2.4 GHz; this measures the clock the chip keeps under that load.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_shape.hip -o tools/mfma_shape && tools/mfma_shape 1 1 1
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// random != 0: operands are per-lane pseudo-random values of magnitude ~1 (what attention feeds the
// pipe); 0: a smooth ramp (few bits toggle between lanes).
__global__ __launch_bounds__(256) void burn(int iters, float* sink, unsigned long long* clk, int random) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.25f); }
  if (random) {
    unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    for (int i = 0; i < 8; ++i) {
      s = s * 1664525u + 1013904223u; a[i] = (__bf16)(((int)(s >> 8) % 4096 - 2048) / 1024.0f);
      s = s * 1664525u + 1013904223u; b[i] = (__bf16)(((int)(s >> 8) % 4096 - 2048) / 1024.0f);
    }
  }
  f32x16 d0, d1, d2, d3;
  for (int i = 0; i < 16; ++i) { d0[i] = 0.f; d1[i] = 0.f; d2[i] = 0.f; d3[i] = 0.f; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d2, 0, 0, 0);
      d3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d3, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float acc = 0.f;
  for (int i = 0; i < 16; ++i) acc += d0[i] + d1[i] + d2[i] + d3[i];
  sink[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ __launch_bounds__(256) void burn16(int iters, float* sink, unsigned long long* clk, int random) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.25f); }
  if (random) {
    unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    for (int i = 0; i < 8; ++i) {
      s = s * 1664525u + 1013904223u; a[i] = (__bf16)(((int)(s >> 8) % 4096 - 2048) / 1024.0f);
      s = s * 1664525u + 1013904223u; b[i] = (__bf16)(((int)(s >> 8) % 4096 - 2048) / 1024.0f);
    }
  }
  f32x4 d[8];
  for (int j = 0; j < 8; ++j) d[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int k = 0; k < 8; ++k) d[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d[k], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float acc = 0.f;
  for (int k = 0; k < 8; ++k) acc += d[k][0] + d[k][1] + d[k][2] + d[k][3];
  sink[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
  const int wgs_per_cu = argc > 1 ? atoi(argv[1]) : 1;     // 1: one wave per SIMD, 2: two
  const int random = argc > 2 ? atoi(argv[2]) : 0;
  const int shape16 = argc > 3 ? atoi(argv[3]) : 0;   // 1: v_mfma_f32_16x16x32_bf16 (same FLOPs per loop iteration)
  const int blocks = 256 * wgs_per_cu, iters = 20000;
  float* sink; unsigned long long* clk;
  CK(hipMalloc(&sink, blocks * 256 * 4)); CK(hipMalloc(&clk, blocks * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0));
    if (shape16) hipLaunchKernelGGL(burn16, dim3(blocks), dim3(256), 0, 0, iters, sink, clk, random);
    else hipLaunchKernelGGL(burn, dim3(blocks), dim3(256), 0, 0, iters, sink, clk, random);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2 * 512]; CK(hipMemcpy(h, clk, blocks * 16, hipMemcpyDeviceToHost));
    double t = 0, r = 0; for (int i = 0; i < blocks; ++i) { t += h[2 * i]; r += h[2 * i + 1]; }
    const double flops = (double)blocks * 4 * iters * 16 * 32768.0;
    if (rep == 5) printf("RESULT {\"operands\": \"%s\", \"waves_per_simd\": %d, \"tflops\": %.1f, \"shader_mhz\": %.0f}\n",
                         random ? "random" : "smooth", wgs_per_cu, flops / (ms * 1e-3) / 1e12, 100.0 * t / r);
    printf("%s %s operands, launch %d: %.2f ms, %.0f TFLOP/s (%.1f %% of 2500), shader clock %.0f MHz, %.1f cycles per MFMA per SIMD\n", shape16 ? "16x16x32" : "32x32x16", random ? "random" : "smooth", rep, ms,
           flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15 * 100, 100.0 * t / r,
           t / blocks / (iters * 16.0) / wgs_per_cu);
  }
  return 0;
}
