// Microbenchmark: cycles per v_mfma_f32_32x32x16_bf16 for different register-file placements of
// its operands, one wave per SIMD, issued back to back from inline asm.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define REP8(x) x x x x x x x x
template <int FORM>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, float* sink) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  f32x16 d0, d1;
  for (int i = 0; i < 16; ++i) { d0[i] = 0.f; d1[i] = 0.f; }
  asm volatile("" ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15",
               "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31",
               "a32","a33","a34","a35");
  asm volatile("v_accvgpr_write_b32 a32, %0\n v_accvgpr_write_b32 a33, %0\n v_accvgpr_write_b32 a34, %0\n v_accvgpr_write_b32 a35, %0\n s_nop 7" ::"v"(1.0f));
  for (int i = 0; i < 32; ++i) asm volatile("v_accvgpr_write_b32 a0, %0" ::"v"(0.f));
  asm volatile("s_nop 7");
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < 64; ++it) {
    if (FORM == 0) {        // D/C vgpr, A vgpr, B agpr; two chains
      REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, a[32:35], %0\n v_mfma_f32_32x32x16_bf16 %1, %2, a[32:35], %1" : "+v"(d0), "+v"(d1) : "v"(a));)
    } else if (FORM == 1) { // D/C agpr, A,B vgpr; two chains
      REP8(asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]\n v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" ::"v"(a), "v"(b));)
    } else if (FORM == 2) { // all vgpr
      REP8(asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1" : "+v"(d0), "+v"(d1) : "v"(a), "v"(b));)
    } else if (FORM == 3) { // as FORM 1 but separate asm statements (compiler may pad between them)
      REP8(asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" ::"v"(a), "v"(b)); asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" ::"v"(a), "v"(b));)
    } else if (FORM == 4) { // builtin, compiler-allocated
      REP8(d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d1, 0, 0, 0);)
    }
  }
  asm volatile("s_nop 15");
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  float acc = 0; for (int i = 0; i < 16; ++i) acc += d0[i] + d1[i];
  sink[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int F> void run(const char* name, unsigned long long* d, float* s) {
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<F>, dim3(256), dim3(256), 0, 0, d, s);
  hipDeviceSynchronize();
  unsigned long long h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double tot = 0; for (int i = 0; i < 256; ++i) tot += h[i];
  printf("%-44s %.1f cycles per MFMA\n", name, tot / 256 / (64 * 16));
}
int main() {
  unsigned long long* d; float* s; hipMalloc(&d, 256 * 8); hipMalloc(&s, 256 * 256 * 4);
  run<0>("D/C vgpr, A vgpr, B agpr (one asm, 2 chains)", d, s);
  run<1>("D/C agpr, A vgpr, B vgpr (one asm, 2 chains)", d, s);
  run<2>("all vgpr (one asm, 2 chains)", d, s);
  run<3>("D/C agpr, separate asm statements", d, s);
  run<4>("builtin", d, s);
  return 0;
}
