// Diagnostic harness (not shipped): builds the prefill kernel with s_memtime stamps and prints the
// share of wave time spent in each loop segment. Build and run:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DLV_STAMPS -Iinclude -Ilight_vllm_amd/csrc \
//         tools/prefill_diag.hip -o tools/prefill_diag && tools/prefill_diag [seqlen [kernel_hint [heads kv_heads]]]
// kernel_hint: an LV_PREFILL_KERNEL_* value (0 = the library's choice, 3 = phase-locked, 5 = 64 rows per wave).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../light_vllm_amd/csrc/prefill_attn.hip"  // defines LV_ABL (0 unless -DLV_ABL=n)

namespace lv {
int fail(int code, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr);
  return code;
}
}  // namespace lv

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static uint16_t rnd_bf16(uint64_t& s) {
  s = s * 6364136223846793005ull + 1442695040888963407ull;
  float f = ((int)((s >> 40) & 0xffff) - 32768) / 16384.0f;  // roughly uniform [-2, 2)
  uint32_t u; memcpy(&u, &f, 4);
  return (uint16_t)(u >> 16);
}

int main(int argc, char** argv) {
  const int B = 4, L = argc > 1 ? atoi(argv[1]) : 4096, D = 128;
  const int Hq = argc > 3 ? atoi(argv[3]) : 32, Hkv = argc > 4 ? atoi(argv[4]) : 8;
  const int hint = argc > 2 ? atoi(argv[2]) : LV_PREFILL_KERNEL_AUTO;
  const size_t nq = (size_t)B * L * Hq * D, nk = (size_t)B * L * Hkv * D;
  std::vector<uint16_t> hq(nq), hk(nk), hv(nk);
  uint64_t seed = 1;
  for (auto& x : hq) x = rnd_bf16(seed);
  for (auto& x : hk) x = rnd_bf16(seed);
  for (auto& x : hv) x = rnd_bf16(seed);
  std::vector<int> cu(B + 1);
  for (int i = 0; i <= B; ++i) cu[i] = i * L;
  uint16_t *dq, *dk, *dv, *dout; int* dcu; unsigned long long* dst;
  CK(hipMalloc(&dq, nq * 2)); CK(hipMalloc(&dk, nk * 2)); CK(hipMalloc(&dv, nk * 2)); CK(hipMalloc(&dout, nq * 2));
  CK(hipMalloc(&dcu, (B + 1) * 4));
  CK(hipMemcpy(dq, hq.data(), nq * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dk, hk.data(), nk * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dv, hv.data(), nk * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dcu, cu.data(), (B + 1) * 4, hipMemcpyHostToDevice));
  const int qtiles = (L + 127) / 128;
  const size_t nst = (size_t)B * Hq * qtiles * 128;
  CK(hipMalloc(&dst, nst * 8)); CK(hipMemset(dst, 0, nst * 8));
#ifdef LV_STAMPS
  CK(hipMemcpyToSymbol(HIP_SYMBOL(lv::g_lv_stamp_buf), &dst, sizeof(dst)));
#endif
  lv::PrefillParams p{};
  p.q = dq; p.k = dk; p.v = dv; p.o = dout;
  p.q_stride0 = p.o_stride0 = Hq * D; p.k_stride0 = p.v_stride0 = Hkv * D;
  p.cu_q = dcu; p.cu_k = dcu; p.block_table = nullptr; p.bt_stride = 0; p.block_size = 0;
  p.batch = B; p.max_seqlen_q = L; p.num_heads = Hq; p.num_kv_heads = Hkv;
  p.kernel_hint = hint;
  p.nsplit = 1; p.ws_o = nullptr; p.ws_ml = nullptr; p.total_q = (int64_t)B * L;
  p.scale_log2e = 0.08838834764f * 1.4426950408889634f;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> times;
  for (int it = 0; it < 30; ++it) {
    CK(hipEventRecord(e0));
    if (lv::launch_prefill_attn(p, nullptr) != 0) return 1;
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    times.push_back(ms);
  }
  std::sort(times.begin(), times.end());
  printf("30 launches: min %.3f ms, median %.3f ms (diagnostic build LV_ABL=%d)\n", times[0], times[15], LV_ABL);
#ifndef LV_STAMPS
  return 0;
#endif
  std::vector<unsigned long long> st(nst);
  CK(hipMemcpy(st.data(), dst, nst * 8, hipMemcpyDeviceToHost));
  double seg[16] = {0}; double tot = 0;
  for (size_t i = 0; i < nst; ++i) { seg[i % 16] += (double)st[i]; tot += (double)st[i]; }
  const char* names[16] = {"barrier (wait for all waves)", "ds_write stage", "-", "issue next tile's loads",
                           "K_A reads, QK_A, K_B+V_A reads", "region 2: PV_A | exp(B)", "PV_B", "-",
                           "max(A) (+rescale)", "region 1: QK_B | exp(A)", "V_B reads issue, max(B)", "-", "-", "-", "-", "-"};
  const char* names_pp[16] = {"L_K: rescale + exp half", "C_QK: 16 MFMA | exp", "L_V: mask, max, decision",
                              "C_PV: 16 MFMA | exp", "barrier after L_K", "barrier after C_QK", "barrier after L_V",
                              "barrier after C_PV", "L_V: K publish + loads", "L_V: V reads (+wait)", "L_K: V publish + loads", "L_K: K reads (+wait)", "-", "-", "-", "-"};
  const char* names_w64[16] = {"barrier", "pending rescale + DMA wait", "-", "tile bases", "QK gaps 17-31",
                               "PV gaps 17-31", "PV: V reads + gap 0 (settle, mask)", "-", "PV gaps 1-2 (decisions)", "QK: K reads + gap 0", "QK gaps 1-16 (4 DMA)", "PV gaps 3-16", "-", "-", "-", "-"};
  const bool is_pp = hint == LV_PREFILL_KERNEL_PP || (hint == LV_PREFILL_KERNEL_AUTO && L >= 2048);
  if (is_pp) for (int i = 0; i < 16; ++i) names[i] = names_pp[i];
  if (hint == LV_PREFILL_KERNEL_W64) { for (int i = 0; i < 16; ++i) names[i] = names_w64[i]; tot -= seg[14]; seg[14] = 0; }
  if (seg[13] > 0) printf("in-kernel clock: %.0f MHz (shader ticks per 100 MHz reference tick, averaged over waves)\n", 100.0 * seg[12] / seg[13]);
  tot -= seg[12] + seg[13]; seg[12] = seg[13] = 0;
  tot -= seg[15];
  for (int i = 0; i < 15; ++i)
    if (seg[i] > 0) printf("%-34s %6.2f %%   %8.0f ticks per (wave, tile)\n", names[i], 100.0 * seg[i] / tot, seg[i] / seg[15]);
  printf("total %.0f ticks per (wave, tile); %.0f (wave, tile) units\n", tot / seg[15], seg[15]);
  return 0;
}
