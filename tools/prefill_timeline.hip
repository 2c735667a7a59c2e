// Diagnostic harness (not shipped): the life of the phase-locked prefill kernel's workgroups on a compute
// unit - prologue (Q fragments, first K/V tiles), tile loop, epilogue (output stores), and the gap until
// the next workgroup starts on the same compute unit. Build and run:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DLV_TIMELINE -Iinclude -Ilight_vllm_amd/csrc \
//         tools/prefill_timeline.hip -o tools/prefill_timeline && tools/prefill_timeline [seqlen [heads kv_heads]]
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


#include <map>
int main(int argc, char** argv) {
  const int B = 4, L = argc > 1 ? atoi(argv[1]) : 4096, D = 128;
  const int Hq = argc > 2 ? atoi(argv[2]) : 32, Hkv = argc > 3 ? atoi(argv[3]) : 8;
  const size_t nq = (size_t)B * L * Hq * D, nk = (size_t)B * L * Hkv * D;
  std::vector<uint16_t> hq(nq), hk(nk), hv(nk);
  uint64_t seed = 1;
  for (auto& x : hq) x = rnd_bf16(seed);
  for (auto& x : hk) x = rnd_bf16(seed);
  for (auto& x : hv) x = rnd_bf16(seed);
  std::vector<int> cu(B + 1);
  for (int i = 0; i <= B; ++i) cu[i] = i * L;
  uint16_t *dq, *dk, *dv, *dout; int* dcu; unsigned long long* dtl;
  CK(hipMalloc(&dq, nq * 2)); CK(hipMalloc(&dk, nk * 2)); CK(hipMalloc(&dv, nk * 2)); CK(hipMalloc(&dout, nq * 2));
  CK(hipMalloc(&dcu, (B + 1) * 4));
  CK(hipMemcpy(dq, hq.data(), nq * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dk, hk.data(), nk * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dv, hv.data(), nk * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dcu, cu.data(), (B + 1) * 4, hipMemcpyHostToDevice));
  const int qtiles = (L + 255) / 256;
  const size_t nwg = (size_t)B * Hq * qtiles;
  CK(hipMalloc(&dtl, nwg * 64)); CK(hipMemset(dtl, 0, nwg * 64));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(lv::g_lv_tl_buf), &dtl, sizeof(dtl)));
  lv::PrefillParams p{};
  p.q = dq; p.k = dk; p.v = dv; p.o = dout;
  p.q_stride0 = p.o_stride0 = Hq * D; p.k_stride0 = p.v_stride0 = Hkv * D;
  p.cu_q = dcu; p.cu_k = dcu; p.block_table = nullptr; p.bt_stride = 0; p.block_size = 0;
  p.batch = B; p.max_seqlen_q = L; p.num_heads = Hq; p.num_kv_heads = Hkv;
  p.kernel_hint = LV_PREFILL_KERNEL_PP;
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
  const float last_ms = times.back();
  std::sort(times.begin(), times.end());
  printf("30 launches: min %.3f ms, median %.3f ms, last %.3f ms\n", times[0], times[15], last_ms);
  std::vector<unsigned long long> tl(nwg * 8);
  CK(hipMemcpy(tl.data(), dtl, nwg * 64, hipMemcpyDeviceToHost));
  // group the last launch's workgroups by compute unit, order by entry time
  std::map<unsigned long long, std::vector<size_t>> by_cu;
  unsigned long long tmin = ~0ull, tmax = 0;
  for (size_t w = 0; w < nwg; ++w) {
    const unsigned long long* d = &tl[w * 8];
    if (d[0] == 0) continue;
    by_cu[((d[7] & 0xf) << 16) | ((d[6] >> 8) & 0xff)].push_back(w);
    tmin = std::min(tmin, d[0]); tmax = std::max(tmax, d[5]);
  }
  double seg[6] = {0}; double n = 0, gap = 0, ngap = 0, first = 0, lastend = 0;
  for (auto& kv : by_cu) {
    auto& v = kv.second;
    std::sort(v.begin(), v.end(), [&](size_t a, size_t b) { return tl[a * 8] < tl[b * 8]; });
    for (size_t i = 0; i < v.size(); ++i) {
      const unsigned long long* d = &tl[v[i] * 8];
      for (int s = 0; s < 5; ++s) seg[s] += (double)(d[s + 1] - d[s]);
      n += 1;
      if (i + 1 < v.size()) { gap += (double)(tl[v[i + 1] * 8] - d[5]); ngap += 1; }
    }
    first += (double)(tl[v.front() * 8] - tmin);
    lastend += (double)(tmax - tl[v.back() * 8 + 5]);
  }
  const char* names[5] = {"entry -> Q fragments ready", "-> first K/V tiles published, loop starts", "tile loop", "-> output stores issued", "-> stores retired"};
  printf("%zu compute units, %.1f workgroups each; launch spans %.1f us (100 MHz ticks)\n", by_cu.size(), n / by_cu.size(), (tmax - tmin) / 100.0);
  for (int s = 0; s < 5; ++s) printf("%-44s %8.2f us per workgroup, %6.2f %% of the span\n", names[s], seg[s] / n / 100.0, 100.0 * seg[s] / by_cu.size() / (double)(tmax - tmin));
  printf("%-44s %8.2f us per hand-over, %6.2f %% of the span\n", "gap to the next workgroup on the same CU", gap / ngap / 100.0, 100.0 * gap / by_cu.size() / (double)(tmax - tmin));
  printf("%-44s %8.2f us, idle after its last workgroup %.2f us\n", "first workgroup after the launch's first", first / by_cu.size() / 100.0, lastend / by_cu.size() / 100.0);
  return 0;
}
