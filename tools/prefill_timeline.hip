// Diagnostic harness (not shipped): the life of the phase-locked prefill kernel's workgroups on a compute
// unit - prologue (Q fragments, first K/V tiles), tile loop, epilogue (output stores), and the gap until
// the next workgroup starts on the same compute unit. Build and run:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DLV_TIMELINE -Iinclude -Ilight_vllm_amd/csrc \
//         tools/prefill_timeline.hip -o tools/prefill_timeline && tools/prefill_timeline [seqlen [heads kv_heads [persistent]]]
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
  p.tile_order = nullptr; p.order_cols = 1; p.order_len = 0; p.order_rows = 0; p.order_lanes = 0;
  const bool persistent = argc > 4 && atoi(argv[4]) != 0;
  size_t n_slots = nwg;
  if (persistent) {
    // item table as prefill_order.build_item_table makes it for equal lengths: tiles dealt heaviest
    // first to the lightest lane (8 kv heads: one column)
    const int cols = lv::prefill_order_columns(Hkv), lanes = lv::prefill_order_lanes(Hq, Hkv);
    if (cols != 1) { printf("persistent timeline: 8 kv heads only\n"); return 1; }
    std::vector<std::vector<int>> lane_items(lanes);
    std::vector<long> load(lanes, 0);
    for (int t = qtiles - 1; t >= 0; --t)
      for (int b = 0; b < B; ++b) {
        int best = 0;
        for (int r = 1; r < lanes; ++r) if (load[r] < load[best]) best = r;
        lane_items[best].push_back(b * qtiles + t);
        load[best] += (t + 1) * 4;
      }
    size_t depth = 0;
    for (auto& v : lane_items) depth = std::max(depth, v.size());
    std::vector<int> table(depth * lanes * 8, 0);
    for (int r = 0; r < lanes; ++r)
      for (size_t k = 0; k < lane_items[r].size(); ++k) {
        const int b = lane_items[r][k] / qtiles, t = lane_items[r][k] % qtiles;
        int* e = &table[(k * lanes + r) * 8];
        e[0] = b * L + t * 256; e[1] = std::min(256, L - t * 256); e[2] = t * 256; e[3] = b * L; e[4] = L;
      }
    int* dtab; CK(hipMalloc(&dtab, table.size() * 4));
    CK(hipMemcpy(dtab, table.data(), table.size() * 4, hipMemcpyHostToDevice));
    p.tile_order = dtab; p.order_len = (int64_t)table.size(); p.order_rows = 256; p.order_lanes = lanes;
    n_slots = (size_t)Hq * lanes;
    printf("persistent: %d lanes x %zu tiles each, %zu workgroups\n", lanes, depth, n_slots);
  }
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
  CK(hipMemcpy(tl.data(), dtl, nwg * 64, hipMemcpyDeviceToHost));   // persistent launches fill the first n_slots entries
  if (persistent) {
    // per workgroup: [0] entry, [1] first tile arrived, [2] sum of seams, [3] sum of tile loops, [4] sum of
    // (wait for the copies + output stores), [5] end (100 MHz ticks)
    double seam = 0, loop = 0, fin = 0, cold = 0, life = 0; unsigned long long t0 = ~0ull, t1 = 0, first_end = ~0ull;
    for (size_t w = 0; w < n_slots; ++w) {
      const unsigned long long* d = &tl[w * 8];
      cold += (double)(d[1] - d[0]); seam += (double)d[2]; loop += (double)d[3]; fin += (double)d[4]; life += (double)(d[5] - d[0]);
      t0 = std::min(t0, d[0]); t1 = std::max(t1, d[5]); first_end = std::min(first_end, d[5]);
    }
    const double n = (double)n_slots, span = (double)(t1 - t0);
    printf("launch spans %.1f us; per workgroup: life %.1f us (first to end %.1f us before the last), cold start %.2f us,\n"
           "  seams %.2f us, tile loops %.2f us, copy wait + output stores %.2f us  (%.2f / %.2f / %.2f %% of the span)\n",
           span / 100, life / n / 100, (t1 - first_end) / 100.0, cold / n / 100, seam / n / 100, loop / n / 100, fin / n / 100,
           100 * seam / n / span, 100 * loop / n / span, 100 * fin / n / span);
    return 0;
  }
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
