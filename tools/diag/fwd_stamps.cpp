// Diagnostic build of the forward kernel with in-kernel stamps (s_memtime): where a wave of dense_fwd_kernel spends its
// cycles.  Built by tools/diag/build.sh into tools/diag/fwd_stamps.bin (travels to the GPU box); never part of the library.
//   usage: fwd_stamps.bin [K N batch act drop]
#define BR_STAMPS 1
#include "../../binary-recommendation_amd/csrc/api.cpp"
#include "../../binary-recommendation_amd/csrc/dense_fwd.hip"

#include <algorithm>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 128, N = argc > 2 ? atoi(argv[2]) : 100;
  const int64_t B = argc > 3 ? atoll(argv[3]) : 65536;
  const int act = argc > 4 ? atoi(argv[4]) : 1;
  const float drop = argc > 5 ? (float)atof(argv[5]) : 0.2f;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> hx((size_t)B * K), hW((size_t)K * N), hb(N), hs(K), hh(K);
  for (auto& v : hx) v = nd(rng);
  for (auto& v : hW) v = 0.1f * nd(rng);
  for (auto& v : hb) v = 0.1f * nd(rng);
  for (auto& v : hs) v = 1.f + 0.1f * nd(rng);
  for (auto& v : hh) v = 0.1f * nd(rng);
  float *x, *W, *b, *y, *sc, *sh; uint32_t* keep; double* stats; unsigned long long* st;
  const int64_t nwaves = ((B + 15) / 16 + 7) / 8 * 8;
  CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&W, hW.size() * 4)); CK(hipMalloc(&b, N * 4)); CK(hipMalloc(&y, (size_t)B * ((N + 3) & ~3) * 4));
  CK(hipMalloc(&sc, K * 4)); CK(hipMalloc(&sh, K * 4)); CK(hipMalloc(&keep, (size_t)brDropoutKeepWords(B, K) * 4));
  CK(hipMalloc(&stats, 8 * 2 * N * 8)); CK(hipMalloc(&st, (size_t)nwaves * br::kStampSlots * 8));
  CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sc, hs.data(), K * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(sh, hh.data(), K * 4, hipMemcpyHostToDevice));
  CK(hipMemset(stats, 0, 8 * 2 * N * 8)); CK(hipMemset(st, 0, (size_t)nwaves * br::kStampSlots * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(br::g_stamp_buf), &st, sizeof(st)));
  if (drop > 0.f) {
    const uint32_t sites[1] = {0}; const int widths[1] = {K}; uint32_t* outs[1] = {keep};
    if (brDropoutKeepBits(drop, 1234, 1, 0, B, 1, sites, widths, outs, nullptr) != 0) { printf("%s\n", brGetLastError()); return 1; }
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // warm the clocks with back-to-back launches, then time and keep the LAST launch's stamps
  const int iters = 200;
  for (int i = 0; i < 20; ++i) brDenseForward(x, K, W, b, y, (N + 3) & ~3, B, K, N, act, sc, sh, nullptr, drop, drop > 0.f ? keep : nullptr, stats, nullptr);
  CK(hipEventRecord(e0, nullptr));
  for (int i = 0; i < iters; ++i)
    if (brDenseForward(x, K, W, b, y, (N + 3) & ~3, B, K, N, act, sc, sh, nullptr, drop, drop > 0.f ? keep : nullptr, stats, nullptr) != 0) { printf("%s\n", brGetLastError()); return 1; }
  CK(hipEventRecord(e1, nullptr)); CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("{\"K\": %d, \"N\": %d, \"batch\": %lld, \"act\": %d, \"drop\": %.2f, \"us_per_launch\": %.2f,\n", K, N, (long long)B, act, drop, ms * 1e3 / iters);
  std::vector<unsigned long long> h((size_t)nwaves * br::kStampSlots);
  CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
  const char* names[] = {"entry->A loads issued", "W image loads+writes issued", "barrier", "first T() (waits for A)", "pass1 mfma", "pass1 epilogue", "pass2 mfma",
                         "pass2 epilogue", "tail (2nd tile, stats)"};
  const int idx[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9};
  // s_memrealtime (100 MHz, one counter for the chip) for cross-wave offsets; s_memtime (shader clock) inside a wave
  unsigned long long t0min = ~0ull, t9max = 0;
  std::vector<double> clk;
  for (int64_t w = 0; w < nwaves; ++w) if (h[w * br::kStampSlots]) {
    t0min = std::min(t0min, h[w * br::kStampSlots + 10]); t9max = std::max(t9max, h[w * br::kStampSlots + 11]);
    const double rt = (double)(h[w * br::kStampSlots + 11] - h[w * br::kStampSlots + 10]);
    if (rt > 0) clk.push_back((double)(h[w * br::kStampSlots + 9] - h[w * br::kStampSlots]) / rt * 0.1);
  }
  std::sort(clk.begin(), clk.end());
  printf(" \"kernel_span_us(first entry -> last exit, s_memrealtime)\": %.2f, \"in_kernel_clock_GHz_median\": %.3f,\n \"segments_cycles [p10,p50,p90]\": {", (t9max - t0min) * 0.01,
         clk.empty() ? 0.0 : clk[clk.size() / 2]);
  for (int s = 0; s < 9; ++s) {
    std::vector<long long> d;
    for (int64_t w = 0; w < nwaves; ++w) {
      const unsigned long long a = h[w * br::kStampSlots + idx[s]], c = h[w * br::kStampSlots + idx[s + 1]];
      if (a && c) d.push_back((long long)(c - a));
    }
    std::sort(d.begin(), d.end());
    printf("%s\"%s\": [%lld, %lld, %lld]", s ? ", " : "", names[s], d.empty() ? 0 : d[d.size() / 10], d.empty() ? 0 : d[d.size() / 2], d.empty() ? 0 : d[d.size() * 9 / 10]);
  }
  // when do waves start / end relative to the first entry (dispatch skew)
  std::vector<long long> st0, st9;
  for (int64_t w = 0; w < nwaves; ++w) if (h[w * br::kStampSlots]) { st0.push_back((long long)(h[w * br::kStampSlots + 10] - t0min)); st9.push_back((long long)(h[w * br::kStampSlots + 11] - t0min)); }
  std::sort(st0.begin(), st0.end()); std::sort(st9.begin(), st9.end());
  printf("},\n \"wave_entry_offset_10ns [p10,p50,p90,max]\": [%lld, %lld, %lld, %lld], \"wave_exit_offset_10ns [p10,p50,p90,max]\": [%lld, %lld, %lld, %lld]}\n",
         st0[st0.size() / 10], st0[st0.size() / 2], st0[st0.size() * 9 / 10], st0.back(), st9[st9.size() / 10], st9[st9.size() / 2], st9[st9.size() * 9 / 10], st9.back());
  return 0;
}
