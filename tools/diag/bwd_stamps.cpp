// Diagnostic build of the fused backward kernel with in-kernel stamps: where the dx waves (0-3) and the dW waves (4-7) of
// dense_bwd_kernel spend their cycles.  Built by tools/diag/build.sh into tools/diag/bwd_stamps.bin; never part of the library.
//   usage: bwd_stamps.bin [K N batch]
#define BR_STAMPS 1
#include "../../binary-recommendation_amd/csrc/api.cpp"
#include "../../binary-recommendation_amd/csrc/dense_bwd.hip"

#include <algorithm>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 128, N = argc > 2 ? atoi(argv[2]) : 100;
  const int64_t B = argc > 3 ? atoll(argv[3]) : 65536;
  const int ldn = (N + 3) & ~3, ldk = (K + 3) & ~3;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> hgy((size_t)B * ldn), hy((size_t)B * ldn), hx((size_t)B * ldk), hW((size_t)K * N), hn(N);
  for (auto& v : hgy) v = 1e-3f * nd(rng);
  for (auto& v : hy) v = 0.5f + 0.1f * nd(rng);
  for (auto& v : hx) v = nd(rng);
  for (auto& v : hW) v = 0.1f * nd(rng);
  for (auto& v : hn) v = 1.f + 0.1f * nd(rng);
  float *gy, *y, *x, *W, *gx, *slabs, *on; double* osum; uint32_t* keep; unsigned long long* st;
  const int kw = (K + 31) / 32;
  const int grid = br::dense_bwd_fused_grid(B);
  const int64_t slab_el = (int64_t)K * N + N;
  CK(hipMalloc(&gy, hgy.size() * 4)); CK(hipMalloc(&y, hy.size() * 4)); CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&W, hW.size() * 4));
  CK(hipMalloc(&gx, (size_t)B * ldk * 4)); CK(hipMalloc(&slabs, (size_t)grid * slab_el * 4)); CK(hipMalloc(&on, N * 4)); CK(hipMalloc(&osum, 8 * 2 * N * 8));
  CK(hipMalloc(&keep, (size_t)B * kw * 4)); CK(hipMalloc(&st, (size_t)grid * 8 * br::kStampSlots * 8));
  CK(hipMemcpy(gy, hgy.data(), hgy.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(y, hy.data(), hy.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(on, hn.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemset(osum, 0, 8 * 2 * N * 8)); CK(hipMemset(keep, 0xEF, (size_t)B * kw * 4));
  CK(hipMemset(st, 0, (size_t)grid * 8 * br::kStampSlots * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(br::g_stamp_buf), &st, sizeof(st)));
  br::BwdArgs a{};
  a.gy = gy; a.ldgy = ldn; a.y = y; a.ldy = ldn; a.x = x; a.ldx = ldk; a.W = W; a.batch = B; a.K = K; a.N = N; a.act = BR_ACT_SIGMOID;
  a.o_mean = on; a.o_rstd = on; a.o_gamma = on; a.o_sums = osum; a.inv_batch = 1.0f / (float)B;       // a BatchNorm behind this layer (as layer 1 of the NeuMF-A tower)
  a.keep = keep; a.kw = kw; a.inv_keep = 1.25f;
  a.gx = gx; a.ldgx = ldk; a.slabs = slabs; a.slab_elems = slab_el; a.db_off = (int64_t)K * N;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 100;
  for (int i = 0; i < 10; ++i) if (br::dense_backward_fused(a, nullptr) != 0) { printf("%s\n", brGetLastError()); return 1; }
  CK(hipEventRecord(e0, nullptr));
  for (int i = 0; i < iters; ++i) br::dense_backward_fused(a, nullptr);
  CK(hipEventRecord(e1, nullptr)); CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("{\"K\": %d, \"N\": %d, \"batch\": %lld, \"grid\": %d, \"us_per_launch\": %.2f,\n", K, N, (long long)B, grid, ms * 1e3 / iters);
  std::vector<unsigned long long> h((size_t)grid * 8 * br::kStampSlots);
  CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
  const char* dxn[] = {"entry->first gy/y loads issued", "staging issued", "barrier", "dz(0) published (waits for gy/y)", "next gy/y + keep words requested", "pass1 mfma", "pass1 epilogue", "pass2 mfma",
                       "pass2 epilogue", "-"};
  const int dxa[] = {0, 1, 2, 3, 4, 5, 6, 7, 8}, dxb[] = {1, 2, 3, 4, 5, 6, 7, 8, 9};
  for (int role = 0; role < 2; ++role) {
    printf(" \"%s waves, cycles [p10,p50,p90]\": {", role ? "dW" : "dx");
    // dx: 0,1,2,3,4,5,6,7,8(pass2 epi)->9 ; dW: 0 entry,1 loads,2 stage,3 barrier,4 T(x0) written,5 barrier0 passed,6 (it=1 start),7 product done,8 loop end
    const int na = role ? 7 : 9;
    const int A[2][9] = {{0, 1, 2, 3, 4, 9, 5, 6, 7}, {0, 1, 2, 3, 4, 5, 6, 7, 0}};
    const int Bb[2][9] = {{1, 2, 3, 4, 9, 5, 6, 7, 8}, {1, 2, 3, 4, 5, 6, 7, 8, 0}};
    const char* dwn[] = {"entry->first x loads issued", "staging issued", "barrier", "T(x0) written (waits for x)", "barrier S0 (waits for dz0)", "T(x1)+barrier.. to iteration 1 start",
                         "product of tile 0 (224 MFMAs)", "rest of the loop (tiles 1..)"};
    for (int sidx = 0; sidx < na + (role ? 1 : 0); ++sidx) {
      std::vector<long long> d;
      for (int64_t w = 0; w < (int64_t)grid * 8; ++w) {
        if ((int)(w % 8 >= 4) != role) continue;
        const unsigned long long t0 = h[w * br::kStampSlots + A[role][sidx]], t1 = h[w * br::kStampSlots + Bb[role][sidx]];
        if (t0 && t1 && t1 >= t0) d.push_back((long long)(t1 - t0));
      }
      std::sort(d.begin(), d.end());
      printf("%s\"%s\": [%lld, %lld, %lld]", sidx ? ", " : "", role ? dwn[sidx] : dxn[sidx], d.empty() ? 0 : d[d.size() / 10], d.empty() ? 0 : d[d.size() / 2], d.empty() ? 0 : d[d.size() * 9 / 10]);
    }
    printf("},\n");
  }
  unsigned long long t0min = ~0ull, t9max = 0;
  for (int64_t w = 0; w < (int64_t)grid * 8; ++w) if (h[w * br::kStampSlots + 10]) { t0min = std::min(t0min, h[w * br::kStampSlots + 10]); t9max = std::max(t9max, h[w * br::kStampSlots + 11]); }
  std::vector<long long> life;
  for (int64_t w = 0; w < (int64_t)grid * 8; ++w) if (h[w * br::kStampSlots + 10]) life.push_back((long long)(h[w * br::kStampSlots + 11] - h[w * br::kStampSlots + 10]));
  std::sort(life.begin(), life.end());
  printf(" \"kernel_span_us\": %.2f, \"wave_lifetime_10ns [p10,p50,p90]\": [%lld, %lld, %lld]}\n", (t9max - t0min) * 0.01, life[life.size() / 10], life[life.size() / 2], life[life.size() * 9 / 10]);
  return 0;
}
