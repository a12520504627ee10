// Philox4x32-10 counter RNG: the dropout-mask definition of this build.
//
// Keras' Dropout stream (trainers/NFC_plain.py:138,141,144; src/models/NeuMFModel.py:67,71,75)
// is a stateful TF generator that cannot be reproduced, so masks here are a pure function:
// element (global row r, col c) of dropout site `site` at optimizer step `step` is KEPT iff
//   philox(key = (seed_lo, seed_hi), ctr = (r_lo, c >> 2, site, step))[c & 3] >= floor(p * 2^32)
// (rows < 2^32 per step).  oracle/binrec_oracle.py::dropout_mask restates this bit-exactly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace br {

struct Philox4 {
  uint32_t x, y, z, w;
};

__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                          uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}

// threshold: draw < thr => dropped
__host__ __device__ __forceinline__ uint32_t dropout_threshold(float p) {
  double t = (double)p * 4294967296.0;
  if (t >= 4294967295.0) return 0xFFFFFFFFu;
  return (uint32_t)t;  // floor
}

struct DropoutCfg {
  uint32_t k0, k1;   // seed lo/hi
  uint32_t step;
  uint32_t site;
  uint32_t thr;      // 0 => dropout off
  float inv_keep;    // 1/(1-p)
};

__host__ inline DropoutCfg make_dropout(float p, uint64_t seed, uint32_t step, uint32_t site) {
  DropoutCfg c;
  c.k0 = (uint32_t)(seed & 0xFFFFFFFFu);
  c.k1 = (uint32_t)(seed >> 32);
  c.step = step;
  c.site = site;
  c.thr = p > 0.f ? dropout_threshold(p) : 0u;
  c.inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  return c;
}

// 4 keep-flags for columns 4*cq .. 4*cq+3 of global row r.
__device__ __forceinline__ Philox4 dropout_draw4(const DropoutCfg& c, int64_t r, uint32_t cq) {
  return philox4x32_10((uint32_t)r, cq, c.site, c.step, c.k0, c.k1);
}
__device__ __forceinline__ float dropout_scale1(const DropoutCfg& c, int64_t r, uint32_t col) {
  if (c.thr == 0u) return 1.0f;
  Philox4 d = dropout_draw4(c, r, col >> 2);
  uint32_t lane = col & 3u;
  uint32_t v = lane == 0 ? d.x : lane == 1 ? d.y : lane == 2 ? d.z : d.w;
  return v >= c.thr ? c.inv_keep : 0.0f;
}

}  // namespace br
