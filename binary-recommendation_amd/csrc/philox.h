// Philox4x32-10 counter RNG: the dropout-mask definition of this build.
//
// Keras' Dropout stream (trainers/NFC_plain.py:138,141,144; src/models/NeuMFModel.py:67,71,75)
// is a stateful TF generator that cannot be reproduced, so masks here are a pure function.
// One Philox call yields 4 x u32 = 8 x u16 draws for the 8 columns 8q..8q+7 of a row; element
// (global row r, col c) of dropout site `site` at optimizer step `step` is KEPT iff
//   u16 = (philox(key = (seed_lo, seed_hi), ctr = (r_lo, c >> 3, site, step))[(c >> 1) & 3] >> 16*(c & 1)) & 0xFFFF
//   u16 >= floor(p * 65536)
// (rows < 2^32 per step).  oracle/binrec_oracle.py::dropout_mask restates this bit-exactly and
// pins the generator to the published Random123 known-answer vectors.
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

// threshold: u16 draw < thr => dropped
__host__ __device__ __forceinline__ uint32_t dropout_threshold(float p) {
  double t = (double)p * 65536.0;
  if (t >= 65535.0) return 0xFFFFu;
  return (uint32_t)t;  // floor
}

struct DropoutCfg {
  uint32_t k0, k1;   // seed lo/hi
  uint32_t step;
  uint32_t site;
  uint32_t thr;      // 0 => dropout off
  float inv_keep;    // 1/(1-p)
  const uint32_t* step_ptr;   // non-null: the step counter lives in device memory (hipGraph replays)
  uint32_t step_add;          // added to the resolved counter (keep-bit planes prefetched for the next step)
};

__host__ inline DropoutCfg make_dropout(float p, uint64_t seed, uint32_t step, uint32_t site) {
  DropoutCfg c;
  c.k0 = (uint32_t)(seed & 0xFFFFFFFFu);
  c.k1 = (uint32_t)(seed >> 32);
  c.step = step;
  c.site = site;
  c.thr = p > 0.f ? dropout_threshold(p) : 0u;
  c.inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  c.step_ptr = nullptr;
  c.step_add = 0;
  return c;
}

// kernels call this once on their by-value copy
__device__ __forceinline__ void dropout_resolve(DropoutCfg& c) {
  if (c.step_ptr) c.step = *c.step_ptr;
  c.step += c.step_add;
}

// keep-bits (bit i = column 8*c8 + i kept) of the 8 columns of chunk c8 of global row r
__device__ __forceinline__ uint32_t dropout_keep8(const DropoutCfg& c, int64_t r, uint32_t c8) {
  if (c.thr == 0u) return 0xFFu;
  const Philox4 d = philox4x32_10((uint32_t)r, c8, c.site, c.step, c.k0, c.k1);
  uint32_t bits = 0;
  bits |= ((d.x & 0xFFFFu) >= c.thr) ? 1u : 0u;
  bits |= ((d.x >> 16) >= c.thr) ? 2u : 0u;
  bits |= ((d.y & 0xFFFFu) >= c.thr) ? 4u : 0u;
  bits |= ((d.y >> 16) >= c.thr) ? 8u : 0u;
  bits |= ((d.z & 0xFFFFu) >= c.thr) ? 16u : 0u;
  bits |= ((d.z >> 16) >= c.thr) ? 32u : 0u;
  bits |= ((d.w & 0xFFFFu) >= c.thr) ? 64u : 0u;
  bits |= ((d.w >> 16) >= c.thr) ? 128u : 0u;
  return bits;
}
__device__ __forceinline__ float dropout_scale1(const DropoutCfg& c, int64_t r, uint32_t col) {
  if (c.thr == 0u) return 1.0f;
  return ((dropout_keep8(c, r, col >> 3) >> (col & 7u)) & 1u) ? c.inv_keep : 0.0f;
}


// ---- keep-bit planes (dropout masks materialised once per (step, site): uint32 [rows][ceil(K/32)], bit c of row r) ----
struct KeepSite { uint32_t* out; int K, kw; uint32_t site; };
struct KeepArgs {
  KeepSite s[3];
  int n_sites;
  DropoutCfg drop;      // key, step (or step_ptr) (+ step_add), threshold
  int64_t row0, batch;
};
// the planes as extra workgroups of another kernel's grid (horizontal fusion): workgroups [0, blocks[0]) fill site 0, the next
// blocks[1] site 1, ...; total == 0: nothing fused
struct KeepFuse { KeepArgs a; int blocks[3]; int total; };

// one thread per (row, 32-column word) of site `si` (256 threads per block): its 4 Philox calls (one per 8-column chunk) run
// INTERLEAVED, round by round - a Philox round is two dependent 32x32->64 multiplies, so four independent chains per lane are
// what keeps the multiplier busy (one chain at a time behind per-chunk branches measured 9.6 us for the 128-column plane of
// 65 536 rows) - then one coalesced store.  Chunks past K produce bits that nobody reads.  `a.drop` must be resolved.
__device__ __forceinline__ void keep_bits_block(const KeepArgs& a, int si, int64_t block) {
  const KeepSite s = si == 0 ? a.s[0] : (si == 1 ? a.s[1] : a.s[2]);    // (a runtime index would copy the struct to scratch)
  const int64_t idx = block * 256 + threadIdx.x;
  if (idx >= a.batch * s.kw) return;
  const int64_t r = idx / s.kw;
  const uint32_t w = (uint32_t)(idx - r * s.kw);
  uint32_t c0[4], c1[4], c2[4], c3[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { c0[c] = (uint32_t)(a.row0 + r); c1[c] = 4 * w + c; c2[c] = s.site; c3[c] = a.drop.step; }
  uint32_t k0 = a.drop.k0, k1 = a.drop.k1;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // 32x32 -> hi / lo as v_mul_hi_u32 + v_mul_lo_u32: the 64-bit product form compiles to v_mad_u64_u32, which measured
      // ~78 cycles per wave instruction on gfx950
      const uint32_t h0 = __umulhi(0xD2511F53u, c0[c]), l0 = 0xD2511F53u * c0[c];
      const uint32_t h1 = __umulhi(0xCD9E8D57u, c2[c]), l1 = 0xCD9E8D57u * c2[c];
      const uint32_t n0 = h1 ^ c1[c] ^ k0, n2 = h0 ^ c3[c] ^ k1;
      c0[c] = n0; c1[c] = l1; c2[c] = n2; c3[c] = l0;
    }
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  const uint32_t thr = a.drop.thr;
  uint32_t word = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint32_t d[4] = {c0[c], c1[c], c2[c], c3[c]};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      word |= ((d[e] & 0xFFFFu) >= thr ? 1u : 0u) << (8 * c + 2 * e);
      word |= ((d[e] >> 16) >= thr ? 1u : 0u) << (8 * c + 2 * e + 1);
    }
  }
  s.out[idx] = word;
}

// a fused grid's workgroup `wg` (0 <= wg < f.total): which site / which block of it
__device__ __forceinline__ void keep_fuse_block(KeepFuse f, int64_t wg) {
  dropout_resolve(f.a.drop);
  if (wg < f.blocks[0]) keep_bits_block(f.a, 0, wg);
  else if (wg < f.blocks[0] + f.blocks[1]) keep_bits_block(f.a, 1, wg - f.blocks[0]);
  else keep_bits_block(f.a, 2, wg - f.blocks[0] - f.blocks[1]);
}

}  // namespace br
