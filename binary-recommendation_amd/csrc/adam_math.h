// O1 arithmetic shared by the optimizer kernels (sparse_opt.hip) and the catch-up gather (gather.hip).
//
// [TF-sem] Keras Adam: m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; theta -= alpha_t m / (sqrt(v) + eps),
// alpha_t = lr sqrt(1-b2^t)/(1-b1^t).  Its sparse apply is NOT lazy: a row without a gradient still gets the
// g = 0 update every step.  "Deferred" mode keeps that result without sweeping the table every step:
// each row remembers the last step its (theta, m, v) include (`last[row]`); whoever needs the row replays
// the missing g = 0 steps in registers with the SAME fp32 operations the dense sweep would have issued
// (adam_decay with that step's alpha from the ring in StepStateDev), so the values are bit-equal to the
// swept table.
#pragma once
#include "common.h"
#include "rows.h"

namespace br {

struct AdamHp {
  float alpha, b1, omb1, b2, omb2, eps;
  const float* alpha_ptr;   // non-null: alpha_t lives in device memory (hipGraph replays)
};
__device__ __forceinline__ void adam_resolve(AdamHp& h) {
  if (h.alpha_ptr) h.alpha = *h.alpha_ptr;
}

// 1 / (sqrt(v) + eps) on the hardware sqrt / rcp (1 ulp each; the IEEE expansions cost ~20 VALU ops per
// element and the deferred replay is VALU-bound).  The error is relative to the UPDATE (alpha m / (..)),
// i.e. ~1e-7 of a step of size <= alpha: far inside the 1e-5 parity budget on the parameters.
__device__ __forceinline__ float adam_inv_denom(float v, float eps) {
  return __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + eps);
}

// every product / sum below is spelled out (explicit fma, no contraction left to the compiler) so that the
// same row history gives the same bits in every kernel that inlines these.
__device__ __forceinline__ void adam_update1(float& th, float& m, float& v, float g, const AdamHp& h) {
  m = __builtin_fmaf(h.omb1, g, h.b1 * m);
  v = __builtin_fmaf(h.omb2, g * g, h.b2 * v);
  th = __builtin_fmaf(-(h.alpha * m), adam_inv_denom(v, h.eps), th);
}
__device__ __forceinline__ void adam_update(float4& th, float4& m, float4& v, float4 g, const AdamHp& h) {
  adam_update1(th.x, m.x, v.x, g.x, h); adam_update1(th.y, m.y, v.y, g.y, h);
  adam_update1(th.z, m.z, v.z, g.z, h); adam_update1(th.w, m.w, v.w, g.w, h);
}
__device__ __forceinline__ void adam_update(float2& th, float2& m, float2& v, float2 g, const AdamHp& h) {
  adam_update1(th.x, m.x, v.x, g.x, h); adam_update1(th.y, m.y, v.y, g.y, h);
}
__device__ __forceinline__ void adam_update(float& th, float& m, float& v, float g, const AdamHp& h) {
  adam_update1(th, m, v, g, h);
}

// the g = 0 step (rows without a gradient): m = b1 m ; v = b2 v ; theta -= alpha m / (sqrt(v) + eps).
// Two elements per instruction on the packed-fp32 VALU ops (v_pk_mul/fma_f32: same IEEE results as the
// scalar forms, so VEC = 1 tables get the same bits).
typedef float pk2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void adam_decay_pk(pk2& th, pk2& m, pk2& v, float alpha, const AdamHp& h) {
  m = m * h.b1;
  v = v * h.b2;
  // adam_inv_denom on both elements, the + eps as one packed add (same IEEE sum per element)
  pk2 d = {__builtin_amdgcn_sqrtf(v.x), __builtin_amdgcn_sqrtf(v.y)};
  d = d + h.eps;
  const pk2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  th = __builtin_elementwise_fma(-(m * alpha), r, th);
}
__device__ __forceinline__ void adam_decay(float& th, float& m, float& v, float alpha, const AdamHp& h) {
  m = m * h.b1;
  v = v * h.b2;
  th = __builtin_fmaf(-(m * alpha), adam_inv_denom(v, h.eps), th);
}
__device__ __forceinline__ void adam_decay(float2& th, float2& m, float2& v, float alpha, const AdamHp& h) {
  pk2 t = {th.x, th.y}, mm = {m.x, m.y}, vv = {v.x, v.y};
  adam_decay_pk(t, mm, vv, alpha, h);
  th = make_float2(t.x, t.y); m = make_float2(mm.x, mm.y); v = make_float2(vv.x, vv.y);
}
__device__ __forceinline__ void adam_decay(float4& th, float4& m, float4& v, float alpha, const AdamHp& h) {
  pk2 t0 = {th.x, th.y}, m0 = {m.x, m.y}, v0 = {v.x, v.y}, t1 = {th.z, th.w}, m1 = {m.z, m.w}, v1 = {v.z, v.w};
  adam_decay_pk(t0, m0, v0, alpha, h);
  adam_decay_pk(t1, m1, v1, alpha, h);
  th = make_float4(t0.x, t0.y, t1.x, t1.y); m = make_float4(m0.x, m0.y, m1.x, m1.y); v = make_float4(v0.x, v0.y, v1.x, v1.y);
}

__device__ __forceinline__ bool all_zero(float4 a) { return a.x == 0.f && a.y == 0.f && a.z == 0.f && a.w == 0.f; }
__device__ __forceinline__ bool all_zero(float2 a) { return a.x == 0.f && a.y == 0.f; }
__device__ __forceinline__ bool all_zero(float a) { return a == 0.f; }

// Deferred-mode view of one table: moments, per-row `last` step and the device step state.
struct CatchUp {
  const float* m;            // same layout / row stride as the table (NULL: plain table, no replay)
  const float* v;
  const int32_t* last;       // [rows]
  const StepStateDev* ss;
  float b1, b2, eps;
};

// apply the g = 0 steps (from, upto] to one vector of a row.  m == v == 0 (row never touched): the update
// is the identity, skipped exactly.
// `ring` = the alpha ring staged in LDS (stage_alpha_ring): the loop is a dependent chain per step, a global
// load per iteration would put an L2 round trip into each of them.
template <typename V>
__device__ __forceinline__ void adam_replay(V& th, V& m, V& v, uint32_t from, uint32_t upto, const float* ring, const AdamHp& h) {
  if (all_zero(m) && all_zero(v)) return;
  for (uint32_t j = from + 1; j <= upto; ++j) adam_decay(th, m, v, ring[j & (BR_ALPHA_RING - 1)], h);
}

// the same replay when `from` / `upto` are WAVE-UNIFORM (one row per wave): alpha comes through scalar loads from the ring in global
// memory, requested one step ahead (no LDS image per workgroup), and a row nobody has touched yet (m == v == 0 in every lane) is
// skipped by a scalar branch.  No per-lane skip: a lane with m == v == 0 runs theta = fma(-0, 1/eps, theta) = theta, the same bits.
template <typename V>
__device__ __forceinline__ void adam_replay_uniform(V& th, V& m, V& v, uint32_t from, uint32_t upto, const StepStateDev* __restrict__ ss,
                                                    const AdamHp& h) {
  if (__builtin_amdgcn_ballot_w64(!(all_zero(m) && all_zero(v))) == 0) return;
  float a = ss->alpha_hist[(from + 1) & (BR_ALPHA_RING - 1)];
  for (uint32_t j = from + 1; j <= upto; ++j) {
    const float an = ss->alpha_hist[(j + 1) & (BR_ALPHA_RING - 1)];
    adam_decay(th, m, v, a, h);
    a = an;
  }
}

// whole workgroup: copy the ring into LDS (call before any early return; ends with a barrier)
__device__ __forceinline__ void stage_alpha_ring(float* lds_ring, const StepStateDev* ss) {
  for (int k = threadIdx.x; k < BR_ALPHA_RING; k += blockDim.x) lds_ring[k] = ss->alpha_hist[k];
  __syncthreads();
}

static inline AdamHp make_hp(double alpha, double b1, double b2, double eps) {
  AdamHp h;
  h.alpha = (float)alpha;
  h.b1 = (float)b1;
  h.omb1 = (float)(1.0 - b1);
  h.b2 = (float)b2;
  h.omb2 = (float)(1.0 - b2);
  h.eps = (float)eps;
  const StepStateDev* ss = current_step_state();
  h.alpha_ptr = ss ? &ss->alpha_t : nullptr;
  return h;
}

}  // namespace br
