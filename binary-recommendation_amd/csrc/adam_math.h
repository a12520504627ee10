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

// ---- fast replay (StepStateDev::fast, include/binrec.h BR_REPLAY_FAST) --------------------------------------------------------------
// The same recurrence in a cheaper form.  For the g = 0 steps v_j = b2^j v exactly, so sqrt(v_j) = c^j sqrt(v) with c = sqrt(b2) and
// d_j = sqrt(v_j) + eps obeys d_j = c d_{j-1} + eps (1 - c): one v_sqrt per row visit instead of one per step.  1 / d_j comes from
// 1 / d_{j-1} by one Newton step r <- r (2 - d r): d_j / d_{j-1} lies in [c, 1], so the guess is off by <= 1 - c = 5e-4 and the
// result by its square, 2.5e-7 - the error class of the v_rcp_f32 the exact form uses, and no quarter-rate op in the loop
// (measured, tools/diag/replay_bench.cpp: 0.015 against 0.027 ns per row of 128 floats and step).  m decays by b1 per step: after
// `trunc` steps the remaining updates of theta are below one ulp of the first (7 (b1 / c)^trunc / (1 - b1 / c) <= 2^-23, 192 steps at
// the Keras defaults), so theta is replayed over min(lag, trunc) steps only, and the moments of the whole lag are one product each
// with b^lag from the tables in the step state.
struct FastRp {
  float c, eps_c;
  uint32_t trunc;
};
__device__ __forceinline__ FastRp fast_rp(const StepStateDev* __restrict__ ss) { return FastRp{ss->sqrt_b2, ss->eps_c, ss->trunc}; }

__device__ __forceinline__ void fast_open(pk2 v, float eps, pk2& d, pk2& r) {
  d = (pk2){__builtin_amdgcn_sqrtf(v.x), __builtin_amdgcn_sqrtf(v.y)} + eps;
  r = (pk2){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}
__device__ __forceinline__ void fast_open(float v, float eps, float& d, float& r) {
  d = __builtin_amdgcn_sqrtf(v) + eps;
  r = __builtin_amdgcn_rcpf(d);
}
__device__ __forceinline__ void fast_step(pk2& th, pk2& m, pk2& d, pk2& r, float alpha, float b1, const FastRp& f) {
  m = m * b1;
  d = __builtin_elementwise_fma(d, (pk2){f.c, f.c}, (pk2){f.eps_c, f.eps_c});
  r = r * __builtin_elementwise_fma(-d, r, (pk2){2.f, 2.f});
  th = __builtin_elementwise_fma(-(m * alpha), r, th);
}
__device__ __forceinline__ void fast_step(float& th, float& m, float& d, float& r, float alpha, float b1, const FastRp& f) {
  m = m * b1;
  d = __builtin_fmaf(d, f.c, f.eps_c);
  r = r * __builtin_fmaf(-d, r, 2.f);
  th = __builtin_fmaf(-(m * alpha), r, th);
}
// the replay state of one lane's vector of a row: running theta, running m, d = sqrt(v_j) + eps, r = 1 / d
template <typename V> struct FastSt;
template <> struct FastSt<float> {
  float th, m, d, r;
  __device__ __forceinline__ void open(float t, float mm, float v, float eps) { th = t; m = mm; fast_open(v, eps, d, r); }
  __device__ __forceinline__ void step(float a, float b1, const FastRp& f) { fast_step(th, m, d, r, a, b1, f); }
  __device__ __forceinline__ float theta() const { return th; }
};
template <> struct FastSt<float2> {
  pk2 th, m, d, r;
  __device__ __forceinline__ void open(float2 t, float2 mm, float2 v, float eps) { th = (pk2){t.x, t.y}; m = (pk2){mm.x, mm.y}; fast_open((pk2){v.x, v.y}, eps, d, r); }
  __device__ __forceinline__ void step(float a, float b1, const FastRp& f) { fast_step(th, m, d, r, a, b1, f); }
  __device__ __forceinline__ float2 theta() const { return make_float2(th.x, th.y); }
};
template <> struct FastSt<float4> {
  pk2 th0, m0, d0, r0, th1, m1, d1, r1;
  __device__ __forceinline__ void open(float4 t, float4 mm, float4 v, float eps) {
    th0 = (pk2){t.x, t.y}; m0 = (pk2){mm.x, mm.y}; fast_open((pk2){v.x, v.y}, eps, d0, r0);
    th1 = (pk2){t.z, t.w}; m1 = (pk2){mm.z, mm.w}; fast_open((pk2){v.z, v.w}, eps, d1, r1);
  }
  __device__ __forceinline__ void step(float a, float b1, const FastRp& f) { fast_step(th0, m0, d0, r0, a, b1, f); fast_step(th1, m1, d1, r1, a, b1, f); }
  __device__ __forceinline__ float4 theta() const { return make_float4(th0.x, th0.y, th1.x, th1.y); }
};
__device__ __forceinline__ uint32_t pow_index(uint32_t k) { return k < (uint32_t)BR_ALPHA_RING ? k : (uint32_t)BR_ALPHA_RING - 1u; }

// the moments of a row after `lag` g = 0 steps, one product each (every kernel forms them from the STORED m, v this way, so a row's
// moments do not depend on which kernel caught it up)
template <typename V>
__device__ __forceinline__ void fast_moments(V& m, V& v, uint32_t lag, const StepStateDev* __restrict__ ss) {
  const uint32_t k = pow_index(lag);
  m = vmul(m, ss->pow1[k]);
  v = vmul(v, ss->pow2[k]);
}

// fast form of adam_replay_uniform: steps (from, upto] of theta; m, v are left as stored (fast_moments brings them up).  Eight steps'
// alphas per scalar load (s_load_dwordx8 from the mirrored ring): with one load per step the loop waits for a scalar-cache round trip
// in every iteration.
template <typename V>
__device__ __forceinline__ void adam_replay_fast_uniform(V& th, const V& m, const V& v, uint32_t from, uint32_t upto, const StepStateDev* __restrict__ ss,
                                                         const AdamHp& h) {
  if (__builtin_amdgcn_ballot_w64(!(all_zero(m) && all_zero(v))) == 0) return;
  const FastRp f = fast_rp(ss);
  const uint32_t lag = upto - from;
  const uint32_t end = from + (lag < f.trunc ? lag : f.trunc);           // last step theta is replayed for
  FastSt<V> st;
  st.open(th, m, v, h.eps);
  uint32_t j = from + 1;
  for (; j + 7 <= end; j += 8) {
    const float* __restrict__ rp = ss->alpha_hist + (j & (BR_ALPHA_RING - 1));
    float a8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a8[k] = rp[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) st.step(a8[k], h.b1, f);
  }
  for (; j <= end; ++j) st.step(ss->alpha_hist[j & (BR_ALPHA_RING - 1)], h.b1, f);
  th = st.theta();
}
// per-lane form (row-group kernels: several rows with different lags per wave; alphas from the ring image in LDS)
template <typename V>
__device__ __forceinline__ void adam_replay_fast(V& th, const V& m, const V& v, uint32_t from, uint32_t upto, const float* ring, const FastRp& f,
                                                 const AdamHp& h) {
  if (all_zero(m) && all_zero(v)) return;
  const uint32_t lag = upto - from;
  const uint32_t end = from + (lag < f.trunc ? lag : f.trunc);
  FastSt<V> st;
  st.open(th, m, v, h.eps);
  for (uint32_t j = from + 1; j <= end; ++j) st.step(ring[j & (BR_ALPHA_RING - 1)], h.b1, f);
  th = st.theta();
}

// Replay of steps (from, upto] in the form the step state asks for.  MOMENTS: m and v are brought to step `upto` too (the optimizer
// and the flush store them; a lookup only needs theta).
template <bool MOMENTS, typename V>
__device__ __forceinline__ void adam_catch_up_uniform(V& th, V& m, V& v, uint32_t from, uint32_t upto, const StepStateDev* __restrict__ ss, const AdamHp& h) {
  if (ss->fast) {
    adam_replay_fast_uniform(th, m, v, from, upto, ss, h);
    if (MOMENTS) fast_moments(m, v, upto - from, ss);
  } else {
    adam_replay_uniform(th, m, v, from, upto, ss, h);
  }
}
template <bool MOMENTS, typename V>
__device__ __forceinline__ void adam_catch_up(V& th, V& m, V& v, uint32_t from, uint32_t upto, const float* ring, const StepStateDev* __restrict__ ss,
                                              const AdamHp& h) {
  if (ss->fast) {
    adam_replay_fast(th, m, v, from, upto, ring, fast_rp(ss), h);
    if (MOMENTS) fast_moments(m, v, upto - from, ss);
  } else {
    adam_replay(th, m, v, from, upto, ring, h);
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
