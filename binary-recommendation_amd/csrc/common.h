// Shared host/device helpers for libbinrec_hip.so (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/binrec.h"

namespace br {

void set_error(const char* fmt, ...);

// Optional device-resident step state (hipGraph replays bake kernel arguments in, so the two per-step
// scalars — the dropout step counter and Adam's alpha_t — are then read from device memory).
// Set by brNeumfStepRun for the duration of a call when brNeumfStep.step_state != NULL.
struct StepStateDev {
  uint32_t step;
  float alpha_t;
  double pow_b1, pow_b2;             // beta1^step, beta2^step (running products: no pow() on the step's critical path)
  // alpha_j at [j & (BR_ALPHA_RING-1)] for the last BR_ALPHA_RING steps, followed by a mirror of the first BR_RING_MIRROR entries, so that
  // the BR_RING_MIRROR alphas from any index on are contiguous (the fast replay takes eight steps' alphas in one scalar load)
  float alpha_hist[BR_ALPHA_RING + BR_RING_MIRROR];
  // ---- replay form (brStepStateInit; all zero = the exact replay) ----
  uint32_t fast;                     // 1: fast replay (adam_math.h adam_replay_fast*), 0: the exact chain (bit-equal to the dense sweep)
  uint32_t trunc;                    // fast: theta is replayed over at most this many steps of a lag (the rest moves it by < 1 ulp of the first step)
  float sqrt_b2, eps_c;              // sqrt(beta2), eps (1 - sqrt(beta2)): d_j = sqrt(v_j) + eps obeys d_j = sqrt_b2 d_{j-1} + eps_c
  float pow1[BR_ALPHA_RING], pow2[BR_ALPHA_RING];   // beta1^k, beta2^k (k < BR_ALPHA_RING): the moments of a lag of k steps in one product
};
// arguments of the step-state advance (step += 1, alpha_t, ring entry, the step's double scratch zeroed): a launch of its own
// (brStepStateAdvance) or one extra workgroup of the chunk-rank launch (neumf_step.cpp defer_advance)
struct StepAdvance { StepStateDev* st = nullptr; double lr = 0, b1 = 0, b2 = 0; double* zero = nullptr; int64_t n_zero = 0; };
const StepStateDev* current_step_state();
void set_current_step_state(const StepStateDev* p);
void probe_split(int first_tag, hipStream_t s);   // launch probe of the step driver (neumf_step.cpp), a no-op unless a record is open
void probe_mark(hipStream_t s);                   // "the launch this call is tagged for comes next" (graph-resident probe; else a no-op)

#define BR_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      br::set_error(__VA_ARGS__);          \
      return BR_ERR_ARG;                   \
    }                                      \
  } while (0)

#define BR_CHECK_LAUNCH(name)                                                   \
  do {                                                                          \
    hipError_t e_ = hipGetLastError();                                          \
    if (e_ != hipSuccess) {                                                     \
      br::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
      return BR_ERR_HIP;                                                        \
    }                                                                           \
  } while (0)

constexpr int kWave = 64;

// Segmented arrays of the row-sharded exchange: both id streams of a step travel in ONE all-to-all buffer laid out
// [peer][stream][cap] (parallel.py PaddedExchange), so logical position t = peer * cap + j of stream k sits at element
// (peer * 2 + k) * cap + j = seg_phys(t, cap, 2 * cap, k * cap).  seg_len == 0: contiguous.
__device__ __host__ __forceinline__ int64_t seg_phys(int64_t t, int64_t seg_len, int64_t seg_stride, int64_t seg_off) {
  if (seg_len <= 0) return t;
  const int64_t q = t / seg_len;
  return q * seg_stride + seg_off + (t - q * seg_len);
}

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// id load: int32 or int64 storage, identity when ids == nullptr.
template <typename IdT>
__device__ __forceinline__ int64_t load_id(const IdT* ids, int64_t b) {
  return ids ? (int64_t)ids[b] : b;
}

// wave-wide and sub-group sums via DPP/shuffles (64-lane wavefront).
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int off = WIDTH / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, WIDTH);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float sigmoidf_stable(float x) {
  float e = __expf(-fabsf(x));
  // expf via v_exp_f32 (1 ulp-ish); division kept IEEE for the 1e-5 budget
  float r = 1.0f / (1.0f + e);
  return x >= 0.f ? r : e * r;
}

// hidden-layer activation on the hardware exp2 / rcp (v_exp_f32, v_rcp_f32: ~1 ulp each, |rel err| < 4e-7):
// the accurate expf + IEEE division cost ~40 VALU ops per element in the dense_fwd epilogue (2 us of a 36 us
// launch); parity with the oracle's exact sigmoid stays inside the 1e-5 budget (tests/test_gpu_neumf.py).
__device__ __forceinline__ float sigmoidf_hw(float x) {
  const float e = __expf(-fabsf(x));
  const float r = __builtin_amdgcn_rcpf(1.0f + e);
  return x >= 0.f ? r : e * r;
}

// accurate forms used where the loss / gradient tolerance is tight (head logit -> probability -> loss)
__device__ __forceinline__ float sigmoidf_acc(float x) {
  float e = expf(-fabsf(x));
  float r = 1.0f / (1.0f + e);
  return x >= 0.f ? r : e * r;
}

__device__ __forceinline__ float act_apply(float z, int act) {
  if (act == BR_ACT_SIGMOID) return sigmoidf_hw(z);
  if (act == BR_ACT_RELU) return fmaxf(z, 0.f);
  return z;
}
// derivative expressed through the activation OUTPUT a
__device__ __forceinline__ float act_grad_from_out(float a, int act) {
  if (act == BR_ACT_SIGMOID) return a * (1.f - a);
  if (act == BR_ACT_RELU) return a > 0.f ? 1.f : 0.f;
  return 1.f;
}

}  // namespace br
