// Argument blocks and host entries shared between translation units: the fused one-launch backward (dense_bwd.hip, called from
// brDenseBackward in mlp.hip) and the MFMA tail (tail_mfma.hip, called from brNeumfTailFused in tail.hip).
#pragma once
#include "common.h"
#include "philox.h"
#include <stdlib.h>

namespace br {

// ---- fp32 products on the bf16 matrix pipe ("bf16x6") -------------------------------------------------------------------------------
// The tower's GEMMs are fp32 (the 1e-5 bar on logits forbids bf16 DATA), and v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 MFMA rate.
// An fp32 number is the exact sum of three bf16 pieces, x = h + m + l (8 + 8 + 8 significant bits; |x - h - m - l| <= 2^-24 |x|), and a
// product of two bf16 numbers is exact in fp32.  x w = hh + (hm + mh) + (hl + lh + mm) + O(2^-24): six v_mfma_f32_16x16x32_bf16 per
// 32-deep k-block (96 matrix-pipe cycles) replace eight fp32 MFMAs (256 cycles).  The dropped terms (ml, lm, ll and the split residuals)
// are <= 4 x 2^-24 of each product - measured on the layer shapes 4e-9 of sum |x w| against 2.5e-7 for a 128-term fp32 fma chain
// (whose one rounding per term dominates: the bf16 path rounds once per six products of a 32-deep block) - so the result is at least
// as close to the real-number product as the exact-fp32 MFMA's.  BR_MLP_MATH=f32 selects that path instead (A/B runs, tests).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// (x0, x1) -> three packed bf16 pairs (x0 in the low half): h = bf16(x), m = bf16(x - h), l = bf16(x - h - m), round-to-nearest-even
__device__ __forceinline__ void split3(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  // written so that hipcc emits 11 VALU ops per pair (v_cvt_pk_bf16_f32, two bit ops to widen the pieces back, v_pk_add_f32); through
  // (float)(__bf16) round trips it emitted 15
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 x = {x0, x1};
  h = __builtin_bit_cast(uint32_t, (bf2){(__bf16)x.x, (__bf16)x.y});
  const f2 hf = {__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u)};
  const f2 r = x - hf;
  m = __builtin_bit_cast(uint32_t, (bf2){(__bf16)r.x, (__bf16)r.y});
  const f2 mf = {__builtin_bit_cast(float, m << 16), __builtin_bit_cast(float, m & 0xffff0000u)};
  const f2 t = r - mf;
  l = __builtin_bit_cast(uint32_t, (bf2){(__bf16)t.x, (__bf16)t.y});
}
__device__ __forceinline__ bf16x8 frag8(const uint32_t (&p)[4]) {
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  const u4 v = {p[0], p[1], p[2], p[3]};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 frag8(const float4& q) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 v = {q.x, q.y, q.z, q.w};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ f32x4_t mfma_bf16(bf16x8 a, bf16x8 b, f32x4_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#else
  return c;      // host pass of the single-source compile: never called
#endif
}
// 1: bf16x6 (default), 0: exact fp32 MFMA (BR_MLP_MATH=f32)
static inline bool mlp_bf16x6() {
  static const bool on = [] { const char* e = getenv("BR_MLP_MATH"); return !(e && e[0] == 'f'); }();
  return on;
}

struct BwdArgs {
  const float* gy; int64_t ldgy;
  const float* y; int64_t ldy;
  const float* x; int64_t ldx;
  const float* W;
  int64_t batch;
  int K, N, act;
  const float* o_mean; const float* o_rstd; const float* o_gamma; const double* o_sums; float inv_batch;   // BN after this layer, or null
  const float* scale; const float* shift;       // BN affine carried by the input (K), or null
  const uint32_t* keep; int kw; float inv_keep; // keep-bit plane of the input dropout [batch][kw], or null
  const float* i_mean; const float* i_rstd; double* in_sums;   // BN carried by the input: producer's backward sums (IBN)
  float* gx; int64_t ldgx;                      // may be null (first layer of a tower whose input needs no gradient)
  float* slabs; int64_t slab_elems; int64_t db_off;   // slab of workgroup b: slabs + b*slab_elems; db at +db_off (< 0: not written)
  float* zero = nullptr; int64_t zero_n = 0;          // floats the launch clears first (slabs beyond its grid, which the reduction still reads)
};

// dense_fwd.hip: brDropoutKeepBits for the step (resolved step counter + step_add)
int make_keep_args(KeepArgs& a, float drop_p, uint64_t seed, uint32_t step, uint32_t step_add, int64_t row0, int64_t batch, int n_sites,
                   const uint32_t* sites, const int* widths, uint32_t* const* out);
int dropout_keep_bits_ahead(float drop_p, uint64_t seed, uint32_t step, uint32_t step_add, int64_t row0, int64_t batch, int n_sites,
                            const uint32_t* sites, const int* widths, uint32_t* const* out, brStream stream);

// sparse_opt.hip: brAdamRowsSortedPair whose grid also fills the keep-bit planes described by *keep (NULL: plain)
struct AdamPairCall {
  float *table_a, *m_a, *v_a; int64_t rows_a; const void* sorted_ids_a; const int32_t* sorted_pos_a; const float* grads_a; int64_t ldg_a;
  const float* grads_hi_a; int64_t ldg_hi_a; uint8_t* mark_a; int32_t* last_a;
  float *table_b, *m_b, *v_b; int64_t rows_b; const void* sorted_ids_b; const int32_t* sorted_pos_b; const float* grads_b; int64_t ldg_b;
  const float* grads_hi_b; int64_t ldg_hi_b; uint8_t* mark_b; int32_t* last_b;
  int dim, id_type; int64_t n; int split; const float* hi_scale; const void* step_state; double alpha_t, beta1, beta2, eps;
  float *seg_ws_a, *seg_ws_b;
  // deferred tables: the rows as the step's lookup replayed them, by position (columns [0,split) | [split,dim)), or null
  const float *th_lo_a = nullptr, *th_hi_a = nullptr, *th_lo_b = nullptr, *th_hi_b = nullptr; int64_t ld_th = 0;
};
// fin != NULL: the launch may also carry the dense finalize as riders of its grid; *fin_done tells whether it did
struct FinalArgs;
int adam_rows_pair_keep(const AdamPairCall& c, const KeepArgs* keep, brStream stream, const FinalArgs* fin = nullptr, bool* fin_done = nullptr);

// sparse_opt.hip: the deferred NeuMF lookup and both dedup indexes of the step on ONE stream (the chunk sorts ride in the lookup's grid)
struct LookupArgs;
struct IndexPairArgs {
  void* sorted_ids_a; int32_t* sorted_pos_a; void* ws_a; int64_t ws_a_bytes;
  void* sorted_ids_b; int32_t* sorted_pos_b; void* ws_b; int64_t ws_b_bytes;
};
bool lookup_with_index_supported(int dim, int64_t n, int64_t upper_a, int64_t upper_b, int64_t ld_stash, const void* x0, const void* stash_a, const void* stash_b);
int lookup_with_index(const LookupArgs& la, int dim, int id_type, const IndexPairArgs& ix, brStream stream, const StepAdvance* adv = nullptr);

int dense_backward_fused(const BwdArgs& a, hipStream_t s);    // BR_ERR_UNSUPPORTED when the LDS image does not fit
int dense_bwd_fused_grid(int64_t batch);                      // workgroups = slabs written
size_t dense_bwd_fused_lds(int NT, int KT);

// tail_mfma.hip
struct TailMArgs {
  const float* a2; int64_t lda2;
  const float *W3, *b3, *w4, *b4, *dot, *labels;
  // BatchNorm 2: either ready-made constants (scale2 != null) or the forward column sums to finalize in the launch (stats2 != null)
  const float *scale2, *shift2, *mean2, *rstd2;
  const double* stats2; double batch_total; const float *gamma2, *beta2; float bn_eps, bn_momentum;
  float *moving_mean, *moving_var, *out_scale, *out_shift, *out_mean, *out_rstd;     // written by workgroup 0 when finalizing
  const uint32_t* keep; int kw; float inv_keep;
  int64_t batch;
  int n2, n3, act, mf_first, loss;
  float inv_batch;
  float *a3, *logit, *prob, *ddot, *gh2;
  int64_t ldgh2;
  double *msums, *bn_sums;
  float* slabs; int n_slabs;      // slabs beyond the grid are zero-filled by the kernel
};
int tail_mfma_grid(int64_t batch);
void launch_tail_mfma(const TailMArgs& a, int grid, hipStream_t s);

}  // namespace br
