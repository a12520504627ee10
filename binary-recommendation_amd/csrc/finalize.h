// Argument block and workgroup bodies of the dense finalize (finalize.hip), shared with the Adam-rows launch that can carry
// the finalize as extra workgroups of its grid (sparse_opt.hip).
#pragma once
#include "common.h"
#include "adam_math.h"

namespace br {

struct FinalRegion { const float* slabs; int n_slabs; int64_t elems; int64_t grad_off; };
struct FinalBn { const double* sums; int N; int64_t dgamma_off, dbeta_off; };
struct FinalArgs {
  FinalRegion reg[3];
  FinalBn bn[2];
  float *theta, *m, *v, *grad;
  int64_t n;
};

// one slab-part of element e: slabs p, p+16, p+32, ... of its region in that order (four independent loads in flight)
__device__ __forceinline__ float final_part(const FinalArgs& a, int64_t e, int p, bool& is_slab) {
  float acc = 0.f;
  is_slab = false;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const FinalRegion& R = a.reg[r];
    const int64_t i = e - R.grad_off;
    if (!is_slab && i >= 0 && i < R.elems) {
      is_slab = true;
      int s = p;
      for (; s + 48 < R.n_slabs; s += 64) {   // 4 independent loads in flight
        const float a0 = R.slabs[(int64_t)s * R.elems + i], a1 = R.slabs[(int64_t)(s + 16) * R.elems + i];
        const float a2 = R.slabs[(int64_t)(s + 32) * R.elems + i], a3 = R.slabs[(int64_t)(s + 48) * R.elems + i];
        acc += a0; acc += a1; acc += a2; acc += a3;
      }
      for (; s < R.n_slabs; s += 16) acc += R.slabs[(int64_t)s * R.elems + i];
    }
  }
  return acc;
}
// BatchNorm parameter gradients of element e (not in a slab region): sums = [BR_STAT_REPLICAS][2N] = (sum gy | sum gy*xhat)
__device__ __forceinline__ float final_bn(const FinalArgs& a, int64_t e) {
  float acc = 0.f;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const FinalBn& B = a.bn[b];
    const int64_t ig = e - B.dgamma_off, ib = e - B.dbeta_off;
    if (ig >= 0 && ig < B.N) { double s2 = 0.0; for (int r = 0; r < BR_STAT_REPLICAS; ++r) s2 += B.sums[(size_t)r * 2 * B.N + B.N + ig]; acc = (float)s2; }
    if (ib >= 0 && ib < B.N) { double s1 = 0.0; for (int r = 0; r < BR_STAT_REPLICAS; ++r) s1 += B.sums[(size_t)r * 2 * B.N + ib]; acc = (float)s1; }
  }
  return acc;
}
// the 16 parts of a lane's element added in part order, then Adam (theta == NULL: gradients only)
__device__ __forceinline__ void final_apply(const FinalArgs& a, int64_t e, const float (*part)[64], int lane, const AdamHp& h) {
  float g = part[0][lane];
#pragma unroll
  for (int q = 1; q < 16; ++q) g += part[q][lane];
  a.grad[e] = g;
  if (a.theta) {
    float th = a.theta[e], m = a.m[e], v = a.v[e];
    adam_update1(th, m, v, g, h);
    a.theta[e] = th; a.m[e] = m; a.v[e] = v;
  }
}

// The finalize of 64 elements by a workgroup of 256 threads (virtual block vb of ceil(n / 64)): thread group P = 0..3 forms the
// parts P, P+4, P+8, P+12 - the same 16 partial sums in the same order as the 1024-thread kernel, so the same bits.
// `part` = 16 x 64 floats of LDS; every thread of the workgroup must call (barrier inside).  h must be resolved.
__device__ __forceinline__ void finalize_block256(const FinalArgs& a, const AdamHp& h, int64_t vb, float (*part)[64]) {
  const int lane = threadIdx.x & 63, P = threadIdx.x >> 6;
  const int64_t e = vb * 64 + lane;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = P + 4 * k;
    float acc = 0.f;
    if (e < a.n) {
      bool is_slab;
      acc = final_part(a, e, p, is_slab);
      if (!is_slab && p == 0) acc = final_bn(a, e);
    }
    part[p][lane] = acc;
  }
  __syncthreads();
  if (P == 0 && e < a.n) final_apply(a, e, part, lane, h);
}

int make_final_args(FinalArgs& a, const float* const* slabs, const int* n_slabs, const int64_t* slab_elems, const int64_t* grad_off,
                    const double* const* bn_sums, const int* bn_n, const int64_t* dgamma_off, const int64_t* dbeta_off,
                    float* theta, float* m, float* v, float* grad, int64_t n);

}  // namespace br
