// End of the dense backward in ONE launch: the fixed-order reduction of the per-workgroup dW|db slabs of the three
// tower layers (+ head), the BatchNorm parameter gradients from the backward column sums, and O1 (Adam) on the flat
// dense parameter vector.  Replaces reduce_slabs x3 + bn_param_grads + adam_flat (5 launch-bound kernels, ~26 us of a
// 0.53 ms step) on the single-process path; a data-parallel host keeps the separate kernels because the dense
// gradient all-reduce sits between the reduction and the optimizer.
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

// 64 elements per workgroup, 16 slab-parts per element (as reduce_slabs_kernel: same summation order per region)
__global__ __launch_bounds__(1024) void dense_finalize_kernel(FinalArgs a, AdamHp h) {
  __shared__ float part[16][64];
  adam_resolve(h);
  const int lane = threadIdx.x & 63, p = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (e < a.n) {
    bool done = false;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const FinalRegion& R = a.reg[r];
      const int64_t i = e - R.grad_off;
      if (!done && i >= 0 && i < R.elems) {
        done = true;
        int s = p;
        for (; s + 48 < R.n_slabs; s += 64) {   // 4 independent loads in flight
          const float a0 = R.slabs[(int64_t)s * R.elems + i], a1 = R.slabs[(int64_t)(s + 16) * R.elems + i];
          const float a2 = R.slabs[(int64_t)(s + 32) * R.elems + i], a3 = R.slabs[(int64_t)(s + 48) * R.elems + i];
          acc += a0; acc += a1; acc += a2; acc += a3;
        }
        for (; s < R.n_slabs; s += 16) acc += R.slabs[(int64_t)s * R.elems + i];
      }
    }
    if (!done && p == 0) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const FinalBn& B = a.bn[b];
        const int64_t ig = e - B.dgamma_off, ib = e - B.dbeta_off;
        // sums: [BR_STAT_REPLICAS][2N] = (sum gy | sum gy*xhat): dbeta = sum gy, dgamma = sum gy*xhat
        if (ig >= 0 && ig < B.N) { double s2 = 0.0; for (int r = 0; r < BR_STAT_REPLICAS; ++r) s2 += B.sums[(size_t)r * 2 * B.N + B.N + ig]; acc = (float)s2; }
        if (ib >= 0 && ib < B.N) { double s1 = 0.0; for (int r = 0; r < BR_STAT_REPLICAS; ++r) s1 += B.sums[(size_t)r * 2 * B.N + ib]; acc = (float)s1; }
      }
    }
  }
  part[p][lane] = acc;
  __syncthreads();
  if (p == 0 && e < a.n) {
    float g = part[0][lane];
#pragma unroll
    for (int q = 1; q < 16; ++q) g += part[q][lane];
    a.grad[e] = g;
    if (a.theta) {             // NULL: gradients only (a data-parallel host all-reduces them before its optimizer launch)
      float th = a.theta[e], m = a.m[e], v = a.v[e];
      adam_update1(th, m, v, g, h);
      a.theta[e] = th; a.m[e] = m; a.v[e] = v;
    }
  }
}

}  // namespace br

using namespace br;

extern "C" int brDenseFinalize(const float* const* slabs, const int* n_slabs, const int64_t* slab_elems, const int64_t* grad_off,
                               const double* const* bn_sums, const int* bn_n, const int64_t* dgamma_off, const int64_t* dbeta_off,
                               float* theta, float* m, float* v, float* grad, int64_t n, double alpha_t, double beta1, double beta2,
                               double eps, brStream stream) {
  BR_CHECK_ARG(slabs && n_slabs && slab_elems && grad_off && bn_sums && bn_n && dgamma_off && dbeta_off && grad && n >= 1 && (!theta || (m && v)),
               "brDenseFinalize: null pointer");
  FinalArgs a;
  int64_t covered = 0;
  for (int r = 0; r < 3; ++r) {
    BR_CHECK_ARG(slabs[r] && n_slabs[r] >= 1 && slab_elems[r] >= 1 && grad_off[r] >= 0 && grad_off[r] + slab_elems[r] <= n, "brDenseFinalize: bad region %d", r);
    a.reg[r] = FinalRegion{slabs[r], n_slabs[r], slab_elems[r], grad_off[r]};
    covered += slab_elems[r];
  }
  for (int b = 0; b < 2; ++b) {
    BR_CHECK_ARG(bn_sums[b] && bn_n[b] >= 1 && dgamma_off[b] >= 0 && dbeta_off[b] >= 0 && dgamma_off[b] + bn_n[b] <= n && dbeta_off[b] + bn_n[b] <= n,
                 "brDenseFinalize: bad BatchNorm block %d", b);
    a.bn[b] = FinalBn{bn_sums[b], bn_n[b], dgamma_off[b], dbeta_off[b]};
    covered += 2 * (int64_t)bn_n[b];
  }
  BR_CHECK_ARG(covered == n, "brDenseFinalize: regions cover %lld of %lld parameters", (long long)covered, (long long)n);
  a.theta = theta; a.m = m; a.v = v; a.grad = grad; a.n = n;
  const AdamHp h = make_hp(alpha_t, beta1, beta2, eps);
  dense_finalize_kernel<<<(unsigned)ceil_div(n, 64), 1024, 0, (hipStream_t)stream>>>(a, h);
  BR_CHECK_LAUNCH("brDenseFinalize");
  return BR_OK;
}
