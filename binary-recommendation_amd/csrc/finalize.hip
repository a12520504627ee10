// End of the dense backward in ONE launch: the fixed-order reduction of the per-workgroup dW|db slabs of the three
// tower layers (+ head), the BatchNorm parameter gradients from the backward column sums, and O1 (Adam) on the flat
// dense parameter vector.  Replaces reduce_slabs x3 + bn_param_grads + adam_flat (5 launch-bound kernels, ~26 us of a
// 0.53 ms step) on the single-process path; a data-parallel host keeps the separate kernels because the dense
// gradient all-reduce sits between the reduction and the optimizer.  The single-GPU step carries the same work as extra
// workgroups of its Adam-rows launch instead (finalize.h finalize_block256, sparse_opt.hip).
#include "finalize.h"

namespace br {

// 64 elements per workgroup, 16 slab-parts per element (as reduce_slabs_kernel: same summation order per region)
__global__ __launch_bounds__(1024) void dense_finalize_kernel(FinalArgs a, AdamHp h) {
  __shared__ float part[16][64];
  adam_resolve(h);
  const int lane = threadIdx.x & 63, p = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (e < a.n) {
    bool is_slab;
    acc = final_part(a, e, p, is_slab);
    if (!is_slab && p == 0) acc = final_bn(a, e);
  }
  part[p][lane] = acc;
  __syncthreads();
  if (p == 0 && e < a.n) final_apply(a, e, part, lane, h);
}

int make_final_args(FinalArgs& a, const float* const* slabs, const int* n_slabs, const int64_t* slab_elems, const int64_t* grad_off,
                    const double* const* bn_sums, const int* bn_n, const int64_t* dgamma_off, const int64_t* dbeta_off,
                    float* theta, float* m, float* v, float* grad, int64_t n) {
  BR_CHECK_ARG(slabs && n_slabs && slab_elems && grad_off && bn_sums && bn_n && dgamma_off && dbeta_off && grad && n >= 1 && (!theta || (m && v)),
               "brDenseFinalize: null pointer");
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
  return BR_OK;
}

}  // namespace br

using namespace br;

extern "C" int brDenseFinalize(const float* const* slabs, const int* n_slabs, const int64_t* slab_elems, const int64_t* grad_off,
                               const double* const* bn_sums, const int* bn_n, const int64_t* dgamma_off, const int64_t* dbeta_off,
                               float* theta, float* m, float* v, float* grad, int64_t n, double alpha_t, double beta1, double beta2,
                               double eps, brStream stream) {
  FinalArgs a;
  const int rc = make_final_args(a, slabs, n_slabs, slab_elems, grad_off, bn_sums, bn_n, dgamma_off, dbeta_off, theta, m, v, grad, n);
  if (rc != BR_OK) return rc;
  const AdamHp h = make_hp(alpha_t, beta1, beta2, eps);
  dense_finalize_kernel<<<(unsigned)ceil_div(n, 64), 1024, 0, (hipStream_t)stream>>>(a, h);
  BR_CHECK_LAUNCH("brDenseFinalize");
  return BR_OK;
}
