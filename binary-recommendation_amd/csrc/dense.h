// Argument block and host entry of the fused one-launch backward (dense_bwd.hip), called from brDenseBackward (mlp.hip).
#pragma once
#include "common.h"

namespace br {

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
};

int dense_backward_fused(const BwdArgs& a, hipStream_t s);    // BR_ERR_UNSUPPORTED when the LDS image does not fit
int dense_bwd_fused_grid(int64_t batch);                      // workgroups = slabs written
size_t dense_bwd_fused_lds(int NT, int KT);

}  // namespace br
