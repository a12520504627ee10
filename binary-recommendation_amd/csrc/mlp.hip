// T1-T4 + B1: the MLP tower on fp32 MFMA (v_mfma_f32_16x16x4_f32: exact f32 fma chains, the
// only MFMA form that meets the 1e-5 logit/loss tolerance; 157 TFLOP/s peak on MI355X).
//
// One workgroup = 4 waves = one 64-row batch tile (16 rows per wave), looping over tiles
// (grid <= 256: one resident workgroup per CU, W staged into LDS once per workgroup).
//   forward : y = act( T(x)·W + b ), BatchNorm column sums of y in the epilogue.
//   backward: dz = actgrad(BN^-1-backward(dropout-backward(gy))) elementwise into LDS, then
//             dx = dz·W^T (per-wave 16 x K), dW += T(x)^T·dz (K x N split over waves by 16-row
//             strips of K) and db, accumulated in registers across the workgroup's tiles and
//             written once as a per-workgroup slab (fixed-order reduce => reproducible).
// K is contracted in a lane-permuted order (lane group g = lane>>4 owns k = 16j+4g..+3) so one
// ds_read_b128 feeds four MFMA k-steps; A and B use the same permutation.
#include "common.h"
#include "philox.h"

namespace br {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kTM = 64;          // rows per workgroup tile
constexpr int kMaxT = 8;         // max 16-wide tiles along K or N (=> K,N <= 128)
constexpr int kMaxSlabs = 256;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct InXform {       // T(x): BatchNorm affine of the producer + dropout, applied on load
  const float* scale;  // (K) or null
  const float* shift;  // (K) or null
  DropoutCfg drop;
};

// stage a (kTM x K) tile of x through T() into LDS [kTM][ldx] (zero pad to Kp columns)
__device__ __forceinline__ void stage_x_tile(float* Xs, int ldx, const float* __restrict__ x, int64_t ldg, int64_t row_base,
                                             int64_t batch, int K, int Kp, const InXform& t, int64_t row0, bool vec_ok,
                                             float ones_col_val, int ones_col) {
  const int cq_n = Kp >> 2;
  for (int idx = threadIdx.x; idx < kTM * cq_n; idx += blockDim.x) {
    const int r = idx / cq_n, cq = idx - r * cq_n;
    const int c = cq << 2;
    const int64_t gr = row_base + r;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (gr < batch && c < K) {
      const float* src = x + gr * ldg + c;
      if (vec_ok && c + 3 < K) {
        const float4 q = *reinterpret_cast<const float4*>(src);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (c + i < K) v[i] = src[i];
      }
      if (t.scale) {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (c + i < K) v[i] = v[i] * t.scale[c + i] + t.shift[c + i];
      }
      if (t.drop.thr) {
        const Philox4 d = dropout_draw4(t.drop, row0 + gr, (uint32_t)cq);
        v[0] = d.x >= t.drop.thr ? v[0] * t.drop.inv_keep : 0.f;
        v[1] = d.y >= t.drop.thr ? v[1] * t.drop.inv_keep : 0.f;
        v[2] = d.z >= t.drop.thr ? v[2] * t.drop.inv_keep : 0.f;
        v[3] = d.w >= t.drop.thr ? v[3] * t.drop.inv_keep : 0.f;
      }
    }
    if (ones_col >= 0 && gr < batch && c <= ones_col && ones_col < c + 4) v[ones_col - c] = ones_col_val;
    *reinterpret_cast<float4*>(Xs + r * ldx + c) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// W (K x N row-major) -> LDS [Kp][ldw], zero padded
__device__ __forceinline__ void stage_w(float* Ws, int ldw, const float* __restrict__ W, int K, int N, int Kp, int Np) {
  for (int idx = threadIdx.x; idx < Kp * Np; idx += blockDim.x) {
    const int k = idx / Np, n = idx - k * Np;
    Ws[k * ldw + n] = (k < K && n < N) ? W[k * N + n] : 0.f;
  }
}

// ------------------------------------------------------------------------------------ forward
template <int NT>
__global__ __launch_bounds__(256) void dense_fwd_kernel(const float* __restrict__ x, int64_t ldx_g, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ y, int64_t ldy,
                                                         int64_t batch, int K, int N, int act, InXform tin, int64_t row0,
                                                         double* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Kp = (K + 15) & ~15, Np = NT * 16;
  const int ldw = Np + 4, ldx = Kp + 4;
  float* Ws = smem;
  float* Xs = Ws + Kp * ldw;
  __shared__ double red[2][kMaxT * 16];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c16 = lane & 15, g = lane >> 4;
  const bool vec_ok = (ldx_g % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);

  stage_w(Ws, ldw, W, K, N, Kp, Np);
  float bcol[NT], ssum[NT], ssq[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = nt * 16 + c16;
    bcol[nt] = (bias && n < N) ? bias[n] : 0.f;
    ssum[nt] = 0.f; ssq[nt] = 0.f;
  }
  const int64_t n_tiles = (batch + kTM - 1) / kTM;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row_base = tile * kTM;
    __syncthreads();  // previous tile's MFMA reads done (and Ws staged on the first pass)
    stage_x_tile(Xs, ldx, x, ldx_g, row_base, batch, K, Kp, tin, row0, vec_ok, 0.f, -1);
    __syncthreads();
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* xr = Xs + (wave * 16 + c16) * ldx + 4 * g;
    for (int j = 0; j < Kp; j += 16) {
      const float4 a4 = *reinterpret_cast<const float4*>(xr + j);
      const float a[4] = {a4.x, a4.y, a4.z, a4.w};
      const float* wr = Ws + (j + 4 * g) * ldw + c16;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(a[s], wr[s * ldw + nt * 16], acc[nt]);
      }
    }
    // epilogue: lane holds rows 4g..4g+3 of its wave's 16, column nt*16+c16
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = nt * 16 + c16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t gr = row_base + wave * 16 + 4 * g + r;
        const float v = act_apply(acc[nt][r] + bcol[nt], act);
        if (gr < batch && n < N) {
          y[gr * ldy + n] = v;
          ssum[nt] += v;
          ssq[nt] += v * v;
        }
      }
    }
  }
  if (stats) {
    // lanes sharing a column: g = 0..3 -> xor 16, 32; then the 4 waves through LDS
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      double s = (double)ssum[nt], q = (double)ssq[nt];
      s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
      if (wave == 0 && g == 0) { red[0][nt * 16 + c16] = 0.0; red[1][nt * 16 + c16] = 0.0; }
      ssum[nt] = 0.f;  // reuse below via doubles kept in registers
      __syncthreads();
      if (g == 0) { atomicAdd(&red[0][nt * 16 + c16], s); atomicAdd(&red[1][nt * 16 + c16], q); }
      __syncthreads();
    }
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
      atomicAdd(stats + n, red[0][n]);
      atomicAdd(stats + N + n, red[1][n]);
    }
  }
}

// --------------------------------------------------------------------------------- BN helpers
__global__ void bn_finalize_kernel(const double* __restrict__ stats, double batch_total, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum, float* __restrict__ mm,
                                   float* __restrict__ mv, float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ mean, float* __restrict__ rstd, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const double mu = stats[n] / batch_total;
  double var = stats[N + n] / batch_total - mu * mu;  // biased batch variance [TF-sem]
  if (var < 0.0) var = 0.0;
  const float muf = (float)mu, varf = (float)var;
  const float rs = 1.0f / sqrtf(varf + eps);
  const float sc = gamma[n] * rs;
  scale[n] = sc;
  shift[n] = beta[n] - muf * sc;
  mean[n] = muf;
  rstd[n] = rs;
  if (mm) {
    mm[n] = mm[n] * momentum + muf * (1.0f - momentum);
    mv[n] = mv[n] * momentum + varf * (1.0f - momentum);
  }
}

__global__ void bn_inference_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mm,
                                    const float* __restrict__ mv, float eps, float* __restrict__ scale, float* __restrict__ shift, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float sc = gamma[n] / sqrtf(mv[n] + eps);
  scale[n] = sc;
  shift[n] = beta[n] - mm[n] * sc;
}

__global__ void bn_param_grads_kernel(const double* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  dbeta[n] = (float)sums[n];
  dgamma[n] = (float)sums[N + n];
}

// ----------------------------------------------------------------------------------- backward
struct OutXform {            // what sits between this layer's y and its consumer
  const float* mean;         // (N) BN batch mean, null => no BN
  const float* rstd;         // (N)
  const float* gamma;        // (N)
  const double* sums;        // (2N): sum_r dh, sum_r dh*xhat
  float inv_batch;           // 1 / global batch
};
struct InBn {                // BN carried by the input (for the producer's backward sums)
  const float* mean;         // (K) or null
  const float* rstd;
};

template <int KT, int NT>
__global__ __launch_bounds__(256) void dense_bwd_kernel(const float* __restrict__ gy, int64_t ldgy, const float* __restrict__ y, int64_t ldy,
                                                         const float* __restrict__ x, int64_t ldx_g, const float* __restrict__ W,
                                                         int64_t batch, int K, int N, int act, OutXform to, InXform tin, InBn ibn,
                                                         int64_t row0, float* __restrict__ gx, int64_t ldgx, float* __restrict__ slabs,
                                                         double* __restrict__ in_sums) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int Kp = KT * 16, Np = NT * 16;
  constexpr int ldw = Np + 4, ldx = Kp + 4, ldz = Np + 4;
  constexpr int STRIPS = (KT + 3) / 4;  // 16-row strips of K owned by one wave for dW
  float* Ws = smem;                     // [Kp][ldw]   W[k][n]
  float* Xs = Ws + Kp * ldw;            // [kTM][ldx]  T(x)
  float* Zs = Xs + kTM * ldx;           // [kTM][ldz]  dz
  float* Cs = Zs + kTM * ldz;           // [4][Np]     per-column constants of the out BN
  __shared__ double redk[2][kMaxT * 16];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c16 = lane & 15, g = lane >> 4;
  const bool xvec = (ldx_g % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);

  stage_w(Ws, ldw, W, K, N, Kp, Np);
  for (int n = threadIdx.x; n < Np; n += blockDim.x) {
    float c1 = 1.f, c2 = 0.f, c3 = 0.f, mu = 0.f, rs = 0.f;
    if (to.mean && n < N) {
      rs = to.rstd[n]; mu = to.mean[n];
      c1 = to.gamma[n] * rs;
      c2 = (float)(to.sums[n] * (double)to.inv_batch);
      c3 = (float)(to.sums[N + n] * (double)to.inv_batch);
    }
    Cs[0 * Np + n] = c1; Cs[1 * Np + n] = c2; Cs[2 * Np + n] = c3 * rs; Cs[3 * Np + n] = mu;
  }
  f32x4 dW[STRIPS][NT];
#pragma unroll
  for (int s = 0; s < STRIPS; ++s)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) dW[s][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float db_acc = 0.f;                // thread t < Np owns column t of db
  float isum[KT], isq[KT];           // in-BN sums for column kt*16+c16
  float imean[KT], irstd[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    isum[kt] = 0.f; isq[kt] = 0.f;
    const int k = kt * 16 + c16;
    imean[kt] = (ibn.mean && k < K) ? ibn.mean[k] : 0.f;
    irstd[kt] = (ibn.mean && k < K) ? ibn.rstd[k] : 0.f;
  }

  const int64_t n_tiles = (batch + kTM - 1) / kTM;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row_base = tile * kTM;
    __syncthreads();
    // ---- dz tile (elementwise): dropout^T -> BN^T -> act' -------------------------------
    {
      constexpr int cq_n = Np >> 2;
      for (int idx = threadIdx.x; idx < kTM * cq_n; idx += blockDim.x) {
        const int r = idx / cq_n, cq = idx - r * cq_n;
        const int c = cq << 2;
        const int64_t gr = row_base + r;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (gr < batch && c < N) {
          float gyv[4] = {0.f, 0.f, 0.f, 0.f}, yv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (c + i < N) { gyv[i] = gy[gr * ldgy + c + i]; yv[i] = y[gr * ldy + c + i]; }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float da = gyv[i];
            if (to.mean) {
              const int n = c + i;
              // da = gamma*rstd * (dh - mean(dh) - xhat*mean(dh*xhat)), xhat = (y-mu)*rstd
              da = Cs[n] * (gyv[i] - Cs[Np + n] - (yv[i] - Cs[3 * Np + n]) * Cs[2 * Np + n]);
            }
            v[i] = (c + i < N) ? da * act_grad_from_out(yv[i], act) : 0.f;
          }
        }
        *reinterpret_cast<float4*>(Zs + r * ldz + c) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    stage_x_tile(Xs, ldx, x, ldx_g, row_base, batch, K, Kp, tin, row0, xvec, 0.f, -1);
    __syncthreads();

    // ---- dx = dz · W^T : wave owns rows wave*16.., all KT column tiles; contraction over n ----
    if (gx) {
      f32x4 acc[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) acc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* zr = Zs + (wave * 16 + c16) * ldz + 4 * g;
      for (int j = 0; j < Np; j += 16) {
        const float4 a4 = *reinterpret_cast<const float4*>(zr + j);
        const float a[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const float4 b4 = *reinterpret_cast<const float4*>(Ws + (kt * 16 + c16) * ldw + j + 4 * g);
          acc[kt] = mfma16(a[0], b4.x, acc[kt]);
          acc[kt] = mfma16(a[1], b4.y, acc[kt]);
          acc[kt] = mfma16(a[2], b4.z, acc[kt]);
          acc[kt] = mfma16(a[3], b4.w, acc[kt]);
        }
      }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const int k = kt * 16 + c16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t gr = row_base + wave * 16 + 4 * g + r;
          if (gr < batch && k < K) {
            // gx = gradient w.r.t. the producer's BN output h (dropout transposed here)
            const float dh = acc[kt][r] * dropout_scale1(tin.drop, row0 + gr, (uint32_t)k);
            gx[gr * ldgx + k] = dh;
            if (ibn.mean) {
              const float xhat = (x[gr * ldx_g + k] - imean[kt]) * irstd[kt];
              isum[kt] += dh;
              isq[kt] += dh * xhat;
            }
          }
        }
      }
    }
    // ---- dW += T(x)^T · dz : wave owns K-strips {wave, wave+4}; contraction over the tile rows ----
    for (int j = 0; j < kTM; j += 16) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int r = j + 4 * g + s4;
        float bz[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bz[nt] = Zs[r * ldz + nt * 16 + c16];
#pragma unroll
        for (int s = 0; s < STRIPS; ++s) {
          const int strip = wave + 4 * s;
          if (strip < KT) {
            const float a = Xs[r * ldx + strip * 16 + c16];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) dW[s][nt] = mfma16(a, bz[nt], dW[s][nt]);
          }
        }
      }
    }
    // ---- db: thread t < Np sums column t of dz ----
    if (threadIdx.x < Np) {
      float sacc = 0.f;
      for (int r = 0; r < kTM; ++r) sacc += Zs[r * ldz + threadIdx.x];
      db_acc += sacc;
    }
  }

  // ---- slab: [dW (K x N) | db (N)] of this workgroup ----
  float* slab = slabs + (int64_t)blockIdx.x * ((int64_t)K * N + N);
#pragma unroll
  for (int s = 0; s < STRIPS; ++s) {
    const int strip = wave + 4 * s;
    if (strip < KT) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + c16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = strip * 16 + 4 * g + r;
          if (k < K && n < N) slab[(int64_t)k * N + n] = dW[s][nt][r];
        }
      }
    }
  }
  if (threadIdx.x < N) slab[(int64_t)K * N + threadIdx.x] = db_acc;

  if (in_sums && ibn.mean) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      double s = (double)isum[kt], q = (double)isq[kt];
      s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
      if (wave == 0 && g == 0) { redk[0][kt * 16 + c16] = 0.0; redk[1][kt * 16 + c16] = 0.0; }
      __syncthreads();
      if (g == 0) { atomicAdd(&redk[0][kt * 16 + c16], s); atomicAdd(&redk[1][kt * 16 + c16], q); }
      __syncthreads();
    }
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
      atomicAdd(in_sums + k, redk[0][k]);
      atomicAdd(in_sums + K + k, redk[1][k]);
    }
  }
}

// fixed-order slab reduction: out[e] = sum_s slabs[s][e], s ascending (reproducible)
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, int n_slabs, int64_t elems, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= elems) return;
  float acc = 0.f;
  for (int s = 0; s < n_slabs; ++s) acc += slabs[(int64_t)s * elems + e];
  out[e] = acc;
}

// ------------------------------------------------------------------------------------- head
// concat [dot | a3] (mf_first) or [a3 | dot] -> Dense(1) -> sigmoid -> loss (+ grads)
__global__ __launch_bounds__(256) void neumf_head_kernel(const float* __restrict__ a3, int64_t lda3, const float* __restrict__ dot,
                                                          const float* __restrict__ labels, const float* __restrict__ w4,
                                                          const float* __restrict__ b4, int64_t batch, int N3, int mf_first, int loss,
                                                          float inv_batch, float* __restrict__ logit, float* __restrict__ prob,
                                                          double* __restrict__ sums, float* __restrict__ da3, int64_t ldda3,
                                                          float* __restrict__ ddot, float* __restrict__ slabs) {
  // one thread per pair; N3 <= 32
  const int moff = mf_first ? 1 : 0;        // a3 weights start
  const float wdot = w4[mf_first ? 0 : N3];
  const float bias = b4[0];
  float gw[33];
#pragma unroll
  for (int i = 0; i < 33; ++i) gw[i] = 0.f;
  float gb = 0.f;
  double s_loss = 0.0, s_se = 0.0, s_ae = 0.0, s_ok = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < batch; b += stride) {
    float av[32];
    const float d = dot[b];
    float z = bias + d * wdot;
#pragma unroll
    for (int i = 0; i < 32; ++i)
      if (i < N3) { av[i] = a3[b * lda3 + i]; z += av[i] * w4[moff + i]; }
    const float p = sigmoidf_acc(z);
    if (logit) logit[b] = z;
    if (prob) prob[b] = p;
    if (labels) {
      const float yv = labels[b];
      float l, dz;
      if (loss == BR_LOSS_BCE) {
        l = fmaxf(z, 0.f) - z * yv + log1pf(expf(-fabsf(z)));
        dz = (p - yv) * inv_batch;
      } else {
        l = (p - yv) * (p - yv);
        dz = 2.f * (p - yv) * p * (1.f - p) * inv_batch;
      }
      s_loss += (double)l;
      s_se += (double)((p - yv) * (p - yv));
      s_ae += (double)fabsf(p - yv);
      s_ok += ((p > 0.5f) == (yv > 0.5f)) ? 1.0 : 0.0;
      if (da3) {
#pragma unroll
        for (int i = 0; i < 32; ++i)
          if (i < N3) { da3[b * ldda3 + i] = dz * w4[moff + i]; gw[i] += dz * av[i]; }
        ddot[b] = dz * wdot;
        gw[32] += dz * d;
        gb += dz;
      }
    }
  }
  if (!labels) return;
  __shared__ double redd[4][4];
  __shared__ float redf[4][34];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  s_loss = wave_sum_d(s_loss); s_se = wave_sum_d(s_se); s_ae = wave_sum_d(s_ae); s_ok = wave_sum_d(s_ok);
  if (lane == 0) { redd[wave][0] = s_loss; redd[wave][1] = s_se; redd[wave][2] = s_ae; redd[wave][3] = s_ok; }
  if (da3) {
#pragma unroll
    for (int i = 0; i < 33; ++i) {
      const float v = group_sum<64>(gw[i]);
      if (lane == 0) redf[wave][i] = v;
    }
    const float v = group_sum<64>(gb);
    if (lane == 0) redf[wave][33] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4 && sums) atomicAdd(sums + threadIdx.x, redd[0][threadIdx.x] + redd[1][threadIdx.x] + redd[2][threadIdx.x] + redd[3][threadIdx.x]);
  if (da3 && slabs) {
    // slab layout [dW4 (N3+1, concat order) | db4]
    float* slab = slabs + (int64_t)blockIdx.x * (N3 + 2);
    const int t = threadIdx.x;
    if (t < N3) slab[moff + t] = redf[0][t] + redf[1][t] + redf[2][t] + redf[3][t];
    if (t == 32) slab[mf_first ? 0 : N3] = redf[0][32] + redf[1][32] + redf[2][32] + redf[3][32];
    if (t == 33) slab[N3 + 1] = redf[0][33] + redf[1][33] + redf[2][33] + redf[3][33];
  }
}

__global__ __launch_bounds__(256) void bce_logits_kernel(const float* __restrict__ z, const float* __restrict__ y, int64_t batch,
                                                          float inv_batch, float* __restrict__ prob, float* __restrict__ dz,
                                                          double* __restrict__ sums) {
  double s_loss = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < batch; b += stride) {
    const float zv = z[b], yv = y[b];
    const float p = sigmoidf_acc(zv);
    if (prob) prob[b] = p;
    if (dz) dz[b] = (p - yv) * inv_batch;
    s_loss += (double)(fmaxf(zv, 0.f) - zv * yv + log1pf(expf(-fabsf(zv))));
  }
  __shared__ double red[4];
  s_loss = wave_sum_d(s_loss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s_loss;
  __syncthreads();
  if (threadIdx.x == 0 && sums) atomicAdd(sums, red[0] + red[1] + red[2] + red[3]);
}

}  // namespace br

using namespace br;

static inline int tiles16(int v) { return (v + 15) / 16; }
static inline unsigned mlp_grid(int64_t batch) {
  int64_t t = ceil_div(batch, kTM);
  return (unsigned)(t < kMaxSlabs ? (t < 1 ? 1 : t) : kMaxSlabs);
}

extern "C" int brDenseForward(const float* x, int64_t ldx, const float* W, const float* bias, float* y, int64_t ldy,
                              int64_t batch, int K, int N, int act, const float* in_scale, const float* in_shift,
                              float drop_p, uint64_t seed, uint32_t step, uint32_t site, int64_t row0, double* stats,
                              brStream stream) {
  BR_CHECK_ARG(x && W && y && batch >= 0 && K >= 1 && N >= 1, "brDenseForward: bad args");
  BR_CHECK_ARG(K <= kMaxT * 16 && N <= kMaxT * 16, "brDenseForward: K=%d N=%d exceed %d", K, N, kMaxT * 16);
  BR_CHECK_ARG(ldx >= K && ldy >= N, "brDenseForward: bad leading dims");
  BR_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "brDenseForward: in_scale/in_shift both or neither");
  BR_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "brDenseForward: drop_p out of [0,1)");
  if (batch == 0) return BR_OK;
  const int NT = tiles16(N), Kp = tiles16(K) * 16;
  const size_t shmem = ((size_t)Kp * (NT * 16 + 4) + (size_t)kTM * (Kp + 4)) * sizeof(float);
  InXform t{in_scale, in_shift, make_dropout(drop_p, seed, step, site)};
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = mlp_grid(batch);
#define BR_FWD(NTv)                                                                                                   \
  case NTv: {                                                                                                         \
    static bool attr_set = false;                                                                                     \
    if (!attr_set) {                                                                                                  \
      (void)hipFuncSetAttribute((const void*)dense_fwd_kernel<NTv>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096); \
      attr_set = true;                                                                                                \
    }                                                                                                                 \
    dense_fwd_kernel<NTv><<<grid, 256, shmem, s>>>(x, ldx, W, bias, y, ldy, batch, K, N, act, t, row0, stats);        \
  } break;
  switch (NT) {
    BR_FWD(1) BR_FWD(2) BR_FWD(3) BR_FWD(4) BR_FWD(5) BR_FWD(6) BR_FWD(7) BR_FWD(8)
    default: br::set_error("brDenseForward: unsupported N"); return BR_ERR_UNSUPPORTED;
  }
  BR_CHECK_LAUNCH("brDenseForward");
  return BR_OK;
}

extern "C" int brBnFinalize(const double* stats, double batch_total, const float* gamma, const float* beta, float eps,
                            float momentum, float* moving_mean, float* moving_var, float* scale, float* shift, float* mean,
                            float* rstd, int N, brStream stream) {
  BR_CHECK_ARG(stats && gamma && beta && scale && shift && mean && rstd && N >= 1 && batch_total > 0, "brBnFinalize: bad args");
  BR_CHECK_ARG((moving_mean == nullptr) == (moving_var == nullptr), "brBnFinalize: moving stats both or neither");
  bn_finalize_kernel<<<(unsigned)ceil_div(N, 128), 128, 0, (hipStream_t)stream>>>(stats, batch_total, gamma, beta, eps, momentum,
                                                                                   moving_mean, moving_var, scale, shift, mean, rstd, N);
  BR_CHECK_LAUNCH("brBnFinalize");
  return BR_OK;
}

extern "C" int brBnInference(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var, float eps,
                             float* scale, float* shift, int N, brStream stream) {
  BR_CHECK_ARG(gamma && beta && moving_mean && moving_var && scale && shift && N >= 1, "brBnInference: bad args");
  bn_inference_kernel<<<(unsigned)ceil_div(N, 128), 128, 0, (hipStream_t)stream>>>(gamma, beta, moving_mean, moving_var, eps, scale, shift, N);
  BR_CHECK_LAUNCH("brBnInference");
  return BR_OK;
}

extern "C" int brBnParamGrads(const double* bn_sums, float* dgamma, float* dbeta, int N, brStream stream) {
  BR_CHECK_ARG(bn_sums && dgamma && dbeta && N >= 1, "brBnParamGrads: bad args");
  bn_param_grads_kernel<<<(unsigned)ceil_div(N, 128), 128, 0, (hipStream_t)stream>>>(bn_sums, dgamma, dbeta, N);
  BR_CHECK_LAUNCH("brBnParamGrads");
  return BR_OK;
}

extern "C" int brDenseBackwardSlabs(int64_t batch, int K, int N) {
  (void)K; (void)N;
  return (int)mlp_grid(batch);
}

template <int KT, int NT>
static int launch_bwd(unsigned grid, size_t shmem, hipStream_t s, const float* gy, int64_t ldgy, const float* y, int64_t ldy,
                      const float* x, int64_t ldx, const float* W, int64_t batch, int K, int N, int act, OutXform to, InXform tin,
                      InBn ibn, int64_t row0, float* gx, int64_t ldgx, float* slabs, double* in_sums) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)dense_bwd_kernel<KT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
    attr_set = true;
  }
  dense_bwd_kernel<KT, NT><<<grid, 256, shmem, s>>>(gy, ldgy, y, ldy, x, ldx, W, batch, K, N, act, to, tin, ibn, row0, gx, ldgx, slabs, in_sums);
  return 0;
}

extern "C" int brDenseBackward(const float* gy, int64_t ldgy, const float* y, int64_t ldy, const float* x, int64_t ldx,
                               const float* W, int64_t batch, int K, int N, int act, const float* out_mean,
                               const float* out_rstd, const float* out_gamma, const double* bn_sums, double batch_total,
                               const float* in_scale, const float* in_shift,
                               const float* in_mean, const float* in_rstd, float in_drop_p, uint32_t in_site, uint64_t seed,
                               uint32_t step, int64_t row0, float* gx, int64_t ldgx, float* dW_slabs, int n_slabs,
                               double* in_bn_sums, brStream stream) {
  BR_CHECK_ARG(gy && y && x && W && dW_slabs && batch >= 0 && K >= 1 && N >= 1, "brDenseBackward: bad args");
  BR_CHECK_ARG(K <= kMaxT * 16 && N <= kMaxT * 16, "brDenseBackward: K=%d N=%d exceed %d", K, N, kMaxT * 16);
  BR_CHECK_ARG(ldgy >= N && ldy >= N && ldx >= K && (!gx || ldgx >= K), "brDenseBackward: bad leading dims");
  BR_CHECK_ARG((out_mean == nullptr) == (out_rstd == nullptr) && (out_mean == nullptr) == (out_gamma == nullptr) &&
               (out_mean == nullptr) == (bn_sums == nullptr), "brDenseBackward: out BN pointers all or none");
  BR_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "brDenseBackward: in_scale/in_shift both or neither");
  BR_CHECK_ARG((in_mean == nullptr) == (in_rstd == nullptr) && (in_mean == nullptr) == (in_bn_sums == nullptr),
               "brDenseBackward: in BN pointers all or none");
  BR_CHECK_ARG(!in_mean || gx, "brDenseBackward: in BN sums need gx");
  if (batch == 0) return BR_OK;
  const unsigned grid = mlp_grid(batch);
  BR_CHECK_ARG(n_slabs == (int)grid, "brDenseBackward: n_slabs %d != brDenseBackwardSlabs() %u", n_slabs, grid);
  const int KT = tiles16(K), NT = tiles16(N);
  const int Kp = KT * 16, Np = NT * 16;
  const size_t shmem = ((size_t)Kp * (Np + 4) + (size_t)kTM * (Kp + 4) + (size_t)kTM * (Np + 4) + 4 * (size_t)Np) * sizeof(float);
  OutXform to{out_mean, out_rstd, out_gamma, bn_sums, (float)(1.0 / (batch_total > 0 ? batch_total : (double)batch))};
  InXform tin{in_scale, in_shift, make_dropout(in_drop_p, seed, step, in_site)};
  InBn ibn{in_mean, in_rstd};
  hipStream_t s = (hipStream_t)stream;
#define BR_BWD(KTv, NTv) \
  if (KT == KTv && NT == NTv) { launch_bwd<KTv, NTv>(grid, shmem, s, gy, ldgy, y, ldy, x, ldx, W, batch, K, N, act, to, tin, ibn, row0, gx, ldgx, dW_slabs, in_bn_sums); } else
#define BR_BWD_ROW(KTv) BR_BWD(KTv, 1) BR_BWD(KTv, 2) BR_BWD(KTv, 3) BR_BWD(KTv, 4) BR_BWD(KTv, 5) BR_BWD(KTv, 6) BR_BWD(KTv, 7) BR_BWD(KTv, 8)
  BR_BWD_ROW(1) BR_BWD_ROW(2) BR_BWD_ROW(3) BR_BWD_ROW(4) BR_BWD_ROW(5) BR_BWD_ROW(6) BR_BWD_ROW(7) BR_BWD_ROW(8)
  { br::set_error("brDenseBackward: unsupported K/N"); return BR_ERR_UNSUPPORTED; }
  BR_CHECK_LAUNCH("brDenseBackward");
  return BR_OK;
}

extern "C" int brReduceSlabs(const float* slabs, int n_slabs, int64_t slab_elems, float* out, brStream stream) {
  BR_CHECK_ARG(slabs && out && n_slabs >= 1 && slab_elems >= 1, "brReduceSlabs: bad args");
  reduce_slabs_kernel<<<(unsigned)ceil_div(slab_elems, 256), 256, 0, (hipStream_t)stream>>>(slabs, n_slabs, slab_elems, out);
  BR_CHECK_LAUNCH("brReduceSlabs");
  return BR_OK;
}

extern "C" int brHeadSlabs(int64_t batch) {
  int64_t b = ceil_div(batch, 256);
  return (int)(b < 1 ? 1 : (b > 256 ? 256 : b));
}

extern "C" int brNeumfHead(const float* a3, int64_t lda3, const float* dot, const float* labels, const float* w4,
                           const float* b4, int64_t batch, int N3, int mf_first, int loss, float inv_batch, float* logit,
                           float* prob, double* sums, float* da3, int64_t ldda3, float* ddot, float* head_slabs, int n_slabs,
                           brStream stream) {
  BR_CHECK_ARG(a3 && dot && w4 && b4 && batch >= 0 && N3 >= 1 && N3 <= 32 && lda3 >= N3, "brNeumfHead: bad args (N3 <= 32)");
  BR_CHECK_ARG(loss == BR_LOSS_BCE || loss == BR_LOSS_MSE, "brNeumfHead: bad loss");
  BR_CHECK_ARG((da3 == nullptr) == (ddot == nullptr) && (da3 == nullptr) == (head_slabs == nullptr), "brNeumfHead: grads all or none");
  BR_CHECK_ARG(!da3 || (labels && ldda3 >= N3), "brNeumfHead: grads need labels");
  if (batch == 0) return BR_OK;
  const int grid = brHeadSlabs(batch);
  BR_CHECK_ARG(!da3 || n_slabs == grid, "brNeumfHead: n_slabs %d != brHeadSlabs() %d", n_slabs, grid);
  neumf_head_kernel<<<(unsigned)grid, 256, 0, (hipStream_t)stream>>>(a3, lda3, dot, labels, w4, b4, batch, N3, mf_first, loss, inv_batch,
                                                                     logit, prob, sums, da3, ldda3, ddot, head_slabs);
  BR_CHECK_LAUNCH("brNeumfHead");
  return BR_OK;
}

extern "C" int brBceLogits(const float* z, const float* y, int64_t batch, float inv_batch, float* prob, float* dz, double* sums,
                           brStream stream) {
  BR_CHECK_ARG(z && y && batch >= 0, "brBceLogits: bad args");
  if (batch == 0) return BR_OK;
  int64_t grid = ceil_div(batch, 256);
  if (grid > 1024) grid = 1024;
  bce_logits_kernel<<<(unsigned)grid, 256, 0, (hipStream_t)stream>>>(z, y, batch, inv_batch, prob, dz, sums);
  BR_CHECK_LAUNCH("brBceLogits");
  return BR_OK;
}
