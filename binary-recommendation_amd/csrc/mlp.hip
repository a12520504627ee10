// T1-T4 + B1: the MLP tower on fp32 MFMA (v_mfma_f32_16x16x4_f32: exact f32 fma chains, the
// only MFMA form that meets the 1e-5 logit/loss tolerance; 157 TFLOP/s peak on MI355X).
//
// forward (dense_fwd_kernel) and backward-dx (dense_dx_kernel): one 512-thread workgroup per CU.
//   * The layer's W is staged ONCE into LDS (forward: pre-swizzled [j][g][n][s] so a lane's four
//     k-steps are one conflict-free ds_read_b128; dx: row-major, read along n as ds_read_b128).
//   * After that single barrier every wave is independent: it walks its own 16-row tiles, loads the A
//     fragments STRAIGHT from global memory in the MFMA lane layout (lane (c16,g) <- 16 B of row c16 at
//     k = 16j+4g; K is contracted in that lane-permuted order), with branch-free guarded loads and the
//     next tile's loads in flight during the current tile's MFMAs.
//   * BatchNorm-affine + Philox dropout are applied in registers (keep-bytes of the 16 x K tile in a
//     per-wave LDS patch: <= 4 Philox calls per lane per tile, reused by the dx epilogue).
//   * MFMA in passes of <= 4 (dx: 2) output tiles: independent accumulator chains, B fragments of the
//     next k-step read from LDS while the current one is in the matrix pipe.
//   * A wave owns whole output rows, so they leave as back-to-back pieces of the same 128-B lines.
//   * dx also forms dz = act'(y)·BN-backward(gy) in registers and hands it to the dW kernel via HBM.
// backward-dW (dense_dw_kernel): a skinny split-K GEMM over the batch; wave (k-tile, row group) keeps its
//   dW strip in registers across the workgroup's row range -> one slab per (workgroup, row group) ->
//   fixed-order reduce (reproducible, no float atomics).
// Column statistics (BatchNorm sums) leave as double atomics into BR_STAT_REPLICAS replicas.
// Round-1 measurement (tools/mlp_bench.py, 128 x 100 layer, batch 65 536, MI355X): the MFMA passes run
// at the fp32-MFMA rate (14.4 us vs 13.6 us ideal) but the A loads (7 us), W staging (3 us) and stores
// (4 us) of this one-generation launch (2 tiles per wave) are not hidden behind them: 32 us = 25 % of peak.
#include <stdlib.h>

#include "common.h"
#include "dense.h"

namespace br {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kThreads = 512;
constexpr int kMaxT = 8;         // max 16-wide tiles along K or N (=> K,N <= 128)
constexpr int kMaxGrid = 256;    // one 8-wave workgroup per CU, each wave walks several 16-row tiles
constexpr int kRep = BR_STAT_REPLICAS;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct InXform {       // T(x): BatchNorm affine of the producer + dropout, applied on load
  const float* scale;  // (K) or null (global; the kernels stage scale|shift into LDS once)
  const float* shift;  // (K) or null
  const uint8_t* keep; // keep-bit plane of the input dropout (brDropoutKeepBits) viewed as bytes: byte c8 of a row = its 8-column
                       // chunk c8 (already offset to this launch's first column for the second half of a K > 128 layer), or null
  int64_t keep_ld;     // bytes per row of the plane
  float inv_keep;      // 1/(1-p)
};
// keep bits of the 8-column chunk c8 of row r (0xFF without dropout)
__device__ __forceinline__ uint32_t keep8(const InXform& t, int64_t r, int c8) {
  return t.keep ? (uint32_t)t.keep[r * t.keep_ld + c8] : 0xFFu;
}

// Branch-free guarded loads.  Every load below is issued UNCONDITIONALLY from an address clamped
// into the buffer and the out-of-range lanes are zeroed with selects: an `if (in range) load` makes
// hipcc put each load in its own exec-masked block followed by s_waitcnt vmcnt(0), which serialises
// the loads of a tile at one HBM latency each (measured: 9.5 us of a 31 us forward kernel).
// vec = wave-uniform: row stride % 4 == 0, K % 4 == 0, base 16-B aligned.
// VEC is a template parameter and callers branch on the wave-uniform flag ONCE around their whole load
// loop, so all loads of a tile sit in one basic block (one wait at first use, not one per load).
template <bool VEC>
__device__ __forceinline__ float4 ld4_guard(const float* __restrict__ base, int64_t ld, int64_t row, int64_t nrows, int k, int K) {
  const bool rin = row < nrows;
  const int64_t r = rin ? row : nrows - 1;
  float4 v;
  if (VEC) {
    const bool ok = rin && k < K;
    const float4 t = *reinterpret_cast<const float4*>(base + r * ld + (k < K ? k : 0));
    v = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
  } else {
    const float* p = base + r * ld;
    const float t0 = p[k + 0 < K ? k + 0 : K - 1], t1 = p[k + 1 < K ? k + 1 : K - 1];
    const float t2 = p[k + 2 < K ? k + 2 : K - 1], t3 = p[k + 3 < K ? k + 3 : K - 1];
    v = make_float4((rin && k + 0 < K) ? t0 : 0.f, (rin && k + 1 < K) ? t1 : 0.f, (rin && k + 2 < K) ? t2 : 0.f, (rin && k + 3 < K) ? t3 : 0.f);
  }
  return v;
}

// 8 consecutive floats of row gr starting at column c (zeros outside [0,K) x [0,batch))
template <bool VEC>
__device__ __forceinline__ void load8(float (&v)[8], const float* __restrict__ base, int64_t ld, int64_t gr, int64_t batch, int c, int K) {
  const float4 a = ld4_guard<VEC>(base, ld, gr, batch, c, K);
  const float4 b = ld4_guard<VEC>(base, ld, gr, batch, c + 4, K);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// T(): BN affine + dropout on one 8-column chunk; returns the keep bits
// ss = LDS copy [scale (Kp) | shift (Kp)], zero padded, or null
__device__ __forceinline__ uint32_t xform8(float (&v)[8], const InXform& t, const float* ss, int Kp, int64_t grow, int c, int K, bool live) {
  if (!live) return 0u;
  if (ss) {
    const float4 s0 = *reinterpret_cast<const float4*>(ss + c), s1 = *reinterpret_cast<const float4*>(ss + c + 4);
    const float4 h0 = *reinterpret_cast<const float4*>(ss + Kp + c), h1 = *reinterpret_cast<const float4*>(ss + Kp + c + 4);
    v[0] = v[0] * s0.x + h0.x; v[1] = v[1] * s0.y + h0.y; v[2] = v[2] * s0.z + h0.z; v[3] = v[3] * s0.w + h0.w;
    v[4] = v[4] * s1.x + h1.x; v[5] = v[5] * s1.y + h1.y; v[6] = v[6] * s1.z + h1.z; v[7] = v[7] * s1.w + h1.w;
  }
  uint32_t bits = 0xFFu;
  if (t.keep) {
    bits = keep8(t, grow, c >> 3);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = ((bits >> i) & 1u) ? v[i] * t.inv_keep : 0.f;
  }
  return bits;
}

__device__ __forceinline__ void store8_lds(float* dst, const float (&v)[8]) {
  *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// --------------------------------------------------------------------------------- BN helpers
__global__ void bn_finalize_kernel(const double* __restrict__ stats, double batch_total, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum, float* __restrict__ mm,
                                   float* __restrict__ mv, float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ mean, float* __restrict__ rstd, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double s1 = 0.0, s2 = 0.0;
  for (int r = 0; r < kRep; ++r) { s1 += stats[(size_t)r * 2 * N + n]; s2 += stats[(size_t)r * 2 * N + N + n]; }
  const double mu = s1 / batch_total;
  double var = s2 / batch_total - mu * mu;  // biased batch variance [TF-sem]
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

// blockIdx.y selects the layer (the two BatchNorms of a tower share one launch)
__global__ void bn_param_grads_kernel(const double* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta, int N,
                                      const double* __restrict__ sums_b, float* __restrict__ dgamma_b, float* __restrict__ dbeta_b, int Nb) {
  if (blockIdx.y == 1) { sums = sums_b; dgamma = dgamma_b; dbeta = dbeta_b; N = Nb; }
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double s1 = 0.0, s2 = 0.0;
  for (int r = 0; r < kRep; ++r) { s1 += sums[(size_t)r * 2 * N + n]; s2 += sums[(size_t)r * 2 * N + N + n]; }
  dbeta[n] = (float)s1;
  dgamma[n] = (float)s2;
}

// ----------------------------------------------------------------------------------- backward
struct OutXform {            // what sits between this layer's y and its consumer
  const float* mean;         // (N) BN batch mean, null => no BN
  const float* rstd;         // (N)
  const float* gamma;        // (N)
  const double* sums;        // (2N): sum_r gy, sum_r gy*xhat
  float inv_batch;           // 1 / global batch
};
struct InBn {                // BN carried by the input (for the producer's backward sums)
  const float* mean;         // (K) or null
  const float* rstd;
};

// ---- backward, part 1: dz and dx -------------------------------------------------------------
// Same skeleton as the forward: independent waves over 16-row tiles.  The A fragments are dz, formed
// in registers straight from gy / y (BN-backward constants from LDS); dz is also written out
// ([batch][Np], zero padded) for the dW kernel.  dx = dz·W^T against the row-major W image in LDS
// (lane (c16,g) reads W[kt*16+c16][16j+4g..+3] as one ds_read_b128).  Epilogue: dropout transposed
// (keep-bytes in the per-wave LDS patch), gx stored, BN-backward sums of the producer accumulated.
// IBN: the layer's input carries a BatchNorm (backward column sums of the producer + xhat operands) - a template
// parameter because its 32 + 16 extra registers spill the first layer's instantiation, which never needs them
template <int NT, int KT, bool IBN>
__global__ __launch_bounds__(kThreads, 2) void dense_dx_kernel(const float* __restrict__ gy, int64_t ldgy, const float* __restrict__ y, int64_t ldy,
                                                                const float* __restrict__ x, int64_t ldx_g, const float* __restrict__ W,
                                                                int64_t batch, int K, int N, int act, OutXform to, InXform tin, InBn ibn,
                                                                float* __restrict__ gx, int64_t ldgx, float* __restrict__ dzbuf,
                                                                double* __restrict__ in_sums) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int Np = NT * 16, Kp = KT * 16, ldw = Np + 4, NCH = 2 * KT;
  float* Ws = smem;                                   // [Kp][ldw] row-major W
  float* Cs = Ws + Kp * ldw;                          // [4][Np] out-BN constants
  float* Is = Cs + 4 * Np;                            // [mean Kp | rstd Kp] of the in BN
  uint8_t* mk_all = reinterpret_cast<uint8_t*>(Is + 2 * Kp);   // [8 waves][16][NCH]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c16 = lane & 15, g = lane >> 4;
  uint8_t* mk = mk_all + wave * 16 * NCH;

  for (int k = threadIdx.x; k < Kp; k += kThreads) {
    Is[k] = (IBN && k < K) ? ibn.mean[k] : 0.f;
    Is[Kp + k] = (IBN && k < K) ? ibn.rstd[k] : 0.f;
  }
  for (int n = threadIdx.x; n < Np; n += kThreads) {
    float c1 = 1.f, c2 = 0.f, c3 = 0.f, mu = 0.f, rs = 0.f;
    if (to.mean && n < N) {
      double s1 = 0.0, s2 = 0.0;
      for (int r = 0; r < kRep; ++r) { s1 += to.sums[(size_t)r * 2 * N + n]; s2 += to.sums[(size_t)r * 2 * N + N + n]; }
      rs = to.rstd[n]; mu = to.mean[n];
      c1 = to.gamma[n] * rs;
      c2 = (float)(s1 * (double)to.inv_batch);
      c3 = (float)(s2 * (double)to.inv_batch);
    }
    Cs[0 * Np + n] = c1; Cs[1 * Np + n] = c2; Cs[2 * Np + n] = c3 * rs; Cs[3 * Np + n] = mu;
  }
  __syncthreads();                                    // BN constants visible: dz can be formed at load time

  const bool gvec = (ldgy % 4 == 0) && (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(gy) & 15) == 0);
  const bool yvec = (ldy % 4 == 0) && (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  const int64_t n_tiles = (batch + 15) >> 4;
  const int64_t tstride = (int64_t)gridDim.x * 8;

  // dz fragments (A layout) of one tile: lane (c16,g) <- columns 16j+4g..+3 of row c16; also written out for dW
  auto load_dz = [&](float4 (&dz)[NT], int64_t tile) {
    const int64_t arow = tile < n_tiles ? (tile << 4) + c16 : batch;
    const bool live = arow < batch;
    float4 vg[NT], vy[NT];
    if (gvec && yvec) {                   // all loads first, in one basic block; arithmetic after
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        vg[j] = ld4_guard<true>(gy, ldgy, arow, batch, 16 * j + 4 * g, N);
        vy[j] = ld4_guard<true>(y, ldy, arow, batch, 16 * j + 4 * g, N);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        vg[j] = ld4_guard<false>(gy, ldgy, arow, batch, 16 * j + 4 * g, N);
        vy[j] = ld4_guard<false>(y, ldy, arow, batch, 16 * j + 4 * g, N);
      }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = 16 * j + 4 * g;
      const float ge[4] = {vg[j].x, vg[j].y, vg[j].z, vg[j].w}, ye[4] = {vy[j].x, vy[j].y, vy[j].z, vy[j].w};
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float da = ge[e];
        // da = gamma*rstd * (gy - mean(gy) - xhat*mean(gy*xhat)), xhat = (y-mu)*rstd
        if (to.mean) da = Cs[n + e] * (ge[e] - Cs[Np + n + e] - (ye[e] - Cs[3 * Np + n + e]) * Cs[2 * Np + n + e]);
        v[e] = (live && n + e < N) ? da * act_grad_from_out(ye[e], act) : 0.f;
      }
      dz[j] = make_float4(v[0], v[1], v[2], v[3]);
      if (live) *reinterpret_cast<float4*>(dzbuf + arow * Np + n) = dz[j];
    }
  };

  int64_t tile = (int64_t)blockIdx.x * 8 + wave;
  float4 dz[NT], dzn[NT];
  load_dz(dz, tile);                                  // first tile's loads fly while W is staged
  if (gx) {     // all loads of the W image before the first LDS write (see dense_fwd_kernel)
    constexpr int TOT = Kp * (Np / 4), TR = (TOT + kThreads - 1) / kThreads;
    float wv[TR][4];
#pragma unroll
    for (int i = 0; i < TR; ++i) {
      const int idx = threadIdx.x + i * kThreads;
      const int idc = idx < TOT ? idx : 0;
      const int k = idc / (Np / 4), n = (idc - k * (Np / 4)) * 4;
      const float* wr = W + (int64_t)(k < K ? k : K - 1) * N;
#pragma unroll
      for (int q = 0; q < 4; ++q) wv[i][q] = wr[n + q < N ? n + q : N - 1];
    }
#pragma unroll
    for (int i = 0; i < TR; ++i) {
      const int idx = threadIdx.x + i * kThreads;
      const int k = idx / (Np / 4), n = (idx - k * (Np / 4)) * 4;
      const bool kin = k < K;
      if (idx < TOT)
        *reinterpret_cast<float4*>(Ws + k * ldw + n) = make_float4((kin && n + 0 < N) ? wv[i][0] : 0.f, (kin && n + 1 < N) ? wv[i][1] : 0.f,
                                                                    (kin && n + 2 < N) ? wv[i][2] : 0.f, (kin && n + 3 < N) ? wv[i][3] : 0.f);
    }
  }
  __syncthreads();

  float isum[KT], isq[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) { isum[kt] = 0.f; isq[kt] = 0.f; }
  for (; tile < n_tiles; tile += tstride) {
    const int64_t rbase = tile << 4;
    load_dz(dzn, tile + tstride);                     // next tile's gy / y in flight during this tile's MFMAs
    if (!gx) {
#pragma unroll
      for (int j = 0; j < NT; ++j) dz[j] = dzn[j];
      continue;
    }
    if (tin.keep) {
#pragma unroll
      for (int i = 0; i < (16 * NCH + 63) / 64; ++i) {
        const int q = lane + 64 * i;
        const int64_t kr = rbase + q / NCH;
        if (q < 16 * NCH) mk[q] = (8 * (q % NCH) < K && kr < batch) ? (uint8_t)keep8(tin, kr, q % NCH) : (uint8_t)0;
      }
      __builtin_amdgcn_wave_barrier();
    }
    // raw x of this tile's outputs (for xhat in the epilogues): ALL k-tiles are requested here, before the
    // Philox and MFMA work of the tile - loaded per pass they cost one exposed HBM round trip per pass
    float xv[4][KT];
    if (IBN) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t gr = rbase + 4 * g + r;
        const float* xr = x + (gr < batch ? gr : batch - 1) * ldx_g;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const int k = kt * 16 + c16;
          xv[r][kt] = xr[k < K ? k : K - 1];
        }
      }
    }
    // ---- dx = dz · W^T in passes of 2 k-tiles: two independent chains alternate (revisit distance
    //      64 cycles >= the 40-cycle dependent latency), B fragments double-buffered; 2 instead of 4 keeps the
    //      kernel inside the 128-VGPR budget of 4 waves/SIMD without scratch traffic in the loop ----
#pragma unroll
    for (int kt0 = 0; kt0 < KT; kt0 += 2) {
      constexpr int WMAX = 2;
      const int Wn = (KT - kt0) < WMAX ? (KT - kt0) : WMAX;
      f32x4 acc[WMAX];
      float4 bc[WMAX], bn[WMAX];
#pragma unroll
      for (int w = 0; w < WMAX; ++w) {
        acc[w] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (w < Wn) bc[w] = *reinterpret_cast<const float4*>(Ws + ((kt0 + w) * 16 + c16) * ldw + 4 * g);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (j + 1 < NT) {
#pragma unroll
          for (int w = 0; w < WMAX; ++w)
            if (w < Wn) bn[w] = *reinterpret_cast<const float4*>(Ws + ((kt0 + w) * 16 + c16) * ldw + 16 * (j + 1) + 4 * g);
        }
#pragma unroll
        for (int w = 0; w < WMAX; ++w) if (w < Wn) acc[w] = mfma16(dz[j].x, bc[w].x, acc[w]);
#pragma unroll
        for (int w = 0; w < WMAX; ++w) if (w < Wn) acc[w] = mfma16(dz[j].y, bc[w].y, acc[w]);
#pragma unroll
        for (int w = 0; w < WMAX; ++w) if (w < Wn) acc[w] = mfma16(dz[j].z, bc[w].z, acc[w]);
#pragma unroll
        for (int w = 0; w < WMAX; ++w) if (w < Wn) acc[w] = mfma16(dz[j].w, bc[w].w, acc[w]);
#pragma unroll
        for (int w = 0; w < WMAX; ++w) if (w < Wn) bc[w] = bn[w];
      }
      // epilogue of the pass: rows 4g..4g+3, column (kt0+w)*16+c16
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lr = 4 * g + r;
        const int64_t gr = rbase + lr;
        if (gr < batch) {
#pragma unroll
          for (int w = 0; w < WMAX; ++w) {
            if (w < Wn) {
              const int k = (kt0 + w) * 16 + c16;
              if (k < K) {
                // gx = gradient w.r.t. the producer's BN output h (dropout transposed here)
                const bool keep = tin.keep ? ((mk[lr * NCH + (k >> 3)] >> (k & 7)) & 1) : true;
                const float dh = keep ? acc[w][r] * tin.inv_keep : 0.f;
                gx[gr * ldgx + k] = dh;
                if (IBN && keep) {
                  const float xhat = (xv[r][kt0 + w] - Is[k]) * Is[Kp + k];
                  isum[kt0 + w] += dh;
                  isq[kt0 + w] += dh * xhat;
                }
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) dz[j] = dzn[j];
  }
  if (IBN && in_sums) {
    __shared__ double red[2][kMaxT * 16];
    for (int n = threadIdx.x; n < 2 * kMaxT * 16; n += kThreads) (&red[0][0])[n] = 0.0;
    __syncthreads();
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      double sv = (double)isum[kt], q = (double)isq[kt];
      sv += __shfl_xor(sv, 16, 64); sv += __shfl_xor(sv, 32, 64);
      q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
      if (g == 0) { atomicAdd(&red[0][kt * 16 + c16], sv); atomicAdd(&red[1][kt * 16 + c16], q); }
    }
    __syncthreads();
    double* rep = in_sums + (size_t)(blockIdx.x % kRep) * 2 * K;
    for (int k = threadIdx.x; k < K; k += kThreads) {
      atomicAdd(rep + k, red[0][k]);
      atomicAdd(rep + K + k, red[1][k]);
    }
  }
}

// ---- backward, part 2: dW = T(x)^T · dz and db -------------------------------------------------
// A skinny split-K GEMM (K x N output, reduction over the batch): one workgroup per CU sweeps its
// row range in tiles of TM rows; T(x) (BN-affine + dropout re-applied) and dz are staged through a
// single LDS image with the next tile's global loads in flight in registers; wave (k-tile, row group)
// keeps its dW strip in registers for the whole launch -> one slab per (workgroup, row group).
// KTP = pow2 >= number of k-tiles: wave -> (k-tile = wave % KTP, row group = wave / KTP)
template <int NT, int KTP>
__global__ __launch_bounds__(kThreads, 2) void dense_dw_kernel(const float* __restrict__ dzbuf, const float* __restrict__ x, int64_t ldx_g,
                                                                int64_t batch, int K, int N, InXform tin,
                                                                float* __restrict__ slabs, int64_t slab_elems, int64_t db_off) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int Np = NT * 16, ldz = Np + 4;
  constexpr int RGN = 8 / KTP;                      // row groups
  constexpr int TM = 16 * (RGN > 4 ? RGN : 4);      // 64 rows per tile (128 when K <= 16)
  constexpr int RT = TM / 16 / RGN;                 // row tiles per wave per tile
  const int KT = (K + 15) >> 4, Kp = KT << 4, ldx = Kp + 4;
  float* Xs = smem;                                 // [TM][ldx]   T(x)
  float* Zs = Xs + TM * ldx;                        // [TM][ldz]   dz
  float* ssb = Zs + TM * ldz;                       // [scale Kp | shift Kp]
  const float* ss = tin.scale ? ssb : nullptr;
  if (tin.scale)
    for (int k = threadIdx.x; k < Kp; k += kThreads) {
      ssb[k] = k < K ? tin.scale[k] : 0.f;
      ssb[Kp + k] = k < K ? tin.shift[k] : 0.f;
    }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c16 = lane & 15, g = lane >> 4;
  const int kt = wave % KTP, rg = wave / KTP;
  const bool kt_live = kt < KT;
  const int kcol = kt * 16 + c16;
  const bool xvec = (ldx_g % 4 == 0) && (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);

  f32x4 dW[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) dW[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float db_acc = 0.f;                               // thread t < Np owns column t of db

  constexpr int ZCR = Np / 8;
  constexpr int MAXZ = (TM * ZCR + kThreads - 1) / kThreads;
  constexpr int MAXX = (TM * (kMaxT * 16 / 8) + kThreads - 1) / kThreads;
  const int xc_row = Kp >> 3, n_xc = TM * xc_row;
  float pz[MAXZ][8], px[MAXX][8];
  const int64_t n_tiles = (batch + TM - 1) / TM;

  auto load_tile = [&](int64_t tile) {
    const int64_t row_base = tile * TM;
#pragma unroll
    for (int i = 0; i < MAXZ; ++i) {
      const int idx = threadIdx.x + kThreads * i;
      // out-of-tile chunks (idx >= TM*ZCR) clamp to the tile's first chunk; they are never written back
      const int idc = idx < TM * ZCR ? idx : 0;
      const int r = idc / ZCR, c = (idc - r * ZCR) << 3;
      load8<true>(pz[i], dzbuf, Np, row_base + r, batch, c, Np);
    }
    if (xvec) {
#pragma unroll
      for (int i = 0; i < MAXX; ++i) {
        const int idx = threadIdx.x + kThreads * i;
        const int idc = idx < n_xc ? idx : 0;
        const int r = idc / xc_row, c = (idc - r * xc_row) << 3;
        load8<true>(px[i], x, ldx_g, row_base + r, batch, c, K);
      }
    } else {
#pragma unroll
      for (int i = 0; i < MAXX; ++i) {
        const int idx = threadIdx.x + kThreads * i;
        const int idc = idx < n_xc ? idx : 0;
        const int r = idc / xc_row, c = (idc - r * xc_row) << 3;
        load8<false>(px[i], x, ldx_g, row_base + r, batch, c, K);
      }
    }
  };
  auto write_tile = [&](int64_t tile) {
    const int64_t row_base = tile * TM;
#pragma unroll
    for (int i = 0; i < MAXZ; ++i) {
      const int idx = threadIdx.x + kThreads * i;
      if (idx < TM * ZCR) {
        const int r = idx / ZCR, c = (idx - r * ZCR) << 3;
        store8_lds(Zs + r * ldz + c, pz[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < MAXX; ++i) {
      const int idx = threadIdx.x + kThreads * i;
      if (idx < n_xc) {
        const int r = idx / xc_row, c = (idx - r * xc_row) << 3;
        xform8(px[i], tin, ss, Kp, row_base + r, c, K, (row_base + r) < batch && c < K);
        store8_lds(Xs + r * ldx + c, px[i]);
      }
    }
  };

  int64_t tile = blockIdx.x;
  if (tile < n_tiles) load_tile(tile);
  __syncthreads();   // scale|shift visible
  if (tile < n_tiles) write_tile(tile);
  __syncthreads();
  while (tile < n_tiles) {
    const int64_t next = tile + gridDim.x;
    if (next < n_tiles) load_tile(next);
    if (kt_live) {
#pragma unroll
      for (int i = 0; i < RT; ++i) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const int r = (rg + RGN * i) * 16 + 4 * g + s4;
          const float av = Xs[r * ldx + kcol];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) dW[nt] = mfma16(av, Zs[r * ldz + nt * 16 + c16], dW[nt]);
        }
      }
    }
    if (threadIdx.x < Np) {
      float sacc = 0.f;
#pragma unroll 8
      for (int r = 0; r < TM; ++r) sacc += Zs[r * ldz + threadIdx.x];
      db_acc += sacc;
    }
    __syncthreads();
    if (next < n_tiles) write_tile(next);
    __syncthreads();
    tile = next;
  }
  // ---- slab (workgroup, row group): [dW (K x N) | db (N)]; split-K: `slabs` points at this half's first row of dW,
  //      slab_elems is the whole layer's slab, db sits db_off floats behind (db_off < 0: the other half writes db) ----
  if (kt_live) {
    float* slab = slabs + ((int64_t)blockIdx.x * RGN + rg) * slab_elems;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = nt * 16 + c16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = kt * 16 + 4 * g + r;
        if (k < K && n < N) slab[(int64_t)k * N + n] = dW[nt][r];
      }
    }
  }
  if ((int)threadIdx.x < N && db_off >= 0) {
    for (int r = 0; r < RGN; ++r)
      slabs[((int64_t)blockIdx.x * RGN + r) * slab_elems + db_off + threadIdx.x] = (r == 0) ? db_acc : 0.f;
  }
}

// fixed-order slab reduction (reproducible): out[e] = sum over slabs, combined as 16 interleaved
// partial sums (part p takes slabs p, p+16, ...) added in part order.  64 elements x 16 parts per
// workgroup: every load is a coalesced 256-B row segment and 16 waves share the latency.
__global__ __launch_bounds__(1024) void reduce_slabs_kernel(const float* __restrict__ slabs, int n_slabs, int64_t elems, float* __restrict__ out) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, p = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (e < elems) {
    int s = p;
    for (; s + 48 < n_slabs; s += 64) {   // 4 independent loads in flight
      const float a0 = slabs[(int64_t)s * elems + e], a1 = slabs[(int64_t)(s + 16) * elems + e];
      const float a2 = slabs[(int64_t)(s + 32) * elems + e], a3 = slabs[(int64_t)(s + 48) * elems + e];
      acc += a0; acc += a1; acc += a2; acc += a3;
    }
    for (; s < n_slabs; s += 16) acc += slabs[(int64_t)s * elems + e];
  }
  part[p][lane] = acc;
  __syncthreads();
  if (p == 0 && e < elems) {
    float r = part[0][lane];
#pragma unroll
    for (int q = 1; q < 16; ++q) r += part[q][lane];
    out[e] = r;
  }
}

// ------------------------------------------------------------------------------------- head
// concat [dot | a3] (mf_first) or [a3 | dot] -> Dense(1) -> sigmoid -> loss (+ grads)
// One thread per pair; a wave's 64 rows of a3 / da3 are contiguous in memory (lda3 == N3), so they
// move as coalesced flat chunks through a per-wave LDS patch instead of 40-B-strided accesses.
__global__ __launch_bounds__(256) void neumf_head_kernel(const float* __restrict__ a3, int64_t lda3, const float* __restrict__ dot,
                                                          const float* __restrict__ labels, const float* __restrict__ w4,
                                                          const float* __restrict__ b4, int64_t batch, int N3, int mf_first, int loss,
                                                          float inv_batch, float* __restrict__ logit, float* __restrict__ prob,
                                                          double* __restrict__ sums, float* __restrict__ da3, int64_t ldda3,
                                                          float* __restrict__ ddot, float* __restrict__ slabs) {
  __shared__ float patch[4][64 * 33];
  const int moff = mf_first ? 1 : 0;        // a3 weights start
  const float wdot = w4[mf_first ? 0 : N3];
  const float bias = b4[0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* P = patch[wave];
  const int ldp = N3 | 1;                   // odd row stride: conflict-free row reads
  float gw[33];
#pragma unroll
  for (int i = 0; i < 33; ++i) gw[i] = 0.f;
  float gb = 0.f;
  double s_loss = 0.0, s_se = 0.0, s_ae = 0.0, s_ok = 0.0, s_bce = 0.0, s_tp = 0.0, s_fp = 0.0, s_fn = 0.0;
  const bool flat_in = lda3 == N3, flat_out = da3 && ldda3 == N3;
  const int64_t wave_stride = (int64_t)gridDim.x * 4 * 64;
  for (int64_t b0 = ((int64_t)blockIdx.x * 4 + wave) * 64; b0 < batch; b0 += wave_stride) {
    const int64_t b = b0 + lane;
    const int nrow = (int)((batch - b0) < 64 ? (batch - b0) : 64);
    // stage the wave's rows of a3
    if (flat_in) {
      for (int e = lane; e < nrow * N3; e += 64) { const int r = e / N3; P[r * ldp + (e - r * N3)] = a3[b0 * N3 + e]; }
    } else if (b < batch) {
      for (int i = 0; i < N3; ++i) P[lane * ldp + i] = a3[b * lda3 + i];
    }
    __builtin_amdgcn_wave_barrier();
    float av[32];
    float z = bias, d = 0.f;
    if (b < batch) {
      d = dot[b];
      z += d * wdot;
#pragma unroll
      for (int i = 0; i < 32; ++i)
        if (i < N3) { av[i] = P[lane * ldp + i]; z += av[i] * w4[moff + i]; }
    }
    const float p = sigmoidf_acc(z);
    float dz = 0.f;
    if (b < batch) {
      if (logit) logit[b] = z;
      if (prob) prob[b] = p;
      if (labels) {
        const float yv = labels[b];
        float l;
        const float bce = fmaxf(z, 0.f) - z * yv + log1pf(expf(-fabsf(z)));
        if (loss == BR_LOSS_BCE) {
          l = bce;
          dz = (p - yv) * inv_batch;
        } else {
          l = (p - yv) * (p - yv);
          dz = 2.f * (p - yv) * p * (1.f - p) * inv_batch;
        }
        const bool pp = p > 0.5f, yp = yv > 0.5f;
        s_loss += (double)l;
        s_se += (double)((p - yv) * (p - yv));
        s_ae += (double)fabsf(p - yv);
        s_ok += (pp == yp) ? 1.0 : 0.0;
        s_bce += (double)bce;
        s_tp += (pp && yp) ? 1.0 : 0.0; s_fp += (pp && !yp) ? 1.0 : 0.0; s_fn += (!pp && yp) ? 1.0 : 0.0;
      }
    }
    if (da3 && labels) {
      __builtin_amdgcn_wave_barrier();
      if (b < batch) {
#pragma unroll
        for (int i = 0; i < 32; ++i)
          if (i < N3) { const float gv = dz * w4[moff + i]; gw[i] += dz * av[i]; if (flat_out) P[lane * ldp + i] = gv; else da3[b * ldda3 + i] = gv; }
        ddot[b] = dz * wdot;
        gw[32] += dz * d;
        gb += dz;
      }
      __builtin_amdgcn_wave_barrier();
      if (flat_out)
        for (int e = lane; e < nrow * N3; e += 64) { const int r = e / N3; da3[b0 * N3 + e] = P[r * ldp + (e - r * N3)]; }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (!labels) return;
  __shared__ double redd[4][BR_METRIC_SUMS];
  __shared__ float redf[4][34];
  s_loss = wave_sum_d(s_loss); s_se = wave_sum_d(s_se); s_ae = wave_sum_d(s_ae); s_ok = wave_sum_d(s_ok);
  s_bce = wave_sum_d(s_bce); s_tp = wave_sum_d(s_tp); s_fp = wave_sum_d(s_fp); s_fn = wave_sum_d(s_fn);
  if (lane == 0) {
    redd[wave][0] = s_loss; redd[wave][1] = s_se; redd[wave][2] = s_ae; redd[wave][3] = s_ok;
    redd[wave][4] = s_bce; redd[wave][5] = s_tp; redd[wave][6] = s_fp; redd[wave][7] = s_fn;
  }
  if (da3) {
#pragma unroll
    for (int i = 0; i < 33; ++i) {
      if (i < N3 || i == 32) {
        const float v = group_sum<64>(gw[i]);
        if (lane == 0) redf[wave][i] = v;
      }
    }
    const float v = group_sum<64>(gb);
    if (lane == 0) redf[wave][33] = v;
  }
  __syncthreads();
  // BR_METRIC_SUMS metric sums per slot, slot = workgroup & 63: same-line double atomics serialise (~12 ns each)
  if (threadIdx.x < BR_METRIC_SUMS && sums)
    atomicAdd(sums + (size_t)(blockIdx.x & (BR_SUM_SLOTS - 1)) * BR_METRIC_SUMS + threadIdx.x,
              redd[0][threadIdx.x] + redd[1][threadIdx.x] + redd[2][threadIdx.x] + redd[3][threadIdx.x]);
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
static inline int pow2_ge(int v) { int p = 1; while (p < v) p <<= 1; return p; }
static inline unsigned grid_for(int64_t batch, int tm) {
  int64_t t = ceil_div(batch, tm);
  return (unsigned)(t < kMaxGrid ? (t < 1 ? 1 : t) : kMaxGrid);
}
constexpr int kMaxDynLds = 150 * 1024;
// K > 128 layers run as two K-halves (see dense_fwd.hip); any split that is a multiple of 8 keeps the dropout chunks' numbering
static inline int split_k(int K) { return ((K / 2 + 31) / 32) * 32; }

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
  bn_param_grads_kernel<<<(unsigned)ceil_div(N, 128), 128, 0, (hipStream_t)stream>>>(bn_sums, dgamma, dbeta, N, nullptr, nullptr, nullptr, 0);
  BR_CHECK_LAUNCH("brBnParamGrads");
  return BR_OK;
}
extern "C" int brBnParamGradsPair(const double* sums_a, float* dgamma_a, float* dbeta_a, int Na, const double* sums_b, float* dgamma_b,
                                  float* dbeta_b, int Nb, brStream stream) {
  BR_CHECK_ARG(sums_a && dgamma_a && dbeta_a && Na >= 1 && sums_b && dgamma_b && dbeta_b && Nb >= 1, "brBnParamGradsPair: bad args");
  const dim3 grid((unsigned)ceil_div(Na > Nb ? Na : Nb, 128), 2);
  bn_param_grads_kernel<<<grid, 128, 0, (hipStream_t)stream>>>(sums_a, dgamma_a, dbeta_a, Na, sums_b, dgamma_b, dbeta_b, Nb);
  BR_CHECK_LAUNCH("brBnParamGradsPair");
  return BR_OK;
}

static inline int dw_tm(int ktp) { return 16 * ((8 / ktp) > 4 ? (8 / ktp) : 4); }
static inline unsigned dw_grid(int64_t batch, int ktp) {
  int64_t t = ceil_div(batch, dw_tm(ktp));
  return (unsigned)(t < 256 ? (t < 1 ? 1 : t) : 256);     // one workgroup per CU
}

extern "C" int brDenseBackwardSlabs(int64_t batch, int K, int N) {
  (void)N;
  const int KTP = K > kMaxT * 16 ? 8 : pow2_ge(tiles16(K < 1 ? 1 : K));   // K > 128: two K-halves, both as 8 k-tile strips
  return (int)dw_grid(batch, KTP) * (8 / KTP);
}

extern "C" int64_t brDenseBackwardWorkspaceFloats(int64_t batch, int K, int N) {
  (void)K;
  return batch * (int64_t)(tiles16(N) * 16);
}

template <int NT, int KT, bool IBN>
static void launch_dx_v(unsigned grid, size_t shmem, hipStream_t s, const float* gy, int64_t ldgy, const float* y, int64_t ldy, const float* x,
                        int64_t ldx, const float* W, int64_t batch, int K, int N, int act, OutXform to, InXform tin, InBn ibn,
                        float* gx, int64_t ldgx, float* dzbuf, double* in_sums) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)dense_dx_kernel<NT, KT, IBN>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynLds);
    attr_set = true;
  }
  dense_dx_kernel<NT, KT, IBN><<<grid, kThreads, shmem, s>>>(gy, ldgy, y, ldy, x, ldx, W, batch, K, N, act, to, tin, ibn, gx, ldgx, dzbuf, in_sums);
}
template <int NT, int KT>
static void launch_dx(unsigned grid, size_t shmem, hipStream_t s, const float* gy, int64_t ldgy, const float* y, int64_t ldy, const float* x,
                      int64_t ldx, const float* W, int64_t batch, int K, int N, int act, OutXform to, InXform tin, InBn ibn,
                      float* gx, int64_t ldgx, float* dzbuf, double* in_sums) {
  if (ibn.mean) launch_dx_v<NT, KT, true>(grid, shmem, s, gy, ldgy, y, ldy, x, ldx, W, batch, K, N, act, to, tin, ibn, gx, ldgx, dzbuf, in_sums);
  else launch_dx_v<NT, KT, false>(grid, shmem, s, gy, ldgy, y, ldy, x, ldx, W, batch, K, N, act, to, tin, ibn, gx, ldgx, dzbuf, in_sums);
}

template <int NT, int KTP>
static void launch_dw(unsigned grid, size_t shmem, hipStream_t s, int64_t slab_elems, int64_t db_off, const float* dzbuf, const float* x, int64_t ldx, int64_t batch, int K, int N,
                      InXform tin, float* slabs) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)dense_dw_kernel<NT, KTP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
    attr_set = true;
  }
  dense_dw_kernel<NT, KTP><<<grid, kThreads, shmem, s>>>(dzbuf, x, ldx, batch, K, N, tin, slabs, slab_elems, db_off);
}

// one K-range of a layer's backward on the two-kernel path (any K, N <= 128, any alignment): dx for columns [k0, k0+Kc) of gx,
// dW rows [k0, k0+Kc)
static int dense_backward_part(const float* gy, int64_t ldgy, const float* y, int64_t ldy, const float* x, int64_t ldx, const float* W,
                               int64_t batch, int Kc, int N, int act, const OutXform& to, InXform tin, InBn ibn, float* gx,
                               int64_t ldgx, float* dz_ws, float* slabs, int64_t slab_elems, int64_t db_off, int KTP, double* in_bn_sums,
                               bool run_dx, bool run_dw, hipStream_t s) {
  const int KT = tiles16(Kc), NT = tiles16(N);
  const int Kp = KT * 16, Np = NT * 16;
  const float* W_ = W;
  if (run_dx) {
    const unsigned grid = grid_for(batch, 16 * 8);
    const size_t shmem = ((size_t)Kp * (Np + 4) + 4 * (size_t)Np + 2 * (size_t)Kp) * sizeof(float) + (size_t)8 * 16 * 2 * KT + 16;
#define BR_DX_K(NTv, KTv) case KTv: launch_dx<NTv, KTv>(grid, shmem, s, gy, ldgy, y, ldy, x, ldx, W_, batch, Kc, N, act, to, tin, ibn, gx, ldgx, dz_ws, in_bn_sums); break;
#define BR_DX(NTv) \
  case NTv:        \
    switch (KT) { BR_DX_K(NTv, 1) BR_DX_K(NTv, 2) BR_DX_K(NTv, 3) BR_DX_K(NTv, 4) BR_DX_K(NTv, 5) BR_DX_K(NTv, 6) BR_DX_K(NTv, 7) BR_DX_K(NTv, 8) default: break; } \
    break;
    switch (NT) {
      BR_DX(1) BR_DX(2) BR_DX(3) BR_DX(4) BR_DX(5) BR_DX(6) BR_DX(7) BR_DX(8)
      default: br::set_error("brDenseBackward: unsupported N"); return BR_ERR_UNSUPPORTED;
    }
    BR_CHECK_LAUNCH("brDenseBackward(dx)");
  }
  if (run_dw) {
    const unsigned grid = dw_grid(batch, KTP);
    const int tm = dw_tm(KTP);
    const size_t shmem = ((size_t)tm * (Kp + 4) + (size_t)tm * (Np + 4) + 2 * (size_t)Kp) * sizeof(float);
#define BR_DW_K(NTv, KTPv) case KTPv: launch_dw<NTv, KTPv>(grid, shmem, s, slab_elems, db_off, dz_ws, x, ldx, batch, Kc, N, tin, slabs); break;
#define BR_DW(NTv) \
  case NTv:        \
    switch (KTP) { BR_DW_K(NTv, 1) BR_DW_K(NTv, 2) BR_DW_K(NTv, 4) BR_DW_K(NTv, 8) default: break; } \
    break;
    switch (NT) {
      BR_DW(1) BR_DW(2) BR_DW(3) BR_DW(4) BR_DW(5) BR_DW(6) BR_DW(7) BR_DW(8)
      default: br::set_error("brDenseBackward: unsupported N"); return BR_ERR_UNSUPPORTED;
    }
    BR_CHECK_LAUNCH("brDenseBackward(dW)");
  }
  return BR_OK;
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int brDenseBackward(const float* gy, int64_t ldgy, const float* y, int64_t ldy, const float* x, int64_t ldx,
                               const float* W, int64_t batch, int K, int N, int act, const float* out_mean,
                               const float* out_rstd, const float* out_gamma, const double* bn_sums, double batch_total,
                               const float* in_scale, const float* in_shift, const float* in_mean, const float* in_rstd,
                               float in_drop_p, const uint32_t* keep, float* gx, int64_t ldgx, float* dz_ws, float* dW_slabs,
                               int n_slabs, double* in_bn_sums, brStream stream) {
  BR_CHECK_ARG(gy && y && x && W && dW_slabs && dz_ws && batch >= 0 && K >= 1 && N >= 1, "brDenseBackward: bad args");
  BR_CHECK_ARG(K <= 2 * kMaxT * 16 && N <= kMaxT * 16, "brDenseBackward: K=%d N=%d exceed %d / %d", K, N, 2 * kMaxT * 16, kMaxT * 16);
  BR_CHECK_ARG(ldgy >= N && ldy >= N && ldx >= K && (!gx || ldgx >= K), "brDenseBackward: bad leading dims");
  BR_CHECK_ARG((out_mean == nullptr) == (out_rstd == nullptr) && (out_mean == nullptr) == (out_gamma == nullptr) &&
               (out_mean == nullptr) == (bn_sums == nullptr), "brDenseBackward: out BN pointers all or none");
  BR_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "brDenseBackward: in_scale/in_shift both or neither");
  BR_CHECK_ARG((in_mean == nullptr) == (in_rstd == nullptr) && (in_mean == nullptr) == (in_bn_sums == nullptr),
               "brDenseBackward: in BN pointers all or none");
  BR_CHECK_ARG(!in_mean || gx, "brDenseBackward: in BN sums need gx");
  BR_CHECK_ARG(K <= kMaxT * 16 || !in_mean, "brDenseBackward: K > %d is supported for inputs without a BatchNorm (first layer of a tower)", kMaxT * 16);
  BR_CHECK_ARG((reinterpret_cast<uintptr_t>(dz_ws) & 15) == 0, "brDenseBackward: dz_ws must be 16-byte aligned");
  BR_CHECK_ARG(in_drop_p >= 0.f && in_drop_p < 1.f, "brDenseBackward: in_drop_p out of [0,1)");
  BR_CHECK_ARG((in_drop_p > 0.f) == (keep != nullptr), "brDenseBackward: keep bits (brDropoutKeepBits) are required exactly when in_drop_p > 0");
  if (batch == 0) return BR_OK;
  const int want = brDenseBackwardSlabs(batch, K, N);
  BR_CHECK_ARG(n_slabs == want, "brDenseBackward: n_slabs %d != brDenseBackwardSlabs() %d", n_slabs, want);
  hipStream_t s = (hipStream_t)stream;
  const int64_t slab_elems = (int64_t)K * N + N;
  const float inv_batch = (float)(1.0 / (batch_total > 0 ? batch_total : (double)batch));
  const float inv_keep = in_drop_p > 0.f ? 1.0f / (1.0f - in_drop_p) : 1.0f;
  const int kw = (K + 31) / 32;
  const int Ka = K <= kMaxT * 16 ? K : split_k(K), Kb = K - Ka;

  // ---- fused one-launch path: 16-B aligned rows everywhere (row strides multiples of 4 floats) and an LDS image that fits ----
  const bool vec = ldgy % 4 == 0 && ldy % 4 == 0 && ldx % 4 == 0 && (!gx || ldgx % 4 == 0) && al16(gy) && al16(y) && al16(x) && (!gx || al16(gx)) &&
                   (N % 4 != 0 || al16(W)) && (Kb == 0 || Ka % 4 == 0);
  if (vec && br::dense_bwd_fused_lds(tiles16(N), tiles16(Ka)) <= 160 * 1024) {
    br::BwdArgs a{gy, ldgy, y, ldy, x, ldx, W, batch, Ka, N, act, out_mean, out_rstd, out_gamma, bn_sums, inv_batch, in_scale, in_shift,
                  keep, kw, inv_keep, in_mean, in_rstd, in_bn_sums, gx, ldgx, dW_slabs, slab_elems, (int64_t)K * N};
    const int fg = br::dense_bwd_fused_grid(batch);
    if (fg < n_slabs) {     // the slab count is sized for the two-kernel path: unused slabs must read as zeros (cleared by the launch itself)
      a.zero = dW_slabs + (int64_t)fg * slab_elems;
      a.zero_n = (int64_t)(n_slabs - fg) * slab_elems;
    }
    int rc = br::dense_backward_fused(a, s);
    if (rc != BR_OK || Kb == 0) return rc;
    a.zero = nullptr; a.zero_n = 0;
    // second K-half of a K > 128 layer: dz is formed again (identically); dW rows / gx columns / keep words of that half; db done
    a.x = x + Ka; a.W = W + (int64_t)Ka * N; a.K = Kb;
    if (in_scale) { a.scale = in_scale + Ka; a.shift = in_shift + Ka; }
    if (keep) a.keep = keep + Ka / 32;
    if (gx) a.gx = gx + Ka;
    a.slabs = dW_slabs + (int64_t)Ka * N; a.db_off = -1;
    return br::dense_backward_fused(a, s);
  }

  // ---- general two-kernel path ----
  OutXform to{out_mean, out_rstd, out_gamma, bn_sums, inv_batch};
  InXform tin{in_scale, in_shift, reinterpret_cast<const uint8_t*>(keep), (int64_t)kw * 4, inv_keep};
  InBn ibn{in_mean, in_rstd};
  if (Kb == 0)
    return dense_backward_part(gy, ldgy, y, ldy, x, ldx, W, batch, K, N, act, to, tin, ibn, gx, ldgx, dz_ws, dW_slabs, slab_elems,
                               (int64_t)K * N, pow2_ge(tiles16(K)), in_bn_sums, true, true, s);
  // two K-halves (see brDenseForward): dx A, dx B (dz is formed - identically - by both), then dW A, dW B
  InXform tb = tin;
  if (in_scale) { tb.scale = in_scale + Ka; tb.shift = in_shift + Ka; }
  if (keep) tb.keep = tin.keep + Ka / 8;
  int rc = BR_OK;
  if (gx) {
    rc = dense_backward_part(gy, ldgy, y, ldy, x, ldx, W, batch, Ka, N, act, to, tin, ibn, gx, ldgx, dz_ws, nullptr, 0, 0, 8, nullptr, true, false, s);
    if (rc != BR_OK) return rc;
    rc = dense_backward_part(gy, ldgy, y, ldy, x + Ka, ldx, W + (int64_t)Ka * N, batch, Kb, N, act, to, tb, ibn, gx + Ka, ldgx, dz_ws, nullptr, 0, 0,
                             8, nullptr, true, false, s);
  } else {   // no input gradient wanted: one dx launch still forms dz for the dW kernels
    rc = dense_backward_part(gy, ldgy, y, ldy, x, ldx, W, batch, Ka, N, act, to, tin, ibn, nullptr, ldgx, dz_ws, nullptr, 0, 0, 8, nullptr, true, false, s);
  }
  if (rc != BR_OK) return rc;
  rc = dense_backward_part(gy, ldgy, y, ldy, x, ldx, W, batch, Ka, N, act, to, tin, ibn, nullptr, ldgx, dz_ws, dW_slabs, slab_elems, (int64_t)K * N, 8,
                           nullptr, false, true, s);
  if (rc != BR_OK) return rc;
  return dense_backward_part(gy, ldgy, y, ldy, x + Ka, ldx, W, batch, Kb, N, act, to, tb, ibn, nullptr, ldgx, dz_ws, dW_slabs + (int64_t)Ka * N, slab_elems,
                             -1, 8, nullptr, false, true, s);
}

extern "C" int brReduceSlabs(const float* slabs, int n_slabs, int64_t slab_elems, float* out, brStream stream) {
  BR_CHECK_ARG(slabs && out && n_slabs >= 1 && slab_elems >= 1, "brReduceSlabs: bad args");
  reduce_slabs_kernel<<<(unsigned)ceil_div(slab_elems, 64), 1024, 0, (hipStream_t)stream>>>(slabs, n_slabs, slab_elems, out);
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
