// T1-T4 + B1: the MLP tower on fp32 MFMA (v_mfma_f32_16x16x4_f32: exact f32 fma chains, the
// only MFMA form that meets the 1e-5 logit/loss tolerance; 157 TFLOP/s peak on MI355X).
//
// Structure (both kernels): 512-thread workgroups (8 waves), TWO resident per CU (4 waves per SIMD,
// <= 128 VGPRs) so one workgroup's load / epilogue phases overlap the other's MFMA phase — a single
// barrier-synchronised workgroup per CU left the matrix pipe idle 2/3 of the time (rocprofv3 PMC:
// SQ_VALU_MFMA_BUSY_CYCLES = 32 % of the kernel, round-1 profile).
//   * W never goes through LDS: every wave keeps the <= 32 fragment registers of the 16 weight
//     columns (forward: n-tile, backward: k-tile) it owns for the whole launch.
//   * One LDS image of the current batch tile; the raw global loads of tile t+1 are issued before
//     the MFMA phase of tile t and wait in registers (global -> reg -> LDS staging); BatchNorm-affine
//     + Philox dropout are applied on the way in.
//   * K is contracted in a lane-permuted order (lane group g = lane>>4 owns k = 16j+4g..+3) so one
//     ds_read_b128 feeds four MFMA k-steps; A and B use the same permutation.
//   forward : wave (n-tile nt, row group) : y[rows][16 cols] = act(T(x)·W + b) ; BN column sums.
//   backward: dz = act'(y)·BN-backward(gy) elementwise into LDS; wave (k-tile kt, row group):
//             dx[rows][16 cols of kt] = dz·W^T  and  dW[16 rows of kt][N] += T(x)^T·dz from the
//             same LDS tiles; dW/db live in registers across the workgroup's tiles and leave as
//             one slab per (workgroup, row group) -> fixed-order reduce (reproducible, no atomics).
// Column statistics (BatchNorm sums) leave as double atomics into BR_STAT_REPLICAS replicas
// (replica = workgroup % 8) so no address takes more than grid/8 serialised adds.
#include <stdlib.h>

#include "common.h"
#include "philox.h"

namespace br {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kThreads = 512;
constexpr int kMaxT = 8;         // max 16-wide tiles along K or N (=> K,N <= 128)
constexpr int kMaxGrid = 512;    // two workgroups per CU
constexpr int kRep = BR_STAT_REPLICAS;

// rows per tile: every one of the 8 waves must own at least one 16-row tile
__host__ __device__ constexpr int fwd_tm(int ntp) { return 16 * ((8 / ntp) > 4 ? (8 / ntp) : 4); }   // 64 (128 when N <= 16)
__host__ __device__ constexpr int bwd_tm(int ktp) { return 16 * ((8 / ktp) > 2 ? (8 / ktp) : 2); }   // 32 (64 / 128 for K <= 32 / 16)

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct InXform {       // T(x): BatchNorm affine of the producer + dropout, applied on load
  const float* scale;  // (K) or null (global; the kernels stage scale|shift into LDS once)
  const float* shift;  // (K) or null
  DropoutCfg drop;
};

// 8 consecutive floats of row gr starting at column c (zeros outside [0,K) x [0,batch))
__device__ __forceinline__ void load8(float (&v)[8], const float* __restrict__ base, int64_t ld, int64_t gr, int64_t batch,
                                      int c, int K, bool vec_ok) {
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 0.f;
  if (gr >= batch || c >= K) return;
  const float* src = base + gr * ld + c;
  if (vec_ok && c + 7 < K) {
    const float4 a = *reinterpret_cast<const float4*>(src);
    const float4 b = *reinterpret_cast<const float4*>(src + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) if (c + i < K) v[i] = src[i];
  }
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
  if (t.drop.thr) {
    bits = dropout_keep8(t.drop, grow, (uint32_t)(c >> 3));
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = ((bits >> i) & 1u) ? v[i] * t.drop.inv_keep : 0.f;
  }
  return bits;
}

__device__ __forceinline__ void store8_lds(float* dst, const float (&v)[8]) {
  *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// ------------------------------------------------------------------------------------ forward
// NTP = n-tiles covered by the 8 waves (1,2,4,8): wave -> (n-tile = wave % NTP, row group = wave / NTP)
// KJ  = K tiles of 16 (compile-time so the k-loop has no branches and LDS reads hoist freely)
template <int NTP, int KJ>
__global__ __launch_bounds__(kThreads, 4) void dense_fwd_kernel(const float* __restrict__ x, int64_t ldx_g, const float* __restrict__ W,
                                                                 const float* __restrict__ bias, float* __restrict__ y, int64_t ldy,
                                                                 int64_t batch, int K, int N, int act, InXform tin, int64_t row0,
                                                                 double* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int RGN = 8 / NTP;            // row groups
  constexpr int TM = fwd_tm(NTP);         // rows per tile
  constexpr int RT = TM / 16 / RGN;       // row tiles per wave (4, 2, 1, 1)
  constexpr int Kp = KJ * 16, ldx = Kp + 4;
  constexpr int CPR = Kp / 8;             // 8-float chunks per row
  constexpr int MAXC = (TM * CPR + kThreads - 1) / kThreads;
  float* Xs = smem;
  float* ssb = smem + TM * ldx;           // [scale Kp | shift Kp]
  const float* ss = tin.scale ? ssb : nullptr;
  if (tin.scale)
    for (int k = threadIdx.x; k < Kp; k += blockDim.x) {
      ssb[k] = k < K ? tin.scale[k] : 0.f;
      ssb[Kp + k] = k < K ? tin.shift[k] : 0.f;
    }

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c16 = lane & 15, g = lane >> 4;
  const int nt = wave % NTP, rgp = wave / NTP;
  const int ncol = nt * 16 + c16;
  const bool vec_ok = (ldx_g % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);

  // this wave's W fragments: bw[4j+s] = W[16j+4g+s][ncol]
  float bw[4 * KJ];
#pragma unroll
  for (int j = 0; j < KJ; ++j)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = 16 * j + 4 * g + s;
      bw[4 * j + s] = (k < K && ncol < N) ? W[k * N + ncol] : 0.f;
    }
  const float bcol = (bias && ncol < N) ? bias[ncol] : 0.f;
  float ssum = 0.f, ssq = 0.f;

  float pre[MAXC][8];
  const int64_t n_tiles = (batch + TM - 1) / TM;

  auto load_tile = [&](int64_t tile) {
    const int64_t row_base = tile * TM;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int idx = threadIdx.x + kThreads * i;
      if (idx < TM * CPR) {
        const int r = idx / CPR, c = (idx - r * CPR) << 3;
        load8(pre[i], x, ldx_g, row_base + r, batch, c, K, vec_ok);
      }
    }
  };
  auto write_tile = [&](int64_t tile) {
    const int64_t row_base = tile * TM;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int idx = threadIdx.x + kThreads * i;
      if (idx < TM * CPR) {
        const int r = idx / CPR, c = (idx - r * CPR) << 3;
        xform8(pre[i], tin, ss, Kp, row0 + row_base + r, c, K, (row_base + r) < batch && c < K);
        store8_lds(Xs + r * ldx + c, pre[i]);
      }
    }
  };

  int64_t tile = blockIdx.x;
  if (tile < n_tiles) load_tile(tile);
  __syncthreads();                 // scale|shift staged
  if (tile < n_tiles) write_tile(tile);
  __syncthreads();
  while (tile < n_tiles) {
    const int64_t next = tile + gridDim.x;
    if (next < n_tiles) load_tile(next);
    // ---- MFMA: RT row tiles (independent accumulator chains) x this wave's n-tile ----
    const int64_t row_base = tile * TM;
    f32x4 acc[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const float4 a4 = *reinterpret_cast<const float4*>(Xs + ((rgp + RGN * i) * 16 + c16) * ldx + 16 * j + 4 * g);
        acc[i] = mfma16(a4.x, bw[4 * j + 0], acc[i]);
        acc[i] = mfma16(a4.y, bw[4 * j + 1], acc[i]);
        acc[i] = mfma16(a4.z, bw[4 * j + 2], acc[i]);
        acc[i] = mfma16(a4.w, bw[4 * j + 3], acc[i]);
      }
    }
    // ---- epilogue: lane holds rows 4g..4g+3 of each of its row tiles, column ncol ----
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int rt = rgp + RGN * i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t gr = row_base + rt * 16 + 4 * g + r;
        const float v = act_apply(acc[i][r] + bcol, act);
        if (gr < batch && ncol < N) {
          y[gr * ldy + ncol] = v;
          ssum += v;
          ssq += v * v;
        }
      }
    }
    __syncthreads();               // everyone done reading the LDS image
    if (next < n_tiles) write_tile(next);
    __syncthreads();
    tile = next;
  }
  if (stats) {
    double s = (double)ssum, q = (double)ssq;
    s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
    q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
    if (g == 0 && ncol < N) {
      double* rep = stats + (size_t)(blockIdx.x % kRep) * 2 * N;
      atomicAdd(rep + ncol, s);
      atomicAdd(rep + N + ncol, q);
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

__global__ void bn_param_grads_kernel(const double* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta, int N) {
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

// NT = n-tiles (1..8); KTP = pow2 >= number of k-tiles: wave -> (k-tile = wave % KTP, row group = wave / KTP)
template <int NT, int KTP>
__global__ __launch_bounds__(kThreads, 4) void dense_bwd_kernel(const float* __restrict__ gy, int64_t ldgy, const float* __restrict__ y, int64_t ldy,
                                                                 const float* __restrict__ x, int64_t ldx_g, const float* __restrict__ W,
                                                                 int64_t batch, int K, int N, int act, OutXform to, InXform tin, InBn ibn,
                                                                 int64_t row0, float* __restrict__ gx, int64_t ldgx, float* __restrict__ slabs,
                                                                 double* __restrict__ in_sums) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int Np = NT * 16, ldz = Np + 4;
  constexpr int RGN = 8 / KTP;                      // row groups
  constexpr int TM = bwd_tm(KTP);                   // rows per tile
  constexpr int RT = TM / 16 / RGN;                 // row tiles per wave (2, 1, 1, 1)
  const int KT = (K + 15) >> 4, Kp = KT << 4, ldx = Kp + 4;
  const int mkld = Kp >> 3;                         // mask bytes per row
  float* Xs = smem;                                 // [TM][ldx]   T(x)
  float* Zs = Xs + TM * ldx;                        // [TM][ldz]   dz
  uint8_t* Mk = reinterpret_cast<uint8_t*>(Zs + TM * ldz);   // [TM][mkld] dropout keep bits
  float* Cs = Zs + TM * ldz + ((TM * mkld + 3) >> 2);        // [4][Np] per-column constants of the out BN
  float* ssb = Cs + 4 * Np;                         // [scale Kp | shift Kp] of the in transform
  const float* ss = tin.scale ? ssb : nullptr;
  if (tin.scale)
    for (int k = threadIdx.x; k < Kp; k += blockDim.x) {
      ssb[k] = k < K ? tin.scale[k] : 0.f;
      ssb[Kp + k] = k < K ? tin.shift[k] : 0.f;
    }

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c16 = lane & 15, g = lane >> 4;
  const int kt = wave % KTP, rg = wave / KTP;
  const bool kt_live = kt < KT;
  const int kcol = kt * 16 + c16;
  const bool xvec = (ldx_g % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  const bool gvec = (ldgy % 4 == 0) && ((reinterpret_cast<uintptr_t>(gy) & 15) == 0);
  const bool yvec = (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);

  for (int n = threadIdx.x; n < Np; n += blockDim.x) {
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
  // W^T fragments of this wave's k-tile: bwt[4j+s] = W[kcol][16j+4g+s]
  float bwt[4 * NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int n = 16 * j + 4 * g + s;
      bwt[4 * j + s] = (kt_live && kcol < K && n < N) ? W[(int64_t)kcol * N + n] : 0.f;
    }
  f32x4 dW[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) dW[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float db_acc = 0.f;                 // thread t < Np owns column t of db
  float isum = 0.f, isq = 0.f;        // in-BN sums of column kcol
  const float imean = (ibn.mean && kt_live && kcol < K) ? ibn.mean[kcol] : 0.f;
  const float irstd = (ibn.mean && kt_live && kcol < K) ? ibn.rstd[kcol] : 0.f;

  constexpr int ZCR = Np / 8;                       // dz chunks per row
  constexpr int MAXZ = (TM * ZCR + kThreads - 1) / kThreads;
  constexpr int MAXX = (TM * (kMaxT * 16 / 8) + kThreads - 1) / kThreads;   // bound for Kp <= 128
  const int xc_row = Kp >> 3, n_xc = TM * xc_row;
  float pz[MAXZ][8], px[MAXX][8];     // dz is formed at load time: one register set instead of gy + y
  const int64_t n_tiles = (batch + TM - 1) / TM;

  auto load_tile = [&](int64_t tile) {
    const int64_t row_base = tile * TM;
#pragma unroll
    for (int i = 0; i < MAXZ; ++i) {
      const int idx = threadIdx.x + kThreads * i;
      if (idx < TM * ZCR) {
        const int r = idx / ZCR, c = (idx - r * ZCR) << 3;
        const bool live = (row_base + r) < batch;
        float vg[8], vy[8];
        load8(vg, gy, ldgy, row_base + r, batch, c, N, gvec);
        load8(vy, y, ldy, row_base + r, batch, c, N, yvec);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int n = c + e;
          float da = vg[e];
          // da = gamma*rstd * (gy - mean(gy) - xhat*mean(gy*xhat)), xhat = (y-mu)*rstd
          if (to.mean) da = Cs[n] * (vg[e] - Cs[Np + n] - (vy[e] - Cs[3 * Np + n]) * Cs[2 * Np + n]);
          pz[i][e] = (live && n < N) ? da * act_grad_from_out(vy[e], act) : 0.f;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < MAXX; ++i) {
      const int idx = threadIdx.x + kThreads * i;
      if (idx < n_xc) {
        const int r = idx / xc_row, c = (idx - r * xc_row) << 3;
        load8(px[i], x, ldx_g, row_base + r, batch, c, K, xvec);
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
        const uint32_t bits = xform8(px[i], tin, ss, Kp, row0 + row_base + r, c, K, (row_base + r) < batch && c < K);
        store8_lds(Xs + r * ldx + c, px[i]);
        Mk[r * mkld + (c >> 3)] = (uint8_t)bits;
      }
    }
  };

  int64_t tile = blockIdx.x;
  __syncthreads();   // Cs, scale|shift visible
  if (tile < n_tiles) load_tile(tile);
  if (tile < n_tiles) write_tile(tile);
  __syncthreads();
  while (tile < n_tiles) {
    const int64_t next = tile + gridDim.x;
    if (next < n_tiles) load_tile(next);
    const int64_t row_base = tile * TM;
    if (kt_live) {
      // ---- dx[rows of my row tiles][k-tile] = dz · W^T ; contraction over n ----
      if (gx) {
        f32x4 acc[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
          for (int i = 0; i < RT; ++i) {
            const float4 a4 = *reinterpret_cast<const float4*>(Zs + ((rg + RGN * i) * 16 + c16) * ldz + 16 * j + 4 * g);
            acc[i] = mfma16(a4.x, bwt[4 * j + 0], acc[i]);
            acc[i] = mfma16(a4.y, bwt[4 * j + 1], acc[i]);
            acc[i] = mfma16(a4.z, bwt[4 * j + 2], acc[i]);
            acc[i] = mfma16(a4.w, bwt[4 * j + 3], acc[i]);
          }
        }
        if (kcol < K) {
#pragma unroll
          for (int i = 0; i < RT; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int lr = (rg + RGN * i) * 16 + 4 * g + r;
              const int64_t gr = row_base + lr;
              if (gr < batch) {
                // gx = gradient w.r.t. the producer's BN output h (dropout transposed here)
                const bool keep = (Mk[lr * mkld + (kcol >> 3)] >> (kcol & 7)) & 1;
                const float dh = keep ? acc[i][r] * tin.drop.inv_keep : 0.f;
                gx[gr * ldgx + kcol] = dh;
                if (ibn.mean && keep) {
                  const float xhat = (x[gr * ldx_g + kcol] - imean) * irstd;
                  isum += dh;
                  isq += dh * xhat;
                }
              }
            }
          }
        }
      }
      // ---- dW[k-tile rows][N] += T(x)^T · dz ; contraction over my row tiles' rows ----
#pragma unroll
      for (int i = 0; i < RT; ++i) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int r = (rg + RGN * i) * 16 + 4 * g + s;
          const float a = Xs[r * ldx + kcol];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) dW[nt] = mfma16(a, Zs[r * ldz + nt * 16 + c16], dW[nt]);
        }
      }
    }
    // ---- db: thread t < Np sums column t of dz ----
    if (threadIdx.x < Np) {
      float sacc = 0.f;
#pragma unroll 8
      for (int r = 0; r < TM; ++r) sacc += Zs[r * ldz + threadIdx.x];
      db_acc += sacc;
    }
    __syncthreads();               // everyone done reading the LDS images
    if (next < n_tiles) write_tile(next);
    __syncthreads();
    tile = next;
  }

  // ---- slab (workgroup, row group): [dW (K x N) | db (N)] ----
  const int64_t slab_elems = (int64_t)K * N + N;
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
  if ((int)threadIdx.x < N) {
    for (int r = 0; r < RGN; ++r)
      slabs[((int64_t)blockIdx.x * RGN + r) * slab_elems + (int64_t)K * N + threadIdx.x] = (r == 0) ? db_acc : 0.f;
  }
  if (in_sums && ibn.mean && kt_live) {
    double s = (double)isum, q = (double)isq;
    s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
    q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
    if (g == 0 && kcol < K) {
      double* rep = in_sums + (size_t)(blockIdx.x % kRep) * 2 * K;
      atomicAdd(rep + kcol, s);
      atomicAdd(rep + K + kcol, q);
    }
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
  double s_loss = 0.0, s_se = 0.0, s_ae = 0.0, s_ok = 0.0;
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
  __shared__ double redd[4][4];
  __shared__ float redf[4][34];
  s_loss = wave_sum_d(s_loss); s_se = wave_sum_d(s_se); s_ae = wave_sum_d(s_ae); s_ok = wave_sum_d(s_ok);
  if (lane == 0) { redd[wave][0] = s_loss; redd[wave][1] = s_se; redd[wave][2] = s_ae; redd[wave][3] = s_ok; }
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
  // 4 metric sums per slot, slot = workgroup & 63: same-line double atomics serialise (~12 ns each)
  if (threadIdx.x < 4 && sums)
    atomicAdd(sums + (size_t)(blockIdx.x & (BR_SUM_SLOTS - 1)) * 4 + threadIdx.x,
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
constexpr int kMaxDynLds = 80 * 1024 - 1024;   // two workgroups per CU

template <int NTP, int KJ>
static void launch_fwd(unsigned grid, size_t shmem, hipStream_t s, const float* x, int64_t ldx, const float* W, const float* bias, float* y,
                       int64_t ldy, int64_t batch, int K, int N, int act, InXform t, int64_t row0, double* stats) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)dense_fwd_kernel<NTP, KJ>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynLds);
    attr_set = true;
  }
  dense_fwd_kernel<NTP, KJ><<<grid, kThreads, shmem, s>>>(x, ldx, W, bias, y, ldy, batch, K, N, act, t, row0, stats);
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
  const int NTP = pow2_ge(tiles16(N)), KJ = tiles16(K), Kp = KJ * 16;
  const int tm = fwd_tm(NTP);
  const size_t shmem = ((size_t)tm * (Kp + 4) + 2 * (size_t)Kp) * sizeof(float);
  InXform t{in_scale, in_shift, make_dropout(drop_p, seed, step, site)};
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = grid_for(batch, tm);
#define BR_FWD_KJ(NTPv, KJv) case KJv: launch_fwd<NTPv, KJv>(grid, shmem, s, x, ldx, W, bias, y, ldy, batch, K, N, act, t, row0, stats); break;
#define BR_FWD(NTPv)                                                                                            \
  case NTPv:                                                                                                    \
    switch (KJ) { BR_FWD_KJ(NTPv, 1) BR_FWD_KJ(NTPv, 2) BR_FWD_KJ(NTPv, 3) BR_FWD_KJ(NTPv, 4) BR_FWD_KJ(NTPv, 5) \
                  BR_FWD_KJ(NTPv, 6) BR_FWD_KJ(NTPv, 7) BR_FWD_KJ(NTPv, 8) default: break; }                   \
    break;
  switch (NTP) {
    BR_FWD(1) BR_FWD(2) BR_FWD(4) BR_FWD(8)
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
  (void)N;
  const int KTP = pow2_ge(tiles16(K < 1 ? 1 : K));
  return (int)grid_for(batch, bwd_tm(KTP)) * (8 / KTP);
}

template <int NT, int KTP>
static void launch_bwd(unsigned grid, size_t shmem, hipStream_t s, const float* gy, int64_t ldgy, const float* y, int64_t ldy, const float* x,
                       int64_t ldx, const float* W, int64_t batch, int K, int N, int act, OutXform to, InXform tin, InBn ibn, int64_t row0,
                       float* gx, int64_t ldgx, float* slabs, double* in_sums) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)dense_bwd_kernel<NT, KTP>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynLds);
    attr_set = true;
  }
  dense_bwd_kernel<NT, KTP><<<grid, kThreads, shmem, s>>>(gy, ldgy, y, ldy, x, ldx, W, batch, K, N, act, to, tin, ibn, row0, gx, ldgx, slabs, in_sums);
}

extern "C" int brDenseBackward(const float* gy, int64_t ldgy, const float* y, int64_t ldy, const float* x, int64_t ldx,
                               const float* W, int64_t batch, int K, int N, int act, const float* out_mean,
                               const float* out_rstd, const float* out_gamma, const double* bn_sums, double batch_total,
                               const float* in_scale, const float* in_shift, const float* in_mean, const float* in_rstd,
                               float in_drop_p, uint32_t in_site, uint64_t seed, uint32_t step, int64_t row0, float* gx,
                               int64_t ldgx, float* dW_slabs, int n_slabs, double* in_bn_sums, brStream stream) {
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
  const int KT = tiles16(K), NT = tiles16(N);
  const int Kp = KT * 16, Np = NT * 16;
  const int KTP = pow2_ge(KT), tm = bwd_tm(KTP);
  const unsigned grid = grid_for(batch, tm);
  const int want = brDenseBackwardSlabs(batch, K, N);
  BR_CHECK_ARG(n_slabs == want, "brDenseBackward: n_slabs %d != brDenseBackwardSlabs() %d", n_slabs, want);
  const size_t shmem = ((size_t)tm * (Kp + 4) + (size_t)tm * (Np + 4) + (size_t)((tm * (Kp / 8) + 3) / 4) + 4 * (size_t)Np + 2 * (size_t)Kp) * sizeof(float);
  OutXform to{out_mean, out_rstd, out_gamma, bn_sums, (float)(1.0 / (batch_total > 0 ? batch_total : (double)batch))};
  InXform tin{in_scale, in_shift, make_dropout(in_drop_p, seed, step, in_site)};
  InBn ibn{in_mean, in_rstd};
  hipStream_t s = (hipStream_t)stream;
#define BR_BWD_K(NTv, KTPv) case KTPv: launch_bwd<NTv, KTPv>(grid, shmem, s, gy, ldgy, y, ldy, x, ldx, W, batch, K, N, act, to, tin, ibn, row0, gx, ldgx, dW_slabs, in_bn_sums); break;
#define BR_BWD(NTv) \
  case NTv:         \
    switch (KTP) { BR_BWD_K(NTv, 1) BR_BWD_K(NTv, 2) BR_BWD_K(NTv, 4) BR_BWD_K(NTv, 8) default: break; } \
    break;
  switch (NT) {
    BR_BWD(1) BR_BWD(2) BR_BWD(3) BR_BWD(4) BR_BWD(5) BR_BWD(6) BR_BWD(7) BR_BWD(8)
    default: br::set_error("brDenseBackward: unsupported N"); return BR_ERR_UNSUPPORTED;
  }
  BR_CHECK_LAUNCH("brDenseBackward");
  return BR_OK;
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
