// The tail of the NeuMF step on fp32 MFMA, one launch (trainers/NFC_plain.py:143-155, src/models/NeuMFModel.py:75-93):
//   [BatchNorm-2 finalize] -> T(a2) = keep/(1-p)·(a2·scale2 + shift2) -> Dense(n3) -> concat [GMF dot | a3] -> Dense(1) ->
//   sigmoid -> loss, and the backward of all of it down to gh2 = dL/d(BN2 output), with the parameter gradients
//   dW3 / db3 / dW4 / db4 (one slab per workgroup) and the BatchNorm-2 backward column sums.
// Round 1 ran this as plain VALU through LDS tiles (one launch, 41-44 us for 26 MB of traffic).  Here every wave owns
// 16-row tiles and everything with a contraction is an MFMA whose operands are already where the previous product left them:
//   z3   = T(a2)·W3            A = a2 straight from global in the A layout (as dense_fwd.hip), B = W3 image in LDS
//   a3   = act(z3 + b3)        C layout: lane (c16 = n, g) holds rows 4g..4g+3 of column n
//   lz   = [dot | a3]·w4 + b4  a3 through the wave's LDS patch into the A layout, B = w4 broadcast over the 16 output columns:
//                              every lane of a row group receives the logits of ITS four rows -> p, loss, dlogit in the C layout
//   dz3  = dlogit·w4·act'(a3)  C layout, no data movement
//   gh2  = keep/(1-p)·dz3·W3^T A = dz3 through the patch, B = W3 rows; epilogue: keep bits, BatchNorm-2 backward sums, patch
//                              transpose, 16-B row stores
//   dW3 += T(a2)^T·dz3         contraction over the tile's rows: k-step r <-> rows 4g+r, so B = dz3's C-layout registers as they
//                              stand and A = T() of the raw a2 values the gh2 epilogue loads for xhat anyway; the 64 x 16
//                              accumulator stays in registers over the wave's tiles
// 52 MFMAs per 16 rows; the launch is bound by its 26 MB of HBM traffic.
// Shapes: n2 <= 64, n3 <= 16, rows of a2 / gh2 16-B aligned (row strides multiples of 4 floats); others: tail.hip.
#include "dense.h"

namespace br {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kTmThreads = 512;
constexpr int kTmWaves = kTmThreads / 64;
constexpr int kTmRep = BR_STAT_REPLICAS;
constexpr int kTmPatchLd = 20;
constexpr int kTmKT = 4;                 // k-tiles of the 64-column padded input

__device__ __forceinline__ f32x4 mfma16t(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float keep_bit_t(float x, uint32_t w, uint32_t pos) {
  return __int_as_float(__float_as_int(x) & __builtin_amdgcn_sbfe((int)w, pos, 1u));
}

// one tile's operands: a2 in the A layout (row c16), raw a2 + keep words + dot / labels of the C-layout rows 4g..4g+3.
// Issued for the wave's first tile BEFORE the staging (BatchNorm finalize, W images, barrier) so that the tile's HBM latency runs
// beside the staging's two memory round trips instead of behind them.
struct TailTile {
  float4 av[kTmKT];
  uint32_t ka[2];
  float xr[4][kTmKT], dv[4], yv[4];
  uint32_t kc[4][2];
  int vm[4];
};
__device__ __forceinline__ void tail_load_tile(TailTile& t, const TailMArgs& a, int64_t tile, int c16, int g) {
  const int64_t batch = a.batch, rbase = tile << 4;
  const int n2 = a.n2;
  const int64_t arow = rbase + c16 < batch ? rbase + c16 : batch - 1;
  const float* pa = a.a2 + arow * a.lda2 + 4 * g;
#pragma unroll
  for (int j = 0; j < kTmKT; ++j) t.av[j] = *reinterpret_cast<const float4*>(16 * j + 4 * g < n2 ? pa + 16 * j : pa - 4 * g);
  t.ka[0] = t.ka[1] = 0xFFFFFFFFu;
  if (a.keep) { t.ka[0] = a.keep[arow * a.kw]; t.ka[1] = a.keep[arow * a.kw + (a.kw > 1 ? 1 : 0)]; }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = rbase + 4 * g + r;
    t.vm[r] = row < batch ? -1 : 0;
    const int64_t rc = row < batch ? row : batch - 1;
    const float* px = a.a2 + rc * a.lda2;
#pragma unroll
    for (int kt = 0; kt < kTmKT; ++kt) { const int k = kt * 16 + c16; t.xr[r][kt] = px[k < n2 ? k : 0]; }
    t.kc[r][0] = a.keep ? a.keep[rc * a.kw] : 0xFFFFFFFFu;
    t.kc[r][1] = a.keep ? a.keep[rc * a.kw + (a.kw > 1 ? 1 : 0)] : 0xFFFFFFFFu;
    t.dv[r] = a.dot[rc];
    t.yv[r] = a.labels[rc];
  }
}

__global__ __launch_bounds__(kTmThreads, 2) void neumf_tail_mfma_kernel(const TailMArgs a) {
  __shared__ __attribute__((aligned(16))) float Wf[4 * 4 * 16 * 4];      // z3 image [j][g][n][4]: W3[16j+4g+s][n]
  __shared__ __attribute__((aligned(16))) float Wb[64 * 16];             // gh2 image [k][n] row-major (n padded to 16)
  __shared__ float cst[4][64];                                            // scale2 | shift2 | mean2 | rstd2 (zero padded)
  __shared__ __attribute__((aligned(16))) float hv[2][16];               // b3 | w4 (a3 part), zero padded
  __shared__ __attribute__((aligned(16))) float patches[kTmWaves][16 * kTmPatchLd];
  __shared__ float wsum[kTmWaves][64 * 16];                               // per-wave dW3 partials [k][n]
  __shared__ float wcol[kTmWaves][2][16 + 2];                             // per-wave db3 | dW4(a3) columns, [16]: dW4(dot), [17]: db4
  __shared__ double wbn[2][64];                                           // BatchNorm-2 backward sums of the workgroup
  __shared__ double wmet[kTmWaves][BR_METRIC_SUMS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c16 = lane & 15, g = lane >> 4;
  const int n2 = a.n2, n3 = a.n3;
  const int64_t batch = a.batch;
  const int moff = a.mf_first ? 1 : 0;
  const int sigmask = a.act == BR_ACT_SIGMOID ? -1 : 0, relumask = a.act == BR_ACT_RELU ? -1 : 0;
  const float floor_ = a.act == BR_ACT_RELU ? 0.f : -__builtin_inff();

  const int64_t n_tiles = (batch + 15) >> 4;
  const int64_t tile0 = (int64_t)blockIdx.x * kTmWaves + wave;
  TailTile tt;
  if (tile0 < n_tiles) tail_load_tile(tt, a, tile0, c16, g);
  // ---------------- staging ----------------
  for (int t = threadIdx.x; t < 64; t += kTmThreads) {
    float sc = 0.f, sh = 0.f, mu = 0.f, rs = 0.f;
    if (t < n2) {
      if (a.stats2) {        // BatchNorm-2 finalize (brBnFinalize): biased batch variance, moving stats by workgroup 0
        double s1 = 0.0, s2 = 0.0;
        for (int r = 0; r < kTmRep; ++r) { s1 += a.stats2[(size_t)r * 2 * n2 + t]; s2 += a.stats2[(size_t)r * 2 * n2 + n2 + t]; }
        const double m = s1 / a.batch_total;
        double var = s2 / a.batch_total - m * m;
        if (var < 0.0) var = 0.0;
        mu = (float)m;
        const float varf = (float)var;
        rs = 1.0f / sqrtf(varf + a.bn_eps);
        sc = a.gamma2[t] * rs;
        sh = a.beta2[t] - mu * sc;
        if (blockIdx.x == 0) {
          a.out_scale[t] = sc; a.out_shift[t] = sh; a.out_mean[t] = mu; a.out_rstd[t] = rs;
          if (a.moving_mean) {
            a.moving_mean[t] = a.moving_mean[t] * a.bn_momentum + mu * (1.0f - a.bn_momentum);
            a.moving_var[t] = a.moving_var[t] * a.bn_momentum + varf * (1.0f - a.bn_momentum);
          }
        }
      } else { sc = a.scale2[t]; sh = a.shift2[t]; mu = a.mean2[t]; rs = a.rstd2[t]; }
    }
    cst[0][t] = sc; cst[1][t] = sh; cst[2][t] = mu; cst[3][t] = rs;
    wbn[0][t] = 0.0; wbn[1][t] = 0.0;
  }
  for (int t = threadIdx.x; t < 16; t += kTmThreads) {
    hv[0][t] = t < n3 ? a.b3[t] : 0.f;
    hv[1][t] = t < n3 ? a.w4[moff + t] : 0.f;
  }
  for (int t = threadIdx.x; t < 64 * 16; t += kTmThreads) {
    const int k = t >> 4, n = t & 15;
    const float w = (k < n2 && n < n3) ? a.W3[k * n3 + n] : 0.f;
    Wb[t] = w;
    Wf[(((k >> 4) * 4 + ((k >> 2) & 3)) * 16 + n) * 4 + (k & 3)] = w;
  }
  __syncthreads();

  const float wdot = a.w4[a.mf_first ? 0 : n3], bias4 = a.b4[0];
  float* patch = patches[wave];
  f32x4 dW[kTmKT];
#pragma unroll
  for (int kt = 0; kt < kTmKT; ++kt) dW[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float db3 = 0.f, dw4 = 0.f, dw4dot = 0.f, db4 = 0.f;              // C-layout column sums (column c16) / row-group sums
  float isum[kTmKT], isq[kTmKT];
#pragma unroll
  for (int kt = 0; kt < kTmKT; ++kt) { isum[kt] = 0.f; isq[kt] = 0.f; }
  double s_loss = 0.0, s_se = 0.0, s_ae = 0.0, s_ok = 0.0, s_bce = 0.0, s_tp = 0.0, s_fp = 0.0, s_fn = 0.0;
  const float4 w4q = *reinterpret_cast<const float4*>(&hv[1][4 * g]);    // B operand of the logit product: w4[4g+q]
  const float b3c = hv[0][c16], w4c = hv[1][c16];
  const float ik = a.inv_keep;

  for (int64_t tile = tile0; tile < n_tiles; tile += (int64_t)gridDim.x * kTmWaves) {
    const int64_t rbase = tile << 4;
    if (tile != tile0) tail_load_tile(tt, a, tile, c16, g);
    float4 (&av)[kTmKT] = tt.av;
    uint32_t (&ka)[2] = tt.ka;
    float (&xr)[4][kTmKT] = tt.xr;
    float (&dv)[4] = tt.dv;
    float (&yv)[4] = tt.yv;
    uint32_t (&kc)[4][2] = tt.kc;
    int (&vm)[4] = tt.vm;
    // ---- z3 = T(a2)·W3 ----
    f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < kTmKT; ++j) {
      const int k = 16 * j + 4 * g;
      const float4 sc = *reinterpret_cast<const float4*>(&cst[0][k]), sh = *reinterpret_cast<const float4*>(&cst[1][k]);
      const uint32_t w = ka[j >> 1], p0 = 16u * (j & 1) + 4u * g;
      const float t0 = keep_bit_t(fmaf(av[j].x, sc.x, sh.x), w, p0), t1 = keep_bit_t(fmaf(av[j].y, sc.y, sh.y), w, p0 + 1);
      const float t2 = keep_bit_t(fmaf(av[j].z, sc.z, sh.z), w, p0 + 2), t3 = keep_bit_t(fmaf(av[j].w, sc.w, sh.w), w, p0 + 3);
      const float4 b = *reinterpret_cast<const float4*>(Wf + ((j * 4 + g) * 16 + c16) * 4);
      z = mfma16t(t0, b.x, z); z = mfma16t(t1, b.y, z); z = mfma16t(t2, b.z, z); z = mfma16t(t3, b.w, z);
    }
    // ---- a3 (C layout: column n = c16, rows 4g+r), into the patch for the logit product ----
    const int cm = c16 < n3 ? -1 : 0;
    float a3v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float zz = fmaf(z[r], ik, b3c);
      const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(zz * -1.44269504088896340736f));
      const float m = fmaxf(zz, floor_);
      a3v[r] = __int_as_float(((__float_as_int(s) & sigmask) | (__float_as_int(m) & ~sigmask)) & cm);
      patch[(4 * g + r) * kTmPatchLd + c16] = a3v[r];
    }
    __builtin_amdgcn_wave_barrier();
    const float4 ar = *reinterpret_cast<const float4*>(patch + c16 * kTmPatchLd + 4 * g);      // a3[row c16][4g..4g+3]
    __builtin_amdgcn_wave_barrier();
    f32x4 lzv = (f32x4){0.f, 0.f, 0.f, 0.f};
    lzv = mfma16t(ar.x, w4q.x, lzv); lzv = mfma16t(ar.y, w4q.y, lzv); lzv = mfma16t(ar.z, w4q.z, lzv); lzv = mfma16t(ar.w, w4q.w, lzv);
    // ---- head: logit, p, loss, dlogit for this lane's 4 rows (the 16 lanes of a row group hold the same values) ----
    float dzl[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float lz = lzv[r] + bias4 + dv[r] * wdot;
      const float e = expf(-fabsf(lz));
      const float rinv = 1.0f / (1.0f + e);
      const float p = lz >= 0.f ? rinv : e * rinv;
      float l, d;
      const float bce = fmaxf(lz, 0.f) - lz * yv[r] + log1pf(e);
      if (a.loss == BR_LOSS_BCE) { l = bce; d = (p - yv[r]) * a.inv_batch; }
      else { l = (p - yv[r]) * (p - yv[r]); d = 2.f * (p - yv[r]) * p * (1.f - p) * a.inv_batch; }
      dzl[r] = __int_as_float(__float_as_int(d) & vm[r]);
      if (c16 == 0 && vm[r]) {
        const int64_t row = rbase + 4 * g + r;
        a.logit[row] = lz; a.prob[row] = p; a.ddot[row] = dzl[r] * wdot;
        s_loss += (double)l; s_se += (double)((p - yv[r]) * (p - yv[r])); s_ae += (double)fabsf(p - yv[r]);
        const bool pp = p > 0.5f, yp = yv[r] > 0.5f;
        s_ok += (pp == yp) ? 1.0 : 0.0;
        s_bce += (double)bce; s_tp += (pp && yp) ? 1.0 : 0.0; s_fp += (pp && !yp) ? 1.0 : 0.0; s_fn += (!pp && yp) ? 1.0 : 0.0;
        dw4dot = fmaf(dzl[r], dv[r], dw4dot);
        db4 += dzl[r];
      }
    }
    // ---- dz3 (C layout), head / layer-3 bias gradients, a3 out ----
    float dz3[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ds = a3v[r] * (1.f - a3v[r]);
      const float dr = a3v[r] > 0.f ? 1.f : 0.f;
      const float dl = __int_as_float((__float_as_int(ds) & sigmask) | (__float_as_int(dr) & relumask) | (0x3f800000 & ~(sigmask | relumask)));
      dz3[r] = __int_as_float(__float_as_int(dzl[r] * w4c * dl) & cm);
      dw4 = fmaf(dzl[r], a3v[r], dw4);
      db3 += dz3[r];
      patch[(4 * g + r) * kTmPatchLd + c16] = dz3[r];
      if (a.a3 && c16 < n3 && vm[r]) a.a3[(rbase + 4 * g + r) * n3 + c16] = a3v[r];
    }
    __builtin_amdgcn_wave_barrier();
    const float4 dzr = *reinterpret_cast<const float4*>(patch + c16 * kTmPatchLd + 4 * g);     // dz3[row c16][4g..4g+3]
    __builtin_amdgcn_wave_barrier();
    // ---- gh2 = keep/(1-p)·dz3·W3^T and dW3 += T(a2)^T·dz3, k-tile by k-tile ----
    const bool row_ok = rbase + c16 < batch;
    float* ghrow = a.gh2 + (rbase + (row_ok ? c16 : 0)) * a.ldgh2 + 4 * g;
#pragma unroll
    for (int kt = 0; kt < kTmKT; ++kt) {
      const int k = kt * 16 + c16;
      const float4 b = *reinterpret_cast<const float4*>(Wb + k * 16 + 4 * g);                   // W3[k][4g..4g+3]
      f32x4 gh = (f32x4){0.f, 0.f, 0.f, 0.f};
      gh = mfma16t(dzr.x, b.x, gh); gh = mfma16t(dzr.y, b.y, gh); gh = mfma16t(dzr.z, b.z, gh); gh = mfma16t(dzr.w, b.w, gh);
      const float sck = cst[0][k], shk = cst[1][k], muk = cst[2][k], rsk = cst[3][k];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t w = kc[r][kt >> 1], pos = 16u * (kt & 1) + c16;
        const float dh = keep_bit_t(gh[r] * ik, w, pos);                                        // (0 for rows past the batch: dz3 is)
        isum[kt] += dh;
        isq[kt] = fmaf(dh, (xr[r][kt] - muk) * rsk, isq[kt]);
        patch[(4 * g + r) * kTmPatchLd + c16] = dh;
        // A operand of dW3: T() of the raw value, rows 4g+r as k-step r
        const float tx = keep_bit_t(fmaf(xr[r][kt], sck, shk), w, pos);
        dW[kt] = mfma16t(tx, dz3[r], dW[kt]);
      }
      __builtin_amdgcn_wave_barrier();
      const float4 o = *reinterpret_cast<const float4*>(patch + c16 * kTmPatchLd + 4 * g);
      __builtin_amdgcn_wave_barrier();
      if (row_ok && kt * 16 + 4 * g < n2) *reinterpret_cast<float4*>(ghrow + kt * 16) = o;
    }
  }

  // ---------------- workgroup reductions ----------------
#pragma unroll
  for (int kt = 0; kt < kTmKT; ++kt) {
    // dW3 tile: lane (n = c16, g), register r <-> k = kt*16 + 4g + r
#pragma unroll
    for (int r = 0; r < 4; ++r) wsum[wave][(kt * 16 + 4 * g + r) * 16 + c16] = dW[kt][r];
    float sv = isum[kt], q = isq[kt];
    sv += __shfl_xor(sv, 16, 64); sv += __shfl_xor(sv, 32, 64);
    q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
    if (g == 0) { atomicAdd(&wbn[0][kt * 16 + c16], (double)sv); atomicAdd(&wbn[1][kt * 16 + c16], (double)q); }
  }
  {
    float v0 = db3, v1 = dw4;
    v0 += __shfl_xor(v0, 16, 64); v0 += __shfl_xor(v0, 32, 64);
    v1 += __shfl_xor(v1, 16, 64); v1 += __shfl_xor(v1, 32, 64);
    if (g == 0) { wcol[wave][0][c16] = v0; wcol[wave][1][c16] = v1; }
    // row-group sums live in the c16 == 0 lanes
    float u0 = c16 == 0 ? dw4dot : 0.f, u1 = c16 == 0 ? db4 : 0.f;
    u0 = group_sum<64>(u0); u1 = group_sum<64>(u1);
    if (lane == 0) { wcol[wave][0][16] = u0; wcol[wave][0][17] = u1; }
    s_loss = wave_sum_d(s_loss); s_se = wave_sum_d(s_se); s_ae = wave_sum_d(s_ae); s_ok = wave_sum_d(s_ok);
    s_bce = wave_sum_d(s_bce); s_tp = wave_sum_d(s_tp); s_fp = wave_sum_d(s_fp); s_fn = wave_sum_d(s_fn);
    if (lane == 0) {
      wmet[wave][0] = s_loss; wmet[wave][1] = s_se; wmet[wave][2] = s_ae; wmet[wave][3] = s_ok;
      wmet[wave][4] = s_bce; wmet[wave][5] = s_tp; wmet[wave][6] = s_fp; wmet[wave][7] = s_fn;
    }
  }
  __syncthreads();
  // slab of this workgroup: [dW3 (n2 x n3) | db3 (n3) | dW4 (n3 + 1, concat order) | db4]; dW3 carries the 1/(1-p) folded out of T()
  const int64_t slab_el = (int64_t)n2 * n3 + 2 * n3 + 2;
  float* slab = a.slabs + (int64_t)blockIdx.x * slab_el;
  // the caller's slab count is sized for 128-row workgroups: the slabs nobody owns must read as zeros (a memset node of its own
  // in the step's graph would cost more than these stores)
  for (int64_t sl = (int64_t)blockIdx.x + gridDim.x; sl < a.n_slabs; sl += gridDim.x)
    for (int t = threadIdx.x; t < slab_el; t += kTmThreads) a.slabs[sl * slab_el + t] = 0.f;
  for (int t = threadIdx.x; t < 64 * 16; t += kTmThreads) {
    const int k = t >> 4, n = t & 15;
    if (k < n2 && n < n3) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < kTmWaves; ++w) s += wsum[w][t];
      slab[k * n3 + n] = s * ik;
    }
  }
  if (threadIdx.x < 18) {
    const int t = threadIdx.x;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int w = 0; w < kTmWaves; ++w) { s0 += wcol[w][0][t]; s1 += t < 16 ? wcol[w][1][t] : 0.f; }
    if (t < n3) { slab[n2 * n3 + t] = s0; slab[n2 * n3 + n3 + moff + t] = s1; }
    if (t == 16) slab[n2 * n3 + n3 + (a.mf_first ? 0 : n3)] = s0;
    if (t == 17) slab[n2 * n3 + 2 * n3 + 1] = s0;
  }
  if (threadIdx.x >= 64 && threadIdx.x < 64 + BR_METRIC_SUMS && a.msums) {
    const int t = threadIdx.x - 64;
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < kTmWaves; ++w) v += wmet[w][t];
    atomicAdd(a.msums + (size_t)(blockIdx.x & (BR_SUM_SLOTS - 1)) * BR_METRIC_SUMS + t, v);
  }
  if (a.bn_sums && threadIdx.x >= 128 && threadIdx.x < 128 + 64) {
    const int k = threadIdx.x - 128;
    if (k < n2) {
      double* rep = a.bn_sums + (size_t)(blockIdx.x % kTmRep) * 2 * n2;
      atomicAdd(rep + k, wbn[0][k]);
      atomicAdd(rep + n2 + k, wbn[1][k]);
    }
  }
}

int tail_mfma_grid(int64_t batch) {
  const int64_t t = ceil_div(ceil_div(batch > 0 ? batch : 1, 16), kTmWaves);
  // 200+ VGPRs: 2 waves/SIMD = ONE 512-thread workgroup per CU.  A grid of 512 ran as two generations of workgroups, each with its
  // own staging and its own slab / statistics tail; 256 workgroups loop over two tiles per wave instead
  return (int)(t < 1 ? 1 : (t > 256 ? 256 : t));
}
void launch_tail_mfma(const TailMArgs& a, int grid, hipStream_t s) { neumf_tail_mfma_kernel<<<(unsigned)grid, kTmThreads, 0, s>>>(a); }

}  // namespace br
