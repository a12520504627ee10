// T3 + T4 + L1/L2 + their backward in ONE launch: the tail of the NeuMF graph that has no batch-wide
// dependency inside it (trainers/NFC_plain.py:143-155: Dropout -> Dense(10) -> concat [GMF dot | MLP] ->
// Dense(1) -> sigmoid -> loss; src/models/NeuMFModel.py:75-93 for variant B).
//
//   per row r:   x   = keep(r,k)/(1-p) * (a2[r,k] * scale2[k] + shift2[k])        (BN2 affine + dropout site 2)
//                a3  = act(x W3 + b3) ; logit = b4 + w4 . [dot | a3] ; p = sigmoid(logit) ; loss, dlogit
//                dz3 = dlogit * w4[a3 part] * act'(a3)
//                gh2 = keep/(1-p) * (dz3 W3^T)                                      (gradient w.r.t. BN2's output)
//   per column:  dW3 += x^T dz3, db3 += dz3, dW4 += dlogit [dot | a3], db4 += dlogit   -> one slab per workgroup
//                BN2 backward sums: sum_r gh2, sum_r gh2 * xhat2                      -> double atomics (8 replicas)
//
// The separate launches this replaces (dense_fwd[50x10] 13 us, head 14 us, reduce 5 us, dense_dx 17 us,
// dense_dw 16 us, reduce 7 us at batch 65 536) move 26 MB in total: the arithmetic is 3 x 33 MFLOP, far below
// what MFMA tiles are worth at N = 10, so this kernel is plain VALU, one thread per row, with every global
// access staged through LDS tiles (coalesced in, coalesced out).  128 rows per workgroup, 4 lanes per row
// (one thread per row left a single wave per SIMD with every LDS latency exposed: 57 us), 2 tiles of LDS.
#include "common.h"
#include "dense.h"

namespace br {

constexpr int kTailRows = 128;      // rows per workgroup
constexpr int kTailLpr = 4;         // lanes per row: lane q owns the 8-column chunks q, q+4, q+8, q+12 of its row
constexpr int kTailThreads = kTailRows * kTailLpr;
constexpr int kTailWaves = kTailThreads / 64;
constexpr int kTailRep = BR_STAT_REPLICAS;

struct TailArgs {
  const float* a2; int64_t lda2;
  const float *W3, *b3, *w4, *b4, *dot, *labels;
  const float *scale2, *shift2, *mean2, *rstd2;
  const uint8_t* keep; int64_t keep_ld; float inv_keep;     // keep-bit plane of dropout site 2 as bytes (byte c8 of a row = its chunk c8), or null
  int64_t batch;
  int n2, n3, act, mf_first, loss;
  float inv_batch;
  float *a3, *logit, *prob, *ddot, *gh2;
  int64_t ldgh2;
  double *msums, *bn_sums;
  float* slabs;
};

// N3P: n3 rounded up to a multiple of 4 (register arrays and the LDS images of W3 / dz3 use it)
template <int N3P>
__global__ __launch_bounds__(kTailThreads) void neumf_tail_kernel(TailArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n2 = a.n2, n3 = a.n3, ld = n2 | 1;             // odd row stride
  float* A = smem;                                          // [128][ld]  raw a2, later gh2
  float* X = A + kTailRows * ld;                            // [128][ld]  T(a2), later gh2 * xhat
  float* Ws = X + kTailRows * ld;                           // [n2][N3P]  W3 (zero padded columns)
  float* DZ = Ws + n2 * N3P;                                // [128][N3P + 1]
  float* cst = DZ + kTailRows * (N3P + 1);                  // [scale | shift | mean | rstd] (n2 each)
  float* hw = cst + 4 * n2;                                 // [b3 (N3P) | w4 for a3 (N3P)]
  float* A3s = hw + 2 * N3P;                                // [128][N3P + 1]  a3 on its way out (coalesced store)
  __shared__ float redf[kTailWaves][N3P + 2];
  __shared__ double redd[kTailWaves][BR_METRIC_SUMS];
  __shared__ double colred[2][128];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = t >> 2, q = t & 3;
  const int64_t base = (int64_t)blockIdx.x * kTailRows;
  const int nrow = (int)((a.batch - base) < kTailRows ? (a.batch - base) : kTailRows);
  const int moff = a.mf_first ? 1 : 0;
  // per-row scalars of the head: requested before the staging barrier, used long after it
  const bool live = r < nrow;
  const int64_t gr = base + r;
  const float d = live ? a.dot[gr] : 0.f, yv = live ? a.labels[gr] : 0.f;
  const float wdot = a.w4[a.mf_first ? 0 : n3], bias4 = a.b4[0];

  // ---- stage: a2 tile (coalesced), W3, per-column constants ----
  // (rows past the batch are zero-filled: they flow through the arithmetic below with dz = 0)
  {
    // a wave per row (no index division, 4*n2-byte runs); all 16 rows of a wave are loaded before the first
    // LDS write - a rolled load -> write loop costs one memory round trip per row
    constexpr int RPW = kTailRows / kTailWaves;
    float v[RPW][2];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int rr = wave + kTailWaves * i;
      const float* src = a.a2 + (base + (rr < nrow ? rr : 0)) * a.lda2;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = lane + 64 * h;
        const float x = src[k < n2 ? k : 0];
        v[i][h] = (rr < nrow && k < n2) ? x : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int rr = wave + kTailWaves * i;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = lane + 64 * h;
        if (k < n2) A[rr * ld + k] = v[i][h];
      }
    }
  }
  for (int e = t; e < n2 * N3P; e += kTailThreads) { const int k = e / N3P, n = e - k * N3P; Ws[e] = n < n3 ? a.W3[k * n3 + n] : 0.f; }
  for (int k = t; k < n2; k += kTailThreads) {
    cst[k] = a.scale2[k]; cst[n2 + k] = a.shift2[k]; cst[2 * n2 + k] = a.mean2[k]; cst[3 * n2 + k] = a.rstd2[k];
  }
  for (int n = t; n < N3P; n += kTailThreads) { hw[n] = n < n3 ? a.b3[n] : 0.f; hw[N3P + n] = n < n3 ? a.w4[moff + n] : 0.f; }
  for (int k = t; k < 256; k += kTailThreads) (&colred[0][0])[k] = 0.0;
  __syncthreads();

  const float ik = a.inv_keep;
  // ---- forward of the row: T(a2) -> X, z3 (this lane's chunks, then summed over the 4 lanes of the row) ----
  float z[N3P];
#pragma unroll
  for (int n = 0; n < N3P; ++n) z[n] = q == 0 ? hw[n] : 0.f;
  uint32_t bits[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k0 = 8 * (q + 4 * i);
    bits[i] = 0u;
    if (k0 < n2) {
      bits[i] = live ? (a.keep ? (uint32_t)a.keep[gr * a.keep_ld + (k0 >> 3)] : 0xFFu) : 0u;
      const int kend = (n2 - k0) < 8 ? (n2 - k0) : 8;
      for (int j = 0; j < kend; ++j) {
        const int k = k0 + j;
        const float h = A[r * ld + k] * cst[k] + cst[n2 + k];
        const float x = ((bits[i] >> j) & 1u) ? h * ik : 0.f;
        X[r * ld + k] = x;
#pragma unroll
        for (int n4 = 0; n4 < N3P; n4 += 4) {
          const float4 w = *reinterpret_cast<const float4*>(Ws + k * N3P + n4);
          z[n4] += x * w.x; z[n4 + 1] += x * w.y; z[n4 + 2] += x * w.z; z[n4 + 3] += x * w.w;
        }
      }
    }
  }
#pragma unroll
  for (int n = 0; n < N3P; ++n) { z[n] += __shfl_xor(z[n], 1, 64); z[n] += __shfl_xor(z[n], 2, 64); }
  // ---- head, loss, dlogit (the 4 lanes of a row compute the same values; lane q == 0 owns the outputs) ----
  float lz = bias4 + d * wdot;
#pragma unroll
  for (int n = 0; n < N3P; ++n) {
    z[n] = n < n3 ? act_apply(z[n], a.act) : 0.f;          // z now holds a3
    lz += z[n] * hw[N3P + n];
  }
  const float p = sigmoidf_acc(lz);
  float dzl = 0.f;
  double s_loss = 0.0, s_se = 0.0, s_ae = 0.0, s_ok = 0.0, s_bce = 0.0, s_tp = 0.0, s_fp = 0.0, s_fn = 0.0;
  const bool own = live && q == 0;
  if (live) {
    float l;
    const float bce = fmaxf(lz, 0.f) - lz * yv + log1pf(expf(-fabsf(lz)));
    if (a.loss == BR_LOSS_BCE) {
      l = bce;
      dzl = (p - yv) * a.inv_batch;
    } else {
      l = (p - yv) * (p - yv);
      dzl = 2.f * (p - yv) * p * (1.f - p) * a.inv_batch;
    }
    if (own) {
      s_loss = (double)l; s_se = (double)((p - yv) * (p - yv)); s_ae = (double)fabsf(p - yv);
      const bool pp = p > 0.5f, yp = yv > 0.5f;
      s_ok = (pp == yp) ? 1.0 : 0.0;
      s_bce = (double)bce; s_tp = (pp && yp) ? 1.0 : 0.0; s_fp = (pp && !yp) ? 1.0 : 0.0; s_fn = (!pp && yp) ? 1.0 : 0.0;
      a.logit[gr] = lz;
      a.prob[gr] = p;
      a.ddot[gr] = dzl * wdot;
    }
  }
  if (q == 0) {
#pragma unroll
    for (int n = 0; n < N3P; ++n) A3s[r * (N3P + 1) + n] = z[n];
  }
  float dz[N3P];
#pragma unroll
  for (int n = 0; n < N3P; ++n) {
    dz[n] = (live && n < n3) ? dzl * hw[N3P + n] * act_grad_from_out(z[n], a.act) : 0.f;
    if (q == 0) DZ[r * (N3P + 1) + n] = dz[n];
  }
  // head parameter gradients: per-row terms (lane q == 0 only), reduced over the workgroup below
  const float dzo = own ? dzl : 0.f;
#pragma unroll
  for (int n = 0; n < N3P; ++n) {
    const float v = group_sum<64>(dzo * z[n]);
    if (lane == 0) redf[wave][n] = v;
  }
  {
    const float v0 = group_sum<64>(dzo * d), v1 = group_sum<64>(dzo);
    if (lane == 0) { redf[wave][N3P] = v0; redf[wave][N3P + 1] = v1; }
    s_loss = wave_sum_d(s_loss); s_se = wave_sum_d(s_se); s_ae = wave_sum_d(s_ae); s_ok = wave_sum_d(s_ok);
    s_bce = wave_sum_d(s_bce); s_tp = wave_sum_d(s_tp); s_fp = wave_sum_d(s_fp); s_fn = wave_sum_d(s_fn);
    if (lane == 0) {
      redd[wave][0] = s_loss; redd[wave][1] = s_se; redd[wave][2] = s_ae; redd[wave][3] = s_ok;
      redd[wave][4] = s_bce; redd[wave][5] = s_tp; redd[wave][6] = s_fp; redd[wave][7] = s_fn;
    }
  }
  __syncthreads();                                           // X, DZ, redf, redd complete

  // ---- slab of this workgroup: [dW3 (n2 x n3) | db3 (n3) | dW4 (n3 + 1, concat order) | db4] ----
  float* slab = a.slabs + (int64_t)blockIdx.x * ((int64_t)n2 * n3 + 2 * n3 + 2);
  for (int o = t; o < n2 * n3; o += kTailThreads) {
    const int k = o / n3, n = o - k * n3;
    float acc = 0.f;
    for (int rr = 0; rr < kTailRows; ++rr) acc += X[rr * ld + k] * DZ[rr * (N3P + 1) + n];   // dead rows hold dz = 0
    slab[o] = acc;
  }
  if (a.a3)      // rows of the tile are contiguous in a3 (row stride n3)
    for (int e = t; e < nrow * n3; e += kTailThreads) { const int rr = e / n3; a.a3[base * n3 + e] = A3s[rr * (N3P + 1) + (e - rr * n3)]; }
  auto wsum = [&](int i) { float v = 0.f; for (int w = 0; w < kTailWaves; ++w) v += redf[w][i]; return v; };
  if (t < n3) {
    float acc = 0.f;
    for (int rr = 0; rr < kTailRows; ++rr) acc += DZ[rr * (N3P + 1) + t];
    slab[n2 * n3 + t] = acc;
    slab[n2 * n3 + n3 + moff + t] = wsum(t);
  }
  if (t == 64) slab[n2 * n3 + n3 + (a.mf_first ? 0 : n3)] = wsum(N3P);
  if (t == 65) slab[n2 * n3 + 2 * n3 + 1] = wsum(N3P + 1);
  if (t >= 96 && t < 96 + BR_METRIC_SUMS && a.msums) {
    double v = 0.0;
    for (int w = 0; w < kTailWaves; ++w) v += redd[w][t - 96];
    atomicAdd(a.msums + (size_t)(blockIdx.x & (BR_SUM_SLOTS - 1)) * BR_METRIC_SUMS + (t - 96), v);
  }
  __syncthreads();                                           // every reader of X is done

  // ---- dx of the row: gh2 -> A (in place of the raw a2), gh2 * xhat2 -> X ----
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k0 = 8 * (q + 4 * i);
    if (k0 < n2) {
      const int kend = (n2 - k0) < 8 ? (n2 - k0) : 8;
      for (int j = 0; j < kend; ++j) {
        const int k = k0 + j;
        float acc = 0.f;
#pragma unroll
        for (int n4 = 0; n4 < N3P; n4 += 4) {
          const float4 w = *reinterpret_cast<const float4*>(Ws + k * N3P + n4);
          acc += dz[n4] * w.x + dz[n4 + 1] * w.y + dz[n4 + 2] * w.z + dz[n4 + 3] * w.w;
        }
        const float dh = ((bits[i] >> j) & 1u) ? acc * ik : 0.f;
        const float xhat = (A[r * ld + k] - cst[2 * n2 + k]) * cst[3 * n2 + k];
        A[r * ld + k] = dh;
        X[r * ld + k] = dh * xhat;
      }
    }
  }
  __syncthreads();
  // ---- BN2 backward column sums (4 row quarters per column -> LDS doubles -> one global atomic per column)
  //      and the coalesced store of gh2 ----
  if (a.bn_sums) {
    for (int idx = t; idx < 4 * n2; idx += kTailThreads) {
      const int part = idx / n2, k = idx - part * n2;
      float s1 = 0.f, s2 = 0.f;
      for (int rr = part * (kTailRows / 4); rr < (part + 1) * (kTailRows / 4); ++rr) { s1 += A[rr * ld + k]; s2 += X[rr * ld + k]; }
      atomicAdd(&colred[0][k], (double)s1);
      atomicAdd(&colred[1][k], (double)s2);
    }
  }
  for (int rr = wave; rr < nrow; rr += kTailWaves) {
    float* dst = a.gh2 + (base + rr) * a.ldgh2;
    for (int k = lane; k < n2; k += 64) dst[k] = A[rr * ld + k];
  }
  if (a.bn_sums) {
    __syncthreads();
    double* rep = a.bn_sums + (size_t)(blockIdx.x % kTailRep) * 2 * n2;
    for (int k = t; k < n2; k += kTailThreads) {
      atomicAdd(rep + k, colred[0][k]);
      atomicAdd(rep + n2 + k, colred[1][k]);
    }
  }
}

static size_t tail_lds_bytes(int n2, int n3p) {
  const int ld = n2 | 1;
  return sizeof(float) * ((size_t)2 * kTailRows * ld + (size_t)n2 * n3p + (size_t)2 * kTailRows * (n3p + 1) + 4 * (size_t)n2 + 2 * (size_t)n3p);
}

}  // namespace br

using namespace br;

extern "C" int brNeumfTailSlabs(int64_t batch) { return (int)ceil_div(batch > 0 ? batch : 1, kTailRows); }

extern "C" int64_t brNeumfTailSlabElems(int n2, int n3) { return (int64_t)n2 * n3 + 2 * n3 + 2; }

extern "C" int brNeumfTailFused(const float* a2, int64_t lda2, const float* W3, const float* b3, const float* w4, const float* b4,
                                const float* dot, const float* labels, const float* scale2, const float* shift2, const float* mean2,
                                const float* rstd2, const brBnFold* bn2, float drop_p, const uint32_t* keep,
                                int64_t batch, int n2, int n3, int act, int mf_first, int loss, float inv_batch, float* a3,
                                float* logit, float* prob, double* sums, float* ddot, float* gh2, int64_t ldgh2, double* bn_sums,
                                float* slabs, int n_slabs, brStream stream) {
  BR_CHECK_ARG(a2 && W3 && b3 && w4 && b4 && dot && labels && logit && prob && ddot && gh2 && slabs, "brNeumfTailFused: null pointer");
  BR_CHECK_ARG(bn2 ? (!scale2 && !shift2 && !mean2 && !rstd2 && bn2->stats && bn2->gamma && bn2->beta && bn2->scale && bn2->shift && bn2->mean && bn2->rstd &&
                      bn2->batch_total > 0 && (bn2->moving_mean == nullptr) == (bn2->moving_var == nullptr))
                   : (scale2 && shift2 && mean2 && rstd2),
               "brNeumfTailFused: BatchNorm 2 either as scale2/shift2/mean2/rstd2 or as bn2 (stats, gamma, beta and the four outputs)");
  BR_CHECK_ARG(batch >= 0 && n2 >= 1 && n2 <= 128 && n3 >= 1 && n3 <= 32 && lda2 >= n2 && ldgh2 >= n2, "brNeumfTailFused: bad sizes");
  BR_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "brNeumfTailFused: drop_p out of [0,1)");
  BR_CHECK_ARG((drop_p > 0.f) == (keep != nullptr), "brNeumfTailFused: keep bits (brDropoutKeepBits) are required exactly when drop_p > 0");
  BR_CHECK_ARG(loss == BR_LOSS_BCE || loss == BR_LOSS_MSE, "brNeumfTailFused: bad loss");
  if (batch == 0) return BR_OK;
  BR_CHECK_ARG(n_slabs == brNeumfTailSlabs(batch), "brNeumfTailFused: n_slabs must be brNeumfTailSlabs(batch)");
  hipStream_t s = (hipStream_t)stream;
  const auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (n2 <= 64 && n3 <= 16 && lda2 % 4 == 0 && ldgh2 % 4 == 0 && al16(a2) && al16(gh2)) {
    TailMArgs m{};
    m.a2 = a2; m.lda2 = lda2; m.W3 = W3; m.b3 = b3; m.w4 = w4; m.b4 = b4; m.dot = dot; m.labels = labels;
    m.scale2 = scale2; m.shift2 = shift2; m.mean2 = mean2; m.rstd2 = rstd2;
    if (bn2) {
      m.stats2 = bn2->stats; m.batch_total = bn2->batch_total; m.gamma2 = bn2->gamma; m.beta2 = bn2->beta; m.bn_eps = bn2->eps; m.bn_momentum = bn2->momentum;
      m.moving_mean = bn2->moving_mean; m.moving_var = bn2->moving_var; m.out_scale = bn2->scale; m.out_shift = bn2->shift; m.out_mean = bn2->mean; m.out_rstd = bn2->rstd;
    }
    m.keep = keep; m.kw = (n2 + 31) / 32; m.inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    m.batch = batch; m.n2 = n2; m.n3 = n3; m.act = act; m.mf_first = mf_first; m.loss = loss; m.inv_batch = inv_batch;
    m.a3 = a3; m.logit = logit; m.prob = prob; m.ddot = ddot; m.gh2 = gh2; m.ldgh2 = ldgh2; m.msums = sums; m.bn_sums = bn_sums; m.slabs = slabs;
    int grid = tail_mfma_grid(batch);
    if (grid > n_slabs) grid = n_slabs;
    m.n_slabs = n_slabs;
    launch_tail_mfma(m, grid, s);
    BR_CHECK_LAUNCH("brNeumfTailFused(mfma)");
    return BR_OK;
  }
  if (bn2) {   // VALU form: finalize first
    int rc = brBnFinalize(bn2->stats, bn2->batch_total, bn2->gamma, bn2->beta, bn2->eps, bn2->momentum, bn2->moving_mean, bn2->moving_var, bn2->scale,
                          bn2->shift, bn2->mean, bn2->rstd, n2, stream);
    if (rc != BR_OK) return rc;
    scale2 = bn2->scale; shift2 = bn2->shift; mean2 = bn2->mean; rstd2 = bn2->rstd;
  }
  TailArgs a;
  a.a2 = a2; a.lda2 = lda2; a.W3 = W3; a.b3 = b3; a.w4 = w4; a.b4 = b4; a.dot = dot; a.labels = labels;
  a.scale2 = scale2; a.shift2 = shift2; a.mean2 = mean2; a.rstd2 = rstd2;
  a.keep = reinterpret_cast<const uint8_t*>(keep); a.keep_ld = 4 * (int64_t)((n2 + 31) / 32); a.inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  a.batch = batch; a.n2 = n2; a.n3 = n3; a.act = act; a.mf_first = mf_first; a.loss = loss; a.inv_batch = inv_batch;
  a.a3 = a3; a.logit = logit; a.prob = prob; a.ddot = ddot; a.gh2 = gh2; a.ldgh2 = ldgh2; a.msums = sums; a.bn_sums = bn_sums;
  a.slabs = slabs;
  const int n3p = (n3 + 3) & ~3;
  const size_t lds = tail_lds_bytes(n2, n3p);
  const unsigned grid = (unsigned)n_slabs;
#define BR_TAIL(NP)                                                                                                       \
  case NP: {                                                                                                              \
    static bool attr = false;                                                                                             \
    if (!attr) { (void)hipFuncSetAttribute((const void*)neumf_tail_kernel<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr = true; } \
    neumf_tail_kernel<NP><<<grid, kTailThreads, lds, s>>>(a);                                                                \
  } break;
  switch (n3p) {
    BR_TAIL(4) BR_TAIL(8) BR_TAIL(12) BR_TAIL(16) BR_TAIL(20) BR_TAIL(24) BR_TAIL(28) BR_TAIL(32)
    default: set_error("brNeumfTailFused: unsupported n3"); return BR_ERR_UNSUPPORTED;
  }
#undef BR_TAIL
  BR_CHECK_LAUNCH("brNeumfTailFused");
  return BR_OK;
}
