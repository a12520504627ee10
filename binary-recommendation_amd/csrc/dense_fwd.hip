// T1-T4 forward of one tower layer on fp32 MFMA (v_mfma_f32_16x16x4_f32: exact f32 fma chains, the only MFMA
// form inside the 1e-5 logit/loss tolerance; 157 TFLOP/s peak on MI355X) + the dropout keep-bit planes.
//
// dense_fwd_kernel: ONE 1024-thread workgroup per CU = 16 waves = 4 per SIMD, <= 128 VGPRs.
//   * The layer's W is staged once per workgroup into LDS, pre-swizzled [j][g][n][s] so that a lane's four k-steps of
//     one n-tile are one conflict-free ds_read_b128.
//   * Every wave is then independent: it owns 16-row tiles, loads the A fragments straight from global memory in the
//     MFMA lane layout (lane (c16,g) <- 16 B of row c16 at k = 16j+4g; K is contracted in that lane-permuted order),
//     applies BatchNorm-affine and the dropout keep bits in registers (v_bfe_i32 + v_and per element; the 1/(1-p)
//     scale is folded into the epilogue's fma with the bias), and runs the n-tiles in passes of 4 independent
//     accumulator chains.  Each finished 16x16 output tile is transposed through a 1.3-KB per-wave LDS patch so that it
//     leaves as ONE 16-B store per lane (7 stores per 16 x 100 tile instead of 28 dword stores: the store tail is
//     issue-bound, MI355X guide T21).
//   * STATIC PRIORITIES: the 4 waves that share a SIMD (w, w+4, w+8, w+12) run at s_setprio 3, 2, 1, 0.  At equal priority
//     they take turns on the matrix pipe, reach the end of a pass together and then execute the (cold, VALU-paced)
//     epilogue code together with the pipe idle: in-kernel stamps showed 7.7 K MFMA cycles inside 50 K wave cycles.  With
//     distinct priorities the leader runs its MFMA passes at full rate and the next wave takes the pipe over whenever the
//     leader is in loads / VALU / stores; the followers also find the leader's code in the instruction cache.
//   * Column statistics (BatchNorm sums of y, y^2): per-wave partials in LDS, summed over the waves in double, then one
//     double atomic per column and workgroup into BR_STAT_REPLICAS replicas.
// keep_bits_kernel: the Philox4x32-10 masks of csrc/philox.h materialised ONCE per (step, site) as bit planes
//   (uint32 [rows][ceil(K/32)]: 2.3 MB for the three sites of the NeuMF tower at batch 65 536) instead of being
//   regenerated inside the forward, the dx and the dW kernel of each layer (~120 VALU per 8 elements, three times).
#include "common.h"
#include <stdlib.h>
#include "philox.h"
#include "dense.h"
#include "stamps.h"

namespace br {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kRep = BR_STAT_REPLICAS;
// tools/diag/build_ablate.sh (-DBR_ABLATE_MFMA): a TIMING experiment that issues 3 of every 8 MFMAs (wrong results) - the matrix-pipe time a
// 6-product bf16 emulation of the fp32 product would have; never defined in the library build
#ifdef BR_ABLATE_MFMA
#define BR_ABLATE_KEEP(c) (c)
#else
#define BR_ABLATE_KEEP(c) true
#endif

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------ fp32 products on the bf16 matrix pipe
// (BR_MLP_MATH, dense.h split3 / mfma_bf16: x = h + m + l in three bf16 pieces, x w = hh + hm + mh + hl + lh + mm - six
//  v_mfma_f32_16x16x32_bf16 per 32-deep k-block and n-tile = 96 matrix-pipe cycles where eight v_mfma_f32_16x16x4_f32 take 256.)

// ------------------------------------------------------------------------------ keep-bit planes
// (KeepSite / KeepArgs / keep_bits_block: philox.h - the optimizer launch can carry the planes of the next step in its own grid)
__global__ __launch_bounds__(256) void keep_bits_kernel(KeepArgs a) {
  dropout_resolve(a.drop);
  keep_bits_block(a, (int)blockIdx.y, (int64_t)blockIdx.x);
}

// ------------------------------------------------------------------------------------ forward
struct FwdArgs {
  const float* x; int64_t ldx;
  const float* W; const float* bias;
  float* y; int64_t ldy;
  int64_t batch;
  int K, N, act;
  const float* scale; const float* shift;   // (K) BatchNorm affine of the producer, or null
  brBnFold bn;                              // bn.stats != null: the affine is finalized here from the producer's column sums
  const uint32_t* keep; int kw;             // keep-bit plane [batch][kw] of this layer's input dropout, or null
  float inv_keep;                           // 1/(1-p) (1 without dropout)
  double* stats;                            // [kRep][2N] or null
  const float* yin;                         // split-K: the other K-half's partial sums (same element of y), or null
};

// x & (bit `pos` of w ? ~0 : 0): v_bfe_i32 + v_and_b32
__device__ __forceinline__ float keep_if(float x, uint32_t w, uint32_t pos) {
  const int m = __builtin_amdgcn_sbfe((int)w, pos, 1u);
  return __int_as_float(__float_as_int(x) & m);
}

// Activation, branch-free and without copies of the epilogue (the instruction footprint matters: every wave runs this code
// once, cold): both candidates are formed and blended with a bit mask (v_bfi_b32) -
//   sigmoid: 1/(1+2^(-z log2 e)) = v_mul, v_exp_f32, v_add, v_rcp_f32 (z -> -inf gives rcp(inf) = 0, z -> +inf gives 1;
//            |rel err| < 4e-7: hardware exp2 / rcp are ~1 ulp each, inside the 1e-5 budget, tests/test_gpu_neumf.py);
//   relu / linear: max(z, floor) with floor = 0 / -inf.
// (`cond ? act(z) : 0` in C makes hipcc build an exec-masked branch around every element's exp/rcp.)
__device__ __forceinline__ float act_fwd(float z, int sigmask, float floor_) {
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.44269504088896340736f));
  const float m = fmaxf(z, floor_);
  return __int_as_float((__float_as_int(s) & sigmask) | (__float_as_int(m) & ~sigmask));
}

// A fragments + keep words of one tile.  ONE base pointer per lane (row clamped into the batch) and compile-time byte
// offsets; only the last 16-column block can reach past K (KJ = ceil(K/16)) and is loaded from a clamped column.
// Nothing is zeroed here: columns >= K meet zero rows of the W image (and a zero BN affine), rows >= batch are
// never stored or counted by the epilogue.  VEC: 16-B loads and stores (launch_fwd).
template <int KJ, bool VEC>
__device__ __forceinline__ void fwd_load_tile(float4 (&av)[KJ], uint32_t (&kb)[(KJ + 1) / 2], const FwdArgs& a, int64_t tile, int c16, int g) {
  int64_t arow = (tile << 4) + c16;
  arow = arow < a.batch ? arow : a.batch - 1;
  const float* p = a.x + arow * a.ldx + 4 * g;
  const int klast = 16 * (KJ - 1) + 4 * g;
  if (VEC) {
#pragma unroll
    for (int j = 0; j < KJ - 1; ++j) av[j] = *reinterpret_cast<const float4*>(p + 16 * j);
    av[KJ - 1] = *reinterpret_cast<const float4*>(klast < a.K ? p + 16 * (KJ - 1) : p - 4 * g);
    // (a last 16-B group that straddles K is cleaned in T(): touching the loaded values here made hipcc wait for every A load
    //  before it issued the W image's loads)
  } else {
#pragma unroll
    for (int j = 0; j < KJ - 1; ++j) av[j] = make_float4(p[16 * j], p[16 * j + 1], p[16 * j + 2], p[16 * j + 3]);
    const float* q = p - 4 * g;     // row start
    const int K1 = a.K - 1;
    av[KJ - 1] = make_float4(q[klast < K1 ? klast : K1], q[klast + 1 < K1 ? klast + 1 : K1], q[klast + 2 < K1 ? klast + 2 : K1], q[klast + 3 < K1 ? klast + 3 : K1]);
  }
  if (a.keep) {
    const uint32_t* kr = a.keep + arow * a.kw;      // kw == (KJ + 1) / 2 words per row
#pragma unroll
    for (int w = 0; w < (KJ + 1) / 2; ++w) kb[w] = kr[w];
  }
}

constexpr int kPatchLd = 20;      // floats per row of the per-wave 16x16 transposition patch: 16-B aligned rows, conflict-free dword writes

// epilogue of one n-tile.  In: lane (c16, g) holds rows 4g..4g+3 of column nt*16+c16 (MFMA C layout).  bias + 1/(1-p) +
// activation + column sums in that layout (elements outside the batch / past N are ANDed to 0), then through the wave's LDS
// patch into row layout: lane (c16, g) <- columns nt*16+4g..+3 of row c16, one 16-B store.  vmask[r]: ~0 where row 4g+r is
// inside the batch.
template <bool VEC>
__device__ __forceinline__ void fwd_epilogue(const f32x4& acc, const FwdArgs& a, float* patch, float* yrow, int64_t rbase, int nt, int c16, int g,
                                             const int (&vmask)[4], bool row_ok, float bv, int sigmask, float floor_, float& ssum, float& ssq) {
  const int n = nt * 16 + c16;
  const bool last = nt * 16 + 16 > a.N;                    // only the last n-tile can be ragged (wave-uniform)
  const int cmask = (!last || n < a.N) ? -1 : 0;
  float yv[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.yin) {                                             // wave-uniform: split-K second half (K > 128)
#pragma unroll
    for (int r = 0; r < 4; ++r) yv[r] = a.yin[(rbase + (vmask[r] ? 4 * g + r : 0)) * a.ldy + (cmask ? n : 0)];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float z = fmaf(acc[r], a.inv_keep, bv) + yv[r];
    const float v = __int_as_float(__float_as_int(act_fwd(z, sigmask, floor_)) & vmask[r] & cmask);
    ssum += v;
    ssq = fmaf(v, v, ssq);
    patch[(4 * g + r) * kPatchLd + c16] = v;
  }
  __builtin_amdgcn_wave_barrier();
  const float4 o = *reinterpret_cast<const float4*>(patch + c16 * kPatchLd + 4 * g);
  __builtin_amdgcn_wave_barrier();
  float* dst = yrow + nt * 16;                              // &y[rbase + c16][nt*16 + 4g]
  const int n0 = nt * 16 + 4 * g;
  if (VEC) {                                                // rows are padded to 4 floats: a group that starts inside N is stored whole
    if (row_ok && (!last || n0 < a.N)) *reinterpret_cast<float4*>(dst) = o;
  } else if (row_ok) {
    if (n0 + 0 < a.N) dst[0] = o.x;
    if (n0 + 1 < a.N) dst[1] = o.y;
    if (n0 + 2 < a.N) dst[2] = o.z;
    if (n0 + 3 < a.N) dst[3] = o.w;
  }
}

// the same for up to 4 n-tiles at once (WPS == 2: the VGPR budget allows it): every tile has its own patch, so the 16 activations,
// the patch writes, the patch reads and the stores of a pass each go out back to back instead of tile by tile
template <bool VEC, int WN, int NT>
__device__ __forceinline__ void fwd_epilogue_multi(const f32x4 (&acc)[4], const FwdArgs& a, float* patch4, float* yrow, int64_t rbase, int nt0, int c16, int g,
                                                   const int (&vmask)[4], bool row_ok, const float* bs, int sigmask, float floor_, float (&ssum)[NT], float (&ssq)[NT]) {
#pragma unroll
  for (int w = 0; w < WN; ++w) {
    const int nt = nt0 + w, n = nt * 16 + c16;
    const bool last = nt * 16 + 16 > a.N;
    const int cmask = (!last || n < a.N) ? -1 : 0;
    const float bv = bs[n];
    float* patch = patch4 + w * 16 * kPatchLd;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float z = fmaf(acc[w][r], a.inv_keep, bv);
      if (a.yin) z += a.yin[(rbase + (vmask[r] ? 4 * g + r : 0)) * a.ldy + (cmask ? n : 0)];      // wave-uniform: split-K second half
      const float v = __int_as_float(__float_as_int(act_fwd(z, sigmask, floor_)) & vmask[r] & cmask);
      ssum[nt] += v;
      ssq[nt] = fmaf(v, v, ssq[nt]);
      patch[(4 * g + r) * kPatchLd + c16] = v;
    }
  }
  __builtin_amdgcn_wave_barrier();
  // (named values, not an array: hipcc kept a float4 o[WN] behind the lane-conditional stores in scratch)
  const float* pr = patch4 + c16 * kPatchLd + 4 * g;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 o0 = *reinterpret_cast<const float4*>(pr);
  const float4 o1 = WN > 1 ? *reinterpret_cast<const float4*>(pr + 1 * 16 * kPatchLd) : z4;
  const float4 o2 = WN > 2 ? *reinterpret_cast<const float4*>(pr + 2 * 16 * kPatchLd) : z4;
  const float4 o3 = WN > 3 ? *reinterpret_cast<const float4*>(pr + 3 * 16 * kPatchLd) : z4;
  auto put = [&](const float4& o, int w) {
    const int nt = nt0 + w, n0 = nt * 16 + 4 * g;
    float* dst = yrow + nt * 16;
    if (VEC) {                                              // rows are padded to 4 floats: a group that starts inside N is stored whole
      if (n0 < a.N) *reinterpret_cast<float4*>(dst) = o;
    } else {
      if (n0 + 0 < a.N) dst[0] = o.x;
      if (n0 + 1 < a.N) dst[1] = o.y;
      if (n0 + 2 < a.N) dst[2] = o.z;
      if (n0 + 3 < a.N) dst[3] = o.w;
    }
  };
  if (row_ok) {
    put(o0, 0);
    if (WN > 1) put(o1, 1);
    if (WN > 2) put(o2, 2);
    if (WN > 3) put(o3, 3);
  }
}

// WPS = waves per SIMD: 4 (one 16-row tile per wave at batch 65 536: every wave of the chip loads, multiplies and stores at the
// same time, the matrix pipe idles through the two memory phases) or 2 (half the waves, two tiles each, the second tile's loads
// issued before the first tile's MFMA passes and the first tile's stores behind them: the memory phases of one tile overlap the
// MFMA phase of the other)
template <int NT, int KJ, bool VEC, int WPS, bool EMU>
__global__ __launch_bounds__(256 * WPS, WPS) void dense_fwd_kernel(const FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kFwdThreads = 256 * WPS, kFwdWaves = 4 * WPS;
  constexpr int Np = NT * 16, Kp = KJ * 16, KWJ = (KJ + 1) / 2;
  constexpr int KB = (KJ + 1) / 2;                      // 32-deep k-blocks of the bf16 path
  // W image.  fp32 path: [KJ][4][Np][4] floats.  bf16 path: [KB][3 pieces][4 g][Np][8 bf16] = KB * 3 * 4 * Np * 16 bytes (1.5 x the fp32 image)
  constexpr int kWsFloats = EMU ? KB * 3 * 4 * Np * 4 : Kp * Np;
  float* Ws = smem;
  float* ssb = Ws + kWsFloats;                          // [scale Kp | shift Kp]
  float* bs = ssb + 2 * Kp;                             // [bias Np]
  float* patches = bs + Np;                             // [waves][16][kPatchLd]
  constexpr int kPatchesPerWave = WPS == 2 ? 4 : 1;
  float* redw = patches + kFwdWaves * kPatchesPerWave * 16 * kPatchLd;    // [waves][2][Np] per-wave column sums
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c16 = lane & 15, g = lane >> 4;
  const int K = a.K, N = a.N;
  const int64_t batch = a.batch;
  const int64_t n_tiles = (batch + 15) >> 4;
  const int64_t tstride = (int64_t)gridDim.x * kFwdWaves;
  {   // static priority per SIMD slot (waves w, w+4, w+8, w+12 share a SIMD): see the file header
    const int slot = __builtin_amdgcn_readfirstlane(wave >> 2);
    if (slot == 0) __builtin_amdgcn_s_setprio(3);
    else if (slot == 1) __builtin_amdgcn_s_setprio(2);
    else if (slot == 2) __builtin_amdgcn_s_setprio(1);     // (WPS == 2: slots 0 and 1 only)
  }

  BR_STAMP_DECL;
  BR_STAMP_RT(10);
  BR_STAMP(0);
  // EMU: the W loads are requested BEFORE the first tile's loads and consumed behind them: the in-order vmcnt lets the wave wait for W alone
  // (L2 hits, ~2 k cycles) and build the piece image while the tile streams in at the CU's share of HBM; behind the tile they arrived last.
  const bool wquad = EMU && (N & 3) == 0;               // rows of W 16-B aligned: 16-B loads
  constexpr int NQ = Np / 4, TOTQ = KB * 4 * NQ, TRQ = EMU ? (TOTQ + kFwdThreads - 1) / kFwdThreads : 1;
  constexpr int TOTS = KB * 4 * Np, TRS = EMU ? (TOTS + kFwdThreads - 1) / kFwdThreads : 1;
  float4 wq[TRQ][8];
  float wv1[TRS][8];
  if constexpr (EMU) {
    if (wquad) {
#pragma unroll
      for (int i = 0; i < TRQ; ++i) {
        const int idx = threadIdx.x + i * kFwdThreads;
        const int idc = idx < TOTQ ? idx : 0;
        const int Jg = idc / NQ, n = (idc - Jg * NQ) * 4;
        const int J = Jg >> 2, gg = Jg & 3;
        const int nc = n < N ? n : 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int k = 32 * J + 16 * (q >> 2) + 4 * gg + (q & 3);
          wq[i][q] = *reinterpret_cast<const float4*>(a.W + (int64_t)(k < K ? k : 0) * N + nc);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < TRS; ++i) {
        const int idx = threadIdx.x + i * kFwdThreads;
        const int idc = idx < TOTS ? idx : 0;
        const int Jg = idc / Np, n = idc - Jg * Np;
        const int J = Jg >> 2, gg = Jg & 3;
        const int nc = n < N ? n : N - 1;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int k = 32 * J + 16 * (q >> 2) + 4 * gg + (q & 3);
          wv1[i][q] = a.W[(k < K ? k : K - 1) * N + nc];
        }
      }
    }
  }
  float4 av[KJ];
  uint32_t kb[KWJ];
  int64_t tile = (int64_t)blockIdx.x * kFwdWaves + wave;
  fwd_load_tile<KJ, VEC>(av, kb, a, tile, c16, g);      // the first tile's loads fly while W is staged (row clamped)
  BR_STAMP(1);

  if (wquad) {
    // bf16 image, rows of W 16-B aligned: thread -> (J, g, four consecutive n): eight 16-B loads (the rows k = 32J + 16(q >> 2) + 4g + (q & 3) of
    // its fragment) instead of 32 dword loads - the same bytes through a quarter of the address-pipe slots - then four fragments' pieces,
    // 64 contiguous bytes per piece
#pragma unroll
    for (int i = 0; i < TRQ; ++i) {
      const int idx = threadIdx.x + i * kFwdThreads;
      const int Jg = idx / NQ, n = (idx - Jg * NQ) * 4;
      const int J = Jg >> 2, gg = Jg & 3;
      const int nin = n < N ? -1 : 0;
      uint32_t* wsu = reinterpret_cast<uint32_t*>(Ws);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        uint32_t ph[4], pm[4], pl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k0 = 32 * J + 16 * (q >> 1) + 4 * gg + 2 * (q & 1);
          const float4 v0 = wq[i][2 * q], v1 = wq[i][2 * q + 1];
          const float f0 = e == 0 ? v0.x : (e == 1 ? v0.y : (e == 2 ? v0.z : v0.w)), f1 = e == 0 ? v1.x : (e == 1 ? v1.y : (e == 2 ? v1.z : v1.w));
          split3(__int_as_float(__float_as_int(f0) & nin & (k0 < K ? -1 : 0)), __int_as_float(__float_as_int(f1) & nin & (k0 + 1 < K ? -1 : 0)), ph[q], pm[q], pl[q]);
        }
        if (idx < TOTQ) {
          *reinterpret_cast<uint4*>(wsu + ((((J * 3 + 0) * 4 + gg) * Np + n + e) << 2)) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
          *reinterpret_cast<uint4*>(wsu + ((((J * 3 + 1) * 4 + gg) * Np + n + e) << 2)) = make_uint4(pm[0], pm[1], pm[2], pm[3]);
          *reinterpret_cast<uint4*>(wsu + ((((J * 3 + 2) * 4 + gg) * Np + n + e) << 2)) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (EMU) {
    // bf16 image: thread -> (J, g, n): the 8 k-values a lane's fragment of k-block J holds (k = 32J + 16(i >> 2) + 4g + (i & 3): the
    // columns of the two float4 A loads of 16-column blocks 2J and 2J+1), split into three bf16 pieces, one 16-B LDS write per piece
    constexpr int TOT = TOTS, TR = TRS;
    auto& wv = wv1;
#pragma unroll
    for (int i = 0; i < TR; ++i) {
      const int idx = threadIdx.x + i * kFwdThreads;
      const int Jg = idx / Np, n = idx - Jg * Np;
      const int J = Jg >> 2, gg = Jg & 3;
      uint32_t ph[4], pm[4], pl[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k0 = 32 * J + 16 * (q >> 1) + 4 * gg + 2 * (q & 1);
        const float w0 = (n < N && k0 < K) ? wv[i][2 * q] : 0.f, w1 = (n < N && k0 + 1 < K) ? wv[i][2 * q + 1] : 0.f;
        split3(w0, w1, ph[q], pm[q], pl[q]);
      }
      if (idx < TOT) {
        uint32_t* wsu = reinterpret_cast<uint32_t*>(Ws);
        *reinterpret_cast<uint4*>(wsu + ((((J * 3 + 0) * 4 + gg) * Np + n) << 2)) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        *reinterpret_cast<uint4*>(wsu + ((((J * 3 + 1) * 4 + gg) * Np + n) << 2)) = make_uint4(pm[0], pm[1], pm[2], pm[3]);
        *reinterpret_cast<uint4*>(wsu + ((((J * 3 + 2) * 4 + gg) * Np + n) << 2)) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else
  // W image: thread -> (j, g, n): 4 coalesced loads (k = 16j+4g+s) -> one conflict-free ds_write_b128.  All loads of the
  // image are issued before the first LDS write (compile-time trip count, clamped addresses).
  {
    constexpr int TOT = KJ * 4 * Np, TR = (TOT + kFwdThreads - 1) / kFwdThreads;
    float wv[TR][4];
#pragma unroll
    for (int i = 0; i < TR; ++i) {
      const int idx = threadIdx.x + i * kFwdThreads;
      const int idc = idx < TOT ? idx : 0;
      const int jg = idc / Np, n = idc - jg * Np;
      const int k0 = 4 * jg;
      const int nc = n < N ? n : N - 1;
#pragma unroll
      for (int q = 0; q < 4; ++q) wv[i][q] = a.W[(k0 + q < K ? k0 + q : K - 1) * N + nc];
    }
#pragma unroll
    for (int i = 0; i < TR; ++i) {
      const int idx = threadIdx.x + i * kFwdThreads;
      const int jg = idx / Np, n = idx - jg * Np;
      const int k0 = 4 * jg;
      const bool nin = n < N;
      if (idx < TOT)
        *reinterpret_cast<float4*>(Ws + idx * 4) = make_float4((nin && k0 + 0 < K) ? wv[i][0] : 0.f, (nin && k0 + 1 < K) ? wv[i][1] : 0.f,
                                                                (nin && k0 + 2 < K) ? wv[i][2] : 0.f, (nin && k0 + 3 < K) ? wv[i][3] : 0.f);
    }
  }
  for (int k = threadIdx.x; k < Kp; k += kFwdThreads) {     // identity where there is no BatchNorm, zero past K
    float sc = k < K ? 1.f : 0.f, sh = 0.f;
    if (k < K && a.bn.stats) {      // BatchNorm finalize (as bn_finalize_kernel): biased batch variance [TF-sem]; workgroup 0 publishes
      double s1 = 0.0, s2 = 0.0;
      for (int r = 0; r < kRep; ++r) { s1 += a.bn.stats[(size_t)r * 2 * K + k]; s2 += a.bn.stats[(size_t)r * 2 * K + K + k]; }
      const double mu = s1 / a.bn.batch_total;
      double var = s2 / a.bn.batch_total - mu * mu;
      if (var < 0.0) var = 0.0;
      const float muf = (float)mu, varf = (float)var;
      const float rs = 1.0f / sqrtf(varf + a.bn.eps);
      sc = a.bn.gamma[k] * rs;
      sh = a.bn.beta[k] - muf * sc;
      if (blockIdx.x == 0) {
        a.bn.scale[k] = sc; a.bn.shift[k] = sh; a.bn.mean[k] = muf; a.bn.rstd[k] = rs;
        if (a.bn.moving_mean) {
          a.bn.moving_mean[k] = a.bn.moving_mean[k] * a.bn.momentum + muf * (1.0f - a.bn.momentum);
          a.bn.moving_var[k] = a.bn.moving_var[k] * a.bn.momentum + varf * (1.0f - a.bn.momentum);
        }
      }
    } else if (k < K && a.scale) { sc = a.scale[k]; sh = a.shift[k]; }
    ssb[k] = sc;
    ssb[Kp + k] = sh;
  }
  for (int n = threadIdx.x; n < Np; n += kFwdThreads) bs[n] = (a.bias && n < N) ? a.bias[n] : 0.f;
  BR_STAMP(2);
  __syncthreads();
  BR_STAMP(3);

  float* patch = patches + wave * kPatchesPerWave * 16 * kPatchLd;
  float ssum[NT], ssq[NT];             // this lane's column sums of y, y^2 (column nt*16+c16, its 4 rows of every tile)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { ssum[nt] = 0.f; ssq[nt] = 0.f; }
  const int sigmask = a.act == BR_ACT_SIGMOID ? -1 : 0;
  const float floor_ = a.act == BR_ACT_RELU ? 0.f : -__builtin_inff();

  while (tile < n_tiles) {
    const int64_t rbase = tile << 4;
    // ---- T(): BN affine + dropout keep bits, in registers (fenced per 16-column block: left alone the compiler hoists every
    //      block's scale / shift reads and spills) ----
    const bool affine = a.scale || a.bn.stats || 16 * KJ != K;      // the affine also zeroes the columns >= K of the last block
    if (VEC && (K & 3)) {      // wave-uniform: the last 16-B group may straddle K (padded rows): what it read past K must not be NaN
      const int klast = 16 * (KJ - 1) + 4 * g;
      if (klast + 1 >= K) av[KJ - 1].y = 0.f;
      if (klast + 2 >= K) av[KJ - 1].z = 0.f;
      if (klast + 3 >= K) av[KJ - 1].w = 0.f;
    }
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      if (affine) {
        const int k = 16 * j + 4 * g;
        const float4 sc = *reinterpret_cast<const float4*>(ssb + k), sh = *reinterpret_cast<const float4*>(ssb + Kp + k);
        av[j].x = av[j].x * sc.x + sh.x; av[j].y = av[j].y * sc.y + sh.y; av[j].z = av[j].z * sc.z + sh.z; av[j].w = av[j].w * sc.w + sh.w;
      }
      if (a.keep) {
        const uint32_t w = kb[j >> 1], p0 = 16u * (j & 1) + 4u * g;
        av[j].x = keep_if(av[j].x, w, p0); av[j].y = keep_if(av[j].y, w, p0 + 1);
        av[j].z = keep_if(av[j].z, w, p0 + 2); av[j].w = keep_if(av[j].w, w, p0 + 3);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    BR_STAMP(4);       // first tile: operands transformed (includes the wait for the A loads)
    const int64_t left = batch - rbase;
    const int rows_left16 = left > 16 ? 16 : (int)left;               // rows of this tile inside the batch
    int vmask[4];                                                     // ~0 where C-layout row 4g+r is inside the batch
#pragma unroll
    for (int r = 0; r < 4; ++r) vmask[r] = (4 * g + r < rows_left16) ? -1 : 0;
    const bool row_ok = c16 < rows_left16;
    float* yrow = a.y + (rbase + (row_ok ? c16 : 0)) * a.ldy + 4 * g;      // row layout: &y[rbase + c16][4g]
    // WPS == 2: the next tile's operands are requested now and fly during this tile's MFMA passes and epilogue
    float4 avn[WPS == 2 ? KJ : 1];
    uint32_t kbn[WPS == 2 ? KWJ : 1];
    const bool has_next = tile + tstride < n_tiles;                        // wave-uniform
    if (WPS == 2 && has_next) fwd_load_tile<KJ, VEC>(reinterpret_cast<float4 (&)[KJ]>(avn), reinterpret_cast<uint32_t (&)[KWJ]>(kbn), a, tile + tstride, c16, g);
    // bf16 path: the transformed operands split once per tile into three bf16 pieces (h + m + l = x to 2^-24), packed as the A
    // fragments of the 32-deep k-blocks: elements 0-3 from 16-column block 2J, 4-7 from block 2J+1 (zeros past KJ)
    // WPS == 2 (256 VGPRs): all k-blocks split here, once per tile.  WPS == 4 (128 VGPRs): 48 registers of pieces do not fit beside the
    // accumulators and fragments - each k-block is split where it is used, once per pass (NT > 4: twice per tile).
    constexpr bool PRESPLIT = EMU && WPS == 2;
    uint32_t ah[PRESPLIT ? KB : 1][4], am[PRESPLIT ? KB : 1][4], al[PRESPLIT ? KB : 1][4];
    auto split_block = [&](int J, uint32_t (&h)[4], uint32_t (&m)[4], uint32_t (&l)[4]) {
      const int j1 = 2 * J + 1 < KJ ? 2 * J + 1 : 0;
      split3(av[2 * J].x, av[2 * J].y, h[0], m[0], l[0]);
      split3(av[2 * J].z, av[2 * J].w, h[1], m[1], l[1]);
      if (2 * J + 1 < KJ) {
        split3(av[j1].x, av[j1].y, h[2], m[2], l[2]);
        split3(av[j1].z, av[j1].w, h[3], m[3], l[3]);
      } else {
        h[2] = h[3] = m[2] = m[3] = l[2] = l[3] = 0u;
      }
    };
    if constexpr (PRESPLIT) {
#pragma unroll
      for (int J = 0; J < KB; ++J) {
        split_block(J, ah[J], am[J], al[J]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- MFMA in passes of <= 4 n-tiles: 4 independent accumulator chains, each revisited every 4th MFMA ----
#pragma unroll
    for (int nt0 = 0; nt0 < NT; nt0 += 4) {
      constexpr int WMAX = 4;
      const int Wn = (NT - nt0) < WMAX ? (NT - nt0) : WMAX;     // compile-time after unrolling
      f32x4 acc[WMAX];
#pragma unroll
      for (int w = 0; w < WMAX; ++w) acc[w] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (EMU) {
        // per k-block and n-tile: three 16-B B fragments (h, m, l) and six products, the small ones first: l h, h l, m m, m h, h m, h h.
        // The fragments of step (J, w) + 1 are requested before the six MFMAs of step (J, w) (two buffers of 12 VGPRs; the fence
        // after every step keeps hipcc from hoisting a whole k-block's 48 fragment registers, which spilled the A pieces).
        // (read as float4, the type the epilogue's patch stores use: a uint4 view lets type-based alias analysis hoist every fragment
        //  read of the image out of the tile loop - 336 VGPRs of "loop invariants")
        float4 bq[2][3];
        // one per-lane base + compile-time offsets (which fold into the ds_read immediates): left as (..g..) * Np + n, hipcc keeps one
        // address VGPR per fragment
        const float* wl_ = Ws + ((g * Np + c16) << 2);
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[0][p] = *reinterpret_cast<const float4*>(wl_ + (((0 * 3 + p) * 4 * Np + nt0 * 16) << 2));
#pragma unroll
        for (int J = 0; J < KB; ++J) {
          uint32_t th[4], tm[4], tl[4];
          if constexpr (!PRESPLIT) { split_block(J, th, tm, tl); __builtin_amdgcn_sched_barrier(0); }
          const bf16x8 xh = frag8(PRESPLIT ? ah[J] : th), xm = frag8(PRESPLIT ? am[J] : tm), xl = frag8(PRESPLIT ? al[J] : tl);
#pragma unroll
          for (int w = 0; w < WMAX; ++w) {
            if (w < Wn) {
              const int st = J * Wn + w, cur = st & 1;
              const int Jn = (w + 1 < Wn) ? J : J + 1, wn = (w + 1 < Wn) ? w + 1 : 0;
              if (Jn < KB) {
#pragma unroll
                for (int p = 0; p < 3; ++p) bq[cur ^ 1][p] = *reinterpret_cast<const float4*>(wl_ + (((Jn * 3 + p) * 4 * Np + (nt0 + wn) * 16) << 2));
              }
              const bf16x8 wh = frag8(bq[cur][0]), wm = frag8(bq[cur][1]), wl = frag8(bq[cur][2]);
              acc[w] = mfma_bf16(xl, wh, acc[w]);
              acc[w] = mfma_bf16(xh, wl, acc[w]);
              acc[w] = mfma_bf16(xm, wm, acc[w]);
              acc[w] = mfma_bf16(xm, wh, acc[w]);
              acc[w] = mfma_bf16(xh, wm, acc[w]);
              acc[w] = mfma_bf16(xh, wh, acc[w]);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      } else
      // One k-block at a time: 4 B fragments (ds_read_b128), then their 16 MFMAs.  No double buffer: while this wave waits for
      // its fragments the lower-priority waves of the SIMD own the matrix pipe, and 16 more VGPRs of fragments do not fit the
      // 128-VGPR budget.  The fence keeps hipcc from hoisting every block's reads to the top of the pass (~200 spills).
#pragma unroll
      for (int j = 0; j < KJ; ++j) {
        float4 bc[WMAX];
#pragma unroll
        for (int w = 0; w < WMAX; ++w)
          if (w < Wn) bc[w] = *reinterpret_cast<const float4*>(Ws + ((j * 4 + g) * Np + c16) * 4 + (nt0 + w) * 64);
#pragma unroll
        for (int w = 0; w < WMAX; ++w) if (w < Wn) acc[w] = mfma16(av[j].x, bc[w].x, acc[w]);
#pragma unroll
        for (int w = 0; w < WMAX; ++w) if (w < Wn && BR_ABLATE_KEEP((j & 1) == 0)) acc[w] = mfma16(av[j].y, bc[w].y, acc[w]);
#pragma unroll
        for (int w = 0; w < WMAX; ++w) if (w < Wn && BR_ABLATE_KEEP(false)) acc[w] = mfma16(av[j].z, bc[w].z, acc[w]);
#pragma unroll
        for (int w = 0; w < WMAX; ++w) if (w < Wn && BR_ABLATE_KEEP(false)) acc[w] = mfma16(av[j].w, bc[w].w, acc[w]);
        __builtin_amdgcn_sched_barrier(0);
      }
      BR_STAMP(5 + 2 * (nt0 / 4));       // pass MFMAs issued
      if (WPS == 2) {
        if (Wn == 4) fwd_epilogue_multi<VEC, 4, NT>(acc, a, patch, yrow, rbase, nt0, c16, g, vmask, row_ok, bs, sigmask, floor_, ssum, ssq);
        else if (Wn == 3) fwd_epilogue_multi<VEC, 3, NT>(acc, a, patch, yrow, rbase, nt0, c16, g, vmask, row_ok, bs, sigmask, floor_, ssum, ssq);
        else if (Wn == 2) fwd_epilogue_multi<VEC, 2, NT>(acc, a, patch, yrow, rbase, nt0, c16, g, vmask, row_ok, bs, sigmask, floor_, ssum, ssq);
        else fwd_epilogue_multi<VEC, 1, NT>(acc, a, patch, yrow, rbase, nt0, c16, g, vmask, row_ok, bs, sigmask, floor_, ssum, ssq);
      } else
#pragma unroll
      for (int w = 0; w < WMAX; ++w) {
        if (w < Wn) {
          const float bv = bs[(nt0 + w) * 16 + c16];
          fwd_epilogue<VEC>(acc[w], a, patch, yrow, rbase, nt0 + w, c16, g, vmask, row_ok, bv, sigmask, floor_, ssum[nt0 + w], ssq[nt0 + w]);
          if (WPS == 4) __builtin_amdgcn_sched_barrier(0);      // one n-tile at a time: interleaved, the four epilogues' temporaries spill at 128 VGPRs
        }
      }
      BR_STAMP(6 + 2 * (nt0 / 4));       // pass epilogue issued
    }
    tile += tstride;
    if (WPS == 2) {
      if (has_next) {
#pragma unroll
        for (int j = 0; j < KJ; ++j) av[j] = avn[j];
#pragma unroll
        for (int w = 0; w < KWJ; ++w) kb[w] = kbn[w];
      }
    } else if (tile < n_tiles) fwd_load_tile<KJ, VEC>(av, kb, a, tile, c16, g);
  }

  if (a.stats) {      // lane -> wave (the 4 row groups) -> per-wave LDS partials -> workgroup sum in double -> one global atomic per column
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float sv = ssum[nt], q = ssq[nt];
      sv += __shfl_xor(sv, 16, 64); sv += __shfl_xor(sv, 32, 64);
      q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
      if (g == 0) { redw[wave * 2 * Np + nt * 16 + c16] = sv; redw[wave * 2 * Np + Np + nt * 16 + c16] = q; }
    }
    __syncthreads();
    double* rep = a.stats + (size_t)(blockIdx.x % kRep) * 2 * N;
    for (int t = threadIdx.x; t < 2 * Np; t += kFwdThreads) {
      const int half = t >= Np, n = t - half * Np;
      if (n < N) {
        double acc = 0.0;
#pragma unroll
        for (int w = 0; w < kFwdWaves; ++w) acc += (double)redw[w * 2 * Np + t];
        atomicAdd(rep + half * N + n, acc);
      }
    }
  }
  BR_STAMP(9);
  BR_STAMP_RT(11);
  BR_STAMP_FLUSH(blockIdx.x * kFwdWaves + wave);
}

}  // namespace br

using namespace br;

static inline int tiles16(int v) { return (v + 15) / 16; }
constexpr int kMaxT = 8;         // max 16-wide tiles along K or N (=> K,N <= 128 per launch)

extern "C" int64_t brDropoutKeepWords(int64_t batch, int K) { return (batch > 0 ? batch : 0) * (int64_t)((K + 31) / 32); }

extern "C" int brDropoutKeepBits(float drop_p, uint64_t seed, uint32_t step, int64_t row0, int64_t batch, int n_sites, const uint32_t* sites,
                                 const int* widths, uint32_t* const* out, brStream stream) {
  return br::dropout_keep_bits_ahead(drop_p, seed, step, 0, row0, batch, n_sites, sites, widths, out, stream);
}

// step_add: masks of step (resolved counter + step_add) - the step driver prefetches the next step's planes
int br::make_keep_args(KeepArgs& a, float drop_p, uint64_t seed, uint32_t step, uint32_t step_add, int64_t row0, int64_t batch, int n_sites,
                       const uint32_t* sites, const int* widths, uint32_t* const* out) {
  BR_CHECK_ARG(n_sites >= 1 && n_sites <= 3 && sites && widths && out && batch >= 0, "brDropoutKeepBits: bad args (1..3 sites)");
  BR_CHECK_ARG(drop_p > 0.f && drop_p < 1.f, "brDropoutKeepBits: drop_p must be in (0,1)");
  a.n_sites = n_sites;
  for (int i = 0; i < 3; ++i) {
    const int j = i < n_sites ? i : 0;
    BR_CHECK_ARG(widths[j] >= 1 && out[j], "brDropoutKeepBits: bad site %d", j);
    a.s[i].out = out[j]; a.s[i].K = widths[j]; a.s[i].kw = (widths[j] + 31) / 32; a.s[i].site = sites[j];
  }
  a.drop = make_dropout(drop_p, seed, step, 0);
  if (const StepStateDev* ss = current_step_state()) a.drop.step_ptr = &ss->step;
  a.drop.step_add = step_add;
  a.row0 = row0; a.batch = batch;
  return BR_OK;
}

int br::dropout_keep_bits_ahead(float drop_p, uint64_t seed, uint32_t step, uint32_t step_add, int64_t row0, int64_t batch, int n_sites,
                                const uint32_t* sites, const int* widths, uint32_t* const* out, brStream stream) {
  KeepArgs a;
  const int rc = make_keep_args(a, drop_p, seed, step, step_add, row0, batch, n_sites, sites, widths, out);
  if (rc != BR_OK) return rc;
  if (batch == 0) return BR_OK;
  int kwmax = 0;
  for (int i = 0; i < n_sites; ++i) kwmax = a.s[i].kw > kwmax ? a.s[i].kw : kwmax;
  const dim3 grid((unsigned)ceil_div(batch * kwmax, 256), (unsigned)n_sites);
  keep_bits_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
  BR_CHECK_LAUNCH("brDropoutKeepBits");
  return BR_OK;
}

template <int NT, int KJ, bool VEC, int WPS, bool EMU>
static void launch_fwd_v(unsigned grid, size_t shmem, hipStream_t s, const FwdArgs& a) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)dense_fwd_kernel<NT, KJ, VEC, WPS, EMU>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr_set = true;
  }
  dense_fwd_kernel<NT, KJ, VEC, WPS, EMU><<<grid, 256 * WPS, shmem, s>>>(a);
}
// waves per SIMD of a forward launch: 2 (two tiles per wave, software-pipelined) once every wave of a one-workgroup-per-CU grid would
// get two tiles anyway, else 4 (one tile per wave, more waves to hide latency); BR_FWD_WPS=2|4 forces one (experiments)
static int fwd_wps(int64_t batch, bool vec, int KJ, int NT, bool emu) {
  static const int forced = [] { const char* e = getenv("BR_FWD_WPS"); return e ? atoi(e) : 0; }();
  if (!vec) return 4;
  if (forced == 2 || forced == 4) return forced;
  // measured at batch 65 536: 128 x 100 36.2 -> 33.8 us with two pipelined tiles per wave, 100 x 50 23.0 -> 23.9 us (its loads are
  // short: the extra waves hide more than the prefetch does)
  // bf16x6 with K > 64 and more than one pass of n-tiles (or a ragged one): the 4-waves form spills (128 x 100: 135 VGPRs; N = 49..64
  // with its single full pass does not, and keeps the form that measured faster for 100 x 50)
  if (emu && KJ >= 5 && NT >= 3 && NT != 4) return 2;
  return (KJ >= 8 && ceil_div(batch, (int64_t)16) >= (int64_t)2 * 256 * 8) ? 2 : 4;
}

template <int NT, int KJ>
static void launch_fwd(size_t shmem_words_fixed, int Np, hipStream_t s, const FwdArgs& a, bool emu) {
  // VEC: 16-B accesses on both sides: x rows / y rows 16-B aligned with row strides that are multiples of 4 floats (rows of K
  // or N floats are then padded to 4, and a 16-B access that starts inside a row stays inside its allocation)
  const bool vec = (a.ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0) && (a.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0) &&
                   (a.yin == nullptr || a.K % 4 == 0);
  const int wps = fwd_wps(a.batch, vec, KJ, NT, emu);
  const int waves = 4 * wps;
  const size_t shmem = (shmem_words_fixed + (size_t)waves * (wps == 2 ? 4 : 1) * 16 * kPatchLd + (size_t)waves * 2 * Np) * sizeof(float);
  const int64_t wgs = ceil_div(ceil_div(a.batch, (int64_t)16), (int64_t)waves);
  const unsigned grid = (unsigned)(wgs < 1 ? 1 : (wgs > 256 ? 256 : wgs));      // one workgroup per CU
  if (emu) {
    if (vec && wps == 2) launch_fwd_v<NT, KJ, true, 2, true>(grid, shmem, s, a);
    else if (vec) launch_fwd_v<NT, KJ, true, 4, true>(grid, shmem, s, a);
    else launch_fwd_v<NT, KJ, false, 4, true>(grid, shmem, s, a);
    return;
  }
  if (vec && wps == 2) launch_fwd_v<NT, KJ, true, 2, false>(grid, shmem, s, a);
  else if (vec) launch_fwd_v<NT, KJ, true, 4, false>(grid, shmem, s, a);
  else launch_fwd_v<NT, KJ, false, 4, false>(grid, shmem, s, a);
}

static int dense_forward_one(FwdArgs a, hipStream_t s) {
  const int NT = tiles16(a.N), KJ = tiles16(a.K), Kp = KJ * 16, Np = NT * 16;
  const bool emu = br::mlp_bf16x6();
  const size_t wimg = emu ? (size_t)((KJ + 1) / 2) * 3 * 4 * Np * 4 : (size_t)Kp * Np;      // W image in floats: bf16x6 keeps three bf16 pieces per element
  const size_t fixed = wimg + 2 * (size_t)Kp + (size_t)Np;                      // W image, affine, bias (floats); + per-wave patches / sums
#define BR_FWD_KJ(NTv, KJv) case KJv: launch_fwd<NTv, KJv>(fixed, Np, s, a, emu); break;
#define BR_FWD(NTv)                                                                                        \
  case NTv:                                                                                                \
    switch (KJ) { BR_FWD_KJ(NTv, 1) BR_FWD_KJ(NTv, 2) BR_FWD_KJ(NTv, 3) BR_FWD_KJ(NTv, 4) BR_FWD_KJ(NTv, 5) \
                  BR_FWD_KJ(NTv, 6) BR_FWD_KJ(NTv, 7) BR_FWD_KJ(NTv, 8) default: break; }                  \
    break;
  switch (NT) {
    BR_FWD(1) BR_FWD(2) BR_FWD(3) BR_FWD(4) BR_FWD(5) BR_FWD(6) BR_FWD(7) BR_FWD(8)
    default: br::set_error("brDenseForward: unsupported N"); return BR_ERR_UNSUPPORTED;
  }
  BR_CHECK_LAUNCH("brDenseForward");
  return BR_OK;
}

// K > 128 (config 5: 2 x embed_dim 128 = 256 inputs) runs as two K-halves: the first launch leaves the raw partial
// sums in y, the second adds them in its epilogue (bias, activation, BatchNorm column sums only there).  The split
// point is a multiple of 32 so that each half starts on a word of the keep-bit plane.
static inline int split_k(int K) { return ((K / 2 + 31) / 32) * 32; }

extern "C" int brDenseForward(const float* x, int64_t ldx, const float* W, const float* bias, float* y, int64_t ldy,
                              int64_t batch, int K, int N, int act, const float* in_scale, const float* in_shift,
                              const brBnFold* in_bn, float drop_p, const uint32_t* keep, double* stats, brStream stream) {
  BR_CHECK_ARG(x && W && y && batch >= 0 && K >= 1 && N >= 1, "brDenseForward: bad args");
  BR_CHECK_ARG(K <= 2 * kMaxT * 16 && N <= kMaxT * 16, "brDenseForward: K=%d N=%d exceed %d / %d", K, N, 2 * kMaxT * 16, kMaxT * 16);
  BR_CHECK_ARG(ldx >= K && ldy >= N, "brDenseForward: bad leading dims");
  BR_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "brDenseForward: in_scale/in_shift both or neither");
  BR_CHECK_ARG(!in_bn || (!in_scale && in_bn->stats && in_bn->gamma && in_bn->beta && in_bn->scale && in_bn->shift && in_bn->mean && in_bn->rstd &&
                          in_bn->batch_total > 0 && (in_bn->moving_mean == nullptr) == (in_bn->moving_var == nullptr) && K <= kMaxT * 16),
               "brDenseForward: in_bn needs stats, gamma, beta and the four outputs, no in_scale/in_shift, K <= %d", kMaxT * 16);
  BR_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "brDenseForward: drop_p out of [0,1)");
  BR_CHECK_ARG((drop_p > 0.f) == (keep != nullptr), "brDenseForward: keep bits (brDropoutKeepBits) are required exactly when drop_p > 0");
  if (batch == 0) return BR_OK;
  hipStream_t s = (hipStream_t)stream;
  const int kw = (K + 31) / 32;
  brBnFold nofold{};
  FwdArgs a{x, ldx, W, bias, y, ldy, batch, K, N, act, in_scale, in_shift, in_bn ? *in_bn : nofold, keep, kw, drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f, stats, nullptr};
  if (K <= kMaxT * 16) return dense_forward_one(a, s);
  const int Ka = split_k(K);
  FwdArgs h = a;
  h.K = Ka; h.bias = nullptr; h.act = BR_ACT_LINEAR; h.stats = nullptr;
  int rc = dense_forward_one(h, s);
  if (rc != BR_OK) return rc;
  h = a;
  h.x = x + Ka; h.W = W + (int64_t)Ka * N; h.K = K - Ka; h.yin = y;
  if (in_scale) { h.scale = in_scale + Ka; h.shift = in_shift + Ka; }
  if (keep) h.keep = keep + Ka / 32;        // same row stride kw, first word of the second half
  return dense_forward_one(h, s);
}
