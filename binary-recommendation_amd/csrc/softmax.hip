// L4: in-batch softmax of the TwoTower retrieval task (tfrs.tasks.Retrieval, twoTower.py:47,82-83)
// and E1 full-catalogue scoring - the only MFMA-bound pieces of the path (2*B^2*semb flop per GEMM).
//
//   S = Q C^T (Bq x Bc) is NEVER materialised for the loss.  One kernel, four modes; in each the workgroup OWNS 128 rows of one
//   operand (32 per wave, kept in registers as the stationary MFMA operand for the whole sweep) and STREAMS 64-row tiles of the
//   other operand through LDS:
//     MODE_LSE    owned = queries     streamed = candidates   online logsumexp per query, loss_sum += sum_i (lse_i - S_i,diag)
//     MODE_GRAD_R owned = queries     streamed = candidates   dQ_i = sum_j (exp(S_ij - lse_i) - [j = diag(i)]) C_j
//     MODE_GRAD_C owned = candidates  streamed = queries      dC_j = sum_i (exp(S_ij - lse_i) - [j = diag(i)]) Q_i
//     MODE_SCORES owned = candidates  streamed = queries      scores[q][c] (top-k scoring: topKmetrics.py:17-43, twoTower.py:64-69)
//     MODE_LSE_GRAD_R = MODE_LSE and MODE_GRAD_R in ONE sweep (the training step needs both): the accumulator of dQ is kept relative
//                 to the running row maximum and rescaled when it grows (online softmax); dQ_i = acc_i / l_i - C_diag(i) at the end.
//                 The score tiles are formed twice per step instead of three times (4 GEMMs instead of 5 for lse + dQ + dC)
//   MFMA: v_mfma_f32_32x32x2_f32.  The score tile is formed TRANSPOSED, D[m = streamed entity][n = owned row]: in the result
//   layout a lane then holds ONE owned row (n = lane % 32) and its 16 registers hold 16 streamed entities
//   (m = 8 (r / 4) + 4 (lane / 32) + r % 4), so
//     * everything that is per owned row (its lse, its id, the running max / sum, the diagonal column) is one register per lane,
//       and the softmax reductions run over a lane's own registers - no cross-lane traffic until the very end;
//     * P = exp(S - lse) - [diag] feeds the second GEMM straight from those registers as the B operand
//       (k = streamed entity = register index, n = owned row): out^T[f][owned] += sum_k T[k][f] P[k][owned]; the A operand
//       T[k][f] comes from a transposed copy of the streamed tile in LDS (one ds_read_b128 = 4 consecutive k).  P never goes
//       through LDS (the previous kernel wrote it out and read it back in the A layout).
//   Contraction order of the score product: k = KH * (lane / 32) + kk (each lane half walks its own half of the feature
//   axis), so both operands are read as whole 16-byte vectors.  Two 32-entity sub-tiles are interleaved so that consecutive
//   MFMAs never depend on each other.
//   Occupancy: 128 owned rows per workgroup would leave most of the chip idle at the batch sizes of the reference (8 192 rows =
//   64 workgroups), so the streamed axis is SPLIT over blockIdx.y: each split writes partial results (per-row (max, sum, diag) or
//   a dQ / dC slab) into the caller's workspace and a small second kernel combines them in a fixed order (deterministic; no
//   float atomics).  Without a workspace the split count is 1 and results are written directly.
// [TF-sem] accidental hits: S_ij += FLT_MIN_TFRS (= float32 min / 100) when cand_ids[j] equals the
// id of query i's own positive and j is not that positive's column; SUM reduction over rows.
//   Round 3 (EMU, "bf16x6": csrc/dense.h): both GEMMs as fp32 products on the bf16 matrix pipe - v_mfma_f32_32x32x16_bf16, six
//   per 16-deep block on three-piece operands, the same result layout as the fp32 32x32x2 (so everything above stays as it is).  Every
//   operand is split once: the owned rows at kernel start (registers), the streamed tile by the thread that stages it - into TP
//   [sub-tile][k-block][piece][lane half][entity][8 bf16] (A fragments of the score product, k = 16 kb + 8 hb + j) and TT
//   [sub-tile][16-entity group][feature tile][piece][lane half][feature][8 bf16] (A fragments of the second GEMM: slot j of lane half hb
//   <-> the streamed entity register 8 g2 + j of the score tile holds, m = 8 (r / 4) + 4 hb + r % 4) - and P by the lane that holds it
//   (consecutive registers = consecutive k slots).  BR_MLP_MATH=f32 keeps the fp32 MFMAs.
#include "common.h"
#include "dense.h"

namespace br {

using f32x16 = __attribute__((ext_vector_type(16))) float;
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma32b(bf16x8 a, bf16x8 b, f32x16 c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#else
  return c;
#endif
}
// x w as six bf16 products, smallest first (x = xh + xm + xl, w = wh + wm + wl)
__device__ __forceinline__ f32x16 mfma32x6(bf16x8 xh, bf16x8 xm, bf16x8 xl, bf16x8 wh, bf16x8 wm, bf16x8 wl, f32x16 c) {
  c = mfma32b(xl, wh, c); c = mfma32b(xh, wl, c); c = mfma32b(xm, wm, c);
  c = mfma32b(xm, wh, c); c = mfma32b(xh, wm, c); c = mfma32b(xh, wh, c);
  return c;
}
constexpr int kGs = 33;                      // 16-B slots per (..., lane half) group of the piece images: 32 + 1 (the staging's half-word stores of
                                             // the 8 groups a wave touches would otherwise hit the same banks)

enum { MODE_SCORES = 0, MODE_LSE = 1, MODE_GRAD_R = 2, MODE_GRAD_C = 3, MODE_LSE_GRAD_R = 4 };

// exp of a non-positive argument (score - running max, score - lse) on the hardware exp2: one evaluation per score made
// these kernels VALU-bound with the accurate expf (~30 instructions each).  |x| <= ~88; the relative
// error grows like |x| * 6e-8, i.e. it is largest on the terms that contribute least to the sums.
__device__ __forceinline__ float exp_hw(float x) { return __expf(x); }
constexpr int kOwn = 128;                    // owned rows per workgroup (32 per wave)
constexpr int kTs = 64;                      // streamed entities per step (two sub-tiles of 32)
constexpr int kLdT = kTs + 4;                // row pitch of the transposed tile
constexpr float kMinFloat = -3.4028234663852886e+36f;   // np.finfo(float32).min / 100

struct InbatchArgs {
  const float* R; const float* T;            // owned / streamed operand (n_r x dim, n_t x dim, row-major)
  int64_t n_r, n_t; int dim;
  const void* own_ids; const void* str_ids;  // id of each owned / streamed entity (both or neither)
  int64_t diag;                              // streamed index of owned row o's diagonal partner = o + diag
  const float* lse_in;                       // GRAD_R: per owned row; GRAD_C: per streamed row
  float* out; int64_t ldo;                   // GRAD: [split][n_r][ldo] (split 0 only when n_split == 1); SCORES: scores[streamed][owned]
  float* part;                               // LSE with splits: [3][n_split][n_r] (max, sum, diag score)
  float* lse_out; double* loss_sum;          // LSE without splits
  int64_t t_per_split;                       // streamed rows per blockIdx.y (multiple of kTs)
};

// KQ = 16-byte vectors per lane half of the feature axis: features are padded to Kp = 8 KQ
template <int MODE, int KQ, typename IdT, bool EMU>
__global__ __launch_bounds__(256, 2) void inbatch_kernel(const InbatchArgs a) {
  constexpr int KH = 4 * KQ, Kp = 8 * KQ, ldt = Kp + 4;
  constexpr int FT = (Kp + 31) / 32;                     // 32-feature tiles of the second GEMM
  constexpr int KB = (Kp + 15) / 16;                     // EMU: 16-deep k-blocks of the score product
  constexpr bool GRAD = MODE == MODE_GRAD_R || MODE == MODE_GRAD_C || MODE == MODE_LSE_GRAD_R;
  constexpr bool FUSED = MODE == MODE_LSE_GRAD_R;
  constexpr int NPRE = (2 * KQ + 3) / 4;                 // 16-byte vectors each thread stages per step
  constexpr int kTpFloats = 2 * KB * 3 * 2 * kGs * 4, kTtFloats = GRAD ? 2 * 2 * FT * 3 * 2 * kGs * 4 : 0;      // EMU piece images
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ts = smem;                                      // [64][ldt]     streamed tile, row-major (A operand of the score product).  EMU: TP
  float* TsT = Ts + (EMU ? kTpFloats : kTs * ldt);       // [FT*32][68]   transposed (A operand of the second GEMM), GRAD only.  EMU: TT
  float* tlse = TsT + (EMU ? kTtFloats : (GRAD ? FT * 32 * kLdT : 0));       // [64]          GRAD_C: lse of the streamed queries
  IdT* tid = reinterpret_cast<IdT*>(tlse + kTs);         // [64]          ids of the streamed entities

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l32 = lane & 31, hb = lane >> 5;
  const int dim = a.dim;
  const int64_t own = (int64_t)blockIdx.x * kOwn + wave * 32 + l32;       // this lane's owned row
  const bool own_ok = own < a.n_r;
  const bool has_ids = a.own_ids != nullptr;
  const int64_t t_begin = (int64_t)blockIdx.y * a.t_per_split;
  const int64_t t_end = t_begin + a.t_per_split < a.n_t ? t_begin + a.t_per_split : a.n_t;

  // stationary operand: R[own][KH * hb + kk].  EMU: R[own][16 kb + 8 hb + j] as three bf16 pieces per k-block
  float rf[EMU ? 1 : KH];
  bf16x8 rh[EMU ? KB : 1], rm[EMU ? KB : 1], rl[EMU ? KB : 1];
  if constexpr (EMU) {
    const float* rp = a.R + (own_ok ? own : 0) * dim;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      float e8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int k = 16 * kb + 8 * hb + j; e8[j] = (own_ok && k < dim) ? rp[k] : 0.f; }
      uint32_t ph[4], pm[4], pl[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) split3(e8[2 * q], e8[2 * q + 1], ph[q], pm[q], pl[q]);
      rh[kb] = frag8(ph); rm[kb] = frag8(pm); rl[kb] = frag8(pl);
    }
  } else {
    const float* rp = a.R + (own_ok ? own : 0) * dim;
    const bool v4 = (dim & 3) == 0 && (reinterpret_cast<uintptr_t>(a.R) & 15) == 0;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int k = KH * hb + 4 * q;
      if (v4) {
        const float4 t = (own_ok && k < dim) ? *reinterpret_cast<const float4*>(rp + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        rf[4 * q] = t.x; rf[4 * q + 1] = t.y; rf[4 * q + 2] = t.z; rf[4 * q + 3] = t.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) rf[4 * q + e] = (own_ok && k + e < dim) ? rp[k + e] : 0.f;
      }
    }
  }
  IdT own_id = (IdT)-1;
  float own_lse = 0.f;
  if (own_ok) {
    if (has_ids) own_id = reinterpret_cast<const IdT*>(a.own_ids)[own];
    if (MODE == MODE_GRAD_R) own_lse = a.lse_in[own];
  }
  float run_m = -INFINITY, run_l = 0.f, diag_s = 0.f;    // LSE
  f32x16 gacc[GRAD ? FT : 1];
#pragma unroll
  for (int t = 0; t < (GRAD ? FT : 1); ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) gacc[t][r] = 0.f;
  if constexpr (EMU) {      // slots no step writes (features past Kp, the pad slot of every group) stay zero
    for (int idx = threadIdx.x; idx < kTpFloats + kTtFloats; idx += 256) Ts[idx] = 0.f;
  } else if (GRAD) {     // feature rows of the transposed tile that no step writes (Kp <= f < 32 FT) stay zero
    for (int idx = threadIdx.x; idx < (FT * 32 - Kp) * kLdT; idx += 256) TsT[Kp * kLdT + idx] = 0.f;
  }

  // staging: thread -> (streamed row sr = tid % 64, 16-byte feature chunks c4 = tid / 64 + 4 q): the transposed stores of a
  // wave hit 64 consecutive floats, the row-major ones 64 rows at a pitch of ldt floats (ldt % 64 in {36, 44, 60, 4}: conflict free)
  const int sr = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const bool tvec = (dim & 3) == 0 && (reinterpret_cast<uintptr_t>(a.T) & 15) == 0;
  float4 pre[NPRE];
  IdT pre_id = (IdT)-2;
  float pre_lse = 0.f;
  auto fetch = [&](int64_t c0) {
    const int64_t gr = c0 + sr;
    const bool rin = gr < t_end;
    const float* tp = a.T + (rin ? gr : 0) * dim;
#pragma unroll
    for (int q = 0; q < NPRE; ++q) {
      const int c = 4 * (cg + 4 * q);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tvec) {
        if (rin && c < dim) v = *reinterpret_cast<const float4*>(tp + c);
      } else if (rin) {
        v.x = c + 0 < dim ? tp[c + 0] : 0.f; v.y = c + 1 < dim ? tp[c + 1] : 0.f;
        v.z = c + 2 < dim ? tp[c + 2] : 0.f; v.w = c + 3 < dim ? tp[c + 3] : 0.f;
      }
      pre[q] = v;
    }
    if (cg == 0) {
      pre_id = (rin && has_ids) ? reinterpret_cast<const IdT*>(a.str_ids)[gr] : (IdT)-2;
      if (MODE == MODE_GRAD_C) pre_lse = rin ? a.lse_in[gr] : 0.f;
    }
  };
  if (t_begin < t_end) fetch(t_begin);

  // EMU: out^T[f][owned] += sum_k T[k][f] P[k][owned] for one 32-entity sub-tile whose P the lane holds in pv[0..15]: two 16-entity groups
  // (registers 8 g2 + j = k slot j), P split in registers, the A fragments (T^T pieces) from TT
  auto second_gemm_emu = [&](int sub, const float (&pv)[16]) {
    const float* tt = TsT + ((hb * kGs + l32) << 2);      // + ((((sub * 2 + g2) * FT + t) * 3 + p) * 2 * kGs) * 4
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
      uint32_t ph[4], pm[4], pl[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) split3(pv[8 * g2 + 2 * q], pv[8 * g2 + 2 * q + 1], ph[q], pm[q], pl[q]);
      const bf16x8 wh = frag8(ph), wm = frag8(pm), wl = frag8(pl);
#pragma unroll
      for (int t = 0; t < (GRAD ? FT : 1); ++t) {
        bf16x8 x[3];
#pragma unroll
        for (int p3 = 0; p3 < 3; ++p3) x[p3] = frag8(*reinterpret_cast<const float4*>(tt + ((((((sub * 2 + g2) * FT + t) * 3 + p3) * 2) * kGs) << 2)));
        gacc[t] = mfma32x6(x[0], x[1], x[2], wh, wm, wl, gacc[t]);
      }
    }
  };
  // diagonal: streamed index of this lane's partner, as an offset into the current tile (compared against the tile-local index)
  const int64_t partner = own + a.diag;
  const int jb = 4 * hb;                                  // lane-half part of the tile-local streamed index
  for (int64_t c0 = t_begin; c0 < t_end; c0 += kTs) {
    __syncthreads();                                      // the previous step's readers of the tile are done
#pragma unroll
    for (int q = 0; q < NPRE; ++q) {
      const int c = 4 * (cg + 4 * q);
      if constexpr (EMU) {
        if (c < Kp) {
          uint32_t h0, m0, l0, h1, m1, l1;
          split3(pre[q].x, pre[q].y, h0, m0, l0);
          split3(pre[q].z, pre[q].w, h1, m1, l1);
          const int sub = sr >> 5, mm = sr & 31;
          {   // TP: features c..c+3 are slots j0..j0+3 of entity mm in k-block c / 16, lane half (c / 8) & 1
            const int slot = (((sub * KB + (c >> 4)) * 3) * 2 + ((c >> 3) & 1)) * kGs + mm;
            float* d = Ts + slot * 4 + ((c & 7) >> 1);
            *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(d + 2 * kGs * 4) = make_uint2(m0, m1);
            *reinterpret_cast<uint2*>(d + 4 * kGs * 4) = make_uint2(l0, l1);
          }
          if (GRAD) {   // TT: entity mm is slot j of lane half (mm / 4) & 1 in the 16-entity group mm / 16 - the register 8 g2 + j that holds it
            const int g2 = mm >> 4, hb2 = (mm >> 2) & 1, j = 4 * ((mm >> 3) & 1) + (mm & 3);
            uint16_t* tt = reinterpret_cast<uint16_t*>(TsT);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int f = c + e;
              const int slot = ((((sub * 2 + g2) * FT + (f >> 5)) * 3) * 2 + hb2) * kGs + (f & 31);
              uint16_t* d = tt + slot * 8 + j;
              const uint32_t vh = e < 2 ? h0 : h1, vm = e < 2 ? m0 : m1, vl = e < 2 ? l0 : l1;
              d[0] = (uint16_t)((e & 1) ? (vh >> 16) : (vh & 0xffffu));
              d[2 * kGs * 8] = (uint16_t)((e & 1) ? (vm >> 16) : (vm & 0xffffu));
              d[4 * kGs * 8] = (uint16_t)((e & 1) ? (vl >> 16) : (vl & 0xffffu));
            }
          }
        }
      } else
      if (c < Kp) {
        *reinterpret_cast<float4*>(Ts + sr * ldt + c) = pre[q];
        if (GRAD) {
          TsT[(c + 0) * kLdT + sr] = pre[q].x; TsT[(c + 1) * kLdT + sr] = pre[q].y;
          TsT[(c + 2) * kLdT + sr] = pre[q].z; TsT[(c + 3) * kLdT + sr] = pre[q].w;
        }
      }
    }
    if (cg == 0) {
      tid[sr] = pre_id;
      if (MODE == MODE_GRAD_C) tlse[sr] = pre_lse;
    }
    __syncthreads();
    if (c0 + kTs < t_end) fetch(c0 + kTs);                // in flight during this step's MFMA work

    // ---- score tiles: D[m = streamed][n = owned], two sub-tiles interleaved ----
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
    if constexpr (EMU) {
      const float* tp = Ts + ((hb * kGs + l32) << 2);      // + (((sub * KB + kb) * 3 + p) * 2 * kGs) * 4
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        bf16x8 x0[3], x1[3];
#pragma unroll
        for (int p3 = 0; p3 < 3; ++p3) {
          x0[p3] = frag8(*reinterpret_cast<const float4*>(tp + ((((0 * KB + kb) * 3 + p3) * 2 * kGs) << 2)));
          x1[p3] = frag8(*reinterpret_cast<const float4*>(tp + ((((1 * KB + kb) * 3 + p3) * 2 * kGs) << 2)));
        }
        s0 = mfma32x6(x0[0], x0[1], x0[2], rh[kb], rm[kb], rl[kb], s0);
        s1 = mfma32x6(x1[0], x1[1], x1[2], rh[kb], rm[kb], rl[kb], s1);
      }
    } else {
    const float* ap = Ts + l32 * ldt + KH * hb;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const float4 a0 = *reinterpret_cast<const float4*>(ap + 4 * q);
      const float4 a1 = *reinterpret_cast<const float4*>(ap + 32 * ldt + 4 * q);
      s0 = mfma32(a0.x, rf[4 * q + 0], s0); s1 = mfma32(a1.x, rf[4 * q + 0], s1);
      s0 = mfma32(a0.y, rf[4 * q + 1], s0); s1 = mfma32(a1.y, rf[4 * q + 1], s1);
      s0 = mfma32(a0.z, rf[4 * q + 2], s0); s1 = mfma32(a1.z, rf[4 * q + 2], s1);
      s0 = mfma32(a0.w, rf[4 * q + 3], s0); s1 = mfma32(a1.w, rf[4 * q + 3], s1);
    }
    }
    if (MODE == MODE_SCORES) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t gs = c0 + 32 * sub + 8 * (r >> 2) + jb + (r & 3);
          if (gs < t_end && own_ok) a.out[gs * a.ldo + own] = sub ? s1[r] : s0[r];
        }
      continue;
    }
    const int dd = (int)((partner - c0 < -1) ? -1 : (partner - c0 > 4096 ? 4096 : partner - c0)) - jb;   // tile-local diag index minus jb
    if (MODE == MODE_LSE) {
      const int lim = (int)(t_end - c0) - jb;             // tile-local validity bound (only the last tile is partial)
      float v[32];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          IdT ids4[4];
          if (has_ids) {
#pragma unroll
            for (int e = 0; e < 4; ++e) ids4[e] = tid[32 * sub + 8 * rq + jb + e];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int cj = 32 * sub + 8 * rq + e;         // tile-local streamed index minus jb
            float x = sub ? s1[4 * rq + e] : s0[4 * rq + e];
            const bool dg = cj == dd;
            if (has_ids) x += (!dg && ids4[e] == own_id) ? kMinFloat : 0.f;
            diag_s = dg ? x : diag_s;
            v[16 * sub + 4 * rq + e] = cj < lim ? x : -INFINITY;
          }
        }
      float m = v[0];
#pragma unroll
      for (int i = 1; i < 32; ++i) m = fmaxf(m, v[i]);
      const float nm = fmaxf(run_m, m);                   // > -inf: every tile has at least one valid entity
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 32; ++i) sum += exp_hw(v[i] - nm);
      run_l = run_l * exp_hw(run_m - nm) + sum;
      run_m = nm;
      continue;
    }
    if (FUSED) {
      // ---- online softmax + dQ accumulator relative to the running maximum ----
      float v[32];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          IdT ids4[4];
          if (has_ids) {
#pragma unroll
            for (int e = 0; e < 4; ++e) ids4[e] = tid[32 * sub + 8 * rq + jb + e];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int cj = 32 * sub + 8 * rq + e;
            float x = sub ? s1[4 * rq + e] : s0[4 * rq + e];
            const bool dg = cj == dd;
            if (has_ids) x += (!dg && ids4[e] == own_id) ? kMinFloat : 0.f;
            diag_s = dg ? x : diag_s;
            v[16 * sub + 4 * rq + e] = cj < (int)(t_end - c0) - jb ? x : -INFINITY;
          }
        }
      float m = v[0];
#pragma unroll
      for (int i = 1; i < 32; ++i) m = fmaxf(m, v[i]);
      m = fmaxf(m, __shfl_xor(m, 32, 64));               // both lane halves of an owned row feed the same accumulator: one maximum
      const float nm = fmaxf(run_m, m);                   // > -inf: every tile has at least one valid entity
      const float sc = exp_hw(run_m - nm);                // 0 on the first step (run_m = -inf)
      run_m = nm;
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 32; ++i) { v[i] = exp_hw(v[i] - nm); sum += v[i]; }
      run_l = run_l * sc + sum;
#pragma unroll
      for (int t = 0; t < FT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) gacc[t][r] *= sc;
      if constexpr (EMU) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
          float pv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) pv[r] = v[16 * sub + r];
          second_gemm_emu(sub, pv);
        }
        continue;
      }
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int j0 = 32 * sub + 8 * rq + jb;
#pragma unroll
          for (int t = 0; t < FT; ++t) {
            const float4 a4 = *reinterpret_cast<const float4*>(TsT + (32 * t + l32) * kLdT + j0);
            gacc[t] = mfma32(a4.x, v[16 * sub + 4 * rq + 0], gacc[t]);
            gacc[t] = mfma32(a4.y, v[16 * sub + 4 * rq + 1], gacc[t]);
            gacc[t] = mfma32(a4.z, v[16 * sub + 4 * rq + 2], gacc[t]);
            gacc[t] = mfma32(a4.w, v[16 * sub + 4 * rq + 3], gacc[t]);
          }
        }
      continue;
    }
    // ---- GRAD: P in registers -> B operand of out^T[f][owned] += sum_k T[k][f] P[k][owned] ----
    // (streamed rows past the end and owned rows past the end need no mask: their T rows are zero / their results are not stored)
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      float pv[16];
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int j0 = 32 * sub + 8 * rq + jb;
        IdT ids4[4];
        float l4[4];
        if (has_ids) {
#pragma unroll
          for (int e = 0; e < 4; ++e) ids4[e] = tid[j0 + e];
        }
        if (MODE == MODE_GRAD_C) {
          const float4 t = *reinterpret_cast<const float4*>(tlse + j0);
          l4[0] = t.x; l4[1] = t.y; l4[2] = t.z; l4[3] = t.w;
        }
        float p[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int cj = 32 * sub + 8 * rq + e;
          float x = sub ? s1[4 * rq + e] : s0[4 * rq + e];
          const bool dg = cj == dd;
          if (has_ids) x += (!dg && ids4[e] == own_id) ? kMinFloat : 0.f;
          p[e] = exp_hw(x - (MODE == MODE_GRAD_R ? own_lse : l4[e])) - (dg ? 1.f : 0.f);
          pv[4 * rq + e] = p[e];
        }
        if constexpr (!EMU) {
#pragma unroll
          for (int t = 0; t < FT; ++t) {
            const float4 a4 = *reinterpret_cast<const float4*>(TsT + (32 * t + l32) * kLdT + j0);
            gacc[t] = mfma32(a4.x, p[0], gacc[t]);
            gacc[t] = mfma32(a4.y, p[1], gacc[t]);
            gacc[t] = mfma32(a4.z, p[2], gacc[t]);
            gacc[t] = mfma32(a4.w, p[3], gacc[t]);
          }
        }
      }
      if constexpr (EMU) second_gemm_emu(sub, pv);
    }
  }

  if (FUSED) {
    // run_m is already common to the two lane halves of a row; l and the diagonal score are per half
    const float l = run_l + __shfl_xor(run_l, 32, 64);
    const float d = diag_s + __shfl_xor(diag_s, 32, 64);
    const bool v4 = (a.ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 && (dim & 3) == 0 && (reinterpret_cast<uintptr_t>(a.T) & 15) == 0;
    if (a.part) {                                         // split: unnormalised accumulator slab + (max, sum, diag) per row
      if (hb == 0 && own_ok) {
        const int64_t ns = gridDim.y;
        a.part[(0 * ns + blockIdx.y) * a.n_r + own] = run_m;
        a.part[(1 * ns + blockIdx.y) * a.n_r + own] = l;
        a.part[(2 * ns + blockIdx.y) * a.n_r + own] = d;
      }
    } else {
      const float lse = run_m + logf(l);
      double part = 0.0;
      if (hb == 0 && own_ok) { a.lse_out[own] = lse; part = (double)lse - (double)d; }
      part = wave_sum_d(part);
      __shared__ double red[4];
      if (lane == 0) red[wave] = part;
      __syncthreads();
      if (threadIdx.x == 0 && a.loss_sum) atomicAdd(a.loss_sum + (blockIdx.x & (BR_SUM_SLOTS - 1)), red[0] + red[1] + red[2] + red[3]);
    }
    const float inv_l = a.part ? 1.f : 1.f / l;           // direct: dQ = acc / l - C[diag]
    const int64_t partner_row = own + a.diag;
    const bool has_diag = !a.part && partner_row >= 0 && partner_row < a.n_t;
    float* op = a.out + ((int64_t)blockIdx.y * a.n_r + (own_ok ? own : 0)) * a.ldo;
    const float* cp = a.T + (has_diag ? partner_row : 0) * dim;
#pragma unroll
    for (int t = 0; t < FT; ++t)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int f = 32 * t + 8 * rq + jb;
        if (!own_ok || f >= dim) continue;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = gacc[t][4 * rq + e] * inv_l - ((has_diag && f + e < dim) ? cp[f + e] : 0.f);
        if (v4 && f + 3 < dim) {
          *reinterpret_cast<float4*>(op + f) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (f + e < dim) op[f + e] = o[e];
        }
      }
    return;
  }
  if (MODE == MODE_LSE) {
    // lanes l and l + 32 hold disjoint streamed entities of the same owned row
    const float om = __shfl_xor(run_m, 32, 64), ol = __shfl_xor(run_l, 32, 64), od = __shfl_xor(diag_s, 32, 64);
    const float nm = fmaxf(run_m, om);
    const float l = (run_m > -INFINITY ? run_l * exp_hw(run_m - nm) : 0.f) + (om > -INFINITY ? ol * exp_hw(om - nm) : 0.f);
    const float d = diag_s + od;                          // exactly one lane-step saw the diagonal (others hold 0)
    double part = 0.0;
    if (hb == 0 && own_ok) {
      if (a.part) {
        const int64_t ns = gridDim.y;
        a.part[(0 * ns + blockIdx.y) * a.n_r + own] = nm;
        a.part[(1 * ns + blockIdx.y) * a.n_r + own] = l;
        a.part[(2 * ns + blockIdx.y) * a.n_r + own] = d;
      } else {
        const float lse = nm + logf(l);
        a.lse_out[own] = lse;
        part = (double)lse - (double)d;
      }
    }
    if (!a.part) {
      part = wave_sum_d(part);
      __shared__ double red[4];
      if (lane == 0) red[wave] = part;
      __syncthreads();
      if (threadIdx.x == 0 && a.loss_sum) atomicAdd(a.loss_sum + (blockIdx.x & (BR_SUM_SLOTS - 1)), red[0] + red[1] + red[2] + red[3]);
    }
    return;
  }
  if (GRAD) {
    // gacc[t][r] = out[own][32 t + 8 (r / 4) + 4 hb + r % 4]: four consecutive features per register group
    float* op = a.out + ((int64_t)blockIdx.y * a.n_r + (own_ok ? own : 0)) * a.ldo;
    const bool v4 = (a.ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0;
#pragma unroll
    for (int t = 0; t < FT; ++t)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int f = 32 * t + 8 * rq + jb;
        if (!own_ok || f >= dim) continue;
        if (v4 && f + 3 < dim) {
          *reinterpret_cast<float4*>(op + f) = make_float4(gacc[t][4 * rq], gacc[t][4 * rq + 1], gacc[t][4 * rq + 2], gacc[t][4 * rq + 3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (f + e < dim) op[f + e] = gacc[t][4 * rq + e];
        }
      }
  }
}

// combine the per-split (max, sum, diag) of every query: lse, loss_sum += sum_i (lse_i - diag_i)
__global__ __launch_bounds__(256) void lse_combine_kernel(const float* __restrict__ part, int n_split, int64_t n_r, float* __restrict__ lse_out,
                                                           double* __restrict__ loss_sum) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double mine = 0.0;
  if (i < n_r) {
    float m = -INFINITY, d = 0.f;
    for (int s = 0; s < n_split; ++s) m = fmaxf(m, part[(0 * (int64_t)n_split + s) * n_r + i]);
    float l = 0.f;
    for (int s = 0; s < n_split; ++s) {
      const float ms = part[(0 * (int64_t)n_split + s) * n_r + i];
      if (ms > -INFINITY) l += part[(1 * (int64_t)n_split + s) * n_r + i] * exp_hw(ms - m);
      d += part[(2 * (int64_t)n_split + s) * n_r + i];
    }
    const float lse = m + logf(l);
    lse_out[i] = lse;
    mine = (double)lse - (double)d;
  }
  mine = wave_sum_d(mine);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0 && loss_sum) atomicAdd(loss_sum + (blockIdx.x & (BR_SUM_SLOTS - 1)), red[0] + red[1] + red[2] + red[3]);
}

// fused sweep with splits: per row M = max_s m_s, l = sum_s l_s e^(m_s - M), lse = M + log l, loss_sum += lse - diag;
// dQ[i][f] = sum_s slab[s][i][f] e^(m_s - M) / l - C[i + diag][f]
__global__ __launch_bounds__(256) void lse_grad_combine_kernel(const float* __restrict__ part, const float* __restrict__ slabs, int n_split, int64_t n_r,
                                                                int dim, int64_t ldo, const float* __restrict__ C, int64_t n_c, int64_t diag,
                                                                float* __restrict__ lse_out, double* __restrict__ loss_sum, float* __restrict__ dQ) {
  // one 64-lane wave per row: lane -> features lane, lane + 64
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  double mine = 0.0;
  if (i < n_r) {
    float M = -INFINITY, d = 0.f;
    for (int s = 0; s < n_split; ++s) M = fmaxf(M, part[(0 * (int64_t)n_split + s) * n_r + i]);
    float l = 0.f;
    for (int s = 0; s < n_split; ++s) {
      const float ms = part[(0 * (int64_t)n_split + s) * n_r + i];
      if (ms > -INFINITY) l += part[(1 * (int64_t)n_split + s) * n_r + i] * exp_hw(ms - M);
      d += part[(2 * (int64_t)n_split + s) * n_r + i];
    }
    const float lse = M + logf(l);
    if (lane == 0) { lse_out[i] = lse; mine = (double)lse - (double)d; }
    const float inv_l = 1.f / l;
    const int64_t pr = i + diag;
    const bool has_diag = pr >= 0 && pr < n_c;
    for (int f = lane; f < dim; f += 64) {
      float acc = 0.f;
      for (int s = 0; s < n_split; ++s) {
        const float ms = part[(0 * (int64_t)n_split + s) * n_r + i];
        if (ms > -INFINITY) acc += slabs[((int64_t)s * n_r + i) * ldo + f] * exp_hw(ms - M);
      }
      dQ[i * ldo + f] = acc * inv_l - (has_diag ? C[pr * dim + f] : 0.f);
    }
  }
  mine = wave_sum_d(mine);
  __shared__ double red[4];
  if (lane == 0) red[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0 && loss_sum) atomicAdd(loss_sum + (blockIdx.x & (BR_SUM_SLOTS - 1)), red[0] + red[1] + red[2] + red[3]);
}

// out[i] = sum_s slabs[s][i] in split order (n = rows * ld floats, n % 4 == 0 or scalar tail)
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ slabs, int n_split, int64_t n, float* __restrict__ out) {
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 + 3 < n && ((reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 && (n & 3) == 0) {
    float4 acc = *reinterpret_cast<const float4*>(slabs + i4);
    for (int s = 1; s < n_split; ++s) {
      const float4 v = *reinterpret_cast<const float4*>(slabs + (int64_t)s * n + i4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(out + i4) = acc;
  } else {
    for (int64_t i = i4; i < n && i < i4 + 4; ++i) {
      float acc = slabs[i];
      for (int s = 1; s < n_split; ++s) acc += slabs[(int64_t)s * n + i];
      out[i] = acc;
    }
  }
}

// E1: stable top-k of one score row per workgroup (ties keep the lower item position).
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ scores, int64_t n_items, int k,
                                                         float* __restrict__ out_s, int32_t* __restrict__ out_i) {
  const float* row = scores + (int64_t)blockIdx.x * n_items;
  __shared__ float bs[4];
  __shared__ int bi[4];
  __shared__ float sel_s;
  __shared__ int sel_i;
  float prev_s = INFINITY;
  int prev_i = -1;
  for (int t = 0; t < k; ++t) {
    // best (score desc, index asc) strictly after (prev_s, prev_i)
    float best = -INFINITY;
    int besti = 0x7FFFFFFF;
    for (int64_t i = threadIdx.x; i < n_items; i += blockDim.x) {
      const float s = row[i];
      const bool after = (s < prev_s) || (s == prev_s && (int)i > prev_i);
      if (after && (s > best || (s == best && (int)i < besti))) { best = s; besti = (int)i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float os = __shfl_xor(best, off, 64);
      const int oi = __shfl_xor(besti, off, 64);
      if (os > best || (os == best && oi < besti)) { best = os; besti = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bs[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float b = bs[0]; int ix = bi[0];
      for (int w = 1; w < 4; ++w) if (bs[w] > b || (bs[w] == b && bi[w] < ix)) { b = bs[w]; ix = bi[w]; }
      sel_s = b; sel_i = ix;
      out_s[(int64_t)blockIdx.x * k + t] = b;
      out_i[(int64_t)blockIdx.x * k + t] = (ix == 0x7FFFFFFF) ? -1 : ix;
    }
    __syncthreads();
    prev_s = sel_s; prev_i = sel_i;
    __syncthreads();
  }
}

}  // namespace br

using namespace br;

namespace {
constexpr int kTargetWgs = 512;      // two workgroups per CU

// splits of the streamed axis so that row_blocks * splits ~ kTargetWgs (each split: a whole number of 64-row steps)
int choose_splits(int64_t n_r, int64_t n_t, int64_t max_splits) {
  const int64_t rb = ceil_div(n_r, (int64_t)kOwn), steps = ceil_div(n_t, (int64_t)kTs);
  int64_t s = ceil_div((int64_t)kTargetWgs, rb);
  if (s > steps) s = steps;
  if (s > max_splits) s = max_splits;
  if (s > 64) s = 64;
  return (int)(s < 1 ? 1 : s);
}

template <int MODE, int KQ, typename IdT, bool EMU>
void launch_inbatch_e(const InbatchArgs& a, int n_split, hipStream_t s) {
  constexpr int Kp = 8 * KQ, FT = (Kp + 31) / 32, KB = (Kp + 15) / 16;
  constexpr bool GRAD = MODE == MODE_GRAD_R || MODE == MODE_GRAD_C || MODE == MODE_LSE_GRAD_R;
  const size_t img = EMU ? (size_t)2 * KB * 3 * 2 * kGs * 4 + (GRAD ? (size_t)2 * 2 * FT * 3 * 2 * kGs * 4 : 0)
                         : (size_t)kTs * (Kp + 4) + (GRAD ? (size_t)FT * 32 * kLdT : 0);
  const size_t shmem = (img + kTs) * sizeof(float) + kTs * sizeof(IdT);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)inbatch_kernel<MODE, KQ, IdT, EMU>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    attr = true;
  }
  const dim3 grid((unsigned)ceil_div(a.n_r, (int64_t)kOwn), (unsigned)n_split);
  inbatch_kernel<MODE, KQ, IdT, EMU><<<grid, 256, shmem, s>>>(a);
}
template <int MODE, int KQ, typename IdT>
void launch_inbatch_k(const InbatchArgs& a, int n_split, hipStream_t s) {
  // bf16x6 where it measured faster (8 192 x 8 192 x 64: lse 120 -> 93 us, lse + dQ one sweep 185 -> 145, dC 163 -> 125): not for short streamed
  // axes (2 048: the piece staging costs more than the MFMAs save), not for the gradient modes past 64 features (the stationary pieces +
  // the accumulators of four feature tiles spill: 264 -> 340 us at 100 features)
  constexpr bool GRAD = MODE == MODE_GRAD_R || MODE == MODE_GRAD_C || MODE == MODE_LSE_GRAD_R;
  if (mlp_bf16x6() && a.n_t >= 4096 && (KQ <= 8 || !GRAD)) launch_inbatch_e<MODE, KQ, IdT, true>(a, n_split, s);
  else launch_inbatch_e<MODE, KQ, IdT, false>(a, n_split, s);
}

template <int MODE, typename IdT>
void launch_inbatch_t(const InbatchArgs& a, int n_split, hipStream_t s) {
  const int kq = (a.dim + 7) / 8;
  if (kq <= 4) launch_inbatch_k<MODE, 4, IdT>(a, n_split, s);
  else if (kq <= 7) launch_inbatch_k<MODE, 7, IdT>(a, n_split, s);
  else if (kq <= 8) launch_inbatch_k<MODE, 8, IdT>(a, n_split, s);
  else if (kq <= 13) launch_inbatch_k<MODE, 13, IdT>(a, n_split, s);
  else launch_inbatch_k<MODE, 16, IdT>(a, n_split, s);
}

template <int MODE>
void launch_inbatch(InbatchArgs a, int n_split, int id_type, hipStream_t s) {
  const int64_t steps = ceil_div(a.n_t, (int64_t)kTs);
  a.t_per_split = ceil_div(steps, (int64_t)n_split) * kTs;
  n_split = (int)ceil_div(a.n_t, a.t_per_split);       // no empty split
  if (id_type == BR_IDS_I64) launch_inbatch_t<MODE, int64_t>(a, n_split, s);
  else launch_inbatch_t<MODE, int32_t>(a, n_split, s);
}
}  // namespace

extern "C" int64_t brInBatchSoftmaxWorkspaceBytes(int64_t Bq, int64_t Bc, int dim) {
  if (Bq <= 0 || Bc <= 0 || dim <= 0) return 0;
  const int64_t lse = (int64_t)3 * choose_splits(Bq, Bc, 64) * Bq;
  const int64_t gq = choose_splits(Bq, Bc, 64) > 1 ? (int64_t)choose_splits(Bq, Bc, 64) * Bq * dim : 0;
  const int64_t gc = choose_splits(Bc, Bq, 64) > 1 ? (int64_t)choose_splits(Bc, Bq, 64) * Bc * dim : 0;
  const int64_t fused = gq ? gq + lse : 0;             // brInBatchSoftmaxLseGradQ: slabs + (max, sum, diag)
  int64_t fl = lse > gq ? (lse > gc ? lse : gc) : (gq > gc ? gq : gc);
  fl = fused > fl ? fused : fl;
  return fl * (int64_t)sizeof(float);
}

extern "C" int brInBatchSoftmaxLse(const float* Q, const float* C, const void* q_pos_ids, const void* cand_ids, int id_type, int64_t Bq,
                                   int64_t Bc, int dim, int64_t diag_offset, float* row_lse, double* loss_sum, void* ws, int64_t ws_bytes,
                                   brStream stream) {
  BR_CHECK_ARG(Q && C && row_lse && Bq >= 0 && Bc >= 1 && dim >= 1 && dim <= 128, "brInBatchSoftmaxLse: bad args (dim <= 128)");
  BR_CHECK_ARG((q_pos_ids == nullptr) == (cand_ids == nullptr), "brInBatchSoftmaxLse: ids both or neither");
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brInBatchSoftmaxLse: bad id_type");
  BR_CHECK_ARG(ws_bytes >= 0 && (ws != nullptr || ws_bytes == 0), "brInBatchSoftmaxLse: bad workspace");
  if (Bq == 0) return BR_OK;
  hipStream_t s = (hipStream_t)stream;
  const int n_split = choose_splits(Bq, Bc, ws ? ws_bytes / (int64_t)(3 * Bq * sizeof(float)) : 1);
  InbatchArgs a{Q, C, Bq, Bc, dim, q_pos_ids, cand_ids, diag_offset, nullptr, nullptr, 0, n_split > 1 ? (float*)ws : nullptr, row_lse, loss_sum, 0};
  launch_inbatch<MODE_LSE>(a, n_split, id_type, s);
  BR_CHECK_LAUNCH("brInBatchSoftmaxLse");
  if (n_split > 1) {
    const int64_t tps = ceil_div(ceil_div(Bc, (int64_t)kTs), (int64_t)n_split) * kTs;
    lse_combine_kernel<<<(unsigned)ceil_div(Bq, (int64_t)256), 256, 0, s>>>((const float*)ws, (int)ceil_div(Bc, tps), Bq, row_lse, loss_sum);
    BR_CHECK_LAUNCH("brInBatchSoftmaxLse(combine)");
  }
  return BR_OK;
}

extern "C" int brInBatchSoftmaxLseGradQ(const float* Q, const float* C, const void* q_pos_ids, const void* cand_ids, int id_type, int64_t Bq,
                                        int64_t Bc, int dim, int64_t diag_offset, float* row_lse, double* loss_sum, float* dQ, void* ws,
                                        int64_t ws_bytes, brStream stream) {
  BR_CHECK_ARG(Q && C && row_lse && dQ && Bq >= 0 && Bc >= 1 && dim >= 1 && dim <= 128, "brInBatchSoftmaxLseGradQ: bad args (dim <= 128)");
  BR_CHECK_ARG((q_pos_ids == nullptr) == (cand_ids == nullptr), "brInBatchSoftmaxLseGradQ: ids both or neither");
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brInBatchSoftmaxLseGradQ: bad id_type");
  BR_CHECK_ARG(ws_bytes >= 0 && (ws != nullptr || ws_bytes == 0), "brInBatchSoftmaxLseGradQ: bad workspace");
  if (Bq == 0) return BR_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dim > 64) {      // the fused sweep's registers (stationary rows + two score tiles + the dQ accumulator + 32 probabilities) only fit up to 64 features
    const int rc = brInBatchSoftmaxLse(Q, C, q_pos_ids, cand_ids, id_type, Bq, Bc, dim, diag_offset, row_lse, loss_sum, ws, ws_bytes, stream);
    if (rc != BR_OK) return rc;
    return brInBatchSoftmaxGrad(Q, C, q_pos_ids, cand_ids, id_type, Bq, Bc, dim, diag_offset, row_lse, dQ, nullptr, ws, ws_bytes, stream);
  }
  // workspace per split: the unnormalised dQ slab (Bq x dim) + (max, sum, diag) per row
  const int64_t per_split = (Bq * dim + 3 * Bq) * (int64_t)sizeof(float);
  const int n_split = choose_splits(Bq, Bc, ws ? ws_bytes / per_split : 1);
  const int64_t tps = ceil_div(ceil_div(Bc, (int64_t)kTs), (int64_t)n_split) * kTs;
  const int ns = (int)ceil_div(Bc, tps);                 // splits actually launched (launch_inbatch recomputes the same)
  float* slabs = (float*)ws;
  float* part = n_split > 1 ? slabs + (int64_t)ns * Bq * dim : nullptr;
  InbatchArgs a{Q, C, Bq, Bc, dim, q_pos_ids, cand_ids, diag_offset, nullptr, n_split > 1 ? slabs : dQ, dim, part, row_lse, loss_sum, 0};
  launch_inbatch<MODE_LSE_GRAD_R>(a, n_split, id_type, s);
  BR_CHECK_LAUNCH("brInBatchSoftmaxLseGradQ");
  if (n_split > 1) {
    lse_grad_combine_kernel<<<(unsigned)ceil_div(Bq, (int64_t)4), 256, 0, s>>>(part, slabs, ns, Bq, dim, dim, C, Bc, diag_offset, row_lse, loss_sum, dQ);
    BR_CHECK_LAUNCH("brInBatchSoftmaxLseGradQ(combine)");
  }
  return BR_OK;
}

// one gradient sweep: rows of `own` are owned, `str` is streamed; direct or through slabs in ws
template <int MODE>
static int grad_sweep(const float* own, const float* str, int64_t n_own, int64_t n_str, int dim, const void* own_ids, const void* str_ids, int id_type,
                      int64_t diag, const float* lse, float* out, void* ws, int64_t ws_bytes, hipStream_t s) {
  const int n_split = choose_splits(n_own, n_str, ws ? ws_bytes / (int64_t)(n_own * dim * sizeof(float)) : 1);
  InbatchArgs a{own, str, n_own, n_str, dim, own_ids, str_ids, diag, lse, n_split > 1 ? (float*)ws : out, dim, nullptr, nullptr, nullptr, 0};
  launch_inbatch<MODE>(a, n_split, id_type, s);
  BR_CHECK_LAUNCH("brInBatchSoftmaxGrad");
  if (n_split > 1) {
    const int64_t tps = ceil_div(ceil_div(n_str, (int64_t)kTs), (int64_t)n_split) * kTs;
    const int64_t n = n_own * dim;
    slab_sum_kernel<<<(unsigned)ceil_div(ceil_div(n, (int64_t)4), (int64_t)256), 256, 0, s>>>((const float*)ws, (int)ceil_div(n_str, tps), n, out);
    BR_CHECK_LAUNCH("brInBatchSoftmaxGrad(slab sum)");
  }
  return BR_OK;
}

extern "C" int brInBatchSoftmaxGrad(const float* Q, const float* C, const void* q_pos_ids, const void* cand_ids, int id_type, int64_t Bq,
                                    int64_t Bc, int dim, int64_t diag_offset, const float* row_lse, float* dQ, float* dC, void* ws,
                                    int64_t ws_bytes, brStream stream) {
  BR_CHECK_ARG(Q && C && row_lse && Bq >= 0 && Bc >= 1 && dim >= 1 && dim <= 128, "brInBatchSoftmaxGrad: bad args (dim <= 128)");
  BR_CHECK_ARG(dQ || dC, "brInBatchSoftmaxGrad: nothing to compute");
  BR_CHECK_ARG((q_pos_ids == nullptr) == (cand_ids == nullptr), "brInBatchSoftmaxGrad: ids both or neither");
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brInBatchSoftmaxGrad: bad id_type");
  BR_CHECK_ARG(ws_bytes >= 0 && (ws != nullptr || ws_bytes == 0), "brInBatchSoftmaxGrad: bad workspace");
  if (Bq == 0) return BR_OK;
  hipStream_t s = (hipStream_t)stream;
  // dQ: queries own, candidates stream; the partner of query i is candidate i + diag_offset
  if (dQ) { const int rc = grad_sweep<MODE_GRAD_R>(Q, C, Bq, Bc, dim, q_pos_ids, cand_ids, id_type, diag_offset, row_lse, dQ, ws, ws_bytes, s); if (rc) return rc; }
  // dC: candidates own, queries stream; the partner of candidate j is query j - diag_offset (the slabs of dQ are consumed by then)
  if (dC) { const int rc = grad_sweep<MODE_GRAD_C>(C, Q, Bc, Bq, dim, cand_ids, q_pos_ids, id_type, -diag_offset, row_lse, dC, ws, ws_bytes, s); if (rc) return rc; }
  return BR_OK;
}

extern "C" int brScoreMatrix(const float* Q, const float* C, int64_t n_q, int64_t n_c, int dim, float* scores, int64_t ld_scores, brStream stream) {
  BR_CHECK_ARG(Q && C && scores && n_q >= 0 && n_c >= 1 && dim >= 1 && dim <= 128 && ld_scores >= n_c, "brScoreMatrix: bad args (dim <= 128)");
  if (n_q == 0) return BR_OK;
  // candidates own (lanes = consecutive candidates = coalesced score rows), queries stream; the splits write disjoint rows
  InbatchArgs a{C, Q, n_c, n_q, dim, nullptr, nullptr, 0, nullptr, scores, ld_scores, nullptr, nullptr, nullptr, 0};
  launch_inbatch<MODE_SCORES>(a, choose_splits(n_c, n_q, 64), BR_IDS_I32, (hipStream_t)stream);
  BR_CHECK_LAUNCH("brScoreMatrix");
  return BR_OK;
}

extern "C" int brTopKRows(const float* scores, int64_t n_users, int64_t n_items, int k, float* out_scores, int32_t* out_index, brStream stream) {
  BR_CHECK_ARG(scores && out_scores && out_index && n_users >= 0 && n_items >= 1 && k >= 1 && k <= n_items && n_items < ((int64_t)1 << 31),
               "brTopKRows: bad args (1 <= k <= n_items < 2^31)");
  if (n_users == 0) return BR_OK;
  topk_rows_kernel<<<(unsigned)n_users, 256, 0, (hipStream_t)stream>>>(scores, n_items, k, out_scores, out_index);
  BR_CHECK_LAUNCH("brTopKRows");
  return BR_OK;
}
