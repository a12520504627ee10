// L4: in-batch softmax of the TwoTower retrieval task (tfrs.tasks.Retrieval, twoTower.py:47,82-83)
// and E1 full-catalogue scoring — the only MFMA-bound pieces of the path (2*B^2*semb flop).
//
//   S = Q C^T (Bq x Bc) is NEVER materialised for the loss: 64 x 64 tiles of S are formed on fp32
//   MFMA (v_mfma_f32_16x16x4_f32) from LDS-staged Q / C tiles and consumed in registers.
//   MODE_SCORES : write S (candidate scoring for top-k: topKmetrics.py:17-43, twoTower.py:64-69)
//   MODE_LSE    : streaming logsumexp per query row (online max / sum), accidental-hit mask,
//                 loss_sum += sum_i (lse_i - S_i,diag)
//   MODE_GRAD_R : rows = queries      : dQ_i  = sum_j (exp(S_ij - lse_i) - [j = diag(i)]) C_j
//   MODE_GRAD_C : rows = candidates   : dC_j  = sum_i (exp(S_ij - lse_i) - [j = diag(i)]) Q_i
// [TF-sem] accidental hits: S_ij += FLT_MIN_TFRS (= float32 min / 100) when cand_ids[j] equals the
// id of query i's own positive and j is not that positive's column; SUM reduction over rows.
#include "common.h"

namespace br {

using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ f32x4 mfma16s(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

enum { MODE_SCORES = 0, MODE_LSE = 1, MODE_GRAD_R = 2, MODE_GRAD_C = 3 };

// exp of a non-positive argument (score - running max, score - lse) on the hardware exp2: 16 evaluations per lane and
// 64-column tile made these kernels VALU-bound with the accurate expf (~30 instructions each).  |x| <= ~88; the relative
// error grows like |x| * 6e-8, i.e. it is largest on the terms that contribute least to the sums.
__device__ __forceinline__ float exp_hw(float x) { return __expf(x); }
constexpr int kT = 64;                       // tile edge (rows and columns of S per step)
constexpr float kMinFloat = -3.4028234663852886e+36f;   // np.finfo(float32).min / 100

// stage rows [r0, r0+64) of M (n_rows x dim, row-major) into LDS [64][ld], zero padded to Dp columns
__device__ __forceinline__ void stage_rows(float* dst, int ld, const float* __restrict__ M, int64_t n_rows, int64_t r0, int dim, int Dp) {
  const int cq = Dp >> 2;
  for (int idx = threadIdx.x; idx < kT * cq; idx += blockDim.x) {
    const int r = idx / cq, c = (idx - r * cq) << 2;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t gr = r0 + r;
    if (gr < n_rows) {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (c + e < dim) v[e] = M[gr * dim + c + e];
    }
    *reinterpret_cast<float4*>(dst + r * ld + c) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// One workgroup (256 threads = 4 waves) owns 64 rows of R and sweeps all column tiles of Cm.
// wave w: rows 16w..16w+15 of the tile; S accumulators: 4 column tiles x f32x4.
// Accumulator layout: lane (c16, g) holds S[row = 16w + 4g + r][col = ct*16 + c16], r = 0..3.
template <int MODE, typename IdT>
__global__ __launch_bounds__(256) void inbatch_kernel(const float* __restrict__ R, const float* __restrict__ Cm, int64_t n_r, int64_t n_c, int dim,
                                                       const IdT* __restrict__ q_pos_ids, const IdT* __restrict__ cand_ids,
                                                       int64_t diag_offset, const float* __restrict__ lse_in, float* __restrict__ out,
                                                       int64_t ldo, float* __restrict__ lse_out, double* __restrict__ loss_sum) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Dp = (dim + 15) & ~15, ld = Dp + 4, DT = Dp >> 4;
  float* Rs = smem;                 // [64][ld]
  float* Cs = Rs + kT * ld;         // [64][ld]
  float* Ps = Cs + kT * ld;         // [64][68]  P tile (grad modes)
  float* CsT = Ps + kT * 68;        // [Dp][68]  the column tile transposed (grad modes: B operand of P·C as ds_read_b128)
  float* aux = CsT + ((MODE == MODE_GRAD_R || MODE == MODE_GRAD_C) ? Dp * 68 : 0);   // [64] per-column lse (GRAD_C)
  int64_t* cid = reinterpret_cast<int64_t*>(aux + kT);   // [64] ids of the column entities
  constexpr int ldp = 68;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c16 = lane & 15, g = lane >> 4;
  const int64_t r0 = (int64_t)blockIdx.x * kT;
  stage_rows(Rs, ld, R, n_r, r0, dim, Dp);

  // per-lane row constants (rows 16w+4g+r)
  int64_t row_id[4];      // GRAD_C: candidate id of the row; else: id of the query's positive
  float row_lse[4], run_m[4], run_l[4], diag_s[4];
  bool has_ids = cand_ids != nullptr;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t gr = r0 + wave * 16 + 4 * g + r;
    row_id[r] = -1; row_lse[r] = 0.f; run_m[r] = -INFINITY; run_l[r] = 0.f; diag_s[r] = 0.f;
    if (gr < n_r) {
      if (has_ids) row_id[r] = (MODE == MODE_GRAD_C) ? (int64_t)cand_ids[gr] : (int64_t)q_pos_ids[gr];
      if (MODE == MODE_GRAD_R) row_lse[r] = lse_in[gr];
    }
  }
  f32x4 gacc[8];          // grad modes: out[16 rows][Dp] as DT column tiles
#pragma unroll
  for (int t = 0; t < 8; ++t) gacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // The column tile of step c0 + 64 is requested into registers BEFORE the MFMA work of step c0 and written to LDS after
  // it (the staging used to sit between two barriers with its global latency exposed on every one of the n_c / 64 steps).
  constexpr int MAXQ = kT * (128 / 4) / 256;        // float4 per thread at dim 128
  const int cq = Dp >> 2, n_q4 = kT * cq;
  const bool cvec = (dim % 4 == 0) && ((reinterpret_cast<uintptr_t>(Cm) & 15) == 0);
  float4 pre[MAXQ];
  auto fetch_tile = [&](int64_t c0n) {
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
      const int idx = threadIdx.x + 256 * q;
      const int idc = idx < n_q4 ? idx : 0;
      const int r = idc / cq, c = (idc - r * cq) << 2;
      const int64_t gr = c0n + r;
      const int64_t grc = gr < n_c ? gr : n_c - 1;
      float4 v;
      if (cvec) {
        const float4 t = *reinterpret_cast<const float4*>(Cm + grc * dim + (c < dim ? c : 0));
        const bool ok = gr < n_c && c < dim;
        v = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
      } else {
        const float* pr_ = Cm + grc * dim;
        const float t0 = pr_[c + 0 < dim ? c + 0 : dim - 1], t1 = pr_[c + 1 < dim ? c + 1 : dim - 1];
        const float t2 = pr_[c + 2 < dim ? c + 2 : dim - 1], t3 = pr_[c + 3 < dim ? c + 3 : dim - 1];
        const bool rin = gr < n_c;
        v = make_float4((rin && c + 0 < dim) ? t0 : 0.f, (rin && c + 1 < dim) ? t1 : 0.f, (rin && c + 2 < dim) ? t2 : 0.f, (rin && c + 3 < dim) ? t3 : 0.f);
      }
      pre[q] = v;
    }
  };
  fetch_tile(0);
  for (int64_t c0 = 0; c0 < n_c; c0 += kT) {
    __syncthreads();                                 // the previous step's readers of Cs / CsT / cid are done
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
      const int idx = threadIdx.x + 256 * q;
      if (idx < n_q4) {
        const int r = idx / cq, c = (idx - r * cq) << 2;
        *reinterpret_cast<float4*>(Cs + r * ld + c) = pre[q];
        if (MODE == MODE_GRAD_R || MODE == MODE_GRAD_C) {
          CsT[(c + 0) * 68 + r] = pre[q].x; CsT[(c + 1) * 68 + r] = pre[q].y;
          CsT[(c + 2) * 68 + r] = pre[q].z; CsT[(c + 3) * 68 + r] = pre[q].w;
        }
      }
    }
    if (threadIdx.x < kT) {
      const int64_t gc = c0 + threadIdx.x;
      int64_t v = -2;
      if (gc < n_c && has_ids) v = (MODE == MODE_GRAD_C) ? (int64_t)q_pos_ids[gc] : (int64_t)cand_ids[gc];
      cid[threadIdx.x] = v;
      if (MODE == MODE_GRAD_C) aux[threadIdx.x] = gc < n_c ? lse_in[gc] : 0.f;
    }
    __syncthreads();
    if (c0 + kT < n_c) fetch_tile(c0 + kT);          // in flight during this step's MFMA work
    // ---- S tile: rows of this wave x 64 columns ----
    f32x4 s[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) s[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* ar = Rs + (wave * 16 + c16) * ld + 4 * g;
    for (int j = 0; j < DT; ++j) {
      const float4 a4 = *reinterpret_cast<const float4*>(ar + 16 * j);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const float4 b4 = *reinterpret_cast<const float4*>(Cs + (ct * 16 + c16) * ld + 16 * j + 4 * g);
        s[ct] = mfma16s(a4.x, b4.x, s[ct]);
        s[ct] = mfma16s(a4.y, b4.y, s[ct]);
        s[ct] = mfma16s(a4.z, b4.z, s[ct]);
        s[ct] = mfma16s(a4.w, b4.w, s[ct]);
      }
    }
    if (MODE == MODE_SCORES) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t gr = r0 + wave * 16 + 4 * g + r, gc = c0 + ct * 16 + c16;
          if (gr < n_r && gc < n_c) out[gr * ldo + gc] = s[ct][r];
        }
      continue;
    }
    // ---- accidental-hit mask, validity, diagonal ----
    float p[4][4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int cl = ct * 16 + c16;
      const int64_t gc = c0 + cl;
      const int64_t col_id = cid[cl];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t gr = r0 + wave * 16 + 4 * g + r;
        // diagonal: query index q and its positive's column: col == q + diag_offset
        const bool is_diag = (MODE == MODE_GRAD_C) ? (gr == gc + diag_offset) : (gc == gr + diag_offset);
        float v = s[ct][r];
        if (has_ids && !is_diag && col_id == row_id[r]) v += kMinFloat;
        const bool valid = gc < n_c && gr < n_r;
        if (MODE == MODE_LSE) {
          if (is_diag && valid) diag_s[r] = v;
          p[ct][r] = valid ? v : -INFINITY;
        } else {
          const float l = (MODE == MODE_GRAD_R) ? row_lse[r] : aux[cl];
          p[ct][r] = valid ? (exp_hw(v - l) - (is_diag ? 1.f : 0.f)) : 0.f;
        }
      }
    }
    if (MODE == MODE_LSE) {
      // online logsumexp over the 64 columns: 4 register tiles, then the 16 lanes that share a row
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float m = fmaxf(fmaxf(p[0][r], p[1][r]), fmaxf(p[2][r], p[3][r]));
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        const float nm = fmaxf(run_m[r], m);
        float sum = 0.f;
        if (nm > -INFINITY) {
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) sum += exp_hw(p[ct][r] - nm);
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        run_l[r] = (nm > -INFINITY ? run_l[r] * exp_hw(run_m[r] - nm) : 0.f) + sum;
        run_m[r] = nm;
      }
      continue;
    }
    // ---- grad modes: out[rows] += P (16 x 64) · Cs (64 x Dp) ----
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ps[(wave * 16 + 4 * g + r) * ldp + ct * 16 + c16] = p[ct][r];
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave reads back only its own 16 rows of Ps
    __builtin_amdgcn_wave_barrier();
    const float* pr = Ps + (wave * 16 + c16) * ldp + 4 * g;
#pragma unroll
    for (int j = 0; j < 4; ++j) {          // contraction over the 64 columns of the tile
      const float4 a4 = *reinterpret_cast<const float4*>(pr + 16 * j);
      const float a[4] = {a4.x, a4.y, a4.z, a4.w};
      // B operand: 4 consecutive column entities (k = 16j + 4g + e) of feature t*16 + c16 = one ds_read_b128 of the
      // transposed tile (was one ds_read_b32 per MFMA)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (t < DT) {
          const float4 b4 = *reinterpret_cast<const float4*>(CsT + (t * 16 + c16) * 68 + 16 * j + 4 * g);
          gacc[t] = mfma16s(a[0], b4.x, gacc[t]);
          gacc[t] = mfma16s(a[1], b4.y, gacc[t]);
          gacc[t] = mfma16s(a[2], b4.z, gacc[t]);
          gacc[t] = mfma16s(a[3], b4.w, gacc[t]);
        }
      }
    }
  }

  if (MODE == MODE_LSE) {
    // every lane of a 16-lane row group holds the same (m, l); the diagonal score sits in ONE of them
    double part = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float d = diag_s[r];
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
      const int64_t gr = r0 + wave * 16 + 4 * g + r;
      if (c16 == 0 && gr < n_r) {
        const float l = run_m[r] + logf(run_l[r]);
        lse_out[gr] = l;
        part += (double)l - (double)d;
      }
    }
    part = wave_sum_d(part);
    __shared__ double red[4];
    if (lane == 0) red[wave] = part;
    __syncthreads();
    if (threadIdx.x == 0 && loss_sum) atomicAdd(loss_sum + (blockIdx.x & (BR_SUM_SLOTS - 1)), red[0] + red[1] + red[2] + red[3]);
    return;
  }
  if (MODE == MODE_GRAD_R || MODE == MODE_GRAD_C) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t < DT) {
        const int col = t * 16 + c16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t gr = r0 + wave * 16 + 4 * g + r;
          if (gr < n_r && col < dim) out[gr * ldo + col] = gacc[t][r];
        }
      }
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

template <int MODE>
static int launch_inbatch(const float* R, const float* Cm, int64_t n_r, int64_t n_c, int dim, const void* q_pos_ids, const void* cand_ids,
                          int id_type, int64_t diag_offset, const float* lse_in, float* out, int64_t ldo, float* lse_out, double* loss_sum,
                          hipStream_t s) {
  const int Dp = (dim + 15) & ~15;
  const bool grad = MODE == MODE_GRAD_R || MODE == MODE_GRAD_C;
  const size_t shmem = ((size_t)2 * kT * (Dp + 4) + (size_t)kT * 68 + (grad ? (size_t)Dp * 68 : 0) + kT) * sizeof(float) + kT * sizeof(int64_t) + 16;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)inbatch_kernel<MODE, int32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute((const void*)inbatch_kernel<MODE, int64_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr = true;
  }
  const unsigned grid = (unsigned)ceil_div(n_r, kT);
  if (id_type == BR_IDS_I64)
    inbatch_kernel<MODE, int64_t><<<grid, 256, shmem, s>>>(R, Cm, n_r, n_c, dim, (const int64_t*)q_pos_ids, (const int64_t*)cand_ids, diag_offset,
                                                            lse_in, out, ldo, lse_out, loss_sum);
  else
    inbatch_kernel<MODE, int32_t><<<grid, 256, shmem, s>>>(R, Cm, n_r, n_c, dim, (const int32_t*)q_pos_ids, (const int32_t*)cand_ids, diag_offset,
                                                            lse_in, out, ldo, lse_out, loss_sum);
  return 0;
}

extern "C" int brInBatchSoftmaxLse(const float* Q, const float* C, const void* q_pos_ids, const void* cand_ids, int id_type, int64_t Bq,
                                   int64_t Bc, int dim, int64_t diag_offset, float* row_lse, double* loss_sum, brStream stream) {
  BR_CHECK_ARG(Q && C && row_lse && Bq >= 0 && Bc >= 1 && dim >= 1 && dim <= 128, "brInBatchSoftmaxLse: bad args (dim <= 128)");
  BR_CHECK_ARG((q_pos_ids == nullptr) == (cand_ids == nullptr), "brInBatchSoftmaxLse: ids both or neither");
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brInBatchSoftmaxLse: bad id_type");
  if (Bq == 0) return BR_OK;
  launch_inbatch<MODE_LSE>(Q, C, Bq, Bc, dim, q_pos_ids, cand_ids, id_type, diag_offset, nullptr, nullptr, 0, row_lse, loss_sum, (hipStream_t)stream);
  BR_CHECK_LAUNCH("brInBatchSoftmaxLse");
  return BR_OK;
}

extern "C" int brInBatchSoftmaxGrad(const float* Q, const float* C, const void* q_pos_ids, const void* cand_ids, int id_type, int64_t Bq,
                                    int64_t Bc, int dim, int64_t diag_offset, const float* row_lse, float* dQ, float* dC, brStream stream) {
  BR_CHECK_ARG(Q && C && row_lse && Bq >= 0 && Bc >= 1 && dim >= 1 && dim <= 128, "brInBatchSoftmaxGrad: bad args (dim <= 128)");
  BR_CHECK_ARG(dQ || dC, "brInBatchSoftmaxGrad: nothing to compute");
  BR_CHECK_ARG((q_pos_ids == nullptr) == (cand_ids == nullptr), "brInBatchSoftmaxGrad: ids both or neither");
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brInBatchSoftmaxGrad: bad id_type");
  if (Bq == 0) return BR_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dQ) launch_inbatch<MODE_GRAD_R>(Q, C, Bq, Bc, dim, q_pos_ids, cand_ids, id_type, diag_offset, row_lse, dQ, dim, nullptr, nullptr, s);
  // dC: rows = candidates, columns = queries (the kernel swaps the roles of the id arrays itself)
  if (dC) launch_inbatch<MODE_GRAD_C>(C, Q, Bc, Bq, dim, q_pos_ids, cand_ids, id_type, diag_offset, row_lse, dC, dim, nullptr, nullptr, s);
  BR_CHECK_LAUNCH("brInBatchSoftmaxGrad");
  return BR_OK;
}

extern "C" int brScoreMatrix(const float* Q, const float* C, int64_t n_q, int64_t n_c, int dim, float* scores, int64_t ld_scores, brStream stream) {
  BR_CHECK_ARG(Q && C && scores && n_q >= 0 && n_c >= 1 && dim >= 1 && dim <= 128 && ld_scores >= n_c, "brScoreMatrix: bad args (dim <= 128)");
  if (n_q == 0) return BR_OK;
  launch_inbatch<MODE_SCORES>(Q, C, n_q, n_c, dim, nullptr, nullptr, BR_IDS_I32, 0, nullptr, scores, ld_scores, nullptr, nullptr, (hipStream_t)stream);
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
