// G1/G2 embedding gather, M1 row dot, NeuMF embed block, BPR fused step.
// HBM-bound kernels: one row group (16 lanes x 16 B at dim 64) per embedding row, all the
// lookups of one (user,item[,neg]) tuple issued before any store so each lane keeps 3-4
// independent 16-B loads in flight; 4096+ workgroups at batch 65 536 to fill 256 CUs.
#include "common.h"
#include "rows.h"
#include "adam_math.h"
#include "lookup_wave.h"

namespace br {

struct GatherArgs {
  const float* tables[4];
  const void* ids[4];
  float* outs[4];
  int64_t rows[4];
};

template <typename IdT, int VEC, int NT>
__global__ __launch_bounds__(256) void gather_rows_kernel(GatherArgs a, int dim, int chunks,
                                                           int lpr_log2, int64_t batch, int* err) {
  using V = typename VecT<VEC>::type;
  const int lpr = 1 << lpr_log2;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t b = tid >> lpr_log2;
  const int lir = (int)(tid & (lpr - 1));
  if (b >= batch) return;
  int64_t id[NT];
  bool ok[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    id[t] = load_id(reinterpret_cast<const IdT*>(a.ids[t]), b);
    ok[t] = (uint64_t)id[t] < (uint64_t)a.rows[t];
    if (!ok[t] && err && lir == 0) *err = 1;
  }
  for (int c = lir; c < chunks; c += lpr) {
    V v[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
      v[t] = ok[t] ? vload<VEC>(a.tables[t] + id[t] * dim + c * VEC) : vzero<VEC>();
#pragma unroll
    for (int t = 0; t < NT; ++t) vstore<VEC>(a.outs[t] + b * dim + c * VEC, v[t]);
  }
}

// out[p] = table[ids[p]] at the physical positions p = seg_phys(b) of a segmented id array (the owner side of the row-sharded exchange
// on tables without deferred Adam), up to two tables of one row width per launch (blockIdx.y)
struct GatherSegJob { const float* table; int64_t rows; const void* ids; float* out; int64_t seg_off; };
struct GatherSegJobs { GatherSegJob j[2]; };
template <typename IdT, int VEC>
__global__ __launch_bounds__(256) void gather_rows_seg_kernel(GatherSegJobs jobs, int dim, int chunks, int lpr_log2, int64_t n, int64_t seg_len, int64_t seg_stride,
                                                               int64_t ld_out, int* err) {
  const GatherSegJob& jb = jobs.j[blockIdx.y];
  const int lpr = 1 << lpr_log2;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t b = tid >> lpr_log2;
  const int lir = (int)(tid & (lpr - 1));
  if (b >= n) return;
  const int64_t p = seg_phys(b, seg_len, seg_stride, jb.seg_off);
  const int64_t id = load_id((const IdT*)jb.ids, p);
  const bool ok = (uint64_t)id < (uint64_t)jb.rows;
  if (!ok && err && lir == 0) *err = 1;
  for (int c = lir; c < chunks; c += lpr)
    vstore<VEC>(jb.out + p * ld_out + c * VEC, ok ? vload<VEC>(jb.table + id * dim + c * VEC) : vzero<VEC>());
}

template <int VEC>
__global__ __launch_bounds__(256) void row_dot_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                       float* __restrict__ out, int dim, int chunks,
                                                       int lpr_log2, int64_t batch) {
  const int lpr = 1 << lpr_log2;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t b = tid >> lpr_log2;
  const int lir = (int)(tid & (lpr - 1));
  const bool live = b < batch;
  if (!live) b = batch - 1;  // keep the whole wave in the shuffles
  float s = 0.f;
  for (int c = lir; c < chunks; c += lpr)
    s += vdot(vload<VEC>(A + b * dim + c * VEC), vload<VEC>(Bm + b * dim + c * VEC));
  s = rowgroup_sum(s, lpr);
  if (live && lir == 0) out[b] = s;
}

template <int VEC>
__global__ __launch_bounds__(256) void row_dot_bwd_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                           const float* __restrict__ dout, float* __restrict__ dA,
                                                           float* __restrict__ dB, int dim, int chunks,
                                                           int lpr_log2, int64_t batch) {
  const int lpr = 1 << lpr_log2;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t b = tid >> lpr_log2;
  const int lir = (int)(tid & (lpr - 1));
  if (b >= batch) return;
  const float g = dout[b];
  for (int c = lir; c < chunks; c += lpr) {
    auto va = vload<VEC>(A + b * dim + c * VEC);
    auto vb = vload<VEC>(Bm + b * dim + c * VEC);
    vstore<VEC>(dA + b * dim + c * VEC, vmul(vb, g));
    vstore<VEC>(dB + b * dim + c * VEC, vmul(va, g));
  }
}

// ---- NeuMF embed block ---------------------------------------------------------------------
template <typename IdT, int VEC>
__global__ __launch_bounds__(256) void neumf_embed_fwd_kernel(
    const float* __restrict__ user_mlp, const float* __restrict__ item_mlp,
    const float* __restrict__ user_mf, const float* __restrict__ item_mf, int64_t ldu, int64_t ldi,
    int64_t user_rows, int64_t item_rows, const IdT* __restrict__ users, const IdT* __restrict__ items,
    int dim, int chunks, int lpr_log2, int64_t batch, int item_first, float* __restrict__ x0,
    float* __restrict__ dot, int* err, float* __restrict__ stash_user, float* __restrict__ stash_item, int64_t ld_stash) {
  using V = typename VecT<VEC>::type;
  const int lpr = 1 << lpr_log2;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t b = tid >> lpr_log2;
  const int lir = (int)(tid & (lpr - 1));
  const bool live = b < batch;
  if (!live) b = batch - 1;
  const int64_t u = load_id(users, b), i = load_id(items, b);
  const bool uok = (uint64_t)u < (uint64_t)user_rows, iok = (uint64_t)i < (uint64_t)item_rows;
  if ((!uok || !iok) && err && lir == 0) *err = 1;
  float* xrow = x0 + b * (2 * (int64_t)dim);
  const int uoff = item_first ? dim : 0, ioff = item_first ? 0 : dim;
  float s = 0.f;
  for (int c = lir; c < chunks; c += lpr) {
    V um = uok ? vload<VEC>(user_mlp + u * ldu + c * VEC) : vzero<VEC>();
    V im = iok ? vload<VEC>(item_mlp + i * ldi + c * VEC) : vzero<VEC>();
    V uf = uok ? vload<VEC>(user_mf + u * ldu + c * VEC) : vzero<VEC>();
    V vf = iok ? vload<VEC>(item_mf + i * ldi + c * VEC) : vzero<VEC>();
    if (live) {
      vstore<VEC>(xrow + uoff + c * VEC, um);
      vstore<VEC>(xrow + ioff + c * VEC, im);
      if (stash_user) {      // the MF rows by batch position: the backward forms ddot * partner row from them (brSegmentSumToSlotsPair)
        vstore<VEC>(stash_user + b * ld_stash + c * VEC, uf);
        vstore<VEC>(stash_item + b * ld_stash + c * VEC, vf);
      }
    }
    s += vdot(uf, vf);
  }
  s = rowgroup_sum(s, lpr);
  if (live && lir == 0) dot[b] = s;
}

// Deferred-Adam variant of the embed forward on the FUSED tables (row = [mlp | mf], stride 2*dim): rows lag
// behind by the g = 0 steps (last[row], t-1]; they are replayed in registers (adam_math.h) and NOT written
// back (duplicate ids in a batch would race) — brAdamRowsSortedDeferred does the authoritative update of
// the unique rows later in the step.  The caught-up MF halves are stashed per pair for the backward.
template <typename IdT, int VEC>
__global__ __launch_bounds__(256) void neumf_embed_fwd_deferred_kernel(
    const float* __restrict__ user_tab, const float* __restrict__ user_m, const float* __restrict__ user_v,
    const int32_t* __restrict__ user_last, const float* __restrict__ item_tab, const float* __restrict__ item_m,
    const float* __restrict__ item_v, const int32_t* __restrict__ item_last, int64_t user_rows, int64_t item_rows,
    const IdT* __restrict__ users, const IdT* __restrict__ items, int dim, int chunks, int lpr_log2, int64_t batch,
    int item_first, const StepStateDev* __restrict__ ss, AdamHp h, float* __restrict__ x0, float* __restrict__ dot,
    float* __restrict__ stash_user, float* __restrict__ stash_item, int64_t ld_stash, int* err) {
  using V = typename VecT<VEC>::type;
  __shared__ float ring[BR_ALPHA_RING];
  stage_alpha_ring(ring, ss);
  const int lpr = 1 << lpr_log2;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t b = tid >> lpr_log2;
  const int lir = (int)(tid & (lpr - 1));
  const bool live = b < batch;
  if (!live) b = batch - 1;
  int64_t u = load_id(users, b), i = load_id(items, b);
  const bool uok = (uint64_t)u < (uint64_t)user_rows, iok = (uint64_t)i < (uint64_t)item_rows;
  if ((!uok || !iok) && err && lir == 0) *err = 1;
  if (!uok) u = 0;
  if (!iok) i = 0;
  const uint32_t t = ss->step;                       // the step being computed: rows must include steps <= t-1
  const uint32_t lu = (uint32_t)user_last[u], li = (uint32_t)item_last[i];
  const int64_t ld = 2 * (int64_t)dim;
  float* xrow = x0 + b * ld;
  const int uoff = item_first ? dim : 0, ioff = item_first ? 0 : dim;
  float s = 0.f;
  for (int c = lir; c < chunks; c += lpr) {
    const int64_t uo = u * ld + c * VEC, io = i * ld + c * VEC;
    V um = vload<VEC>(user_tab + uo), uf = vload<VEC>(user_tab + uo + dim);
    V im = vload<VEC>(item_tab + io), vf = vload<VEC>(item_tab + io + dim);
    if (lu + 1 < t) {
      V m0 = vload<VEC>(user_m + uo), v0 = vload<VEC>(user_v + uo), m1 = vload<VEC>(user_m + uo + dim), v1 = vload<VEC>(user_v + uo + dim);
      adam_catch_up<false>(um, m0, v0, lu, t - 1, ring, ss, h);
      adam_catch_up<false>(uf, m1, v1, lu, t - 1, ring, ss, h);
    }
    if (li + 1 < t) {
      V m0 = vload<VEC>(item_m + io), v0 = vload<VEC>(item_v + io), m1 = vload<VEC>(item_m + io + dim), v1 = vload<VEC>(item_v + io + dim);
      adam_catch_up<false>(im, m0, v0, li, t - 1, ring, ss, h);
      adam_catch_up<false>(vf, m1, v1, li, t - 1, ring, ss, h);
    }
    if (!uok) { um = vzero<VEC>(); uf = vzero<VEC>(); }
    if (!iok) { im = vzero<VEC>(); vf = vzero<VEC>(); }
    if (live) {
      vstore<VEC>(xrow + uoff + c * VEC, um);
      vstore<VEC>(xrow + ioff + c * VEC, im);
      vstore<VEC>(stash_user + b * ld_stash + c * VEC, uf);
      vstore<VEC>(stash_item + b * ld_stash + c * VEC, vf);
    }
    s += vdot(uf, vf);
  }
  s = rowgroup_sum(s, lpr);
  if (live && lir == 0) dot[b] = s;
}

// The same lookup with one WAVE per pair, for embed_dim = 32 * VEC (fused rows of 64 * VEC floats; VEC = 2, 4): a lane owns VEC
// columns of the user row and the same columns of the item row (lanes 0-31 the MLP halves, 32-63 the MF halves), so the two replay
// loops run exactly lag(user) + lag(item) steps with scalar control flow - the row-group form above holds four pairs per wave and
// every lane waits for the longest of four lags (twice the work at a geometric lag distribution) - and a fresh row skips its m / v
// loads by a scalar branch.  x0, the stashes and `dot` have the same bits as the row-group form (vdot's order chained across lanes).
template <typename IdT, int VEC>
__global__ __launch_bounds__(256) void neumf_embed_fwd_deferred_wave_kernel(const LookupArgs a) {
  const int64_t b = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (b >= a.batch) return;
  lookup_wave_pair<IdT, VEC>(a, b, (int)(threadIdx.x & 63));
}

// G1 on a deferred-Adam table (adam_math.h): out[b] = the row of ids[b] brought up to step t-1 in registers,
// nothing written back (the row-sharded owner serves lookups with it; the single-GPU step uses the fused
// neumf_embed_fwd_deferred_kernel).
template <typename IdT, int VEC>
__global__ __launch_bounds__(256) void gather_rows_deferred_kernel(
    const float* __restrict__ table, const float* __restrict__ M, const float* __restrict__ Vv, const int32_t* __restrict__ last,
    int64_t rows, int dim, int chunks, int lpr_log2, const IdT* __restrict__ ids, int64_t n, const StepStateDev* __restrict__ ss,
    AdamHp h, float* __restrict__ out, int64_t ld_out, int* err) {
  using V = typename VecT<VEC>::type;
  __shared__ float ring[BR_ALPHA_RING];
  stage_alpha_ring(ring, ss);
  const int lpr = 1 << lpr_log2;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t b = tid >> lpr_log2;
  const int lir = (int)(tid & (lpr - 1));
  if (b >= n) return;
  int64_t r = load_id(ids, b);
  const bool ok = (uint64_t)r < (uint64_t)rows;
  if (!ok) { if (err && lir == 0) *err = 1; r = 0; }
  const uint32_t t = ss->step, seen = (uint32_t)last[r];
  for (int c = lir; c < chunks; c += lpr) {
    const int64_t off = r * dim + c * VEC;
    V th = vload<VEC>(table + off);
    if (seen + 1 < t) {
      V m = vload<VEC>(M + off), v = vload<VEC>(Vv + off);
      adam_catch_up<false>(th, m, v, seen, t - 1, ring, ss, h);
    }
    vstore<VEC>(out + b * ld_out + c * VEC, ok ? th : vzero<VEC>());
  }
}

// The same gather with one WAVE per row, for rows of 64 * VEC floats: the replay loop runs exactly the row's lag with scalar
// control flow (the row-group form holds 2 - 16 rows per wave and every lane waits for the longest lag), a fresh row skips its
// m / v loads, alpha comes through scalar loads from the ring (adam_replay_uniform).  Same bits.
template <typename IdT, int VEC>
__global__ __launch_bounds__(256) void gather_rows_deferred_wave_kernel(
    const float* __restrict__ table, const float* __restrict__ M, const float* __restrict__ Vv, const int32_t* __restrict__ last,
    int64_t rows, const IdT* __restrict__ ids, int64_t n, const StepStateDev* __restrict__ ss, AdamHp h, float* __restrict__ out,
    int64_t ld_out, int* err) {
  using V = typename VecT<VEC>::type;
  constexpr int dim = 64 * VEC;
  const int64_t b = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (b >= n) return;
  const int lane = (int)(threadIdx.x & 63);
  int64_t r = load_id(ids, b);
  const bool ok = (uint64_t)r < (uint64_t)rows;
  if (!ok) { if (err && lane == 0) *err = 1; r = 0; }
  const uint32_t t = ss->step, seen = (uint32_t)last[r];
  const int64_t off = r * dim + lane * VEC;
  V th = vload<VEC>(table + off);
  if (seen + 1 < t) {
    V m = vload<VEC>(M + off), v = vload<VEC>(Vv + off);
    adam_catch_up_uniform<false>(th, m, v, seen, t - 1, ss, h);
  }
  vstore<VEC>(out + b * ld_out + lane * VEC, ok ? th : vzero<VEC>());
}

// two deferred tables of one geometry served by ONE launch (blockIdx.y): a row-sharded owner's user and item shard - each gather alone
// is a chain of dependent round trips per row and leaves HBM half idle
// (GatherDefJob / GatherDefJobs / gather_deferred_wave_row: lookup_wave.h - the fused gather + chunk-sort launch of sparse_opt.hip shares them)
template <typename IdT, int VEC>
__global__ __launch_bounds__(256) void gather_rows_deferred_wave_pair_kernel(GatherDefJobs jobs, const StepStateDev* __restrict__ ss, AdamHp h,
                                                                              int64_t ld_out, int* err) {
  const GatherDefJob& jb = jobs.j[blockIdx.y];
  const int64_t b = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (b >= jb.n) return;
  gather_deferred_wave_row<IdT, VEC>(jobs, jb, b, (int)(threadIdx.x & 63), ss, h, ld_out, err);
}

// B1 of the GMF dot on the stashed MF rows, in place: (u, i) -> (ddot * i, ddot * u).  No __restrict__: the
// two outputs ARE the two inputs.
template <int VEC>
__global__ __launch_bounds__(256) void mf_grad_inplace_kernel(float* su, float* si, int64_t ld, const float* __restrict__ ddot,
                                                               int64_t batch, int chunks) {
  using V = typename VecT<VEC>::type;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= batch * chunks) return;
  const int64_t b = e / chunks;
  const int c = (int)(e - b * chunks);
  const float g = ddot[b];
  const V u = vload<VEC>(su + b * ld + c * VEC), i = vload<VEC>(si + b * ld + c * VEC);
  vstore<VEC>(su + b * ld + c * VEC, vmul(i, g));
  vstore<VEC>(si + b * ld + c * VEC, vmul(u, g));
}

template <typename IdT, int VEC>
__global__ __launch_bounds__(256) void neumf_embed_bwd_kernel(
    const float* __restrict__ user_mf, const float* __restrict__ item_mf, int64_t ldu, int64_t ldi,
    int64_t user_rows, int64_t item_rows, const IdT* __restrict__ users, const IdT* __restrict__ items,
    int dim, int chunks, int lpr_log2, int64_t batch, int item_first, const float* __restrict__ dx0,
    const float* __restrict__ ddot, float* __restrict__ g_user_mlp, float* __restrict__ g_item_mlp,
    float* __restrict__ g_user_mf, float* __restrict__ g_item_mf, int64_t ldg, int out_by_id) {
  using V = typename VecT<VEC>::type;
  const int lpr = 1 << lpr_log2;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t b = tid >> lpr_log2;
  const int lir = (int)(tid & (lpr - 1));
  if (b >= batch) return;
  const int64_t u = load_id(users, b), i = load_id(items, b);
  const bool uok = (uint64_t)u < (uint64_t)user_rows, iok = (uint64_t)i < (uint64_t)item_rows;
  // out_by_id (row-sharded host: the "tables" are the received rows and the ids their slots): the gradient of pair b goes to row
  // users[b] / items[b] of the outputs = straight into the padded send buffers of the exchange
  const int64_t ou = out_by_id ? (uok ? u : -1) : b, oi = out_by_id ? (iok ? i : -1) : b;
  const float g = ddot[b];
  const int uoff = item_first ? dim : 0, ioff = item_first ? 0 : dim;
  for (int c = lir; c < chunks; c += lpr) {
    V uf = uok ? vload<VEC>(user_mf + u * ldu + c * VEC) : vzero<VEC>();
    V vf = iok ? vload<VEC>(item_mf + i * ldi + c * VEC) : vzero<VEC>();
    if (ou >= 0) vstore<VEC>(g_user_mf + ou * ldg + c * VEC, vmul(vf, g));
    if (oi >= 0) vstore<VEC>(g_item_mf + oi * ldg + c * VEC, vmul(uf, g));
    if (g_user_mlp) {
      const float* xr = dx0 + b * (2 * (int64_t)dim);
      if (ou >= 0) vstore<VEC>(g_user_mlp + ou * ldg + c * VEC, vload<VEC>(xr + uoff + c * VEC));
      if (oi >= 0) vstore<VEC>(g_item_mlp + oi * ldg + c * VEC, vload<VEC>(xr + ioff + c * VEC));
    }
  }
}

// ---- BPR fused step --------------------------------------------------------------------------
template <typename IdT, int VEC>
__global__ __launch_bounds__(256) void bpr_fwd_bwd_kernel(
    const float* __restrict__ user_table, const float* __restrict__ item_table, int64_t user_rows,
    int64_t item_rows, const IdT* __restrict__ users, const IdT* __restrict__ pos,
    const IdT* __restrict__ neg, int dim, int chunks, int lpr_log2, int64_t batch, float inv_batch,
    float* __restrict__ per_triplet, double* __restrict__ loss_sum, float* __restrict__ g_user,
    float* __restrict__ g_item, int* err) {
  using V = typename VecT<VEC>::type;
  const int lpr = 1 << lpr_log2;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t b = tid >> lpr_log2;
  const int lir = (int)(tid & (lpr - 1));
  const bool live = b < batch;
  if (!live) b = batch - 1;
  const int64_t u = load_id(users, b), p = load_id(pos, b), n = load_id(neg, b);
  const bool uok = (uint64_t)u < (uint64_t)user_rows;
  const bool pok = (uint64_t)p < (uint64_t)item_rows, nok = (uint64_t)n < (uint64_t)item_rows;
  if (!(uok && pok && nok) && err && lir == 0) *err = 1;
  float sp = 0.f, sn = 0.f;
  // single-pass fast path keeps the three rows in registers (chunks <= lpr: dim <= 256)
  const bool one_pass = chunks <= lpr;
  V eu = vzero<VEC>(), ep = vzero<VEC>(), en = vzero<VEC>();
  if (one_pass) {
    if (lir < chunks) {
      eu = uok ? vload<VEC>(user_table + u * dim + lir * VEC) : vzero<VEC>();
      ep = pok ? vload<VEC>(item_table + p * dim + lir * VEC) : vzero<VEC>();
      en = nok ? vload<VEC>(item_table + n * dim + lir * VEC) : vzero<VEC>();
    }
    sp = vdot(eu, ep);
    sn = vdot(eu, en);
  } else {
    for (int c = lir; c < chunks; c += lpr) {
      V a = uok ? vload<VEC>(user_table + u * dim + c * VEC) : vzero<VEC>();
      V bp = pok ? vload<VEC>(item_table + p * dim + c * VEC) : vzero<VEC>();
      V bn = nok ? vload<VEC>(item_table + n * dim + c * VEC) : vzero<VEC>();
      sp += vdot(a, bp);
      sn += vdot(a, bn);
    }
  }
  sp = rowgroup_sum(sp, lpr);
  sn = rowgroup_sum(sn, lpr);
  const float x = sp - sn;
  const float s = sigmoidf_acc(x);
  const float l = 1.f - s;
  const float dx = -s * (1.f - s) * inv_batch;
  if (live) {
    if (per_triplet && lir == 0) per_triplet[b] = l;
    float* gu = g_user + b * dim;
    float* gp = g_item + b * dim;
    float* gn = g_item + (batch + b) * dim;
    if (one_pass) {
      if (lir < chunks) {
        vstore<VEC>(gu + lir * VEC, vmul(vsub(ep, en), dx));
        vstore<VEC>(gp + lir * VEC, vmul(eu, dx));
        vstore<VEC>(gn + lir * VEC, vmul(eu, -dx));
      }
    } else {
      for (int c = lir; c < chunks; c += lpr) {
        V a = uok ? vload<VEC>(user_table + u * dim + c * VEC) : vzero<VEC>();
        V bp = pok ? vload<VEC>(item_table + p * dim + c * VEC) : vzero<VEC>();
        V bn = nok ? vload<VEC>(item_table + n * dim + c * VEC) : vzero<VEC>();
        vstore<VEC>(gu + c * VEC, vmul(vsub(bp, bn), dx));
        vstore<VEC>(gp + c * VEC, vmul(a, dx));
        vstore<VEC>(gn + c * VEC, vmul(a, -dx));
      }
    }
  }
  // loss: one contribution per row group -> wave sum -> block sum -> one double atomic per block
  __shared__ double red[4];
  double contrib = (live && lir == 0) ? (double)l : 0.0;
  contrib = wave_sum_d(contrib);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) red[wave] = contrib;
  __syncthreads();
  if (threadIdx.x == 0 && loss_sum) atomicAdd(loss_sum + (blockIdx.x & (BR_SUM_SLOTS - 1)), red[0] + red[1] + red[2] + red[3]);
}

}  // namespace br

using namespace br;

static inline unsigned grid_for_rows(int64_t batch, int lpr_log2) {
  const int64_t rows_per_block = 256 >> lpr_log2;
  return (unsigned)ceil_div(batch, rows_per_block);
}

extern "C" int brGatherRows(int n_tables, const float* const* tables, const int64_t* table_rows,
                            const void* const* ids, float* const* outs, int dim, int64_t batch,
                            int id_type, int* err_flag, brStream stream) {
  BR_CHECK_ARG(n_tables >= 1 && n_tables <= BR_MAX_TABLES, "brGatherRows: n_tables %d out of [1,%d]", n_tables, BR_MAX_TABLES);
  BR_CHECK_ARG(dim >= 1 && batch >= 0, "brGatherRows: bad dim/batch");
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brGatherRows: bad id_type %d", id_type);
  if (batch == 0) return BR_OK;
  for (int t = 0; t < n_tables; ++t)
    BR_CHECK_ARG(tables[t] && outs[t] && table_rows[t] > 0, "brGatherRows: null table/out %d", t);
  const RowGeom g = row_geom(dim);
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = grid_for_rows(batch, g.lpr_log2);
  for (int t0 = 0; t0 < n_tables; t0 += 4) {
    const int nt = (n_tables - t0) < 4 ? (n_tables - t0) : 4;
    GatherArgs a{};
    for (int t = 0; t < nt; ++t) {
      a.tables[t] = tables[t0 + t];
      a.ids[t] = ids ? ids[t0 + t] : nullptr;
      a.outs[t] = outs[t0 + t];
      a.rows[t] = table_rows[t0 + t];
    }
#define LAUNCH_G(IdT, NT) \
  BR_DISPATCH_VEC(g.vec, (gather_rows_kernel<IdT, VEC, NT><<<grid, 256, 0, s>>>(a, dim, g.chunks, g.lpr_log2, batch, err_flag)))
#define LAUNCH_G_ID(NT)                                   \
  do {                                                    \
    if (id_type == BR_IDS_I32) LAUNCH_G(int32_t, NT);     \
    else LAUNCH_G(int64_t, NT);                           \
  } while (0)
    switch (nt) {
      case 1: LAUNCH_G_ID(1); break;
      case 2: LAUNCH_G_ID(2); break;
      case 3: LAUNCH_G_ID(3); break;
      default: LAUNCH_G_ID(4); break;
    }
    BR_CHECK_LAUNCH("brGatherRows");
  }
  return BR_OK;
}

extern "C" int brRowDot(const float* a, const float* b, float* out, int dim, int64_t batch, brStream stream) {
  BR_CHECK_ARG(a && b && out && dim >= 1 && batch >= 0, "brRowDot: bad args");
  if (batch == 0) return BR_OK;
  const RowGeom g = row_geom(dim);
  BR_DISPATCH_VEC(g.vec, (row_dot_kernel<VEC><<<grid_for_rows(batch, g.lpr_log2), 256, 0, (hipStream_t)stream>>>(
                             a, b, out, dim, g.chunks, g.lpr_log2, batch)));
  BR_CHECK_LAUNCH("brRowDot");
  return BR_OK;
}

extern "C" int brRowDotBackward(const float* a, const float* b, const float* dout, float* da, float* db,
                                int dim, int64_t batch, brStream stream) {
  BR_CHECK_ARG(a && b && dout && da && db && dim >= 1 && batch >= 0, "brRowDotBackward: bad args");
  if (batch == 0) return BR_OK;
  const RowGeom g = row_geom(dim);
  BR_DISPATCH_VEC(g.vec, (row_dot_bwd_kernel<VEC><<<grid_for_rows(batch, g.lpr_log2), 256, 0, (hipStream_t)stream>>>(
                             a, b, dout, da, db, dim, g.chunks, g.lpr_log2, batch)));
  BR_CHECK_LAUNCH("brRowDotBackward");
  return BR_OK;
}

extern "C" int brNeumfEmbedForward(const float* user_mlp, const float* item_mlp, const float* user_mf,
                                   const float* item_mf, int64_t ld_user, int64_t ld_item, int64_t user_rows,
                                   int64_t item_rows, const void* users, const void* items, int id_type, int dim,
                                   int64_t batch, int item_first, float* x0, float* dot, int* err_flag, brStream stream) {
  return brNeumfEmbedForwardStash(user_mlp, item_mlp, user_mf, item_mf, ld_user, ld_item, user_rows, item_rows, users, items, id_type, dim, batch, item_first, x0, dot,
                                  nullptr, nullptr, 0, err_flag, stream);
}

extern "C" int brNeumfEmbedForwardStash(const float* user_mlp, const float* item_mlp, const float* user_mf,
                                        const float* item_mf, int64_t ld_user, int64_t ld_item, int64_t user_rows,
                                        int64_t item_rows, const void* users, const void* items, int id_type, int dim,
                                        int64_t batch, int item_first, float* x0, float* dot, float* stash_user, float* stash_item, int64_t ld_stash,
                                        int* err_flag, brStream stream) {
  BR_CHECK_ARG(user_mlp && item_mlp && user_mf && item_mf && x0 && dot, "brNeumfEmbedForward: null pointer");
  BR_CHECK_ARG((stash_user == nullptr) == (stash_item == nullptr) && (!stash_user || ld_stash >= dim), "brNeumfEmbedForwardStash: both stashes (ld >= dim) or neither");
  BR_CHECK_ARG(dim >= 1 && batch >= 0 && user_rows > 0 && item_rows > 0, "brNeumfEmbedForward: bad sizes");
  BR_CHECK_ARG(ld_user >= dim && ld_item >= dim, "brNeumfEmbedForward: row strides < dim");
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brNeumfEmbedForward: bad id_type");
  if (batch == 0) return BR_OK;
  const bool st4 = !stash_user || (ld_stash % 4 == 0 && ((reinterpret_cast<uintptr_t>(stash_user) | reinterpret_cast<uintptr_t>(stash_item)) & 15) == 0);
  const bool st2 = !stash_user || (ld_stash % 2 == 0 && ((reinterpret_cast<uintptr_t>(stash_user) | reinterpret_cast<uintptr_t>(stash_item)) & 7) == 0);
  const RowGeom g = row_geom_ld(dim, (ld_user % 4 == 0 && ld_item % 4 == 0 && st4) ? 4 : (ld_user % 2 == 0 && ld_item % 2 == 0 && st2) ? 2 : 1);
  const unsigned grid = grid_for_rows(batch, g.lpr_log2);
  hipStream_t s = (hipStream_t)stream;
  if (id_type == BR_IDS_I32) {
    BR_DISPATCH_VEC(g.vec, (neumf_embed_fwd_kernel<int32_t, VEC><<<grid, 256, 0, s>>>(
                               user_mlp, item_mlp, user_mf, item_mf, ld_user, ld_item, user_rows, item_rows, (const int32_t*)users,
                               (const int32_t*)items, dim, g.chunks, g.lpr_log2, batch, item_first, x0, dot, err_flag, stash_user, stash_item, ld_stash)));
  } else {
    BR_DISPATCH_VEC(g.vec, (neumf_embed_fwd_kernel<int64_t, VEC><<<grid, 256, 0, s>>>(
                               user_mlp, item_mlp, user_mf, item_mf, ld_user, ld_item, user_rows, item_rows, (const int64_t*)users,
                               (const int64_t*)items, dim, g.chunks, g.lpr_log2, batch, item_first, x0, dot, err_flag, stash_user, stash_item, ld_stash)));
  }
  BR_CHECK_LAUNCH("brNeumfEmbedForward");
  return BR_OK;
}

extern "C" int brNeumfEmbedForwardDeferred(const float* user_tab, const float* user_m, const float* user_v, const int32_t* user_last,
                                           const float* item_tab, const float* item_m, const float* item_v, const int32_t* item_last,
                                           int64_t user_rows, int64_t item_rows, const void* users, const void* items, int id_type,
                                           int dim, int64_t batch, int item_first, const void* step_state, double beta1, double beta2,
                                           double eps, float* x0, float* dot, float* stash_user, float* stash_item, int64_t ld_stash,
                                           int* err_flag, brStream stream) {
  BR_CHECK_ARG(user_tab && user_m && user_v && user_last && item_tab && item_m && item_v && item_last && step_state && x0 && dot &&
                   stash_user && stash_item, "brNeumfEmbedForwardDeferred: null pointer");
  BR_CHECK_ARG(dim >= 1 && batch >= 0 && user_rows > 0 && item_rows > 0 && ld_stash >= dim, "brNeumfEmbedForwardDeferred: bad sizes");
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brNeumfEmbedForwardDeferred: bad id_type");
  if (batch == 0) return BR_OK;
  const RowGeom g = row_geom_ld(dim, ld_stash % 4 == 0 ? 4 : ld_stash % 2 == 0 ? 2 : 1);
  const unsigned grid = grid_for_rows(batch, g.lpr_log2);
  const AdamHp h = make_hp(0.0, beta1, beta2, eps);
  const StepStateDev* ss = (const StepStateDev*)step_state;
  hipStream_t s = (hipStream_t)stream;
  static const bool wave_rows = [] { const char* e = getenv("BR_WAVE_ROWS"); return !(e && e[0] == '0'); }();
  const int wvec = dim / 32;
  if (wave_rows && dim % 32 == 0 && (wvec == 2 || wvec == 4) && ld_stash % wvec == 0 &&
      ((reinterpret_cast<uintptr_t>(stash_user) | reinterpret_cast<uintptr_t>(stash_item) | reinterpret_cast<uintptr_t>(x0)) & (4 * wvec - 1)) == 0) {
    const unsigned wgrid = (unsigned)ceil_div(batch, 4);
    const LookupArgs la{user_tab, user_m, user_v, user_last, item_tab, item_m, item_v, item_last, user_rows, item_rows, users, items, batch, item_first, ss, h,
                        x0, dot, stash_user, stash_item, ld_stash, err_flag};
    if (id_type == BR_IDS_I32) { if (wvec == 2) neumf_embed_fwd_deferred_wave_kernel<int32_t, 2><<<wgrid, 256, 0, s>>>(la); else neumf_embed_fwd_deferred_wave_kernel<int32_t, 4><<<wgrid, 256, 0, s>>>(la); }
    else { if (wvec == 2) neumf_embed_fwd_deferred_wave_kernel<int64_t, 2><<<wgrid, 256, 0, s>>>(la); else neumf_embed_fwd_deferred_wave_kernel<int64_t, 4><<<wgrid, 256, 0, s>>>(la); }
    BR_CHECK_LAUNCH("brNeumfEmbedForwardDeferred(wave)");
    return BR_OK;
  }
  if (id_type == BR_IDS_I32) {
    BR_DISPATCH_VEC(g.vec, (neumf_embed_fwd_deferred_kernel<int32_t, VEC><<<grid, 256, 0, s>>>(
                               user_tab, user_m, user_v, user_last, item_tab, item_m, item_v, item_last, user_rows, item_rows,
                               (const int32_t*)users, (const int32_t*)items, dim, g.chunks, g.lpr_log2, batch, item_first, ss, h, x0, dot,
                               stash_user, stash_item, ld_stash, err_flag)));
  } else {
    BR_DISPATCH_VEC(g.vec, (neumf_embed_fwd_deferred_kernel<int64_t, VEC><<<grid, 256, 0, s>>>(
                               user_tab, user_m, user_v, user_last, item_tab, item_m, item_v, item_last, user_rows, item_rows,
                               (const int64_t*)users, (const int64_t*)items, dim, g.chunks, g.lpr_log2, batch, item_first, ss, h, x0, dot,
                               stash_user, stash_item, ld_stash, err_flag)));
  }
  BR_CHECK_LAUNCH("brNeumfEmbedForwardDeferred");
  return BR_OK;
}

extern "C" int brGatherRowsDeferred(const float* table, const float* m, const float* v, const int32_t* last, int64_t table_rows, int dim,
                                    const void* ids, int id_type, int64_t n, const void* step_state, double beta1, double beta2,
                                    double eps, float* out, int64_t ld_out, int* err_flag, brStream stream) {
  BR_CHECK_ARG(table && m && v && last && step_state && out && dim >= 1 && table_rows > 0 && n >= 0 && ld_out >= dim, "brGatherRowsDeferred: bad args");
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brGatherRowsDeferred: bad id_type");
  if (n == 0) return BR_OK;
  BR_CHECK_ARG(ids != nullptr, "brGatherRowsDeferred: null ids");
  const RowGeom g = row_geom_ld(dim, ld_out);
  const unsigned grid = grid_for_rows(n, g.lpr_log2);
  const AdamHp h = make_hp(0.0, beta1, beta2, eps);
  const StepStateDev* ss = (const StepStateDev*)step_state;
  hipStream_t s = (hipStream_t)stream;
  static const bool wave_rows = [] { const char* e = getenv("BR_WAVE_ROWS"); return !(e && e[0] == '0'); }();
  const int wvec = dim / 64;
  if (wave_rows && dim % 64 == 0 && (wvec == 1 || wvec == 2 || wvec == 4) && ld_out % wvec == 0 && (reinterpret_cast<uintptr_t>(out) & (4 * wvec - 1)) == 0) {
    const unsigned wgrid = (unsigned)ceil_div(n, 4);
    if (id_type == BR_IDS_I32)
      BR_DISPATCH_VEC(wvec, (gather_rows_deferred_wave_kernel<int32_t, VEC><<<wgrid, 256, 0, s>>>(table, m, v, last, table_rows, (const int32_t*)ids, n, ss, h,
                                                                                                   out, ld_out, err_flag)));
    else
      BR_DISPATCH_VEC(wvec, (gather_rows_deferred_wave_kernel<int64_t, VEC><<<wgrid, 256, 0, s>>>(table, m, v, last, table_rows, (const int64_t*)ids, n, ss, h,
                                                                                                   out, ld_out, err_flag)));
    BR_CHECK_LAUNCH("brGatherRowsDeferred(wave)");
    return BR_OK;
  }
  if (id_type == BR_IDS_I32)
    BR_DISPATCH_VEC(g.vec, (gather_rows_deferred_kernel<int32_t, VEC><<<grid, 256, 0, s>>>(table, m, v, last, table_rows, dim, g.chunks, g.lpr_log2,
                                                                                           (const int32_t*)ids, n, ss, h, out, ld_out, err_flag)));
  else
    BR_DISPATCH_VEC(g.vec, (gather_rows_deferred_kernel<int64_t, VEC><<<grid, 256, 0, s>>>(table, m, v, last, table_rows, dim, g.chunks, g.lpr_log2,
                                                                                           (const int64_t*)ids, n, ss, h, out, ld_out, err_flag)));
  BR_CHECK_LAUNCH("brGatherRowsDeferred");
  return BR_OK;
}

extern "C" int brGatherRowsDeferredPair(const float* table_a, const float* m_a, const float* v_a, const int32_t* last_a, int64_t rows_a, const void* ids_a,
                                        float* out_a, const float* table_b, const float* m_b, const float* v_b, const int32_t* last_b, int64_t rows_b,
                                        const void* ids_b, float* out_b, int dim, int id_type, int64_t n, int64_t n_b, const void* step_state, double beta1,
                                        double beta2, double eps, int64_t ld_out, int* err_flag, brStream stream) {
  if (n_b == 0) n_b = n;
  BR_CHECK_ARG(table_a && m_a && v_a && last_a && ids_a && out_a && table_b && m_b && v_b && last_b && ids_b && out_b && step_state && rows_a > 0 && rows_b > 0 &&
                   n >= 0 && n_b >= 0 && ld_out >= dim, "brGatherRowsDeferredPair: bad args");
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brGatherRowsDeferredPair: bad id_type");
  if (n == 0 && n_b == 0) return BR_OK;
  const int wvec = dim / 64;
  const bool wave_ok = dim % 64 == 0 && (wvec == 1 || wvec == 2 || wvec == 4) && ld_out % wvec == 0 &&
                       ((reinterpret_cast<uintptr_t>(out_a) | reinterpret_cast<uintptr_t>(out_b)) & (4 * wvec - 1)) == 0;
  if (!wave_ok) {      // other row widths: the two single-table launches
    const int rc = brGatherRowsDeferred(table_a, m_a, v_a, last_a, rows_a, dim, ids_a, id_type, n, step_state, beta1, beta2, eps, out_a, ld_out, err_flag, stream);
    if (rc != BR_OK) return rc;
    return brGatherRowsDeferred(table_b, m_b, v_b, last_b, rows_b, dim, ids_b, id_type, n_b, step_state, beta1, beta2, eps, out_b, ld_out, err_flag, stream);
  }
  GatherDefJobs J;
  J.j[0] = GatherDefJob{table_a, m_a, v_a, last_a, rows_a, ids_a, out_a, n};
  J.j[1] = GatherDefJob{table_b, m_b, v_b, last_b, rows_b, ids_b, out_b, n_b};
  const AdamHp h = make_hp(0.0, beta1, beta2, eps);
  const StepStateDev* ss = (const StepStateDev*)step_state;
  const dim3 grid((unsigned)ceil_div(n > n_b ? n : n_b, 4), 2);
  hipStream_t s = (hipStream_t)stream;
  if (id_type == BR_IDS_I32)
    BR_DISPATCH_VEC(wvec, (gather_rows_deferred_wave_pair_kernel<int32_t, VEC><<<grid, 256, 0, s>>>(J, ss, h, ld_out, err_flag)));
  else
    BR_DISPATCH_VEC(wvec, (gather_rows_deferred_wave_pair_kernel<int64_t, VEC><<<grid, 256, 0, s>>>(J, ss, h, ld_out, err_flag)));
  BR_CHECK_LAUNCH("brGatherRowsDeferredPair");
  return BR_OK;
}

extern "C" int brGatherRowsDeferredPairSeg(const float* table_a, const float* m_a, const float* v_a, const int32_t* last_a, int64_t rows_a, const float* table_b,
                                           const float* m_b, const float* v_b, const int32_t* last_b, int64_t rows_b, const void* ids, float* out, int dim,
                                           int id_type, int64_t n, int64_t seg_len, int64_t seg_stride, int64_t seg_off_a, int64_t seg_off_b,
                                           const void* step_state, double beta1, double beta2, double eps, int64_t ld_out, int* err_flag, brStream stream) {
  BR_CHECK_ARG(table_a && m_a && v_a && last_a && table_b && m_b && v_b && last_b && ids && out && step_state && rows_a > 0 && rows_b > 0 && n >= 0 && ld_out >= dim,
               "brGatherRowsDeferredPairSeg: bad args");
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brGatherRowsDeferredPairSeg: bad id_type");
  BR_CHECK_ARG(seg_len >= 1 && seg_stride >= seg_len && seg_off_a >= 0 && seg_off_b >= 0, "brGatherRowsDeferredPairSeg: bad segment geometry");
  if (n == 0) return BR_OK;
  const int wvec = dim / 64;
  BR_CHECK_ARG(dim % 64 == 0 && (wvec == 1 || wvec == 2 || wvec == 4) && ld_out % wvec == 0 && (reinterpret_cast<uintptr_t>(out) & (4 * wvec - 1)) == 0,
               "brGatherRowsDeferredPairSeg: rows of 64 / 128 / 256 floats (one wave per row)");
  GatherDefJobs J;
  J.j[0] = GatherDefJob{table_a, m_a, v_a, last_a, rows_a, ids, out, n, seg_off_a};
  J.j[1] = GatherDefJob{table_b, m_b, v_b, last_b, rows_b, ids, out, n, seg_off_b};
  J.seg_len = seg_len; J.seg_stride = seg_stride;
  const AdamHp h = make_hp(0.0, beta1, beta2, eps);
  const StepStateDev* ss = (const StepStateDev*)step_state;
  const dim3 grid((unsigned)ceil_div(n, 4), 2);
  hipStream_t s = (hipStream_t)stream;
  if (id_type == BR_IDS_I32)
    BR_DISPATCH_VEC(wvec, (gather_rows_deferred_wave_pair_kernel<int32_t, VEC><<<grid, 256, 0, s>>>(J, ss, h, ld_out, err_flag)));
  else
    BR_DISPATCH_VEC(wvec, (gather_rows_deferred_wave_pair_kernel<int64_t, VEC><<<grid, 256, 0, s>>>(J, ss, h, ld_out, err_flag)));
  BR_CHECK_LAUNCH("brGatherRowsDeferredPairSeg");
  return BR_OK;
}

extern "C" int brGatherRowsPairSeg(const float* table_a, int64_t rows_a, const float* table_b, int64_t rows_b, const void* ids, float* out, int dim, int id_type,
                                   int64_t n, int64_t seg_len, int64_t seg_stride, int64_t seg_off_a, int64_t seg_off_b, int64_t ld_out, int* err_flag,
                                   brStream stream) {
  BR_CHECK_ARG(table_a && table_b && ids && out && rows_a > 0 && rows_b > 0 && n >= 0 && dim >= 1 && ld_out >= dim, "brGatherRowsPairSeg: bad args");
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brGatherRowsPairSeg: bad id_type");
  BR_CHECK_ARG(seg_len >= 1 && seg_stride >= seg_len && seg_off_a >= 0 && seg_off_b >= 0, "brGatherRowsPairSeg: bad segment geometry");
  if (n == 0) return BR_OK;
  const RowGeom g = row_geom_ld(dim, (ld_out % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) ? 4 : (ld_out % 2 == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0) ? 2 : 1);
  GatherSegJobs J;
  J.j[0] = GatherSegJob{table_a, rows_a, ids, out, seg_off_a};
  J.j[1] = GatherSegJob{table_b, rows_b, ids, out, seg_off_b};
  const dim3 grid(grid_for_rows(n, g.lpr_log2), 2);
  hipStream_t s = (hipStream_t)stream;
  if (id_type == BR_IDS_I32)
    BR_DISPATCH_VEC(g.vec, (gather_rows_seg_kernel<int32_t, VEC><<<grid, 256, 0, s>>>(J, dim, g.chunks, g.lpr_log2, n, seg_len, seg_stride, ld_out, err_flag)));
  else
    BR_DISPATCH_VEC(g.vec, (gather_rows_seg_kernel<int64_t, VEC><<<grid, 256, 0, s>>>(J, dim, g.chunks, g.lpr_log2, n, seg_len, seg_stride, ld_out, err_flag)));
  BR_CHECK_LAUNCH("brGatherRowsPairSeg");
  return BR_OK;
}

extern "C" int brMfGradInplace(float* stash_user, float* stash_item, int64_t ld, const float* ddot, int64_t batch, int dim, brStream stream) {
  BR_CHECK_ARG(stash_user && stash_item && ddot && dim >= 1 && ld >= dim && batch >= 0, "brMfGradInplace: bad args");
  if (batch == 0) return BR_OK;
  const RowGeom g = row_geom_ld(dim, ld % 4 == 0 ? 4 : ld % 2 == 0 ? 2 : 1);
  const unsigned grid = (unsigned)ceil_div(batch * g.chunks, 256);
  BR_DISPATCH_VEC(g.vec, (mf_grad_inplace_kernel<VEC><<<grid, 256, 0, (hipStream_t)stream>>>(stash_user, stash_item, ld, ddot, batch, g.chunks)));
  BR_CHECK_LAUNCH("brMfGradInplace");
  return BR_OK;
}

extern "C" int brNeumfEmbedBackward(const float* user_mf, const float* item_mf, int64_t ld_user, int64_t ld_item,
                                    int64_t user_rows, int64_t item_rows, const void* users, const void* items,
                                    int id_type, int dim, int64_t batch, int item_first, const float* dx0,
                                    const float* ddot, float* g_user_mlp, float* g_item_mlp, float* g_user_mf,
                                    float* g_item_mf, int64_t ldg, int out_rows_by_id, brStream stream) {
  BR_CHECK_ARG(user_mf && item_mf && ddot && g_user_mf && g_item_mf, "brNeumfEmbedBackward: null pointer");
  BR_CHECK_ARG((g_user_mlp == nullptr) == (g_item_mlp == nullptr), "brNeumfEmbedBackward: g_*_mlp both or neither");
  BR_CHECK_ARG(!g_user_mlp || dx0, "brNeumfEmbedBackward: dx0 required for g_*_mlp");
  BR_CHECK_ARG(ld_user >= dim && ld_item >= dim && ldg >= dim, "brNeumfEmbedBackward: row strides < dim");
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brNeumfEmbedBackward: bad id_type");
  if (batch == 0) return BR_OK;
  const int64_t ldmin = (ld_user % 4 == 0 && ld_item % 4 == 0 && ldg % 4 == 0) ? 4 : (ld_user % 2 == 0 && ld_item % 2 == 0 && ldg % 2 == 0) ? 2 : 1;
  const RowGeom g = row_geom_ld(dim, ldmin);
  const unsigned grid = grid_for_rows(batch, g.lpr_log2);
  hipStream_t s = (hipStream_t)stream;
  if (id_type == BR_IDS_I32) {
    BR_DISPATCH_VEC(g.vec, (neumf_embed_bwd_kernel<int32_t, VEC><<<grid, 256, 0, s>>>(
                               user_mf, item_mf, ld_user, ld_item, user_rows, item_rows, (const int32_t*)users, (const int32_t*)items,
                               dim, g.chunks, g.lpr_log2, batch, item_first, dx0, ddot, g_user_mlp, g_item_mlp, g_user_mf, g_item_mf, ldg, out_rows_by_id)));
  } else {
    BR_DISPATCH_VEC(g.vec, (neumf_embed_bwd_kernel<int64_t, VEC><<<grid, 256, 0, s>>>(
                               user_mf, item_mf, ld_user, ld_item, user_rows, item_rows, (const int64_t*)users, (const int64_t*)items,
                               dim, g.chunks, g.lpr_log2, batch, item_first, dx0, ddot, g_user_mlp, g_item_mlp, g_user_mf, g_item_mf, ldg, out_rows_by_id)));
  }
  BR_CHECK_LAUNCH("brNeumfEmbedBackward");
  return BR_OK;
}

extern "C" int brBprForwardBackward(const float* user_table, const float* item_table, int64_t user_rows,
                                    int64_t item_rows, const void* users, const void* pos, const void* neg,
                                    int id_type, int dim, int64_t batch, float inv_batch, float* per_triplet,
                                    double* loss_sum, float* g_user, float* g_item, int* err_flag, brStream stream) {
  BR_CHECK_ARG(user_table && item_table && users && pos && neg && g_user && g_item, "brBprForwardBackward: null pointer");
  BR_CHECK_ARG(dim >= 1 && batch >= 0 && user_rows > 0 && item_rows > 0, "brBprForwardBackward: bad sizes");
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brBprForwardBackward: bad id_type");
  if (batch == 0) return BR_OK;
  const RowGeom g = row_geom(dim);
  const unsigned grid = grid_for_rows(batch, g.lpr_log2);
  hipStream_t s = (hipStream_t)stream;
  if (id_type == BR_IDS_I32) {
    BR_DISPATCH_VEC(g.vec, (bpr_fwd_bwd_kernel<int32_t, VEC><<<grid, 256, 0, s>>>(
                               user_table, item_table, user_rows, item_rows, (const int32_t*)users, (const int32_t*)pos,
                               (const int32_t*)neg, dim, g.chunks, g.lpr_log2, batch, inv_batch, per_triplet, loss_sum,
                               g_user, g_item, err_flag)));
  } else {
    BR_DISPATCH_VEC(g.vec, (bpr_fwd_bwd_kernel<int64_t, VEC><<<grid, 256, 0, s>>>(
                               user_table, item_table, user_rows, item_rows, (const int64_t*)users, (const int64_t*)pos,
                               (const int64_t*)neg, dim, g.chunks, g.lpr_log2, batch, inv_batch, per_triplet, loss_sum,
                               g_user, g_item, err_flag)));
  }
  BR_CHECK_LAUNCH("brBprForwardBackward");
  return BR_OK;
}
