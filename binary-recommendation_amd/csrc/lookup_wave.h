// One pair of the deferred NeuMF lookup by one wave (gather.hip neumf_embed_fwd_deferred_wave_kernel; also the body of the lookup
// workgroups of the fused lookup + chunk-sort launch in sparse_opt.hip).
#pragma once
#include "common.h"
#include "rows.h"
#include "adam_math.h"
#include <stdlib.h>

namespace br {

struct LookupArgs {
  const float *user_tab, *user_m, *user_v; const int32_t* user_last;
  const float *item_tab, *item_m, *item_v; const int32_t* item_last;
  int64_t user_rows, item_rows;
  const void *users, *items;
  int64_t batch; int item_first;
  const StepStateDev* ss; AdamHp h;
  float *x0, *dot, *stash_user, *stash_item; int64_t ld_stash; int* err;
  uint32_t step_add = 0;      // 1: the step state is advanced BEHIND this launch (by the chunk-rank launch): the step being computed is ss->step + 1
  // bit 0 / bit 1: request the user / item row's m and v TOGETHER with theta instead of behind last[] (lookup_spec()): a row that
  // turns out to need no replay wasted 2 row reads, every other row saved one of its three dependent round trips
  uint32_t spec = 0;
};
// which streams of the deferred lookup load m / v speculatively: BR_LOOKUP_SPEC = 0 (default) | 1 (users) | 2 (items) | 3 (both).  Measured at
// config 2 (94 % of the user rows lag): 69-70 us with and without - like two pairs per wave (BR_LOOKUP_PAIRS=2), it does not move the launch:
// neither the dependent round trips nor the bytes in flight per wave bound it.
static inline uint32_t lookup_spec(int64_t, int64_t, int64_t) {
  static const int forced = [] { const char* e = getenv("BR_LOOKUP_SPEC"); return e ? atoi(e) : 0; }();
  return (uint32_t)forced & 3u;
}

// pair b (wave-uniform) of embed_dim = 32 * VEC: a lane owns VEC columns of the fused user row and the same columns of the item row
template <typename IdT, int VEC>
__device__ __forceinline__ void lookup_wave_pair(const LookupArgs& a, int64_t b, int lane) {
  using V = typename VecT<VEC>::type;
  constexpr int dim = 32 * VEC;
  constexpr int64_t ld = 2 * dim;
  const float* __restrict__ user_tab = a.user_tab; const float* __restrict__ item_tab = a.item_tab;
  const StepStateDev* __restrict__ ss = a.ss;
  const int col = lane * VEC;                         // column of the fused row: [0, dim) MLP, [dim, 2 dim) MF
  int64_t u = load_id((const IdT*)a.users, b), i = load_id((const IdT*)a.items, b);
  const bool uok = (uint64_t)u < (uint64_t)a.user_rows, iok = (uint64_t)i < (uint64_t)a.item_rows;
  if ((!uok || !iok) && a.err && lane == 0) *a.err = 1;
  if (!uok) u = 0;
  if (!iok) i = 0;
  const uint32_t t = ss->step + a.step_add;           // the step being computed: rows must include steps <= t-1
  const uint32_t lu = (uint32_t)a.user_last[u], li = (uint32_t)a.item_last[i];
  const int64_t uo = u * ld + col, io = i * ld + col;
  V ur = vload<VEC>(user_tab + uo), ir = vload<VEC>(item_tab + io);
  V um = vzero<VEC>(), uv = vzero<VEC>(), im = vzero<VEC>(), iv = vzero<VEC>();
  if ((a.spec & 1u) || lu + 1 < t) { um = vload<VEC>(a.user_m + uo); uv = vload<VEC>(a.user_v + uo); }
  if ((a.spec & 2u) || li + 1 < t) { im = vload<VEC>(a.item_m + io); iv = vload<VEC>(a.item_v + io); }
  if (li + 1 < t) adam_catch_up_uniform<false>(ir, im, iv, li, t - 1, ss, a.h);
  if (lu + 1 < t) adam_catch_up_uniform<false>(ur, um, uv, lu, t - 1, ss, a.h);
  if (!uok) ur = vzero<VEC>();
  if (!iok) ir = vzero<VEC>();
  const bool mlp = lane < 32;
  const int uoff = a.item_first ? dim : 0, ioff = a.item_first ? 0 : dim;
  vstore<VEC>(mlp ? a.x0 + b * ld + uoff + col : a.stash_user + b * a.ld_stash + (col - dim), ur);
  vstore<VEC>(mlp ? a.x0 + b * ld + ioff + col : a.stash_item + b * a.ld_stash + (col - dim), ir);
  // GMF dot of the MF halves in the row-group form's order: partial of four columns by a chain of fused multiply-adds, then the xor tree
  float s;
  if constexpr (VEC == 4) {
    s = 0.f + vdot(ur, ir);
    s += __shfl_xor(s, 16, 64);
  } else {
    static_assert(VEC == 2, "wave lookup: embed_dim 64 or 128");
    const float e = __builtin_fmaf(ur.y, ir.y, ur.x * ir.x);
    const float prev = __shfl_up(e, 1, 64);          // odd lanes: the first two columns of their group of four
    s = 0.f + __builtin_fmaf(ur.y, ir.y, __builtin_fmaf(ur.x, ir.x, prev));
    s += __shfl_xor(s, 16, 64);
  }
  s += __shfl_xor(s, 8, 64);
  s += __shfl_xor(s, 4, 64);
  s += __shfl_xor(s, 2, 64);
  if constexpr (VEC == 4) s += __shfl_xor(s, 1, 64);
  if (lane == 63) a.dot[b] = s;
}

// embed_dim 64, fast replay: the pair's two fused rows (128 floats each) side by side in ONE wave - lanes 0-31 the user row, lanes 32-63 the
// item row, 16 B per lane - so theta, m and v are one load instruction each for BOTH rows (1 KB per instruction; the form above moves
// 512 B per instruction and needs twice as many), and the two replays run together: max(lag_u, lag_i) iterations instead of their sum, each
// half with its own alpha (the ids and last[] of a pair are wave-uniform, so both alphas are scalar loads; a half whose lag is used up
// gets alpha = 0, which leaves theta as it is).  Same fp32 operations per element as lookup_wave_pair in the same order: same bits
// (x0, MF stash and dot: the dot's lanes hold four columns each, as the row-group kernels' do).
template <typename IdT>
__device__ __forceinline__ void lookup_half_pair(const LookupArgs& a, int64_t b, int lane) {
  constexpr int dim = 64;
  constexpr int64_t ld = 2 * dim;
  const StepStateDev* __restrict__ ss = a.ss;
  const bool it = lane >= 32;                          // this lane's row: the item's
  const int col = (lane & 31) * 4;                     // column of the fused row: [0, dim) MLP, [dim, 2 dim) MF
  int64_t u = load_id((const IdT*)a.users, b), i = load_id((const IdT*)a.items, b);
  const bool uok = (uint64_t)u < (uint64_t)a.user_rows, iok = (uint64_t)i < (uint64_t)a.item_rows;
  if ((!uok || !iok) && a.err && lane == 0) *a.err = 1;
  if (!uok) u = 0;
  if (!iok) i = 0;
  const uint32_t t = ss->step + a.step_add;
  const uint32_t lu = (uint32_t)a.user_last[u], li = (uint32_t)a.item_last[i];
  const FastRp f = fast_rp(ss);
  uint32_t lag_u = lu + 1 < t ? t - 1 - lu : 0u, lag_i = li + 1 < t ? t - 1 - li : 0u;
  const int64_t off = (it ? i : u) * ld + col;
  const float4 th = vload<4>((it ? a.item_tab : a.user_tab) + off);
  float4 m = vzero<4>(), v = vzero<4>();
  if (it ? lag_i > 0 : lag_u > 0) { m = vload<4>((it ? a.item_m : a.user_m) + off); v = vload<4>((it ? a.item_v : a.user_v) + off); }
  float4 out = th;
  lag_u = lag_u < f.trunc ? lag_u : f.trunc; lag_i = lag_i < f.trunc ? lag_i : f.trunc;      // theta is replayed over min(lag, trunc) steps
  const uint32_t steps = lag_u > lag_i ? lag_u : lag_i, mine = it ? lag_i : lag_u;
  if (steps > 0 && __builtin_amdgcn_ballot_w64(!(all_zero(m) && all_zero(v))) != 0) {
    FastSt<float4> st;
    st.open(th, m, v, a.h.eps);
    uint32_t k = 1;
    for (; k + 7 <= steps; k += 8) {      // eight alphas of each row per scalar load (the ring's first eight entries are mirrored behind its end)
      const float* __restrict__ ru = ss->alpha_hist + ((lu + k) & (BR_ALPHA_RING - 1));
      const float* __restrict__ ri = ss->alpha_hist + ((li + k) & (BR_ALPHA_RING - 1));
      float au[8], ai[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { au[q] = ru[q]; ai[q] = ri[q]; }
#pragma unroll
      for (int q = 0; q < 8; ++q) st.step(k + q <= mine ? (it ? ai[q] : au[q]) : 0.f, a.h.b1, f);
    }
    for (; k <= steps; ++k) {
      const float au = ss->alpha_hist[(lu + k) & (BR_ALPHA_RING - 1)], ai = ss->alpha_hist[(li + k) & (BR_ALPHA_RING - 1)];
      st.step(k <= mine ? (it ? ai : au) : 0.f, a.h.b1, f);
    }
    out = st.theta();
  }
  if (it ? !iok : !uok) out = vzero<4>();
  const bool mlp = col < dim;
  const int uoff = a.item_first ? dim : 0, ioff = a.item_first ? 0 : dim;
  float* dst = mlp ? a.x0 + b * ld + (it ? ioff : uoff) + col : (it ? a.stash_item : a.stash_user) + b * a.ld_stash + (col - dim);
  vstore<4>(dst, out);
  // GMF dot of the MF halves (lanes 16-31: user, 48-63: item): four columns per lane by vdot's chain, then the xor tree over the 16 lanes
  float4 other;
  other.x = __shfl_xor(out.x, 32, 64); other.y = __shfl_xor(out.y, 32, 64); other.z = __shfl_xor(out.z, 32, 64); other.w = __shfl_xor(out.w, 32, 64);
  float sdot = 0.f + (it ? vdot(other, out) : vdot(out, other));       // (user, item) operand order of the other kernels
  sdot += __shfl_xor(sdot, 8, 64);
  sdot += __shfl_xor(sdot, 4, 64);
  sdot += __shfl_xor(sdot, 2, 64);
  sdot += __shfl_xor(sdot, 1, 64);
  if (lane == 63) a.dot[b] = sdot;
}

// R consecutive pairs by one wave: every id, then every last[], then every row load of the 2 R rows is requested before the first replay
// (see gather_deferred_wave_rows below for why).  Pairs past the batch are skipped.  Same arithmetic per pair as lookup_wave_pair.
template <typename IdT, int VEC, int R>
__device__ __forceinline__ void lookup_wave_pairs(const LookupArgs& a, int64_t b0, int lane) {
  using V = typename VecT<VEC>::type;
  constexpr int dim = 32 * VEC;
  constexpr int64_t ld = 2 * dim;
  const float* __restrict__ user_tab = a.user_tab; const float* __restrict__ item_tab = a.item_tab;
  const StepStateDev* __restrict__ ss = a.ss;
  const int col = lane * VEC;
  const uint32_t t = ss->step + a.step_add;
  int64_t u[R], i[R], bq[R];
  bool live[R], uok[R], iok[R];
#pragma unroll
  for (int q = 0; q < R; ++q) {
    live[q] = b0 + q < a.batch;
    bq[q] = live[q] ? b0 + q : a.batch - 1;
    u[q] = load_id((const IdT*)a.users, bq[q]); i[q] = load_id((const IdT*)a.items, bq[q]);
  }
  uint32_t lu[R], li[R];
#pragma unroll
  for (int q = 0; q < R; ++q) {
    uok[q] = (uint64_t)u[q] < (uint64_t)a.user_rows; iok[q] = (uint64_t)i[q] < (uint64_t)a.item_rows;
    if ((!uok[q] || !iok[q]) && a.err && lane == 0 && live[q]) *a.err = 1;
    if (!uok[q]) u[q] = 0;
    if (!iok[q]) i[q] = 0;
    lu[q] = (uint32_t)a.user_last[u[q]]; li[q] = (uint32_t)a.item_last[i[q]];
  }
  V ur[R], ir[R], um[R], uv[R], im[R], iv[R];
#pragma unroll
  for (int q = 0; q < R; ++q) {
    const int64_t uo = u[q] * ld + col, io = i[q] * ld + col;
    ur[q] = vload<VEC>(user_tab + uo); ir[q] = vload<VEC>(item_tab + io);
    um[q] = vzero<VEC>(); uv[q] = vzero<VEC>(); im[q] = vzero<VEC>(); iv[q] = vzero<VEC>();
    if (lu[q] + 1 < t) { um[q] = vload<VEC>(a.user_m + uo); uv[q] = vload<VEC>(a.user_v + uo); }
    if (li[q] + 1 < t) { im[q] = vload<VEC>(a.item_m + io); iv[q] = vload<VEC>(a.item_v + io); }
  }
  const bool mlp = lane < 32;
  const int uoff = a.item_first ? dim : 0, ioff = a.item_first ? 0 : dim;
#pragma unroll
  for (int q = 0; q < R; ++q) {
    if (li[q] + 1 < t) adam_catch_up_uniform<false>(ir[q], im[q], iv[q], li[q], t - 1, ss, a.h);
    if (lu[q] + 1 < t) adam_catch_up_uniform<false>(ur[q], um[q], uv[q], lu[q], t - 1, ss, a.h);
    if (!uok[q]) ur[q] = vzero<VEC>();
    if (!iok[q]) ir[q] = vzero<VEC>();
    if (!live[q]) continue;
    const int64_t b = bq[q];
    vstore<VEC>(mlp ? a.x0 + b * ld + uoff + col : a.stash_user + b * a.ld_stash + (col - dim), ur[q]);
    vstore<VEC>(mlp ? a.x0 + b * ld + ioff + col : a.stash_item + b * a.ld_stash + (col - dim), ir[q]);
    float s;
    if constexpr (VEC == 4) {
      s = 0.f + vdot(ur[q], ir[q]);
      s += __shfl_xor(s, 16, 64);
    } else {
      static_assert(VEC == 2 || VEC == 4, "wave lookup: embed_dim 64 or 128");
      const float e = __builtin_fmaf(ur[q].y, ir[q].y, ur[q].x * ir[q].x);
      const float prev = __shfl_up(e, 1, 64);
      s = 0.f + __builtin_fmaf(ur[q].y, ir[q].y, __builtin_fmaf(ur[q].x, ir[q].x, prev));
      s += __shfl_xor(s, 16, 64);
    }
    s += __shfl_xor(s, 8, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 2, 64);
    if constexpr (VEC == 4) s += __shfl_xor(s, 1, 64);
    if (lane == 63) a.dot[b] = s;
  }
}

// one row of a deferred table by one wave (rows of 64 * VEC floats): out[p] = row ids[p] as of step - 1, p = the physical position of
// logical position b (segmented id arrays: common.h seg_phys)
struct GatherDefJob { const float* table; const float* M; const float* Vv; const int32_t* last; int64_t rows; const void* ids; float* out; int64_t n; int64_t seg_off = 0; };
struct GatherDefJobs {
  GatherDefJob j[2];
  int64_t seg_len = 0, seg_stride = 0;
  uint32_t step_add = 0;      // 1: the step state is advanced BEHIND this launch (by the chunk-rank launch): the step being computed is ss->step + 1
};
template <typename IdT, int VEC>
__device__ __forceinline__ void gather_deferred_wave_row(const GatherDefJobs& jobs, const GatherDefJob& jb, int64_t b, int lane, const StepStateDev* __restrict__ ss,
                                                         const AdamHp& h, int64_t ld_out, int* err) {
  using V = typename VecT<VEC>::type;
  constexpr int dim = 64 * VEC;
  b = seg_phys(b, jobs.seg_len, jobs.seg_stride, jb.seg_off);
  int64_t r = load_id((const IdT*)jb.ids, b);
  const bool ok = (uint64_t)r < (uint64_t)jb.rows;
  if (!ok) { if (err && lane == 0) *err = 1; r = 0; }
  const uint32_t t = ss->step + jobs.step_add, seen = (uint32_t)jb.last[r];
  const int64_t off = r * dim + lane * VEC;
  V th = vload<VEC>(jb.table + off);
  if (seen + 1 < t) {
    V m = vload<VEC>(jb.M + off), v = vload<VEC>(jb.Vv + off);
    adam_catch_up_uniform<false>(th, m, v, seen, t - 1, ss, h);
  }
  vstore<VEC>(jb.out + b * ld_out + lane * VEC, ok ? th : vzero<VEC>());
}

// R rows of deferred tables by one wave: every id, then every last[], then every row load of the R rows is requested before the first
// replay starts.  A 256-B row (dim 64) per wave leaves 3 dependent round trips with <= 768 B in flight per wave - at 32 waves per CU half
// of what the CU's share of HBM needs (the BPR gather ran at 0.48 of its memory time); control flow stays scalar per row (a wave replays
// one row at a time).  Rows [0, n_a) belong to job 0, the rest to job 1; rows past the end are skipped.
template <typename IdT, int VEC, int R>
__device__ __forceinline__ void gather_deferred_wave_rows(const GatherDefJobs& jobs, int64_t b0, int lane, const StepStateDev* __restrict__ ss, const AdamHp& h,
                                                          int64_t ld_out, int* err) {
  using V = typename VecT<VEC>::type;
  constexpr int dim = 64 * VEC;
  const int64_t n_a = jobs.j[0].n, n_all = n_a + jobs.j[1].n;
  const uint32_t t = ss->step + jobs.step_add;
  int which[R];
  int64_t pos[R], row[R];
  bool live[R], ok[R];
#pragma unroll
  for (int q = 0; q < R; ++q) {
    const int64_t b = b0 + q;
    live[q] = b < n_all;
    const int64_t bc = live[q] ? b : n_all - 1;
    which[q] = bc < n_a ? 0 : 1;
    const GatherDefJob& jb = jobs.j[which[q]];
    pos[q] = seg_phys(which[q] ? bc - n_a : bc, jobs.seg_len, jobs.seg_stride, jb.seg_off);
    row[q] = load_id((const IdT*)jb.ids, pos[q]);
  }
  uint32_t seen[R];
#pragma unroll
  for (int q = 0; q < R; ++q) {
    const GatherDefJob& jb = jobs.j[which[q]];
    ok[q] = (uint64_t)row[q] < (uint64_t)jb.rows;
    if (!ok[q]) { if (err && lane == 0 && live[q]) *err = 1; row[q] = 0; }
    seen[q] = (uint32_t)jb.last[row[q]];
  }
  V th[R], m[R], v[R];
#pragma unroll
  for (int q = 0; q < R; ++q) {
    const GatherDefJob& jb = jobs.j[which[q]];
    const int64_t off = row[q] * dim + lane * VEC;
    th[q] = vload<VEC>(jb.table + off);
    m[q] = vzero<VEC>(); v[q] = vzero<VEC>();
    if (seen[q] + 1 < t) { m[q] = vload<VEC>(jb.M + off); v[q] = vload<VEC>(jb.Vv + off); }
  }
#pragma unroll
  for (int q = 0; q < R; ++q) {
    if (seen[q] + 1 < t) adam_catch_up_uniform<false>(th[q], m[q], v[q], seen[q], t - 1, ss, h);
    if (live[q]) vstore<VEC>(jobs.j[which[q]].out + pos[q] * ld_out + lane * VEC, ok[q] ? th[q] : vzero<VEC>());
  }
}

// 256-B rows (64 floats), fast replay: FOUR rows per wave side by side - lane group q = lane >> 4 holds row q, 16 B per lane - so theta,
// m, v and the output are one 1 KB instruction each for the four rows (one wave per row moves 256 B per instruction: theta + last[] alone
// took 34 us for the BPR step's 196 608 rows where the plain gather of the same rows takes 12, tools/diag/gather_probe.py).  The four replays
// run together over max(lag) iterations; each group takes ITS alpha (ring index last + k: the four `last` values are read back as scalars,
// four s_load_dwordx8 per eight steps) and alpha = 0 once its lag is used up, which leaves theta as it is.  Per element the operations of
// the one-wave-per-row form in the same order: same bits.
template <typename IdT>
__device__ __forceinline__ void gather_deferred_group4(const GatherDefJobs& jobs, int64_t b0, int lane, const StepStateDev* __restrict__ ss, const AdamHp& h,
                                                       int64_t ld_out, int* err) {
  constexpr int dim = 64;
  const int64_t n_a = jobs.j[0].n, n_all = n_a + jobs.j[1].n;
  const uint32_t t = ss->step + jobs.step_add;
  const int q = lane >> 4, col = (lane & 15) * 4;
  const int64_t b = b0 + q;
  const bool live = b < n_all;
  const int64_t bc = live ? b : n_all - 1;
  const int which = bc < n_a ? 0 : 1;
  const GatherDefJob& ja = jobs.j[0];
  const GatherDefJob& jb = jobs.j[1];
  const int64_t pos = seg_phys(which ? bc - n_a : bc, jobs.seg_len, jobs.seg_stride, which ? jb.seg_off : ja.seg_off);
  int64_t row = which ? load_id((const IdT*)jb.ids, pos) : load_id((const IdT*)ja.ids, pos);
  const int64_t rows = which ? jb.rows : ja.rows;
  const bool ok = (uint64_t)row < (uint64_t)rows;
  if (!ok) { if (err && live && (lane & 15) == 0) *err = 1; row = 0; }
  const uint32_t seen = (uint32_t)(which ? jb.last : ja.last)[row];
  const FastRp f = fast_rp(ss);
  uint32_t lag = seen + 1 < t ? t - 1 - seen : 0u;
  const int64_t off = row * dim + col;
  float4 th = vload<4>((which ? jb.table : ja.table) + off);
  float4 m = vzero<4>(), v = vzero<4>();
  if (lag > 0) { m = vload<4>((which ? jb.M : ja.M) + off); v = vload<4>((which ? jb.Vv : ja.Vv) + off); }
  lag = lag < f.trunc ? lag : f.trunc;
  // wave-uniform: the longest lag, and the four groups' `last` as scalars
  uint32_t steps = lag;
  steps = max(steps, (uint32_t)__shfl_xor((int)steps, 16, 64));
  steps = max(steps, (uint32_t)__shfl_xor((int)steps, 32, 64));
  steps = (uint32_t)__builtin_amdgcn_readfirstlane((int)steps);
  if (steps > 0 && __builtin_amdgcn_ballot_w64(!(all_zero(m) && all_zero(v))) != 0) {
    const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)seen, 0), s1 = (uint32_t)__builtin_amdgcn_readlane((int)seen, 16);
    const uint32_t s2 = (uint32_t)__builtin_amdgcn_readlane((int)seen, 32), s3 = (uint32_t)__builtin_amdgcn_readlane((int)seen, 48);
    FastSt<float4> st;
    st.open(th, m, v, h.eps);
    uint32_t k = 1;
    for (; k + 7 <= steps; k += 8) {
      const float* __restrict__ r0 = ss->alpha_hist + ((s0 + k) & (BR_ALPHA_RING - 1));
      const float* __restrict__ r1 = ss->alpha_hist + ((s1 + k) & (BR_ALPHA_RING - 1));
      const float* __restrict__ r2 = ss->alpha_hist + ((s2 + k) & (BR_ALPHA_RING - 1));
      const float* __restrict__ r3 = ss->alpha_hist + ((s3 + k) & (BR_ALPHA_RING - 1));
      float a0[8], a1[8], a2[8], a3[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { a0[e] = r0[e]; a1[e] = r1[e]; a2[e] = r2[e]; a3[e] = r3[e]; }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float al = q == 0 ? a0[e] : (q == 1 ? a1[e] : (q == 2 ? a2[e] : a3[e]));
        st.step(k + e <= lag ? al : 0.f, h.b1, f);
      }
    }
    for (; k <= steps; ++k) {
      const float a0 = ss->alpha_hist[(s0 + k) & (BR_ALPHA_RING - 1)], a1 = ss->alpha_hist[(s1 + k) & (BR_ALPHA_RING - 1)];
      const float a2 = ss->alpha_hist[(s2 + k) & (BR_ALPHA_RING - 1)], a3 = ss->alpha_hist[(s3 + k) & (BR_ALPHA_RING - 1)];
      const float al = q == 0 ? a0 : (q == 1 ? a1 : (q == 2 ? a2 : a3));
      st.step(k <= lag ? al : 0.f, h.b1, f);
    }
    th = st.theta();
  }
  if (live) vstore<4>((which ? jb.out : ja.out) + pos * ld_out + col, ok ? th : vzero<4>());
}

}  // namespace br
