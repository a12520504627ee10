// S1 duplicate-id handling + O1 TF-form Adam + O2 Keras Adagrad.
//
// Design: no float atomics on the training path.  ids are radix-sorted once per id stream
// together with their batch position (stable => equal ids stay in ascending position); every
// table that shares the stream reuses the index.  The optimizer kernels walk each segment in
// that order, so the duplicate sum is exactly a sequential unsorted_segment_sum ([TF-sem]
// _deduplicate_indexed_slices): bitwise reproducible, and (sum g)^2 feeds Adam's v.
// The per-pair row gradients are written ONCE with plain stores by the backward kernels and
// read back here through the inverted index (cdna_hip_programming.md App. B "Scatter / gather").
#include "common.h"
#include "rows.h"
#include "adam_math.h"
#include "finalize.h"
#include "lookup_wave.h"
#include "dense.h"

#include <algorithm>
#include <vector>
#include <math.h>
#include <stddef.h>
#include <hipcub/hipcub.hpp>

namespace br {

// positions 0..n-1 and the sort keys: ids outside [0, upper) become `upper` so that they sort behind every
// valid id instead of aliasing one in the low key bits (the optimizer kernels skip ids >= table rows)
template <typename IdT>
__global__ void sort_prep_kernel(const IdT* __restrict__ ids, int64_t upper, IdT* __restrict__ keys, int32_t* __restrict__ p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    p[i] = (int32_t)i;
    const int64_t id = (int64_t)ids[i];
    keys[i] = (upper > 0 && (uint64_t)id >= (uint64_t)upper) ? (IdT)upper : (IdT)id;
  }
}

__global__ __launch_bounds__(256) void zero_bytes_kernel(uint8_t* __restrict__ p, int64_t n) {
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (i0 + q < n) p[i0 + q] = 0;
}

// one row group per sorted position; only segment heads do work
template <typename IdT>
__device__ __forceinline__ bool segment_head(const IdT* sid, int64_t i) {
  return i == 0 || sid[i - 1] != sid[i];
}

// ---- long id segments (hot ids: a Zipf batch puts thousands of positions on one row) ------------------------------
// A segment head that adds its duplicates one by one is a chain of dependent loads as long as the segment (measured:
// 2.3 ms for the Adam-rows launch on a Zipf(1.05) batch of 65 536).  With a partial buffer the walk is two-level:
// segment_partials_kernel gives every kSegBlock-aligned block of the SORTED order that continues its predecessor's id
// the ordered sum of its own run (<= 64 adds, all blocks in parallel); the head then adds its own positions up to the
// next block boundary one by one and ONE partial row per later block.  Still a fixed order - (((g_i + ..) + P_b) + P_b+1)
// with P_b = ((g_64b + g_64b+1) + ..) - restated by the oracle (ordered_segment_sum); without a buffer: one by one.
constexpr int kSegBlock = 64;

// a row of per-pair gradients: g[pos], times sc[pos] when the source carries a per-position factor (the MF halves of a NeuMF
// step: the stashed partner row times ddot[pos] - the product the embed backward used to write out as its own launch)
template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::type grow(const float* __restrict__ g, int64_t ldg, const float* __restrict__ sc, int32_t pos) {
  const typename VecT<VEC>::type v = vload<VEC>(g + (int64_t)pos * ldg);
  return sc ? vmul(v, sc[pos]) : v;
}

// acc += rows of positions [j, end) while they belong to `row`, in that order - the additions are sequential (the fp32 order is part
// of the result), but the loads are not: eight ids / positions / gradient rows are requested together (one dependent round trip per eight
// positions instead of per position; a hot id walks thousands).  Rows past the run's end are loaded and dropped.  -> first position
// not added.
template <typename IdT, int VEC, int W>
__device__ __forceinline__ int64_t seg_walk(typename VecT<VEC>::type& acc, const IdT* __restrict__ sid, const int32_t* __restrict__ spos, int64_t j,
                                            int64_t end, IdT row, const float* __restrict__ g, int64_t ldg, const float* __restrict__ sc) {
  using V = typename VecT<VEC>::type;
  for (; j + W <= end; j += W) {
    IdT sv[W];
    int32_t pv[W];
    V v[W];
#pragma unroll
    for (int e = 0; e < W; ++e) sv[e] = sid[j + e];
    if (sv[0] != row) return j;
#pragma unroll
    for (int e = 0; e < W; ++e) pv[e] = spos[j + e];
#pragma unroll
    for (int e = 0; e < W; ++e) v[e] = grow<VEC>(g, ldg, sc, pv[e]);
#pragma unroll
    for (int e = 0; e < W; ++e) {
      if (sv[e] != row) return j + e;
      acc = vadd(acc, v[e]);
    }
  }
  for (; j < end && sid[j] == row; ++j) acc = vadd(acc, grow<VEC>(g, ldg, sc, spos[j]));
  return j;
}

template <typename IdT, int VEC, int W = 1>
__device__ __forceinline__ void seg_acc_from(typename VecT<VEC>::type& acc, const IdT* __restrict__ sid, const int32_t* __restrict__ spos, int64_t n,
                                             int64_t i, int64_t j, IdT row, const float* __restrict__ g, int64_t ldg, const float* __restrict__ sc,
                                             const float* __restrict__ part, int pdim);

template <typename IdT, int VEC>
__device__ __forceinline__ typename VecT<VEC>::type seg_acc(const IdT* __restrict__ sid, const int32_t* __restrict__ spos, int64_t n, int64_t i,
                                                            IdT row, const float* __restrict__ g, int64_t ldg, const float* __restrict__ sc,
                                                            const float* __restrict__ part, int pdim) {
  using V = typename VecT<VEC>::type;
  V acc = grow<VEC>(g, ldg, sc, spos[i]);
  seg_acc_from<IdT, VEC>(acc, sid, spos, n, i, i + 1, row, g, ldg, sc, part, pdim);
  return acc;
}

// the rest of a head's sum from position j on (everything in [i, j) is already in acc, in order); W = positions requested together by the
// one-by-one part of the walk (same additions in the same order)
template <typename IdT, int VEC, int W>
__device__ __forceinline__ void seg_acc_from(typename VecT<VEC>::type& acc, const IdT* __restrict__ sid, const int32_t* __restrict__ spos, int64_t n,
                                             int64_t i, int64_t j, IdT row, const float* __restrict__ g, int64_t ldg, const float* __restrict__ sc,
                                             const float* __restrict__ part, int pdim) {
  using V = typename VecT<VEC>::type;
  const int64_t own_end = part ? ((i / kSegBlock + 1) * kSegBlock < n ? (i / kSegBlock + 1) * kSegBlock : n) : n;
  // (W = 1 in the row-group launches: they are bound by HBM latency at 8 waves / SIMD - the registers of a batched walk cost them more
  //  on ordinary batches than they save on hot ids; the long runs are cut to <= 63 positions by the partials.  The one-wave-per-row
  //  kernel walks eight at a time: its ids / positions are scalar loads, and a Zipf batch is full of runs of 5 - 60 positions whose
  //  heads otherwise pay three dependent round trips per position)
  j = seg_walk<IdT, VEC, W>(acc, sid, spos, j, own_end, row, g, ldg, sc);
  if (part && j == own_end) {
    // one partial per later block of the run, four at a time (same order of additions)
    for (; j + 3 * kSegBlock < n && sid[j + 3 * kSegBlock] == row; j += 4 * kSegBlock) {
      const V p0 = vload<VEC>(part + (j / kSegBlock) * pdim), p1 = vload<VEC>(part + (j / kSegBlock + 1) * pdim);
      const V p2 = vload<VEC>(part + (j / kSegBlock + 2) * pdim), p3 = vload<VEC>(part + (j / kSegBlock + 3) * pdim);
      acc = vadd(vadd(vadd(vadd(acc, p0), p1), p2), p3);
    }
    for (; j < n && sid[j] == row; j += kSegBlock) acc = vadd(acc, vload<VEC>(part + (j / kSegBlock) * pdim));
  }
}

struct SegJob {                // one table's gradient source for the partials
  const void* sid; const int32_t* spos;
  const float* g0; int64_t ldg0;
  const float* g1; int64_t ldg1;
  float* part;                 // [ceil(n / kSegBlock)][dim]
  const float* sc1 = nullptr;  // per-position factor of g1 rows, or null (last: positional initialisers of the other users stay valid)
  int64_t n = 0;               // this job's positions when the jobs of a launch differ in length (0: the launch's n)
};
struct SegJobs { SegJob j[2]; };

template <typename IdT, int VEC>
__global__ __launch_bounds__(256) void segment_partials_kernel(SegJobs jobs, int64_t n_launch, int dim, int chunks, int lpr_log2, int split) {
  using V = typename VecT<VEC>::type;
  const SegJob& jb = jobs.j[blockIdx.y];
  const int64_t n = jb.n ? jb.n : n_launch;
  const IdT* __restrict__ sid = (const IdT*)jb.sid;
  const int32_t* __restrict__ spos = jb.spos;
  const int lpr = 1 << lpr_log2;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t b = (tid >> lpr_log2) + 1;                 // block 0 starts at a head (or is walked by one)
  const int lir = (int)(tid & (lpr - 1));
  const int64_t i = b * kSegBlock;
  if (i >= n) return;
  const IdT row = sid[i];
  if (sid[i - 1] != row) return;                            // a head starts here: nobody reads this block's partial
  const int64_t end = i + kSegBlock < n ? i + kSegBlock : n;
  for (int c = lir; c < chunks; c += lpr) {
    const int col = c * VEC;
    const float* g = col < split ? jb.g0 + col : jb.g1 + (col - split);
    const int64_t ldg = col < split ? jb.ldg0 : jb.ldg1;
    const float* sc = col < split ? nullptr : jb.sc1;
    V acc = grow<VEC>(g, ldg, sc, spos[i]);
    (void)seg_walk<IdT, VEC, 8>(acc, sid, spos, i + 1, end, row, g, ldg, sc);
    vstore<VEC>(jb.part + b * dim + col, acc);
  }
}

// The same partials with one WAVE per 64-block, for rows of 64 * VEC floats: lane l holds the id and the position of sorted position
// 64 b + l (one coalesced round trip for the whole block instead of a dependent id -> position -> row chain per 8 positions), the run's
// length is a ballot, and the rows are requested eight at a time by position (readlane) and added in position order - the order
// seg_walk adds them in.  On a Zipf(1.05) batch ~70 % of the blocks continue a run: 40 us in the row-group form.
template <typename IdT, int VEC>
__global__ __launch_bounds__(256) void segment_partials_wave_kernel(SegJobs jobs, int64_t n_launch, int split) {
  using V = typename VecT<VEC>::type;
  constexpr int dim = 64 * VEC;
  const SegJob& jb = jobs.j[blockIdx.y];
  const int64_t n = jb.n ? jb.n : n_launch;
  const IdT* __restrict__ sid = (const IdT*)jb.sid;
  const int32_t* __restrict__ spos = jb.spos;
  const int64_t b = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) + 1;   // block 0 starts at a head
  const int64_t i = b * kSegBlock;
  if (i >= n) return;
  const int lane = (int)(threadIdx.x & 63);
  const int64_t q = i + lane < n ? i + lane : n - 1;
  const IdT my_id = sid[q];
  const int32_t my_pos = spos[q];
  const IdT row = sid[i];
  if (sid[i - 1] != row) return;                            // a head starts here: nobody reads this block's partial
  // positions of the block that belong to the run: the leading lanes whose id is `row`
  const uint64_t same = __builtin_amdgcn_ballot_w64(i + lane < n && my_id == row);
  const int L = same == ~0ull ? 64 : __builtin_ctzll(~same);
  const int col = lane * VEC;
  const bool lo = col < split;
  const float* __restrict__ gp = lo ? jb.g0 + col : jb.g1 + (col - split);
  const int64_t ldg = lo ? jb.ldg0 : jb.ldg1;
  const float* __restrict__ sc = jb.sc1;
  V acc = vzero<VEC>();
  for (int j0 = 0; j0 < L; j0 += 8) {
    V v[8];
    float scl[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int64_t pos = (int64_t)__builtin_amdgcn_readlane(my_pos, (j0 + e) & 63);     // (past L: a row of the block, loaded and dropped)
      v[e] = vload<VEC>(gp + pos * ldg);
      scl[e] = (sc && !lo) ? sc[pos] : 1.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (j0 + e >= L) break;
      const V t = lo ? v[e] : vmul(v[e], scl[e]);            // grow(): the MF halves are scaled as they are read
      acc = (j0 + e == 0) ? t : vadd(acc, t);
    }
  }
  vstore<VEC>(jb.part + b * dim + col, acc);
}

template <typename IdT, int VEC>
__global__ __launch_bounds__(256) void segment_sum_kernel(const IdT* __restrict__ sid, const int32_t* __restrict__ spos,
                                                           int64_t n, const float* __restrict__ g, int64_t ldg, int dim,
                                                           int chunks, int lpr_log2, float* __restrict__ out,
                                                           int32_t* __restrict__ head_flag, const float* __restrict__ part) {
  using V = typename VecT<VEC>::type;
  const int lpr = 1 << lpr_log2;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = tid >> lpr_log2;
  const int lir = (int)(tid & (lpr - 1));
  if (i >= n) return;
  const bool head = segment_head(sid, i);
  if (lir == 0 && head_flag) head_flag[i] = head ? 1 : 0;
  if (!head) return;
  const IdT row = sid[i];
  for (int c = lir; c < chunks; c += lpr) {
    const V acc = seg_acc<IdT, VEC>(sid, spos, n, i, row, g + c * VEC, ldg, nullptr, part ? part + c * VEC : nullptr, dim);
    vstore<VEC>(out + i * dim + c * VEC, acc);
  }
}

// Row gradients may come from two buffers: columns [0,split) from g0, [split,dim) from g1 (the fused
// NeuMF tables [mlp | mf] take their MLP half from dx0 and their MF half from the embed backward).
// One launch may serve two tables of the same geometry (the user and the item table of a NeuMF step, blockIdx.y):
// each alone leaves HBM half idle (random 512-B rows, a dependent chain per row), together they overlap.
struct AdamRowsJob {
  float* table; float* M; float* Vv;
  int64_t table_rows;
  const void* sid; const int32_t* spos;
  const float* g0; int64_t ldg0;
  const float* g1; int64_t ldg1;
  const float* sc1;           // per-position factor of g1 rows, or null
  const float* part;          // segment partials of this table's gradient source (NULL: one-by-one walk)
  uint8_t* mark; int32_t* last;
  // deferred mode, optional: the row as the step's lookup replayed it (theta at step t-1), per POSITION - columns [0,split) at
  // th0 + pos*ldt0, [split,dim) at th1 + pos*ldt1.  With it the optimizer replays only m and v (two multiplies per step) instead of
  // repeating the lookup's sqrt / rcp chain on theta: same bits (the lookup ran adam_replay on the same stored row).
  const float* th0 = nullptr; int64_t ldt0 = 0;
  const float* th1 = nullptr; int64_t ldt1 = 0;
  int64_t n = 0;              // this job's positions when the two jobs of a launch differ in length (wave kernel only; 0: the launch's n)
};
struct AdamRowsJobs { AdamRowsJob j[2]; };

template <typename IdT, int VEC>
__global__ __launch_bounds__(256) void adam_rows_sorted_kernel(AdamRowsJobs jobs, int dim, int chunks, int lpr_log2, int64_t n, int split,
                                                                AdamHp h, const StepStateDev* __restrict__ ss, const KeepFuse kf, int keep_x) {
  // keep_x columns of the grid fill the next step's dropout planes (ALU-bound Philox beside this HBM-bound kernel:
  // no fork / join around a launch of its own, which costs ~10 us each inside a hipGraph)
  // every P-th column of the grid (P = columns / keep_x) is a plane column, so the two kinds of work share the CUs from start to end
  int64_t bx = blockIdx.x;
  if (keep_x > 0) {
    const int P = (int)gridDim.x / keep_x;
    const int q = (int)blockIdx.x / P;
    if ((int)blockIdx.x - q * P == 0 && q < keep_x) {
      const int64_t wg = (int64_t)q * gridDim.y + blockIdx.y;
      if (wg < kf.total) keep_fuse_block(kf, wg);
      return;
    }
    const int before = ((int)blockIdx.x + P - 1) / P;          // plane columns left of this one
    bx -= before < keep_x ? before : keep_x;
  }
  const AdamRowsJob& jb = jobs.j[blockIdx.y];
  float* __restrict__ table = jb.table; float* __restrict__ M = jb.M; float* __restrict__ Vv = jb.Vv;
  const int64_t table_rows = jb.table_rows;
  const IdT* __restrict__ sid = (const IdT*)jb.sid;
  const int32_t* __restrict__ spos = jb.spos;
  const float* __restrict__ g0 = jb.g0; const float* __restrict__ g1 = jb.g1;
  const int64_t ldg0 = jb.ldg0, ldg1 = jb.ldg1;
  uint8_t* __restrict__ mark = jb.mark;
  int32_t* __restrict__ last = jb.last;
  using V = typename VecT<VEC>::type;
  __shared__ float ring[BR_ALPHA_RING];
  if (last) stage_alpha_ring(ring, ss);      // uniform branch: whole workgroup
  adam_resolve(h);
  const int lpr = 1 << lpr_log2;
  const int64_t tid = bx * blockDim.x + threadIdx.x;
  const int64_t i = tid >> lpr_log2;
  const int lir = (int)(tid & (lpr - 1));
  if (i >= n) return;
  if (!segment_head(sid, i)) return;
  const int64_t row = (int64_t)sid[i];
  if ((uint64_t)row >= (uint64_t)table_rows) return;  // out-of-range ids were flagged by the forward
  if (mark && lir == 0) mark[row] = 1;
  // deferred mode: this row includes the steps <= last[row]; replay the g = 0 steps up to t-1 first
  const uint32_t t = last ? ss->step : 0u;
  const uint32_t seen = last ? (uint32_t)last[row] : 0u;
  for (int c = lir; c < chunks; c += lpr) {
    const int col = c * VEC;
    const float* g = col < split ? g0 + col : g1 + (col - split);
    const int64_t ldg = col < split ? ldg0 : ldg1;
    const V acc = seg_acc<IdT, VEC>(sid, spos, n, i, sid[i], g, ldg, col < split ? nullptr : jb.sc1, jb.part ? jb.part + col : nullptr, dim);
    const int64_t off = row * dim + col;
    V th = vload<VEC>(table + off), m = vload<VEC>(M + off), v = vload<VEC>(Vv + off);
    if (last && seen + 1 < t) adam_catch_up<true>(th, m, v, seen, t - 1, ring, ss, h);
    adam_update(th, m, v, acc, h);
    vstore<VEC>(table + off, th);
    vstore<VEC>(M + off, m);
    vstore<VEC>(Vv + off, v);
  }
  if (last && lir == 0) last[row] = (int32_t)t;
}

__constant__ float kOneF = 1.f;
__constant__ int32_t kZeroI = 0;
// A kernel-argument pointer held in scalar registers from here on, typed as a GLOBAL pointer (behind the asm the compiler no longer
// knows where it came from and would fall back to flat loads, which also count on lgkmcnt: every scalar-load wait would drain them).
typedef __attribute__((address_space(1))) const float gcf;
typedef __attribute__((address_space(1))) float gwf;
typedef __attribute__((address_space(1))) const int32_t gci32;
typedef __attribute__((address_space(1))) int32_t gwi32;
__device__ __forceinline__ gcf* sgpr_g(const float* x) { gcf* y = (gcf*)x; BR_PIN_S(y); return y; }
__device__ __forceinline__ gwf* sgpr_g(float* x) { gwf* y = (gwf*)x; BR_PIN_S(y); return y; }
__device__ __forceinline__ gwi32* sgpr_g(int32_t* x) { gwi32* y = (gwi32*)x; BR_PIN_S(y); return y; }
__device__ __forceinline__ int64_t sgpr(int64_t x) { BR_PIN_S(x); return x; }
// (the HIP vector classes have no constructors from address-space-qualified objects: go through the native vector types)
template <int VEC> struct NatV { typedef float type __attribute__((ext_vector_type(VEC))); };
template <> struct NatV<1> { typedef float type; };
__device__ __forceinline__ float from_nat(float a) { return a; }
__device__ __forceinline__ float2 from_nat(NatV<2>::type a) { return make_float2(a.x, a.y); }
__device__ __forceinline__ float4 from_nat(NatV<4>::type a) { return make_float4(a.x, a.y, a.z, a.w); }
__device__ __forceinline__ float to_nat(float a) { return a; }
__device__ __forceinline__ NatV<2>::type to_nat(float2 a) { NatV<2>::type r = {a.x, a.y}; return r; }
__device__ __forceinline__ NatV<4>::type to_nat(float4 a) { NatV<4>::type r = {a.x, a.y, a.z, a.w}; return r; }
template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::type gvload(gcf* p) {
  typedef __attribute__((address_space(1))) const typename NatV<VEC>::type GV;
  const typename NatV<VEC>::type r = *(GV*)p;
  return from_nat(r);
}
template <int VEC>
__device__ __forceinline__ void gvstore(gwf* p, typename VecT<VEC>::type v) {
  typedef __attribute__((address_space(1))) typename NatV<VEC>::type GV;
  *(GV*)p = to_nat(v);
}


// The same update with one WAVE per row, for rows of 64 * VEC floats (the fused [mlp | mf] rows of embed_dim 32 / 64 / 128): a lane
// owns VEC consecutive columns, so everything that depends on the position - head test, id, duplicate walk, the row's lag in deferred
// mode - is wave-uniform (scalar loads and branches; the row-group form above runs two or four rows with different lags per wave and
// every lane waits for the longest).  A wave takes a STRIP of S consecutive sorted positions: one round trip for the strip's ids and
// positions, then every row load of the strip's heads (m, v, theta) and every gradient row a head of the strip can need are requested
// before the first is used - S rows of 3.5 KB in flight per wave instead of one, which is what an HBM-latency-bound gather / scatter
// of 512-B rows needs - and a head finds its duplicates inside the strip already in registers (added in position order, as seg_acc
// does; a run that leaves the strip continues through seg_acc_from).  Deferred mode takes theta as the lookup replayed it
// (AdamRowsJob::th0 / th1) and replays m and v only: no alpha ring, no sqrt / rcp per replayed step.
// Riders: the grid may carry two other pieces of the step as extra workgroups, spread evenly between the row workgroups
// (every P-th workgroup is a rider) - the NEXT step's dropout keep-bit planes (Philox: ALU work beside an HBM-bound kernel) and the
// dense finalize (slab reductions + BatchNorm gradients + Adam on the flat vector: only needs the backward).  As launches of their
// own they needed a fork / join around this kernel inside the step's hipGraph (~10 us each on the main branch, ROCm 7.2).
struct AdamRiders {
  KeepFuse kf;             // kf.total workgroups of keep-bit planes
  FinalArgs fin;           // n_final workgroups of the dense finalize
  int n_final, total;      // total = kf.total + n_final riders
  int rows_x;              // row workgroups per job
};

template <typename IdT, int VEC, int S>
__global__ __launch_bounds__(256) void adam_rows_wave_kernel(AdamRowsJobs jobs, int64_t n_launch, int split, AdamHp h, const StepStateDev* __restrict__ ss,
                                                              const AdamRiders rd) {
  using V = typename VecT<VEC>::type;
  constexpr int dim = 64 * VEC;
  static_assert(kSegBlock % S == 0, "a strip stays inside one partial block");
  __shared__ float fin_part[16][64];
  int64_t rb = blockIdx.x;                                    // row workgroup: job = rb / rows_x
  if (rd.total > 0) {
    const int P = (int)gridDim.x / rd.total;
    const int q = (int)blockIdx.x / P;
    if ((int)blockIdx.x - q * P == 0 && q < rd.total) {       // rider q
      if (q < rd.kf.total) { keep_fuse_block(rd.kf, q); return; }
      adam_resolve(h);
      finalize_block256(rd.fin, h, q - rd.kf.total, fin_part);
      return;
    }
    const int before = ((int)blockIdx.x + P - 1) / P;          // riders left of this workgroup
    rb -= before < rd.total ? before : rd.total;
  }
  const int job = (int)(rb / rd.rows_x);
  const AdamRowsJob& jb = jobs.j[job];
  const int64_t n = jb.n ? jb.n : n_launch;
  const int64_t base = ((rb - (int64_t)job * rd.rows_x) * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * S;
  if (base >= n) return;
  const IdT* __restrict__ sid = (const IdT*)jb.sid;
  const int32_t* __restrict__ spos = jb.spos;
  // round 1: the strip's ids / positions (+ the predecessor's id)
  IdT id[S];
  int64_t pos[S];
  const IdT prev = sid[base > 0 ? base - 1 : 0];
#pragma unroll
  for (int e = 0; e < S; ++e) {
    const int64_t q = base + e < n ? base + e : n - 1;
    id[e] = sid[q];
    pos[e] = (int64_t)spos[q];
  }
  bool head[S], live[S], need_g[S];
  bool any = false;
#pragma unroll
  for (int e = 0; e < S; ++e) {
    const bool in = base + e < n;
    head[e] = in && (e == 0 ? (base == 0 || prev != id[0]) : id[e] != id[e - 1]);
    live[e] = head[e] && (uint64_t)(int64_t)id[e] < (uint64_t)jb.table_rows;   // out-of-range ids were flagged by the forward
    any = any || head[e];
    need_g[e] = in && any;          // a position before the strip's first head belongs to an earlier strip's head
  }
  if (!any) return;
  adam_resolve(h);
  const int lane = (int)(threadIdx.x & 63);
  const int col = lane * VEC;
  const bool lo = col < split;
  // the job's fields as scalars first (a per-lane choice between two fields of the argument block would otherwise be compiled into a
  // per-lane LOAD of the chosen field: a dependent vector load in front of everything)
  gcf* g0 = sgpr_g(jb.g0); gcf* g1 = sgpr_g(jb.g1);
  const int64_t ldg0 = sgpr(jb.ldg0), ldg1 = sgpr(jb.ldg1);
  gcf* t0 = sgpr_g(jb.th0); gcf* t1 = sgpr_g(jb.th1);
  const int64_t ldt0 = sgpr(jb.ldt0), ldt1 = sgpr(jb.ldt1);
  gwf* const tab = sgpr_g(jb.table); gwf* const Mp = sgpr_g(jb.M); gwf* const Vp = sgpr_g(jb.Vv);
  gcf* sc = sgpr_g(jb.sc1);
  gwi32* const last = sgpr_g(jb.last);
  gcf* gp = lo ? g0 + col : g1 + (col - split);
  const int64_t ldg = lo ? ldg0 : ldg1;
  const bool stashed = t0 != nullptr;
  gcf* tp = lo ? t0 + col : t1 + (col - split);
  const int64_t ldt = lo ? ldt0 : ldt1;
  // round 2: everything the strip needs, requested together.  Branch-free on purpose: behind scalar branches the compiler sinks each
  // load to its first use and the wave is back to one row in flight.  A position that is no live head reads row 0 instead (the
  // same three lines for everybody: cache hits), a position before the strip's first head its own gradient row (dropped).
  V g[S], m[S], v[S], th[S];
  float scale[S];
  int32_t seen_v[S];               // last[row]: a vector load (the kernel writes the array, so no scalar cache) - uniform, read out below
#pragma unroll
  for (int e = 0; e < S; ++e) {
    const int64_t row = live[e] ? (int64_t)id[e] : 0;
    const int64_t off = row * dim + col;
    g[e] = gvload<VEC>(gp + pos[e] * ldg);
    scale[e] = *(sc ? sc + pos[e] : (gcf*)&kOneF);
    m[e] = gvload<VEC>(Mp + off);
    v[e] = gvload<VEC>(Vp + off);
    th[e] = gvload<VEC>(stashed && live[e] ? tp + pos[e] * ldt : (gcf*)(tab + off));
    seen_v[e] = *(last ? last + row : (gwi32*)&kZeroI);
  }
  const uint32_t t = last ? ss->step : 0u;
  const bool fast = last && ss->fast;                          // replay form of the deferred tables (adam_math.h)
  uint32_t seen[S];
#pragma unroll
  for (int e = 0; e < S; ++e) {                                               // one wait for the whole strip
    pin(g[e]), pin(m[e]), pin(v[e]), pin(th[e]);
    BR_PIN_V(seen_v[e]);
    seen[e] = (uint32_t)__builtin_amdgcn_readfirstlane(seen_v[e]);
  }
#pragma unroll
  for (int e = 0; e < S; ++e)
    if (!lo) g[e] = vmul(g[e], scale[e]);                      // the MF halves: stashed partner row times ddot[pos] (grow())
#pragma unroll
  for (int e = 0; e < S; ++e) {
    if (!live[e]) continue;
    const int64_t i = base + e;
    V acc = g[e];
    bool open = true;                                          // the run is still going at the strip's end
#pragma unroll
    for (int f = e + 1; f < S; ++f) {
      open = open && base + f < n && !head[f];
      if (open) acc = vadd(acc, g[f]);
    }
    if (open && base + S < n)
      seg_acc_from<IdT, VEC, 8>(acc, sid, spos, n, i, base + S, id[e], (const float*)gp, ldg, lo ? nullptr : (const float*)sc, jb.part ? jb.part + col : nullptr, dim);
    // deferred: the g = 0 steps (seen, t-1] of the moments (adam_decay's first two products; theta came replayed)
    V mm = m[e], vv = v[e], tt = th[e];
    if (last && seen[e] + 1 < t) {
      if (fast) fast_moments(mm, vv, t - 1 - seen[e], ss);
      else for (uint32_t j = seen[e] + 1; j < t; ++j) { mm = vmul(mm, h.b1); vv = vmul(vv, h.b2); }
    }
    adam_update(tt, mm, vv, acc, h);
    const int64_t off = (int64_t)id[e] * dim + col;
    gvstore<VEC>(tab + off, tt);
    gvstore<VEC>(Mp + off, mm);
    gvstore<VEC>(Vp + off, vv);
    if (lane == 0) {
      if (jb.mark) jb.mark[(int64_t)id[e]] = 1;
      if (last) last[(int64_t)id[e]] = (int32_t)t;
    }
  }
}

// Deferred mode, whole table: bring every row up to the current step (inclusive) — before the table is
// read by anything but the catch-up gather (inference, checkpoint), and at least once per BR_ALPHA_RING steps.
template <int VEC>
__global__ __launch_bounds__(256) void adam_flush_kernel(float* __restrict__ table, float* __restrict__ M, float* __restrict__ Vv,
                                                          int64_t n_vec, int chunks, AdamHp h, int32_t* __restrict__ last,
                                                          const StepStateDev* __restrict__ ss) {
  using V = typename VecT<VEC>::type;
  __shared__ float ring[BR_ALPHA_RING];
  stage_alpha_ring(ring, ss);
  const uint32_t t = ss->step;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_vec; e += stride) {
    const int64_t row = e / chunks;
    const uint32_t seen = (uint32_t)last[row];
    if (seen >= t) continue;
    V m = vload<VEC>(M + e * VEC), v = vload<VEC>(Vv + e * VEC);
    if (all_zero(m) && all_zero(v)) continue;
    V th = vload<VEC>(table + e * VEC);
    adam_catch_up<true>(th, m, v, seen, t, ring, ss, h);
    vstore<VEC>(table + e * VEC, th);
    vstore<VEC>(M + e * VEC, m);
    vstore<VEC>(Vv + e * VEC, v);
  }
}
__global__ __launch_bounds__(256) void fill_last_kernel(int32_t* __restrict__ last, int64_t rows, const StepStateDev* __restrict__ ss) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows) last[i] = (int32_t)ss->step;
}

// Dense sweep: all rows NOT marked get the zero-gradient update.  float4 streaming,
// 2 vectors per thread in flight, grid-stride.
template <int VEC>
__global__ __launch_bounds__(256) void adam_dense_sweep_kernel(float* __restrict__ table, float* __restrict__ M,
                                                                float* __restrict__ Vv, int64_t n_vec, int chunks,
                                                                AdamHp h, const uint8_t* __restrict__ mark) {
  using V = typename VecT<VEC>::type;
  adam_resolve(h);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_vec; e += stride) {
    const int64_t row = e / chunks;
    if (mark && mark[row]) continue;
    V th = vload<VEC>(table + e * VEC), m = vload<VEC>(M + e * VEC), v = vload<VEC>(Vv + e * VEC);
    adam_decay(th, m, v, h.alpha, h);
    vstore<VEC>(table + e * VEC, th);
    vstore<VEC>(M + e * VEC, m);
    vstore<VEC>(Vv + e * VEC, v);
  }
}

__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ th, float* __restrict__ m, float* __restrict__ v,
                                                         const float* __restrict__ g, int64_t n, AdamHp h) {
  adam_resolve(h);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) adam_update1(th[i], m[i], v[i], g[i], h);
}

__device__ __forceinline__ void adagrad_update1(float& th, float& acc, float g, float lr, float eps) {
  acc = acc + g * g;
  th = th - lr * g / (sqrtf(acc) + eps);
}

__device__ __forceinline__ void adagrad_update(float4& th, float4& a, float4 g, float lr, float eps) {
  adagrad_update1(th.x, a.x, g.x, lr, eps); adagrad_update1(th.y, a.y, g.y, lr, eps);
  adagrad_update1(th.z, a.z, g.z, lr, eps); adagrad_update1(th.w, a.w, g.w, lr, eps);
}
__device__ __forceinline__ void adagrad_update(float2& th, float2& a, float2 g, float lr, float eps) {
  adagrad_update1(th.x, a.x, g.x, lr, eps); adagrad_update1(th.y, a.y, g.y, lr, eps);
}
__device__ __forceinline__ void adagrad_update(float& th, float& a, float g, float lr, float eps) { adagrad_update1(th, a, g, lr, eps); }

template <typename IdT, int VEC>
__global__ __launch_bounds__(256) void adagrad_rows_sorted_kernel(float* __restrict__ table, float* __restrict__ A,
                                                                   int64_t table_rows, int dim, int chunks, int lpr_log2,
                                                                   const IdT* __restrict__ sid, const int32_t* __restrict__ spos,
                                                                   int64_t n, const float* __restrict__ g, int64_t ldg,
                                                                   float lr, float eps, const float* __restrict__ part) {
  using V = typename VecT<VEC>::type;
  const int lpr = 1 << lpr_log2;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = tid >> lpr_log2;
  const int lir = (int)(tid & (lpr - 1));
  if (i >= n) return;
  if (!segment_head(sid, i)) return;
  const int64_t row = (int64_t)sid[i];
  if ((uint64_t)row >= (uint64_t)table_rows) return;
  for (int c = lir; c < chunks; c += lpr) {
    const V acc = seg_acc<IdT, VEC>(sid, spos, n, i, sid[i], g + c * VEC, ldg, nullptr, part ? part + c * VEC : nullptr, dim);
    const int64_t off = row * dim + c * VEC;
    V th = vload<VEC>(table + off), a = vload<VEC>(A + off);
    adagrad_update(th, a, acc, lr, eps);
    vstore<VEC>(table + off, th);
    vstore<VEC>(A + off, a);
  }
}

__global__ __launch_bounds__(256) void adagrad_flat_kernel(float* __restrict__ th, float* __restrict__ acc,
                                                            const float* __restrict__ g, int64_t n, float lr, float eps) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) adagrad_update1(th[i], acc[i], g[i], lr, eps);
}

template <typename IdT>
__global__ __launch_bounds__(256) void scatter_add_kernel(float* __restrict__ gt, int64_t table_rows, const IdT* __restrict__ ids,
                                                           int64_t n, const float* __restrict__ rows, int dim, int* err) {
  // one lane per float: a wave covers 64 contiguous floats (256 B) of one row => the atomic
  // shape MI355X_MICROARCH.md "Global float atomics" measures at full rate for dim >= 64.
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * dim) return;
  const int64_t b = e / dim;
  const int d = (int)(e - b * dim);
  const int64_t id = load_id(ids, b);
  if ((uint64_t)id >= (uint64_t)table_rows) {
    if (err) *err = 1;
    return;
  }
  atomicAdd(gt + id * dim + d, rows[e]);
}

// top of a training step: step += 1, alpha_t (thread 0), and the step's double scratch zeroed (all threads) -
// one launch instead of a memset node plus a kernel
__device__ __forceinline__ void step_state_advance_block(const StepAdvance& a) {
  for (int64_t i = threadIdx.x; i < a.n_zero; i += blockDim.x) a.zero[i] = 0.0;
  if (threadIdx.x == 0) {
    StepStateDev* st = a.st;
    const uint32_t t = st->step + 1;
    const double p1 = st->pow_b1 * a.b1, p2 = st->pow_b2 * a.b2;
    const float al = (float)(a.lr * sqrt(1.0 - p2) / (1.0 - p1));
    st->step = t;
    st->pow_b1 = p1;
    st->pow_b2 = p2;
    st->alpha_t = al;
    st->alpha_hist[t & (BR_ALPHA_RING - 1)] = al;
    if ((t & (BR_ALPHA_RING - 1)) < (uint32_t)BR_RING_MIRROR) st->alpha_hist[BR_ALPHA_RING + (t & (BR_ALPHA_RING - 1))] = al;   // the mirror behind the ring's end
  }
}
__global__ __launch_bounds__(256) void step_state_advance_kernel(StepAdvance a) { step_state_advance_block(a); }

template <typename IdT>
__global__ __launch_bounds__(256) void stage_batch_kernel(IdT* __restrict__ du, IdT* __restrict__ di, float* __restrict__ dy,
                                                           const IdT* __restrict__ su, const IdT* __restrict__ si,
                                                           const float* __restrict__ sy, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const IdT u = su[i], it = si[i];
  const float y = sy ? sy[i] : 0.f;
  du[i] = u; di[i] = it;
  if (dy) dy[i] = y;
}

static inline int bits_for(int64_t upper) {
  int bits = 1;
  while (bits < 63 && ((int64_t)1 << bits) < upper) ++bits;
  return bits;
}

template <typename IdT>
static int64_t sort_temp_bytes(int64_t n) {
  size_t bytes = 0;
  (void)hipcub::DeviceRadixSort::SortPairs((void*)nullptr, bytes, (const IdT*)nullptr, (IdT*)nullptr, (const int32_t*)nullptr,
                                     (int32_t*)nullptr, (int)n, 0, (int)sizeof(IdT) * 8, (hipStream_t)0);
  return (int64_t)bytes;
}

}  // namespace br

using namespace br;

static inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

// ---- dedup index for small batches: 2 launches instead of hipcub's ~10 per id stream ------------------------
// hipcub::DeviceRadixSort on 65 536 pairs is a block sort + 6 merge passes (+ iota): ~10 launches of ~5 us each,
// and a hipGraph replay runs them on the critical path (rocprofv3 timeline, ROCm 7.2).  For n <= kRankMaxN:
//   K1 chunk_sort_kernel : every workgroup radix-sorts one chunk of (id, position) pairs in LDS (stable);
//   K2 chunk_rank_kernel : the final rank of an element = its index in its chunk + for every other chunk the
//                          number of keys that sort before it (binary search: "<=" in earlier chunks, "<" in
//                          later ones = stable), then one scatter.
// Chunk size: 2048 pairs (256 threads) up to 16 384 keys, 8192 pairs (1024 threads) above.  Round 1 used 2048 throughout:
// at n = 65 536 that is 31 other chunks x 11 probes = 341 dependent L2 probes per key, 25 us alone and 90 us beside the
// MLP kernels it overlaps (15 % of all GPU time in the rocprofv3 trace).  8 chunks of 8192 need 7 x 13 = 91 probes.
// Both id streams of a step share the two launches (blockIdx.y).  Out-of-range ids get the key `upper`
// (>= table rows: the optimizer kernels skip them), so only bits_for(upper + 2) key bits are sorted.
constexpr int kChunkS = 2048, kThreadsS = 256, kChunkL = 8192, kThreadsL = 1024, kRankMaxChunks = 64;
constexpr int64_t kChunkSwitchN = 16384;
constexpr int64_t kRankMaxN = (int64_t)kChunkL * kRankMaxChunks;
struct IdxJob {
  const void* ids;
  void* sorted_ids;
  int32_t* sorted_pos;
  uint32_t* ck;       // [n] chunk-sorted keys
  uint32_t* cp;       // [n] their positions
  uint32_t upper;     // ids are valid in [0, upper)
  int end_bit;
  // segmented id arrays (brRowIndexBuildPairSeg): logical position t sits at element seg_phys(t) of `ids`, and the index carries
  // that PHYSICAL position (the optimizer reads gradient rows by it); seg_len == 0: contiguous
  int64_t seg_len = 0, seg_stride = 0, seg_off = 0;
  int64_t n = 0;      // this stream's keys when the two streams of a launch differ in length (0: the launch's n)
  int n_chunks = 0;   // its chunk count then
};

struct IdxJobs { IdxJob j[2]; };

// chunk `chunk` of id stream `which` by the calling workgroup of kSortThreads threads
template <typename IdT, int kChunk, int kSortThreads>
__device__ __forceinline__ void chunk_sort_block(const IdxJobs& jobs, int64_t n, int chunk, int which) {
  constexpr int IPT = kChunk / kSortThreads;
  using Sort = hipcub::BlockRadixSort<uint32_t, kSortThreads, IPT, uint32_t>;
  __shared__ typename Sort::TempStorage tmp;
  const IdxJob& job = jobs.j[which];
  if (job.n) n = job.n;
  const IdT* ids = (const IdT*)job.ids;
  const int64_t base = (int64_t)chunk * kChunk + threadIdx.x * IPT;
  uint32_t k[IPT], p[IPT];
#pragma unroll
  for (int q = 0; q < IPT; ++q) {
    const int64_t e = base + q;
    const int64_t pe = seg_phys(e, job.seg_len, job.seg_stride, job.seg_off);
    const int64_t id = e < n ? (int64_t)ids[pe] : -1;
    k[q] = e < n ? (((uint64_t)id < (uint64_t)job.upper) ? (uint32_t)id : job.upper) : job.upper + 1u;   // padding sorts last
    p[q] = (uint32_t)pe;
  }
  Sort(tmp).Sort(k, p, 0, job.end_bit);
#pragma unroll
  for (int q = 0; q < IPT; ++q)
    if (base + q < n) { job.ck[base + q] = k[q]; job.cp[base + q] = p[q]; }
}
template <typename IdT, int kChunk, int kSortThreads>
__global__ __launch_bounds__(kSortThreads) void chunk_sort_kernel(IdxJobs jobs, int64_t n) {
  chunk_sort_block<IdT, kChunk, kSortThreads>(jobs, n, (int)blockIdx.x, (int)blockIdx.y);
}

// The deferred NeuMF lookup and the chunk sorts of the step's two id streams in ONE launch of 1024-thread workgroups: the first
// 2 * n_chunks workgroups each sort a chunk (they start first and run ~37 us on 16 CUs), the rest are lookup workgroups of 16 waves =
// 16 pairs that flow around them.  As a launch of their own on a side stream the sorts needed a fork and a join in the step's
// hipGraph (~10 us each on the main branch, ROCm 7.2) and stretched the lookup they ran beside; the chunk-rank launch follows on the
// same stream.  LDS: the sort's image is reserved by every workgroup (two per CU = 32 waves: the lookup's full occupancy anyway).
template <typename IdT, int VEC, int R>
__global__ __launch_bounds__(kThreadsL, 8) void lookup_sort_kernel(const LookupArgs a, IdxJobs jobs, int64_t n, int n_chunks) {
  const int n_sort = 2 * n_chunks;
  if ((int)blockIdx.x < n_sort) {
    chunk_sort_block<IdT, kChunkL, kThreadsL>(jobs, n, (int)blockIdx.x % n_chunks, (int)blockIdx.x / n_chunks);
    return;
  }
  const int64_t b = (((int64_t)blockIdx.x - n_sort) * (kThreadsL / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * (R < 1 ? 1 : R);
  if (b >= a.batch) return;
  if constexpr (R == 0) {      // embed_dim 64: both rows of the pair side by side (lookup_half_pair); exact replay keeps the form above
    if (a.ss->fast) lookup_half_pair<IdT>(a, b, (int)(threadIdx.x & 63));
    else lookup_wave_pair<IdT, VEC>(a, b, (int)(threadIdx.x & 63));
  } else if constexpr (R == 1) lookup_wave_pair<IdT, VEC>(a, b, (int)(threadIdx.x & 63));
  else lookup_wave_pairs<IdT, VEC, R>(a, b, (int)(threadIdx.x & 63));
}
// form of the fused lookup at embed_dim 64: BR_LOOKUP_PAIRS = 1 (default: one pair per wave, lookup_wave_pair) | 2 (two pairs per wave) |
// 0 (lookup_half_pair: both rows of the pair side by side, 16 B per lane).  Measured at config 2, same bits in all three: 68-69 us | 70.5 us |
// 86-88 us - the half-wave form halves the load instructions but replays four elements per lane over max(lag_u, lag_i) steps with a
// per-lane alpha select: the launch follows its VALU work, not its instruction count.
static int lookup_pairs_per_wave() {
  static const int r = [] { const char* e = getenv("BR_LOOKUP_PAIRS"); const int v = e ? atoi(e) : 1; return (v == 0 || v == 2) ? v : 1; }();
  return r;
}

// Two deferred gathers of one row width (rows of 64 * VEC floats, one wave per row) and the chunk sorts of their two id streams in ONE
// launch of 1024-thread workgroups - the BPR step's user gather (B rows) and [pos | neg] item gather (2 B rows): the first workgroups each
// sort a chunk, the rest gather 16 rows each and flow around them; the chunk-rank launch (+ the step-state advance) follows on the same
// stream.  As launches of their own on two side streams the sorts and ranks were 42 % of the step's kernel time and cost a fork / join
// inside the step's hipGraph.
// rows per wave of the fused gather: 1 KB of row per wave and table (dim 64: four rows)
template <int VEC> constexpr int kGatherRowsPerWave = VEC == 1 ? 4 : (VEC == 2 ? 2 : 1);
template <typename IdT, int VEC>
__global__ __launch_bounds__(kThreadsL) void gather_sort_kernel(const GatherDefJobs gj, const StepStateDev* __restrict__ ss, const AdamHp h, int64_t ld_out, int* err,
                                                                IdxJobs jobs, int n_sort_a, int n_sort_b, bool gather_group4_enabled) {
  const int n_sort = n_sort_a + n_sort_b;
  if ((int)blockIdx.x < n_sort) {
    const int which = (int)blockIdx.x < n_sort_a ? 0 : 1;
    chunk_sort_block<IdT, kChunkL, kThreadsL>(jobs, 0, which ? (int)blockIdx.x - n_sort_a : (int)blockIdx.x, which);
    return;
  }
  constexpr int R = kGatherRowsPerWave<VEC>;
  const int64_t b0 = (((int64_t)blockIdx.x - n_sort) * (kThreadsL / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * R;
  if (b0 >= gj.j[0].n + gj.j[1].n) return;
  if constexpr (VEC == 1) {      // 256-B rows: four rows side by side at 16 B per lane under the fast replay (gather_deferred_group4)
    if (ss->fast && (ld_out & 3) == 0 && gather_group4_enabled) { gather_deferred_group4<IdT>(gj, b0, (int)(threadIdx.x & 63), ss, h, ld_out, err); return; }
  }
  gather_deferred_wave_rows<IdT, VEC, R>(gj, b0, (int)(threadIdx.x & 63), ss, h, ld_out, err);
}

// adv.st != NULL: the grid has one extra column of workgroups, whose y = 0 member advances the step state (nothing in this launch reads it;
// the lookup in front computed its step as ss->step + 1, everything behind sees the advanced state) - one launch less per step
template <typename IdT, int kChunk>
__global__ __launch_bounds__(256) void chunk_rank_kernel(IdxJobs jobs, int64_t n, int n_chunks, const StepAdvance adv) {      // (n: the longer stream's keys)
  if (adv.st && blockIdx.x == gridDim.x - 1) {
    if (blockIdx.y == 0) step_state_advance_block(adv);
    return;
  }
  const IdxJob& job = jobs.j[blockIdx.y];
  if (job.n) { n = job.n; n_chunks = job.n_chunks; }
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int c = (int)(e / kChunk);
  const uint32_t key = job.ck[e];
  uint32_t rank = (uint32_t)(e - (int64_t)c * kChunk);
  constexpr int G = 8;                                   // chunks searched together (independent probes in flight per step)
  for (int c0 = 0; c0 < n_chunks; c0 += G) {
    uint32_t lo[G], len[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int cc = c0 + u;
      const int64_t left = n - (int64_t)cc * kChunk;
      len[u] = (cc < n_chunks && cc != c) ? (uint32_t)(left < kChunk ? left : kChunk) : 0u;
      lo[u] = 0u;
    }
#pragma unroll
    for (int step = kChunk; step > 0; step >>= 1) {
      // the G probes of a step are issued first, then the G bounds move - as arithmetic, not predicated moves
      // (47 -> 38 us for the pair; more chunks per step, more lanes per element or an LDS splitter level were all slower)
      uint32_t v[G];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const uint32_t idx = lo[u] + (uint32_t)step;
        const uint32_t at = idx <= len[u] ? idx - 1u : 0u;                          // clamped probe, branch-free
        v[u] = job.ck[(int64_t)(c0 + u < n_chunks ? c0 + u : 0) * kChunk + at];
      }
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const uint32_t idx = lo[u] + (uint32_t)step;
        const uint32_t lt = (c0 + u < c) ? (v[u] <= key ? 1u : 0u) : (v[u] < key ? 1u : 0u);   // earlier chunk: ties sort before
        lo[u] += (idx <= len[u] ? lt : 0u) * (uint32_t)step;
      }
    }
#pragma unroll
    for (int u = 0; u < G; ++u) rank += lo[u];
  }
  ((IdT*)job.sorted_ids)[rank] = (IdT)key;
  job.sorted_pos[rank] = (int32_t)job.cp[e];
}

static bool rank_path_ok(int64_t n, int64_t upper) { return n <= kRankMaxN && upper > 0 && upper < ((int64_t)1 << 31) - 2; }

static int index_build_rank(IdxJobs& jobs, int n_jobs, int id_type, int64_t n, hipStream_t s) {
  const bool large = n > kChunkSwitchN;
  const int n_chunks = (int)ceil_div(n, large ? kChunkL : kChunkS);
  const dim3 g1((unsigned)n_chunks, (unsigned)n_jobs), g2((unsigned)ceil_div(n, 256), (unsigned)n_jobs);
  if (id_type == BR_IDS_I32) {
    if (large) { chunk_sort_kernel<int32_t, kChunkL, kThreadsL><<<g1, kThreadsL, 0, s>>>(jobs, n); probe_split(BR_TAG_INDEX_SORT, s); chunk_rank_kernel<int32_t, kChunkL><<<g2, 256, 0, s>>>(jobs, n, n_chunks, StepAdvance{}); }
    else { chunk_sort_kernel<int32_t, kChunkS, kThreadsS><<<g1, kThreadsS, 0, s>>>(jobs, n); probe_split(BR_TAG_INDEX_SORT, s); chunk_rank_kernel<int32_t, kChunkS><<<g2, 256, 0, s>>>(jobs, n, n_chunks, StepAdvance{}); }
  } else {
    if (large) { chunk_sort_kernel<int64_t, kChunkL, kThreadsL><<<g1, kThreadsL, 0, s>>>(jobs, n); probe_split(BR_TAG_INDEX_SORT, s); chunk_rank_kernel<int64_t, kChunkL><<<g2, 256, 0, s>>>(jobs, n, n_chunks, StepAdvance{}); }
    else { chunk_sort_kernel<int64_t, kChunkS, kThreadsS><<<g1, kThreadsS, 0, s>>>(jobs, n); probe_split(BR_TAG_INDEX_SORT, s); chunk_rank_kernel<int64_t, kChunkS><<<g2, 256, 0, s>>>(jobs, n, n_chunks, StepAdvance{}); }
  }
  BR_CHECK_LAUNCH("brRowIndexBuild");
  return BR_OK;
}
static IdxJob make_job(const void* ids, void* sorted_ids, int32_t* sorted_pos, void* workspace, int64_t n, int64_t upper) {
  IdxJob j{};
  j.ids = ids; j.sorted_ids = sorted_ids; j.sorted_pos = sorted_pos;
  j.ck = (uint32_t*)workspace;
  j.cp = (uint32_t*)((char*)workspace + align256(n * 4));
  j.upper = (uint32_t)upper;
  j.end_bit = bits_for(upper + 2);
  return j;
}

extern "C" int64_t brRowIndexWorkspaceBytes(int64_t n, int id_type) {
  if (n <= 0) return 256;
  const int64_t iota = align256(n * 4);
  const int64_t tmp = id_type == BR_IDS_I64 ? sort_temp_bytes<int64_t>(n) : sort_temp_bytes<int32_t>(n);
  const int64_t rank = 2 * align256(n * 4);       // chunk-sorted keys + positions (small-batch path)
  const int64_t sort = iota + align256(n * 8) + align256(tmp);
  return (sort > rank ? sort : rank) + 256;
}

extern "C" int brRowIndexBuild(const void* ids, int id_type, int64_t n, int64_t id_upper_bound, void* sorted_ids,
                               int32_t* sorted_pos, void* workspace, int64_t workspace_bytes, brStream stream) {
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brRowIndexBuild: bad id_type");
  BR_CHECK_ARG(n >= 0 && n < ((int64_t)1 << 31), "brRowIndexBuild: n out of range");
  if (n == 0) return BR_OK;
  BR_CHECK_ARG(ids && sorted_ids && sorted_pos && workspace, "brRowIndexBuild: null pointer");
  const int64_t need = brRowIndexWorkspaceBytes(n, id_type);
  if (workspace_bytes < need) {
    set_error("brRowIndexBuild: workspace %lld < required %lld", (long long)workspace_bytes, (long long)need);
    return BR_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  if (rank_path_ok(n, id_upper_bound)) {
    IdxJobs jobs;
    jobs.j[0] = jobs.j[1] = make_job(ids, sorted_ids, sorted_pos, workspace, n, id_upper_bound);
    return index_build_rank(jobs, 1, id_type, n, s);
  }
  int32_t* iota = (int32_t*)workspace;
  void* keys = (char*)workspace + align256(n * 4);
  void* tmp = (char*)keys + align256(n * 8);
  size_t tmp_bytes = (size_t)(workspace_bytes - align256(n * 4) - align256(n * 8));
  const int end_bit_cap = (id_type == BR_IDS_I64 ? 64 : 32);
  int end_bit = id_upper_bound > 0 ? bits_for(id_upper_bound + 1) : end_bit_cap;
  if (end_bit > end_bit_cap) end_bit = end_bit_cap;
  hipError_t e;
  if (id_type == BR_IDS_I64) {
    sort_prep_kernel<int64_t><<<(unsigned)ceil_div(n, 256), 256, 0, s>>>((const int64_t*)ids, id_upper_bound, (int64_t*)keys, iota, n);
    e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, (const int64_t*)keys, (int64_t*)sorted_ids, (const int32_t*)iota,
                                           sorted_pos, (int)n, 0, end_bit, s);
  } else {
    sort_prep_kernel<int32_t><<<(unsigned)ceil_div(n, 256), 256, 0, s>>>((const int32_t*)ids, id_upper_bound, (int32_t*)keys, iota, n);
    e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, (const int32_t*)keys, (int32_t*)sorted_ids, (const int32_t*)iota,
                                           sorted_pos, (int)n, 0, end_bit, s);
  }
  if (e != hipSuccess) {
    set_error("brRowIndexBuild: radix sort failed: %s", hipGetErrorString(e));
    return BR_ERR_HIP;
  }
  BR_CHECK_LAUNCH("brRowIndexBuild");
  return BR_OK;
}

extern "C" int brRowIndexBuildPair(const void* ids_a, int64_t upper_a, void* sorted_ids_a, int32_t* sorted_pos_a, void* ws_a, int64_t ws_a_bytes,
                                   const void* ids_b, int64_t upper_b, void* sorted_ids_b, int32_t* sorted_pos_b, void* ws_b, int64_t ws_b_bytes,
                                   int id_type, int64_t n, brStream stream) {
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brRowIndexBuildPair: bad id_type");
  if (n == 0) return BR_OK;
  if (rank_path_ok(n, upper_a) && rank_path_ok(n, upper_b)) {
    BR_CHECK_ARG(ids_a && ids_b && sorted_ids_a && sorted_ids_b && sorted_pos_a && sorted_pos_b && ws_a && ws_b, "brRowIndexBuildPair: null pointer");
    const int64_t need = brRowIndexWorkspaceBytes(n, id_type);
    if (ws_a_bytes < need || ws_b_bytes < need) {
      set_error("brRowIndexBuildPair: workspace < required %lld", (long long)need);
      return BR_ERR_WORKSPACE;
    }
    IdxJobs jobs;
    jobs.j[0] = make_job(ids_a, sorted_ids_a, sorted_pos_a, ws_a, n, upper_a);
    jobs.j[1] = make_job(ids_b, sorted_ids_b, sorted_pos_b, ws_b, n, upper_b);
    return index_build_rank(jobs, 2, id_type, n, (hipStream_t)stream);
  }
  const int rc = brRowIndexBuild(ids_a, id_type, n, upper_a, sorted_ids_a, sorted_pos_a, ws_a, ws_a_bytes, stream);
  return rc != BR_OK ? rc : brRowIndexBuild(ids_b, id_type, n, upper_b, sorted_ids_b, sorted_pos_b, ws_b, ws_b_bytes, stream);
}

// The same pair of indexes over SEGMENTED id arrays: both streams of a row-sharded step arrive in ONE all-to-all buffer laid out
// [source rank][stream][cap] (parallel.py PaddedExchange), so stream k's logical position t = src * cap + j sits at element
// (src * 2 + k) * cap + j.  sorted_pos holds that physical element index: the owner's optimizer reads the received gradient rows (same
// layout, one buffer for both streams) by it.  Stable in logical = physical order.
extern "C" int brRowIndexBuildPairSeg(const void* ids_a, int64_t upper_a, void* sorted_ids_a, int32_t* sorted_pos_a, void* ws_a, int64_t ws_a_bytes,
                                      const void* ids_b, int64_t upper_b, void* sorted_ids_b, int32_t* sorted_pos_b, void* ws_b, int64_t ws_b_bytes,
                                      int id_type, int64_t n, int64_t seg_len, int64_t seg_stride, int64_t seg_off_a, int64_t seg_off_b, brStream stream) {
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brRowIndexBuildPairSeg: bad id_type");
  BR_CHECK_ARG(seg_len >= 1 && seg_stride >= seg_len && seg_off_a >= 0 && seg_off_b >= 0, "brRowIndexBuildPairSeg: bad segment geometry");
  if (n == 0) return BR_OK;
  BR_CHECK_ARG(ids_a && ids_b && sorted_ids_a && sorted_ids_b && sorted_pos_a && sorted_pos_b && ws_a && ws_b, "brRowIndexBuildPairSeg: null pointer");
  BR_CHECK_ARG(rank_path_ok(n, upper_a) && rank_path_ok(n, upper_b) && seg_phys(n - 1, seg_len, seg_stride, seg_off_a > seg_off_b ? seg_off_a : seg_off_b) < ((int64_t)1 << 31),
               "brRowIndexBuildPairSeg: n <= %lld positions and id bounds < 2^31 - 2", (long long)kRankMaxN);
  const int64_t need = brRowIndexWorkspaceBytes(n, id_type);
  if (ws_a_bytes < need || ws_b_bytes < need) { set_error("brRowIndexBuildPairSeg: workspace < required %lld", (long long)need); return BR_ERR_WORKSPACE; }
  IdxJobs jobs;
  jobs.j[0] = make_job(ids_a, sorted_ids_a, sorted_pos_a, ws_a, n, upper_a);
  jobs.j[1] = make_job(ids_b, sorted_ids_b, sorted_pos_b, ws_b, n, upper_b);
  jobs.j[0].seg_len = jobs.j[1].seg_len = seg_len; jobs.j[0].seg_stride = jobs.j[1].seg_stride = seg_stride;
  jobs.j[0].seg_off = seg_off_a; jobs.j[1].seg_off = seg_off_b;
  return index_build_rank(jobs, 2, id_type, n, (hipStream_t)stream);
}

// ---- the same index when every segment of the array is ALREADY sorted ascending (the fixed-capacity exchange: a requester sends each
// owner its distinct local rows in key order, pads = the spare row = the largest id, behind them): the W segments of a stream are W sorted
// runs, so the index is their merge - rank(e) = position in its own run + the number of smaller keys (earlier runs: smaller or equal) in
// every other run, one binary search each - and the 35 us chunk sort of brRowIndexBuildPairSeg is not needed.  Output identical to it
// (stable in logical order).  A run that is not sorted sets BR_ERRFLAG_RANGE in *err_flag (the index is then wrong).
template <typename IdT>
__global__ __launch_bounds__(256) void run_rank_kernel(IdxJobs jobs, int64_t n, int64_t run_len, int n_runs, int steps, int* err) {
  const IdxJob& job = jobs.j[blockIdx.y];
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const IdT* __restrict__ ids = (const IdT*)job.ids;
  auto key_at = [&](int64_t t) -> uint32_t {
    const int64_t id = (int64_t)ids[seg_phys(t, job.seg_len, job.seg_stride, job.seg_off)];
    return ((uint64_t)id < (uint64_t)job.upper) ? (uint32_t)id : job.upper;
  };
  const int r = (int)(e / run_len);
  const int64_t i = e - (int64_t)r * run_len;
  const uint32_t key = key_at(e);
  if (i > 0 && key_at(e - 1) > key && err) atomicOr(err, BR_ERRFLAG_RANGE);
  uint32_t rank = (uint32_t)i;
  constexpr int G = 8;                                   // runs searched together (independent probes in flight per step)
  for (int r0 = 0; r0 < n_runs; r0 += G) {
    uint32_t lo[G], len[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int rr = r0 + u;
      const int64_t left = n - (int64_t)rr * run_len;
      len[u] = (rr < n_runs && rr != r) ? (uint32_t)(left < run_len ? left : run_len) : 0u;
      lo[u] = 0u;
    }
    for (int st = steps - 1; st >= 0; --st) {
      const uint32_t step = 1u << st;
      uint32_t v[G];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const uint32_t idx = lo[u] + step;
        const uint32_t at = idx <= len[u] ? idx - 1u : 0u;                          // clamped probe, branch-free
        v[u] = key_at((int64_t)(r0 + u < n_runs ? r0 + u : 0) * run_len + at);
      }
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const uint32_t idx = lo[u] + step;
        const uint32_t lt = (r0 + u < r) ? (v[u] <= key ? 1u : 0u) : (v[u] < key ? 1u : 0u);   // earlier run: ties sort before
        lo[u] += (idx <= len[u] ? lt : 0u) * step;
      }
    }
#pragma unroll
    for (int u = 0; u < G; ++u) rank += lo[u];
  }
  ((IdT*)job.sorted_ids)[rank] = (IdT)key;
  job.sorted_pos[rank] = (int32_t)seg_phys(e, job.seg_len, job.seg_stride, job.seg_off);
}

extern "C" int brRowIndexMergePairSeg(const void* ids_a, int64_t upper_a, void* sorted_ids_a, int32_t* sorted_pos_a, const void* ids_b, int64_t upper_b,
                                      void* sorted_ids_b, int32_t* sorted_pos_b, int id_type, int64_t n, int64_t seg_len, int64_t seg_stride, int64_t seg_off_a,
                                      int64_t seg_off_b, int* err_flag, brStream stream) {
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brRowIndexMergePairSeg: bad id_type");
  BR_CHECK_ARG(seg_len >= 1 && seg_stride >= seg_len && seg_off_a >= 0 && seg_off_b >= 0, "brRowIndexMergePairSeg: bad segment geometry");
  if (n == 0) return BR_OK;
  BR_CHECK_ARG(ids_a && ids_b && sorted_ids_a && sorted_ids_b && sorted_pos_a && sorted_pos_b, "brRowIndexMergePairSeg: null pointer");
  BR_CHECK_ARG(upper_a > 0 && upper_b > 0 && upper_a < ((int64_t)1 << 31) - 2 && upper_b < ((int64_t)1 << 31) - 2 && n < ((int64_t)1 << 31) &&
                   seg_phys(n - 1, seg_len, seg_stride, seg_off_a > seg_off_b ? seg_off_a : seg_off_b) < ((int64_t)1 << 31),
               "brRowIndexMergePairSeg: positions and id bounds < 2^31 - 2");
  IdxJobs jobs;
  jobs.j[0].ids = ids_a; jobs.j[0].sorted_ids = sorted_ids_a; jobs.j[0].sorted_pos = sorted_pos_a; jobs.j[0].upper = (uint32_t)upper_a;
  jobs.j[1].ids = ids_b; jobs.j[1].sorted_ids = sorted_ids_b; jobs.j[1].sorted_pos = sorted_pos_b; jobs.j[1].upper = (uint32_t)upper_b;
  jobs.j[0].ck = jobs.j[1].ck = nullptr; jobs.j[0].cp = jobs.j[1].cp = nullptr; jobs.j[0].end_bit = jobs.j[1].end_bit = 0;
  jobs.j[0].seg_len = jobs.j[1].seg_len = seg_len; jobs.j[0].seg_stride = jobs.j[1].seg_stride = seg_stride;
  jobs.j[0].seg_off = seg_off_a; jobs.j[1].seg_off = seg_off_b;
  const int n_runs = (int)ceil_div(n, seg_len);
  int steps = 0;
  while (((int64_t)1 << steps) <= seg_len) ++steps;       // 2^steps > seg_len: the search covers every length <= seg_len
  const dim3 grid((unsigned)ceil_div(n, 256), 2);
  hipStream_t s = (hipStream_t)stream;
  if (id_type == BR_IDS_I32) run_rank_kernel<int32_t><<<grid, 256, 0, s>>>(jobs, n, seg_len, n_runs, steps, err_flag);
  else run_rank_kernel<int64_t><<<grid, 256, 0, s>>>(jobs, n, seg_len, n_runs, steps, err_flag);
  BR_CHECK_LAUNCH("brRowIndexMergePairSeg");
  return BR_OK;
}

static bool wave_rows_enabled();
// neumf_step.cpp: lookup + both dedup indexes on one stream (lookup_sort_kernel, then the chunk-rank launch).  supported(): the wave
// lookup's shapes, the large-chunk sort's range, BR_FUSED_SORT != 0.
bool br::lookup_with_index_supported(int dim, int64_t n, int64_t upper_a, int64_t upper_b, int64_t ld_stash, const void* x0, const void* stash_a,
                                     const void* stash_b) {
  static const bool on = [] { const char* e = getenv("BR_FUSED_SORT"); return !(e && e[0] == '0'); }();
  const int wvec = dim / 32;
  return on && wave_rows_enabled() && dim % 32 == 0 && (wvec == 2 || wvec == 4) && ld_stash % wvec == 0 &&
         ((reinterpret_cast<uintptr_t>(x0) | reinterpret_cast<uintptr_t>(stash_a) | reinterpret_cast<uintptr_t>(stash_b)) & (4 * wvec - 1)) == 0 &&
         n > kChunkSwitchN && rank_path_ok(n, upper_a) && rank_path_ok(n, upper_b);
}
int br::lookup_with_index(const LookupArgs& la, int dim, int id_type, const IndexPairArgs& ix, brStream stream, const StepAdvance* adv) {
  const int64_t n = la.batch;
  BR_CHECK_ARG(ix.sorted_ids_a && ix.sorted_ids_b && ix.sorted_pos_a && ix.sorted_pos_b && ix.ws_a && ix.ws_b, "lookup_with_index: null pointer");
  const int64_t need = brRowIndexWorkspaceBytes(n, id_type);
  if (ix.ws_a_bytes < need || ix.ws_b_bytes < need) { set_error("lookup_with_index: workspace < required %lld", (long long)need); return BR_ERR_WORKSPACE; }
  IdxJobs jobs;
  jobs.j[0] = make_job(la.users, ix.sorted_ids_a, ix.sorted_pos_a, ix.ws_a, n, la.user_rows);
  jobs.j[1] = make_job(la.items, ix.sorted_ids_b, ix.sorted_pos_b, ix.ws_b, n, la.item_rows);
  const int n_chunks = (int)ceil_div(n, kChunkL);
  hipStream_t s = (hipStream_t)stream;
  const int wvec = dim / 32;
  const int ppw = wvec == 2 ? lookup_pairs_per_wave() : 1;
  const unsigned grid = (unsigned)(2 * n_chunks + ceil_div(ceil_div(n, (int64_t)(ppw < 1 ? 1 : ppw)), (int64_t)(kThreadsL / 64)));
  if (id_type == BR_IDS_I32) {
    if (wvec == 2 && ppw == 0) lookup_sort_kernel<int32_t, 2, 0><<<grid, kThreadsL, 0, s>>>(la, jobs, n, n_chunks);
    else if (wvec == 2 && ppw == 2) lookup_sort_kernel<int32_t, 2, 2><<<grid, kThreadsL, 0, s>>>(la, jobs, n, n_chunks);
    else if (wvec == 2) lookup_sort_kernel<int32_t, 2, 1><<<grid, kThreadsL, 0, s>>>(la, jobs, n, n_chunks);
    else lookup_sort_kernel<int32_t, 4, 1><<<grid, kThreadsL, 0, s>>>(la, jobs, n, n_chunks);
  } else {
    if (wvec == 2 && ppw == 0) lookup_sort_kernel<int64_t, 2, 0><<<grid, kThreadsL, 0, s>>>(la, jobs, n, n_chunks);
    else if (wvec == 2 && ppw == 2) lookup_sort_kernel<int64_t, 2, 2><<<grid, kThreadsL, 0, s>>>(la, jobs, n, n_chunks);
    else if (wvec == 2) lookup_sort_kernel<int64_t, 2, 1><<<grid, kThreadsL, 0, s>>>(la, jobs, n, n_chunks);
    else lookup_sort_kernel<int64_t, 4, 1><<<grid, kThreadsL, 0, s>>>(la, jobs, n, n_chunks);
  }
  BR_CHECK_LAUNCH("lookup_with_index(lookup + sort)");
  probe_split(BR_TAG_EMBED_FWD, s);
  const StepAdvance av = adv ? *adv : StepAdvance{};
  const dim3 g2((unsigned)(ceil_div(n, 256) + (av.st ? 1 : 0)), 2);
  if (id_type == BR_IDS_I32) chunk_rank_kernel<int32_t, kChunkL><<<g2, 256, 0, s>>>(jobs, n, n_chunks, av);
  else chunk_rank_kernel<int64_t, kChunkL><<<g2, 256, 0, s>>>(jobs, n, n_chunks, av);
  BR_CHECK_LAUNCH("lookup_with_index(rank)");
  return BR_OK;
}

extern "C" int brGatherRowsDeferredPairWithIndex(const float* table_a, const float* m_a, const float* v_a, const int32_t* last_a, int64_t rows_a, const void* ids_a,
                                                 float* out_a, void* sorted_ids_a, int32_t* sorted_pos_a, void* ws_a, int64_t ws_a_bytes, const float* table_b,
                                                 const float* m_b, const float* v_b, const int32_t* last_b, int64_t rows_b, const void* ids_b, float* out_b,
                                                 void* sorted_ids_b, int32_t* sorted_pos_b, void* ws_b, int64_t ws_b_bytes, int dim, int id_type, int64_t n_a, int64_t n_b,
                                                 void* step_state, int advance, double lr, double beta1, double beta2, double eps, int64_t ld_out, int* err_flag,
                                                 brStream stream) {
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brGatherRowsDeferredPairWithIndex: bad id_type");
  BR_CHECK_ARG(table_a && m_a && v_a && last_a && ids_a && out_a && sorted_ids_a && sorted_pos_a && ws_a && table_b && m_b && v_b && last_b && ids_b && out_b &&
                   sorted_ids_b && sorted_pos_b && ws_b && step_state && rows_a > 0 && rows_b > 0 && n_a > 0 && n_b > 0 && ld_out >= dim,
               "brGatherRowsDeferredPairWithIndex: bad args");
  const int wvec = dim / 64;
  BR_CHECK_ARG(dim % 64 == 0 && (wvec == 1 || wvec == 2 || wvec == 4) && ld_out % wvec == 0 &&
                   ((reinterpret_cast<uintptr_t>(out_a) | reinterpret_cast<uintptr_t>(out_b)) & (4 * wvec - 1)) == 0,
               "brGatherRowsDeferredPairWithIndex: rows of 64 / 128 / 256 floats (one wave per row)");
  BR_CHECK_ARG(rank_path_ok(n_a, rows_a) && rank_path_ok(n_b, rows_b), "brGatherRowsDeferredPairWithIndex: at most %lld ids per stream, table rows < 2^31 - 2", (long long)kRankMaxN);
  if (ws_a_bytes < brRowIndexWorkspaceBytes(n_a, id_type) || ws_b_bytes < brRowIndexWorkspaceBytes(n_b, id_type)) {
    set_error("brGatherRowsDeferredPairWithIndex: index workspace too small");
    return BR_ERR_WORKSPACE;
  }
  GatherDefJobs G;
  G.j[0] = GatherDefJob{table_a, m_a, v_a, last_a, rows_a, ids_a, out_a, n_a};
  G.j[1] = GatherDefJob{table_b, m_b, v_b, last_b, rows_b, ids_b, out_b, n_b};
  G.step_add = advance ? 1u : 0u;
  IdxJobs jobs;
  jobs.j[0] = make_job(ids_a, sorted_ids_a, sorted_pos_a, ws_a, n_a, rows_a);
  jobs.j[1] = make_job(ids_b, sorted_ids_b, sorted_pos_b, ws_b, n_b, rows_b);
  const int ca = (int)ceil_div(n_a, kChunkL), cb = (int)ceil_div(n_b, kChunkL);
  jobs.j[0].n = n_a; jobs.j[0].n_chunks = ca; jobs.j[1].n = n_b; jobs.j[1].n_chunks = cb;
  const AdamHp h = make_hp(0.0, beta1, beta2, eps);
  StepStateDev* ss = (StepStateDev*)step_state;
  hipStream_t s = (hipStream_t)stream;
  const int rpw = wvec == 1 ? 4 : (wvec == 2 ? 2 : 1);      // kGatherRowsPerWave
  static const bool g4env = [] { const char* e = getenv("BR_GATHER_GROUP4"); return !(e && e[0] == '0'); }();
  const bool g4 = g4env && ((reinterpret_cast<uintptr_t>(table_a) | reinterpret_cast<uintptr_t>(table_b) | reinterpret_cast<uintptr_t>(m_a) | reinterpret_cast<uintptr_t>(m_b) |
                             reinterpret_cast<uintptr_t>(v_a) | reinterpret_cast<uintptr_t>(v_b) | reinterpret_cast<uintptr_t>(out_a) | reinterpret_cast<uintptr_t>(out_b)) & 15) == 0;
  const unsigned grid = (unsigned)(ca + cb + ceil_div(ceil_div(n_a + n_b, (int64_t)rpw), (int64_t)(kThreadsL / 64)));
  if (id_type == BR_IDS_I32)
    BR_DISPATCH_VEC(wvec, (gather_sort_kernel<int32_t, VEC><<<grid, kThreadsL, 0, s>>>(G, ss, h, ld_out, err_flag, jobs, ca, cb, g4)));
  else
    BR_DISPATCH_VEC(wvec, (gather_sort_kernel<int64_t, VEC><<<grid, kThreadsL, 0, s>>>(G, ss, h, ld_out, err_flag, jobs, ca, cb, g4)));
  BR_CHECK_LAUNCH("brGatherRowsDeferredPairWithIndex(gather + sort)");
  StepAdvance av;
  if (advance) { av.st = ss; av.lr = lr; av.b1 = beta1; av.b2 = beta2; }
  const int64_t nmax = n_a > n_b ? n_a : n_b;
  const dim3 g2((unsigned)(ceil_div(nmax, 256) + (advance ? 1 : 0)), 2);
  if (id_type == BR_IDS_I32) chunk_rank_kernel<int32_t, kChunkL><<<g2, 256, 0, s>>>(jobs, nmax, ca > cb ? ca : cb, av);
  else chunk_rank_kernel<int64_t, kChunkL><<<g2, 256, 0, s>>>(jobs, nmax, ca > cb ? ca : cb, av);
  BR_CHECK_LAUNCH("brGatherRowsDeferredPairWithIndex(rank)");
  return BR_OK;
}

extern "C" int64_t brSegmentScratchFloats(int64_t n, int dim) { return ceil_div(n > 0 ? n : 1, kSegBlock) * (int64_t)dim; }

// partial rows of every kSegBlock-aligned block that continues a segment (see seg_acc); jobs share n / dim / split
static int launch_partials(const SegJob* jobs, int n_jobs, int id_type, int64_t n, int dim, const RowGeom& g, int split, hipStream_t s) {
  const int64_t blocks = ceil_div(n, kSegBlock) - 1;
  if (blocks <= 0) return BR_OK;
  SegJobs J;
  for (int q = 0; q < 2; ++q) J.j[q] = jobs[q < n_jobs ? q : 0];
  const int wvec = dim / 64;
  if (wave_rows_enabled() && dim % 64 == 0 && (wvec == 1 || wvec == 2 || wvec == 4) && g.vec >= wvec) {     // (g.vec: the widest vector the sources' strides honour)
    const dim3 wgrid((unsigned)ceil_div(blocks, 4), (unsigned)n_jobs);
    if (id_type == BR_IDS_I32)
      BR_DISPATCH_VEC(wvec, (segment_partials_wave_kernel<int32_t, VEC><<<wgrid, 256, 0, s>>>(J, n, split)));
    else
      BR_DISPATCH_VEC(wvec, (segment_partials_wave_kernel<int64_t, VEC><<<wgrid, 256, 0, s>>>(J, n, split)));
    BR_CHECK_LAUNCH("segment partials (wave)");
    return BR_OK;
  }
  const dim3 grid((unsigned)ceil_div(blocks, 256 >> g.lpr_log2), (unsigned)n_jobs);
  if (id_type == BR_IDS_I32)
    BR_DISPATCH_VEC(g.vec, (segment_partials_kernel<int32_t, VEC><<<grid, 256, 0, s>>>(J, n, dim, g.chunks, g.lpr_log2, split)));
  else
    BR_DISPATCH_VEC(g.vec, (segment_partials_kernel<int64_t, VEC><<<grid, 256, 0, s>>>(J, n, dim, g.chunks, g.lpr_log2, split)));
  BR_CHECK_LAUNCH("segment partials");
  return BR_OK;
}

extern "C" int brSegmentSumRows(const void* sorted_ids, int id_type, const int32_t* sorted_pos, int64_t n,
                                const float* row_grads, int64_t ldg, int dim, float* out_rows, int32_t* head_flag,
                                float* seg_ws, brStream stream) {
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brSegmentSumRows: bad id_type");
  if (n == 0) return BR_OK;
  BR_CHECK_ARG(sorted_ids && sorted_pos && row_grads && out_rows && dim >= 1 && ldg >= dim, "brSegmentSumRows: bad args");
  const RowGeom g = row_geom_ld(dim, ldg);
  const unsigned grid = (unsigned)ceil_div(n, 256 >> g.lpr_log2);
  hipStream_t s = (hipStream_t)stream;
  if (seg_ws) {
    const SegJob job{sorted_ids, sorted_pos, row_grads, ldg, row_grads, ldg, seg_ws};
    const int rc = launch_partials(&job, 1, id_type, n, dim, g, dim, s);
    if (rc != BR_OK) return rc;
  }
  if (id_type == BR_IDS_I32)
    BR_DISPATCH_VEC(g.vec, (segment_sum_kernel<int32_t, VEC><<<grid, 256, 0, s>>>((const int32_t*)sorted_ids, sorted_pos, n, row_grads,
                                                                                   ldg, dim, g.chunks, g.lpr_log2, out_rows, head_flag, seg_ws)));
  else
    BR_DISPATCH_VEC(g.vec, (segment_sum_kernel<int64_t, VEC><<<grid, 256, 0, s>>>((const int64_t*)sorted_ids, sorted_pos, n, row_grads,
                                                                                   ldg, dim, g.chunks, g.lpr_log2, out_rows, head_flag, seg_ws)));
  BR_CHECK_LAUNCH("brSegmentSumRows");
  return BR_OK;
}

// ---- requester side of the row-sharded exchange: per-unique-id gradient sums laid into the send slots -------------------------------
// One row group per sorted position of the (owner, id)-sorted index of brShardDedupPlanPair; a head sums its run (two-level order,
// as everywhere) straight from the step's gradient sources - columns [0, split) from g0 (the MLP half of dx0), [split, dim) from g1
// times sc1[position] (the partner's stashed MF row times ddot) - and stores the sum at row slot[position of the head] of the merged
// slot buffer.  Heads whose id found no slot (capacity overflow, id out of range: slot < 0) store nothing.
struct SlotSumJob {
  const void* sid; const int32_t* spos; const int32_t* slot;
  const float* g0; const float* g1; const float* part;
};
struct SlotSumJobs { SlotSumJob j[2]; };
template <typename IdT, int VEC>
__global__ __launch_bounds__(256) void segment_sum_to_slots_kernel(SlotSumJobs jobs, int64_t n, int64_t ldg0, int64_t ldg1, const float* __restrict__ sc1, int dim,
                                                                    int chunks, int lpr_log2, int split, float* __restrict__ out) {
  using V = typename VecT<VEC>::type;
  const SlotSumJob& jb = jobs.j[blockIdx.y];
  const IdT* __restrict__ sid = (const IdT*)jb.sid;
  const int32_t* __restrict__ spos = jb.spos;
  const int lpr = 1 << lpr_log2;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = tid >> lpr_log2;
  const int lir = (int)(tid & (lpr - 1));
  if (i >= n) return;
  if (!segment_head(sid, i)) return;
  const int64_t sl = (int64_t)jb.slot[spos[i]];
  if (sl < 0) return;
  const IdT row = sid[i];
  for (int c = lir; c < chunks; c += lpr) {
    const int col = c * VEC;
    const bool lo = col < split;
    const float* g = lo ? jb.g0 + col : jb.g1 + (col - split);
    const V acc = seg_acc<IdT, VEC>(sid, spos, n, i, row, g, lo ? ldg0 : ldg1, lo ? nullptr : sc1, jb.part ? jb.part + col : nullptr, dim);
    vstore<VEC>(out + sl * dim + col, acc);
  }
}

extern "C" int brSegmentSumToSlotsPair(const void* sorted_ids_a, const int32_t* sorted_pos_a, const int32_t* slot_a, const float* g0_a, const float* g1_a,
                                       const void* sorted_ids_b, const int32_t* sorted_pos_b, const int32_t* slot_b, const float* g0_b, const float* g1_b,
                                       int64_t ldg0, int64_t ldg1, const float* hi_scale, int id_type, int64_t n, int dim, int split, float* out_slots,
                                       float* seg_ws_a, float* seg_ws_b, brStream stream) {
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brSegmentSumToSlotsPair: bad id_type");
  if (n == 0) return BR_OK;
  BR_CHECK_ARG(sorted_ids_a && sorted_pos_a && slot_a && g0_a && sorted_ids_b && sorted_pos_b && slot_b && g0_b && out_slots && dim >= 1, "brSegmentSumToSlotsPair: null pointer");
  if (!g1_a || !g1_b) { split = dim; g1_a = g0_a; g1_b = g0_b; ldg1 = ldg0; hi_scale = nullptr; }
  BR_CHECK_ARG(split >= 1 && split <= dim && ldg0 >= split && ldg1 >= dim - split, "brSegmentSumToSlotsPair: bad split / strides");
  BR_CHECK_ARG((seg_ws_a == nullptr) == (seg_ws_b == nullptr), "brSegmentSumToSlotsPair: seg_ws for both streams or neither");
  const uintptr_t al = reinterpret_cast<uintptr_t>(g0_a) | reinterpret_cast<uintptr_t>(g1_a) | reinterpret_cast<uintptr_t>(g0_b) | reinterpret_cast<uintptr_t>(g1_b) |
                       reinterpret_cast<uintptr_t>(out_slots);
  const int64_t lm = (ldg0 % 4 == 0 && ldg1 % 4 == 0 && split % 4 == 0 && dim % 4 == 0 && (al & 15) == 0) ? 4
                     : (ldg0 % 2 == 0 && ldg1 % 2 == 0 && split % 2 == 0 && dim % 2 == 0 && (al & 7) == 0) ? 2 : 1;
  const RowGeom g = row_geom_ld(dim, lm);
  hipStream_t s = (hipStream_t)stream;
  if (seg_ws_a) {
    const SegJob segs[2] = {SegJob{sorted_ids_a, sorted_pos_a, g0_a, ldg0, g1_a, ldg1, seg_ws_a, hi_scale}, SegJob{sorted_ids_b, sorted_pos_b, g0_b, ldg0, g1_b, ldg1, seg_ws_b, hi_scale}};
    const int rc = launch_partials(segs, 2, id_type, n, dim, g, split, s);
    if (rc != BR_OK) return rc;
  }
  SlotSumJobs J;
  J.j[0] = SlotSumJob{sorted_ids_a, sorted_pos_a, slot_a, g0_a, g1_a, seg_ws_a};
  J.j[1] = SlotSumJob{sorted_ids_b, sorted_pos_b, slot_b, g0_b, g1_b, seg_ws_b};
  const dim3 grid((unsigned)ceil_div(n, 256 >> g.lpr_log2), 2);
  if (id_type == BR_IDS_I32)
    BR_DISPATCH_VEC(g.vec, (segment_sum_to_slots_kernel<int32_t, VEC><<<grid, 256, 0, s>>>(J, n, ldg0, ldg1, hi_scale, dim, g.chunks, g.lpr_log2, split, out_slots)));
  else
    BR_DISPATCH_VEC(g.vec, (segment_sum_to_slots_kernel<int64_t, VEC><<<grid, 256, 0, s>>>(J, n, ldg0, ldg1, hi_scale, dim, g.chunks, g.lpr_log2, split, out_slots)));
  BR_CHECK_LAUNCH("brSegmentSumToSlotsPair");
  return BR_OK;
}

extern "C" int brScatterAddRows(float* g_table, int64_t table_rows, const void* ids, int id_type, int64_t n,
                                const float* rows, int dim, int* err_flag, brStream stream) {
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brScatterAddRows: bad id_type");
  if (n == 0) return BR_OK;
  BR_CHECK_ARG(g_table && rows && dim >= 1 && table_rows > 0, "brScatterAddRows: bad args");
  const unsigned grid = (unsigned)ceil_div(n * dim, 256);
  hipStream_t s = (hipStream_t)stream;
  if (id_type == BR_IDS_I32)
    scatter_add_kernel<int32_t><<<grid, 256, 0, s>>>(g_table, table_rows, (const int32_t*)ids, n, rows, dim, err_flag);
  else
    scatter_add_kernel<int64_t><<<grid, 256, 0, s>>>(g_table, table_rows, (const int64_t*)ids, n, rows, dim, err_flag);
  BR_CHECK_LAUNCH("brScatterAddRows");
  return BR_OK;
}

struct AdamRowsArgs {      // one table's host-side arguments
  float *table, *m, *v;
  int64_t table_rows;
  const void* sorted_ids; const int32_t* sorted_pos;
  const float* row_grads; int64_t ldg;
  const float* row_grads_hi; int64_t ldg_hi;
  uint8_t* mark; int32_t* last;
  float* seg_ws;
  const float* hi_scale = nullptr;     // per-position factor of the row_grads_hi rows
  const float* th_lo = nullptr; const float* th_hi = nullptr; int64_t ld_th = 0;   // replayed theta by position (AdamRowsJob::th0 / th1)
  int64_t n = 0;              // positions of this table when the tables of a launch differ in length (wave kernel only)
};

// BR_WAVE_ROWS=0: keep the row-group kernels (A/B runs)
static bool wave_rows_enabled() {
  static const bool on = [] { const char* e = getenv("BR_WAVE_ROWS"); return !(e && e[0] == '0'); }();
  return on;
}

static int adam_rows_launch(const AdamRowsArgs* a, int n_jobs, int dim, int id_type, int64_t n, int split, double alpha_t, double beta1,
                            double beta2, double eps, const StepStateDev* ss, brStream stream, const KeepArgs* keep = nullptr,
                            const FinalArgs* fin = nullptr, bool* fin_done = nullptr) {
  if (fin_done) *fin_done = false;
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brAdamRowsSorted: bad id_type");
  if (n == 0) return BR_OK;
  AdamRowsJobs jobs;
  SegJob segs[2];
  bool with_partials = true, all_stashed = true, per_job_n = false;
  int64_t ldmin = 4, th_min = 4, n_max = n;
  for (int q = 0; q < n_jobs; ++q) {
    AdamRowsArgs t = a[q];
    BR_CHECK_ARG(t.table && t.m && t.v && t.sorted_ids && t.sorted_pos && t.row_grads && dim >= 1 && t.table_rows > 0, "brAdamRowsSorted: bad args");
    if (!t.row_grads_hi) { if (n_jobs == 1) split = dim; t.ldg_hi = t.ldg; t.row_grads_hi = t.row_grads; }
    BR_CHECK_ARG(split >= 1 && split <= dim && t.ldg >= split && t.ldg_hi >= dim - split, "brAdamRowsSorted: bad split / strides");
    // widest vector (floats) that every row start of both sources and the split honour
    const uintptr_t al = reinterpret_cast<uintptr_t>(t.row_grads) | reinterpret_cast<uintptr_t>(t.row_grads_hi);
    const int64_t lm = (t.ldg % 4 == 0 && t.ldg_hi % 4 == 0 && split % 4 == 0 && (al & 15) == 0) ? 4
                       : (t.ldg % 2 == 0 && t.ldg_hi % 2 == 0 && split % 2 == 0 && (al & 7) == 0) ? 2 : 1;
    ldmin = lm < ldmin ? lm : ldmin;
    jobs.j[q] = AdamRowsJob{t.table, t.m, t.v, t.table_rows, t.sorted_ids, t.sorted_pos, t.row_grads, t.ldg, t.row_grads_hi, t.ldg_hi, t.hi_scale,
                            t.seg_ws, t.mark, t.last};
    if (t.th_lo && t.th_hi && t.last) {
      const uintptr_t ta = reinterpret_cast<uintptr_t>(t.th_lo) | reinterpret_cast<uintptr_t>(t.th_hi);
      const int64_t tm = (t.ld_th % 4 == 0 && (ta & 15) == 0) ? 4 : (t.ld_th % 2 == 0 && (ta & 7) == 0) ? 2 : 1;
      th_min = tm < th_min ? tm : th_min;
      jobs.j[q].th0 = t.th_lo; jobs.j[q].ldt0 = t.ld_th; jobs.j[q].th1 = t.th_hi; jobs.j[q].ldt1 = t.ld_th;
    } else if (t.last) {
      all_stashed = false;
    }
    segs[q] = SegJob{t.sorted_ids, t.sorted_pos, t.row_grads, t.ldg, t.row_grads_hi, t.ldg_hi, t.seg_ws, t.hi_scale, t.n};
    jobs.j[q].n = t.n;
    if (t.n > n_max) n_max = t.n;
    per_job_n = per_job_n || (t.n != 0 && t.n != n);
    with_partials = with_partials && t.seg_ws != nullptr;
  }
  if (n_jobs == 1) jobs.j[1] = jobs.j[0];
  if (!with_partials)
    for (int q = 0; q < 2; ++q) jobs.j[q].part = nullptr;     // all tables or none
  const RowGeom g = row_geom_ld(dim, ldmin);
  const int wvec0 = dim / 64;
  const bool wave_ok = wave_rows_enabled() && dim % 64 == 0 && (wvec0 == 1 || wvec0 == 2 || wvec0 == 4) && ldmin >= wvec0 && th_min >= wvec0 && all_stashed;
  BR_CHECK_ARG(!per_job_n || wave_ok, "brAdamRowsSorted: tables of different lengths in one launch need the one-wave-per-row shapes");
  if (with_partials) {
    const int rc = launch_partials(segs, n_jobs, id_type, n_max, dim, g, split, (hipStream_t)stream);
    if (rc != BR_OK) return rc;
    probe_split(BR_TAG_SEG_PARTIALS, (hipStream_t)stream);
  }
  hipStream_t s = (hipStream_t)stream;
  AdamHp h = make_hp(alpha_t, beta1, beta2, eps);
  if (ss) h.alpha_ptr = &ss->alpha_t;
  probe_mark(s);
  // rows of 64 / 128 / 256 floats: one wave per row (deferred tables only with the lookup's replayed theta at hand)
  const int wvec = dim / 64;
  if (wave_rows_enabled() && dim % 64 == 0 && (wvec == 1 || wvec == 2 || wvec == 4) && ldmin >= wvec && th_min >= wvec && all_stashed) {
    static const int strip = [] { const char* e = getenv("BR_ADAM_STRIP"); return e ? atoi(e) : 4; }();      // knob: 2, 4 (default), 8 - measured at the bench config: 92 / 85 / 115 us with riders
    AdamRiders rd;
    rd.kf.total = 0; rd.kf.blocks[0] = rd.kf.blocks[1] = rd.kf.blocks[2] = 0; rd.kf.a = KeepArgs{};
    rd.n_final = 0;
    if (keep && keep->batch > 0) {
      rd.kf.a = *keep;
      for (int i = 0; i < keep->n_sites; ++i) { rd.kf.blocks[i] = (int)ceil_div(keep->batch * keep->s[i].kw, (int64_t)256); rd.kf.total += rd.kf.blocks[i]; }
    }
    if (fin) { rd.fin = *fin; rd.n_final = (int)ceil_div(fin->n, 64); } else { rd.fin = FinalArgs{}; }
    rd.total = rd.kf.total + rd.n_final;
    const int S = (strip == 2 || strip == 8) && wvec <= 2 ? strip : 4;
    rd.rows_x = (int)ceil_div(n_max, 4 * S);
    const unsigned wgrid = (unsigned)((int64_t)rd.rows_x * n_jobs + rd.total);
#define BR_ADAM_WAVE(IdT, S_) BR_DISPATCH_VEC(wvec, (adam_rows_wave_kernel<IdT, VEC, S_><<<wgrid, 256, 0, s>>>(jobs, n, split, h, ss, rd)))
    if (id_type == BR_IDS_I32) { if (S == 2) BR_ADAM_WAVE(int32_t, 2); else if (S == 8) BR_ADAM_WAVE(int32_t, 8); else BR_ADAM_WAVE(int32_t, 4); }
    else { if (S == 2) BR_ADAM_WAVE(int64_t, 2); else if (S == 8) BR_ADAM_WAVE(int64_t, 8); else BR_ADAM_WAVE(int64_t, 4); }
#undef BR_ADAM_WAVE
    BR_CHECK_LAUNCH("brAdamRowsSorted(wave)");
    if (fin_done) *fin_done = fin != nullptr;
    return BR_OK;
  }
  for (int q = 0; q < 2; ++q) { jobs.j[q].th0 = jobs.j[q].th1 = nullptr; }
  KeepFuse kf;
  kf.total = 0; kf.blocks[0] = kf.blocks[1] = kf.blocks[2] = 0;
  int keep_x = 0;
  if (keep && keep->batch > 0) {
    kf.a = *keep;
    for (int i = 0; i < keep->n_sites; ++i) { kf.blocks[i] = (int)ceil_div(keep->batch * keep->s[i].kw, (int64_t)256); kf.total += kf.blocks[i]; }
    keep_x = (int)ceil_div((int64_t)kf.total, (int64_t)n_jobs);
  } else {
    kf.a = KeepArgs{};
  }
  const dim3 grid((unsigned)(ceil_div(n, 256 >> g.lpr_log2) + keep_x), (unsigned)n_jobs);
  if (id_type == BR_IDS_I32)
    BR_DISPATCH_VEC(g.vec, (adam_rows_sorted_kernel<int32_t, VEC><<<grid, 256, 0, s>>>(jobs, dim, g.chunks, g.lpr_log2, n, split, h, ss, kf, keep_x)));
  else
    BR_DISPATCH_VEC(g.vec, (adam_rows_sorted_kernel<int64_t, VEC><<<grid, 256, 0, s>>>(jobs, dim, g.chunks, g.lpr_log2, n, split, h, ss, kf, keep_x)));
  BR_CHECK_LAUNCH("brAdamRowsSorted");
  return BR_OK;
}

extern "C" int brAdamRowsSorted(float* table, float* m, float* v, int64_t table_rows, int dim, const void* sorted_ids,
                                int id_type, const int32_t* sorted_pos, int64_t n, const float* row_grads, int64_t ldg,
                                const float* row_grads_hi, int64_t ldg_hi, int split, double alpha_t, double beta1,
                                double beta2, double eps, uint8_t* mark, float* seg_ws, brStream stream) {
  const AdamRowsArgs a{table, m, v, table_rows, sorted_ids, sorted_pos, row_grads, ldg, row_grads_hi, ldg_hi, mark, nullptr, seg_ws};
  return adam_rows_launch(&a, 1, dim, id_type, n, split, alpha_t, beta1, beta2, eps, nullptr, stream);
}

extern "C" int brAdamRowsSortedDeferred(float* table, float* m, float* v, int32_t* last, int64_t table_rows, int dim,
                                        const void* sorted_ids, int id_type, const int32_t* sorted_pos, int64_t n,
                                        const float* row_grads, int64_t ldg, const float* row_grads_hi, int64_t ldg_hi, int split,
                                        const void* step_state, double beta1, double beta2, double eps, float* seg_ws, brStream stream) {
  BR_CHECK_ARG(last && step_state, "brAdamRowsSortedDeferred: last / step_state missing");
  const AdamRowsArgs a{table, m, v, table_rows, sorted_ids, sorted_pos, row_grads, ldg, row_grads_hi, ldg_hi, nullptr, last, seg_ws};
  return adam_rows_launch(&a, 1, dim, id_type, n, split, 0.0, beta1, beta2, eps, (const StepStateDev*)step_state, stream);
}

extern "C" int brAdamRowsSortedDeferredReplayed(float* table, float* m, float* v, int32_t* last, int64_t table_rows, int dim,
                                                const void* sorted_ids, int id_type, const int32_t* sorted_pos, int64_t n,
                                                const float* row_grads, int64_t ldg, const float* row_grads_hi, int64_t ldg_hi, int split,
                                                const float* replayed_rows, int64_t ld_replayed, const void* step_state, double beta1,
                                                double beta2, double eps, float* seg_ws, brStream stream) {
  BR_CHECK_ARG(last && step_state, "brAdamRowsSortedDeferredReplayed: last / step_state missing");
  BR_CHECK_ARG(replayed_rows && ld_replayed >= dim, "brAdamRowsSortedDeferredReplayed: replayed rows missing or ld < dim");
  const int sp = row_grads_hi ? split : dim;
  AdamRowsArgs a{table, m, v, table_rows, sorted_ids, sorted_pos, row_grads, ldg, row_grads_hi, ldg_hi, nullptr, last, seg_ws};
  a.th_lo = replayed_rows; a.th_hi = replayed_rows + sp; a.ld_th = ld_replayed;
  return adam_rows_launch(&a, 1, dim, id_type, n, split, 0.0, beta1, beta2, eps, (const StepStateDev*)step_state, stream);
}

// both fused tables of a NeuMF step in one launch (same dim / n / split; `last` arrays: deferred mode, `marks`: sweep mode)
extern "C" int brAdamRowsSortedPair(float* table_a, float* m_a, float* v_a, int64_t rows_a, const void* sorted_ids_a, const int32_t* sorted_pos_a,
                                    const float* grads_a, int64_t ldg_a, const float* grads_hi_a, int64_t ldg_hi_a, uint8_t* mark_a, int32_t* last_a,
                                    float* table_b, float* m_b, float* v_b, int64_t rows_b, const void* sorted_ids_b, const int32_t* sorted_pos_b,
                                    const float* grads_b, int64_t ldg_b, const float* grads_hi_b, int64_t ldg_hi_b, uint8_t* mark_b, int32_t* last_b,
                                    int dim, int id_type, int64_t n, int split, const float* hi_scale, const void* step_state, double alpha_t,
                                    double beta1, double beta2, double eps, float* seg_ws_a, float* seg_ws_b, brStream stream) {
  BR_CHECK_ARG((last_a == nullptr) == (last_b == nullptr) && (last_a == nullptr || step_state), "brAdamRowsSortedPair: last arrays for both tables (with step_state) or neither");
  BR_CHECK_ARG(grads_hi_a && grads_hi_b, "brAdamRowsSortedPair: both gradient halves required");
  const AdamRowsArgs a[2] = {{table_a, m_a, v_a, rows_a, sorted_ids_a, sorted_pos_a, grads_a, ldg_a, grads_hi_a, ldg_hi_a, mark_a, last_a, seg_ws_a, hi_scale},
                             {table_b, m_b, v_b, rows_b, sorted_ids_b, sorted_pos_b, grads_b, ldg_b, grads_hi_b, ldg_hi_b, mark_b, last_b, seg_ws_b, hi_scale}};
  return adam_rows_launch(a, 2, dim, id_type, n, split, alpha_t, beta1, beta2, eps, last_a ? (const StepStateDev*)step_state : nullptr, stream);
}

extern "C" int brAdamRowsSortedPairReplayed(float* table_a, float* m_a, float* v_a, int64_t rows_a, const void* sorted_ids_a, const int32_t* sorted_pos_a,
                                            const float* grads_a, int64_t ldg_a, const float* grads_hi_a, int64_t ldg_hi_a, int32_t* last_a,
                                            const float* replayed_a,
                                            float* table_b, float* m_b, float* v_b, int64_t rows_b, const void* sorted_ids_b, const int32_t* sorted_pos_b,
                                            const float* grads_b, int64_t ldg_b, const float* grads_hi_b, int64_t ldg_hi_b, int32_t* last_b,
                                            const float* replayed_b, int64_t ld_replayed,
                                            int dim, int id_type, int64_t n, int64_t n_b, int split, const void* step_state,
                                            double beta1, double beta2, double eps, float* seg_ws_a, float* seg_ws_b, brStream stream) {
  BR_CHECK_ARG(last_a && last_b && step_state, "brAdamRowsSortedPairReplayed: last arrays / step_state missing");
  BR_CHECK_ARG((grads_hi_a == nullptr) == (grads_hi_b == nullptr), "brAdamRowsSortedPairReplayed: gradient halves for both tables or neither");
  if (!grads_hi_a) { split = dim; grads_hi_a = grads_a; grads_hi_b = grads_b; ldg_hi_a = ldg_a; ldg_hi_b = ldg_b; }     // one source: the whole row is the low part
  BR_CHECK_ARG(split >= 1 && split <= dim, "brAdamRowsSortedPairReplayed: split out of range");
  BR_CHECK_ARG(replayed_a && replayed_b && ld_replayed >= dim && n >= 0 && n_b >= 0, "brAdamRowsSortedPairReplayed: replayed rows missing, ld < dim or n < 0");
  if (n_b == 0) n_b = n;
  if (n == 0 && n_b == 0) return BR_OK;
  AdamRowsArgs a[2] = {{table_a, m_a, v_a, rows_a, sorted_ids_a, sorted_pos_a, grads_a, ldg_a, grads_hi_a, ldg_hi_a, nullptr, last_a, seg_ws_a},
                       {table_b, m_b, v_b, rows_b, sorted_ids_b, sorted_pos_b, grads_b, ldg_b, grads_hi_b, ldg_hi_b, nullptr, last_b, seg_ws_b}};
  a[0].th_lo = replayed_a; a[0].th_hi = replayed_a + (split < dim ? split : 0); a[0].ld_th = ld_replayed;
  a[1].th_lo = replayed_b; a[1].th_hi = replayed_b + (split < dim ? split : 0); a[1].ld_th = ld_replayed;
  if (n_b != n) { a[0].n = n; a[1].n = n_b; }
  return adam_rows_launch(a, 2, dim, id_type, n > n_b ? n : n_b, split, 0.0, beta1, beta2, eps, (const StepStateDev*)step_state, stream);
}

int br::adam_rows_pair_keep(const AdamPairCall& c, const KeepArgs* keep, brStream stream, const FinalArgs* fin, bool* fin_done) {
  BR_CHECK_ARG((c.last_a == nullptr) == (c.last_b == nullptr) && (c.last_a == nullptr || c.step_state), "brAdamRowsSortedPair: last arrays for both tables (with step_state) or neither");
  BR_CHECK_ARG(c.grads_hi_a && c.grads_hi_b, "brAdamRowsSortedPair: both gradient halves required");
  const AdamRowsArgs a[2] = {{c.table_a, c.m_a, c.v_a, c.rows_a, c.sorted_ids_a, c.sorted_pos_a, c.grads_a, c.ldg_a, c.grads_hi_a, c.ldg_hi_a, c.mark_a, c.last_a, c.seg_ws_a, c.hi_scale,
                              c.th_lo_a, c.th_hi_a, c.ld_th},
                             {c.table_b, c.m_b, c.v_b, c.rows_b, c.sorted_ids_b, c.sorted_pos_b, c.grads_b, c.ldg_b, c.grads_hi_b, c.ldg_hi_b, c.mark_b, c.last_b, c.seg_ws_b, c.hi_scale,
                              c.th_lo_b, c.th_hi_b, c.ld_th}};
  return adam_rows_launch(a, 2, c.dim, c.id_type, c.n, c.split, c.alpha_t, c.beta1, c.beta2, c.eps, c.last_a ? (const StepStateDev*)c.step_state : nullptr, stream, keep,
                          fin, fin_done);
}

extern "C" int brAdamFlush(float* table, float* m, float* v, int32_t* last, int64_t table_rows, int dim, const void* step_state,
                           double beta1, double beta2, double eps, brStream stream) {
  BR_CHECK_ARG(table && m && v && last && step_state && dim >= 1 && table_rows > 0, "brAdamFlush: bad args");
  const RowGeom g = row_geom(dim);
  const int64_t n_vec = table_rows * g.chunks;
  const int64_t blocks = std::min<int64_t>(ceil_div(n_vec, 256), 256 * 16);
  const AdamHp h = make_hp(0.0, beta1, beta2, eps);
  const StepStateDev* ss = (const StepStateDev*)step_state;
  hipStream_t s = (hipStream_t)stream;
  BR_DISPATCH_VEC(g.vec, (adam_flush_kernel<VEC><<<(unsigned)blocks, 256, 0, s>>>(table, m, v, n_vec, g.chunks, h, last, ss)));
  BR_CHECK_LAUNCH("brAdamFlush");
  fill_last_kernel<<<(unsigned)ceil_div(table_rows, 256), 256, 0, s>>>(last, table_rows, ss);
  BR_CHECK_LAUNCH("brAdamFlush(last)");
  return BR_OK;
}

extern "C" int brAdamDenseSweep(float* table, float* m, float* v, int64_t table_rows, int dim, double alpha_t,
                                double beta1, double beta2, double eps, uint8_t* mark, brStream stream) {
  BR_CHECK_ARG(table && m && v && dim >= 1 && table_rows > 0, "brAdamDenseSweep: bad args");
  const RowGeom g = row_geom(dim);
  const int64_t n_vec = table_rows * g.chunks;
  int64_t blocks = ceil_div(n_vec, 256);
  const int64_t cap = 256 * 16;  // 16 workgroups per CU, grid-stride beyond
  if (blocks > cap) blocks = cap;
  const AdamHp h = make_hp(alpha_t, beta1, beta2, eps);
  hipStream_t s = (hipStream_t)stream;
  BR_DISPATCH_VEC(g.vec, (adam_dense_sweep_kernel<VEC><<<(unsigned)blocks, 256, 0, s>>>(table, m, v, n_vec, g.chunks, h, mark)));
  BR_CHECK_LAUNCH("brAdamDenseSweep");
  if (mark) {
    // not hipMemsetAsync: a memset NODE of this (odd) size inside a captured hipGraph left garbage in the marks
    // on ROCm 7.2 when the graph started with it (tests/test_gpu_neumf.py, split replay) - a kernel node is safe
    zero_bytes_kernel<<<(unsigned)ceil_div(table_rows, 1024), 256, 0, s>>>(mark, table_rows);
    BR_CHECK_LAUNCH("brAdamDenseSweep(marks)");
  }
  return BR_OK;
}

extern "C" int brAdamFlat(float* theta, float* m, float* v, const float* g, int64_t n, double alpha_t, double beta1,
                          double beta2, double eps, brStream stream) {
  if (n == 0) return BR_OK;
  BR_CHECK_ARG(theta && m && v && g && n > 0, "brAdamFlat: bad args");
  adam_flat_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(theta, m, v, g, n, make_hp(alpha_t, beta1, beta2, eps));
  BR_CHECK_LAUNCH("brAdamFlat");
  return BR_OK;
}

extern "C" int brAdagradRowsSorted(float* table, float* acc, int64_t table_rows, int dim, const void* sorted_ids,
                                   int id_type, const int32_t* sorted_pos, int64_t n, const float* row_grads, int64_t ldg,
                                   double lr, double eps, float* seg_ws, brStream stream) {
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brAdagradRowsSorted: bad id_type");
  if (n == 0) return BR_OK;
  BR_CHECK_ARG(table && acc && sorted_ids && sorted_pos && row_grads && dim >= 1 && ldg >= dim && table_rows > 0,
               "brAdagradRowsSorted: bad args");
  const RowGeom g = row_geom_ld(dim, ldg);
  const unsigned grid = (unsigned)ceil_div(n, 256 >> g.lpr_log2);
  hipStream_t s = (hipStream_t)stream;
  if (seg_ws) {
    const SegJob job{sorted_ids, sorted_pos, row_grads, ldg, row_grads, ldg, seg_ws};
    const int rc = launch_partials(&job, 1, id_type, n, dim, g, dim, s);
    if (rc != BR_OK) return rc;
  }
  if (id_type == BR_IDS_I32)
    BR_DISPATCH_VEC(g.vec, (adagrad_rows_sorted_kernel<int32_t, VEC><<<grid, 256, 0, s>>>(
                               table, acc, table_rows, dim, g.chunks, g.lpr_log2, (const int32_t*)sorted_ids, sorted_pos, n,
                               row_grads, ldg, (float)lr, (float)eps, seg_ws)));
  else
    BR_DISPATCH_VEC(g.vec, (adagrad_rows_sorted_kernel<int64_t, VEC><<<grid, 256, 0, s>>>(
                               table, acc, table_rows, dim, g.chunks, g.lpr_log2, (const int64_t*)sorted_ids, sorted_pos, n,
                               row_grads, ldg, (float)lr, (float)eps, seg_ws)));
  BR_CHECK_LAUNCH("brAdagradRowsSorted");
  return BR_OK;
}

extern "C" int brAdagradFlat(float* theta, float* acc, const float* g, int64_t n, double lr, double eps, brStream stream) {
  if (n == 0) return BR_OK;
  BR_CHECK_ARG(theta && acc && g && n > 0, "brAdagradFlat: bad args");
  adagrad_flat_kernel<<<(unsigned)ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(theta, acc, g, n, (float)lr, (float)eps);
  BR_CHECK_LAUNCH("brAdagradFlat");
  return BR_OK;
}

extern "C" int64_t brStepStateBytes(void) { return (int64_t)sizeof(StepStateDev); }

extern "C" int brStepStateAdvance(void* step_state, double lr, double beta1, double beta2, double* zero, int64_t n_zero, brStream stream) {
  BR_CHECK_ARG(step_state != nullptr && n_zero >= 0 && (zero || n_zero == 0), "brStepStateAdvance: bad args");
  StepAdvance a;
  a.st = (StepStateDev*)step_state; a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.zero = zero; a.n_zero = n_zero;
  step_state_advance_kernel<<<1, 256, 0, (hipStream_t)stream>>>(a);
  BR_CHECK_LAUNCH("brStepStateAdvance");
  return BR_OK;
}

extern "C" int brStepStateInit(void* step_state, double beta1, double beta2, double eps, int replay_mode, brStream stream) {
  BR_CHECK_ARG(step_state != nullptr, "brStepStateInit: null state");
  BR_CHECK_ARG(replay_mode == BR_REPLAY_EXACT || replay_mode == BR_REPLAY_FAST, "brStepStateInit: bad replay_mode %d", replay_mode);
  BR_CHECK_ARG(beta1 >= 0.0 && beta1 < 1.0 && beta2 > 0.0 && beta2 < 1.0 && eps >= 0.0, "brStepStateInit: beta1 in [0,1), beta2 in (0,1), eps >= 0");
  constexpr size_t off = offsetof(StepStateDev, fast);
  static_assert(offsetof(StepStateDev, pow2) + sizeof(float) * BR_ALPHA_RING == sizeof(StepStateDev), "StepStateDev tail layout");
  std::vector<StepStateDev> img(1);                            // a host image of the state; only its tail [fast, end) is copied
  StepStateDev* h = img.data();
  // the kernels multiply by the fp32 roundings of beta1 / beta2 (AdamHp): the closed forms below are powers of THOSE numbers - over a lag of
  // 1000 steps pow(0.999, k) and pow((float)0.999, k) are 1.3e-5 apart
  beta1 = (double)(float)beta1; beta2 = (double)(float)beta2;
  const double c = sqrt(beta2), rho = beta1 / c;
  h->fast = replay_mode == BR_REPLAY_FAST ? 1u : 0u;
  // theta is replayed over the first `trunc` steps of a lag: the steps behind it move theta by at most 7 rho^trunc / (1 - rho) of the first
  // step's update (alpha_j varies by less than 7x over any lag, m decays by beta1 and 1 / d grows by at most 1 / c per step); below 2^-23
  // of it they are under one ulp.  A multiple of 8 (the replay takes eight alphas per scalar load); rho >= 1: never truncated.
  uint32_t trunc = BR_ALPHA_RING;
  if (rho < 1.0) {
    const double need = log(ldexp(1.0, -23) * (1.0 - rho) / 7.0) / log(rho);
    if (need < (double)BR_ALPHA_RING) trunc = (uint32_t)((((int64_t)ceil(need < 1.0 ? 1.0 : need)) + 7) / 8 * 8);
  }
  h->trunc = trunc;
  h->sqrt_b2 = (float)c;
  h->eps_c = (float)(eps * (1.0 - c));
  for (int k = 0; k < BR_ALPHA_RING; ++k) { h->pow1[k] = (float)pow(beta1, (double)k); h->pow2[k] = (float)pow(beta2, (double)k); }
  const hipError_t ce = hipMemcpyAsync((char*)step_state + off, (const char*)h + off, sizeof(StepStateDev) - off, hipMemcpyHostToDevice, (hipStream_t)stream);
  if (ce != hipSuccess) {
    set_error("brStepStateInit: copy failed: %s", hipGetErrorString(ce));
    return BR_ERR_HIP;
  }
  (void)hipStreamSynchronize((hipStream_t)stream);
  return BR_OK;
}

extern "C" int brStepStateSet(void* step_state, uint32_t step, double lr, double beta1, double beta2, brStream stream) {
  BR_CHECK_ARG(step_state != nullptr, "brStepStateSet: null state");
  struct { uint32_t step; float alpha_t; double p1, p2; } h;
  static_assert(sizeof(h) == offsetof(StepStateDev, alpha_hist), "StepStateDev head layout");
  h.step = step;
  h.p1 = pow(beta1, (double)step);
  h.p2 = pow(beta2, (double)step);
  const double tt = step > 0 ? (double)step : 1.0;
  h.alpha_t = (float)(lr * sqrt(1.0 - pow(beta2, tt)) / (1.0 - pow(beta1, tt)));
  // pageable source: hipMemcpyAsync returns after the copy has been staged, `h` may leave scope
  const hipError_t ce = hipMemcpyAsync(step_state, &h, sizeof(h), hipMemcpyHostToDevice, (hipStream_t)stream);
  if (ce != hipSuccess) {
    set_error("brStepStateSet: copy failed: %s", hipGetErrorString(ce));
    return BR_ERR_HIP;
  }
  (void)hipStreamSynchronize((hipStream_t)stream);
  return BR_OK;
}

extern "C" int brStageBatch(void* dst_users, void* dst_items, float* dst_labels, const void* users, const void* items,
                            const float* labels, int id_type, int64_t n, brStream stream) {
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brStageBatch: bad id_type");
  if (n == 0) return BR_OK;
  BR_CHECK_ARG(dst_users && dst_items && users && items && n > 0, "brStageBatch: bad args");
  const unsigned grid = (unsigned)ceil_div(n, 256);
  if (id_type == BR_IDS_I32)
    stage_batch_kernel<int32_t><<<grid, 256, 0, (hipStream_t)stream>>>((int32_t*)dst_users, (int32_t*)dst_items, dst_labels, (const int32_t*)users,
                                                                        (const int32_t*)items, labels, n);
  else
    stage_batch_kernel<int64_t><<<grid, 256, 0, (hipStream_t)stream>>>((int64_t*)dst_users, (int64_t*)dst_items, dst_labels, (const int64_t*)users,
                                                                        (const int64_t*)items, labels, n);
  BR_CHECK_LAUNCH("brStageBatch");
  return BR_OK;
}

// ---- row-sharded exchange planning (parallel.py ShardExchange.plan): owner(id) = id mod W --------------------------
// One id stream: dest = id mod W -> stable sort of (dest, position) with the index machinery above -> order[j] = batch
// position of bucket slot j, inv[b] = bucket slot of position b, send_local[j] = id div W in bucket order, counts[d] =
// rows for owner d.  4 launches for a PAIR of equally long streams instead of ~10 torch ops per stream.
template <typename IdT>
__global__ __launch_bounds__(256) void shard_dest_kernel(const IdT* __restrict__ a, const IdT* __restrict__ b, IdT* __restrict__ da,
                                                          IdT* __restrict__ db, int64_t n, int world) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const IdT* src = blockIdx.y ? b : a;
  IdT* dst = blockIdx.y ? db : da;
  const int64_t id = (int64_t)src[i];
  const int64_t m = id % world;
  dst[i] = (IdT)(m < 0 ? m + world : m);
}

struct ShardFinishJob { const void* ids; const void* sorted_dest; const int32_t* order; int32_t* inv; void* send_local; int64_t* counts; };
struct ShardFinishJobs { ShardFinishJob j[2]; };

template <typename IdT>
__global__ __launch_bounds__(256) void shard_finish_kernel(ShardFinishJobs jobs, int64_t n, int world) {
  const ShardFinishJob& jb = jobs.j[blockIdx.y];
  const IdT* ids = (const IdT*)jb.ids;
  const IdT* sd = (const IdT*)jb.sorted_dest;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < n) {
    const int32_t b = jb.order[j];
    jb.inv[b] = (int32_t)j;
    const int64_t id = (int64_t)ids[b];
    const int64_t m = id % world;
    ((IdT*)jb.send_local)[j] = (IdT)((id - (m < 0 ? m + world : m)) / world);      // floor division for every sign
  }
  if (j < world) {                                   // rows for owner j: [lower_bound(j), lower_bound(j + 1)) of the sorted dests
    auto lb = [&](int64_t d) {
      int64_t lo = 0, hi = n;
      while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)sd[mid] < d) lo = mid + 1; else hi = mid; }
      return lo;
    };
    jb.counts[j] = lb(j + 1) - lb(j);
  }
}

// ---- fixed-capacity form of the exchange plan: every peer gets exactly `cap` slots per stream, so the all-to-alls have equal, static
// splits and no row count ever crosses to the host.  Slot t = d * cap + k holds the k-th row for owner d (bucket order), pad slots
// hold the owner's spare row (local index = its row count: the tables of a padded engine carry one extra row whose gradient is always 0).
struct ShardPadJob {
  const void* sorted_dest; const int32_t* order; const void* send_local; const int64_t* counts;
  void* send_pad; int32_t* slot; int32_t* bpos; int64_t total_rows;
  float* zero_rows; int zero_dim;      // optional [world*cap][zero_dim] buffer whose PAD rows are cleared (gradient send slots)
};
struct ShardPadJobs { ShardPadJob j[2]; };

template <typename IdT>
__global__ __launch_bounds__(256) void shard_pad_kernel(ShardPadJobs jobs, int64_t n, int world, int64_t cap, int* __restrict__ overflow) {
  const ShardPadJob& jb = jobs.j[blockIdx.y];
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < n) {                                        // bucket position t -> its slot
    const int64_t d = (int64_t)((const IdT*)jb.sorted_dest)[t];
    int64_t off = 0;
    for (int64_t q = 0; q < d; ++q) off += jb.counts[q];
    int64_t k = t - off;
    if (k >= cap) { atomicOr(overflow, BR_ERRFLAG_CAPACITY); k = cap - 1; }     // reported by the host's next flag check
    const int64_t sl = d * cap + k;
    const int32_t b = jb.order[t];
    jb.slot[b] = (int32_t)sl;
    if (t - off < cap) { ((IdT*)jb.send_pad)[sl] = ((const IdT*)jb.send_local)[t]; jb.bpos[sl] = b; }
  }
  if (t < (int64_t)world * cap) {                     // pad slots
    const int64_t d = t / cap, k = t - d * cap;
    if (k >= jb.counts[d]) {
      const int64_t spare = jb.total_rows > d ? (jb.total_rows - d + world - 1) / world : 0;   // rows owner d holds = index of its spare row
      ((IdT*)jb.send_pad)[t] = (IdT)spare;
      jb.bpos[t] = -1;
      if (jb.zero_rows)
        for (int c = 0; c < jb.zero_dim; c += 4) *reinterpret_cast<float4*>(jb.zero_rows + t * jb.zero_dim + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

extern "C" int brShardPadPair(const void* sorted_dest_a, const void* sorted_dest_b, const int32_t* order_a, const int32_t* order_b,
                              const void* send_local_a, const void* send_local_b, const int64_t* counts_a, const int64_t* counts_b, int id_type,
                              int64_t n, int world, int64_t cap, int64_t total_rows_a, int64_t total_rows_b, void* send_pad_a, void* send_pad_b,
                              int32_t* slot_a, int32_t* slot_b, int32_t* bpos_a, int32_t* bpos_b, float* zero_a, float* zero_b, int zero_dim, int* err_flag,
                              brStream stream) {
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brShardPadPair: bad id_type");
  BR_CHECK_ARG(!zero_a || (zero_dim >= 4 && zero_dim % 4 == 0 && (reinterpret_cast<uintptr_t>(zero_a) & 15) == 0 && (reinterpret_cast<uintptr_t>(zero_b) & 15) == 0),
               "brShardPadPair: zero rows need a dim that is a multiple of 4 and 16-byte alignment");
  BR_CHECK_ARG(world >= 1 && world <= 256 && n >= 0 && cap >= 1 && err_flag, "brShardPadPair: bad world / n / cap / flag");
  BR_CHECK_ARG(sorted_dest_a && order_a && send_local_a && counts_a && send_pad_a && slot_a && bpos_a, "brShardPadPair: null pointer (stream a)");
  const int n_jobs = sorted_dest_b ? 2 : 1;
  BR_CHECK_ARG(!sorted_dest_b || (order_b && send_local_b && counts_b && send_pad_b && slot_b && bpos_b), "brShardPadPair: null pointer (stream b)");
  ShardPadJobs J;
  J.j[0] = ShardPadJob{sorted_dest_a, order_a, send_local_a, counts_a, send_pad_a, slot_a, bpos_a, total_rows_a, zero_a, zero_dim};
  J.j[1] = sorted_dest_b ? ShardPadJob{sorted_dest_b, order_b, send_local_b, counts_b, send_pad_b, slot_b, bpos_b, total_rows_b, zero_b, zero_dim} : J.j[0];
  const int64_t m = n > (int64_t)world * cap ? n : (int64_t)world * cap;
  const dim3 g((unsigned)ceil_div(m, 256), (unsigned)n_jobs);
  if (id_type == BR_IDS_I32) shard_pad_kernel<int32_t><<<g, 256, 0, (hipStream_t)stream>>>(J, n, world, cap, err_flag);
  else shard_pad_kernel<int64_t><<<g, 256, 0, (hipStream_t)stream>>>(J, n, world, cap, err_flag);
  BR_CHECK_LAUNCH("brShardPadPair");
  return BR_OK;
}

// dst[t] = bpos[t] >= 0 ? src[bpos[t]] : 0 for one or two (src, bpos, dst) sets of equal shape: per-pair rows -> padded send slots
struct PadRowsJob { const float* src; int64_t ld; const int32_t* bpos; float* dst; };
struct PadRowsJobs { PadRowsJob j[2]; };
__global__ __launch_bounds__(256) void rows_to_slots_kernel(PadRowsJobs jobs, int64_t n_slots, int dim) {
  const PadRowsJob& jb = jobs.j[blockIdx.y];
  const int q4 = dim >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_slots * q4) return;
  const int64_t t = i / q4;
  const int c = (int)(i - t * q4) << 2;
  const int32_t b = jb.bpos[t];
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (b >= 0) v = *reinterpret_cast<const float4*>(jb.src + (int64_t)b * jb.ld + c);
  *reinterpret_cast<float4*>(jb.dst + t * dim + c) = v;
}

extern "C" int brRowsToSlotsPair(const float* src_a, const float* src_b, int64_t ld, const int32_t* bpos_a, const int32_t* bpos_b, float* dst_a,
                                 float* dst_b, int64_t n_slots, int dim, brStream stream) {
  BR_CHECK_ARG(src_a && bpos_a && dst_a && n_slots >= 0 && dim >= 4 && dim % 4 == 0 && ld >= dim && ld % 4 == 0, "brRowsToSlotsPair: bad args (dim, ld multiples of 4)");
  BR_CHECK_ARG(((reinterpret_cast<uintptr_t>(src_a) | reinterpret_cast<uintptr_t>(dst_a) | reinterpret_cast<uintptr_t>(src_b) | reinterpret_cast<uintptr_t>(dst_b)) & 15) == 0,
               "brRowsToSlotsPair: rows must be 16-byte aligned");
  BR_CHECK_ARG(!src_b || (bpos_b && dst_b), "brRowsToSlotsPair: null pointer (set b)");
  if (n_slots == 0) return BR_OK;
  PadRowsJobs J;
  J.j[0] = PadRowsJob{src_a, ld, bpos_a, dst_a};
  J.j[1] = src_b ? PadRowsJob{src_b, ld, bpos_b, dst_b} : J.j[0];
  const dim3 g((unsigned)ceil_div(n_slots * (dim >> 2), 256), src_b ? 2u : 1u);
  rows_to_slots_kernel<<<g, 256, 0, (hipStream_t)stream>>>(J, n_slots, dim);
  BR_CHECK_LAUNCH("brRowsToSlotsPair");
  return BR_OK;
}

extern "C" int brShardPlanPair(const void* ids_a, const void* ids_b, int id_type, int64_t n, int world, void* dest_a, void* dest_b,
                               void* sorted_dest_a, void* sorted_dest_b, int32_t* order_a, int32_t* order_b, void* ws_a, void* ws_b,
                               int64_t ws_bytes, int32_t* inv_a, int32_t* inv_b, void* send_local_a, void* send_local_b,
                               int64_t* counts_a, int64_t* counts_b, brStream stream) {
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brShardPlanPair: bad id_type");
  BR_CHECK_ARG(world >= 1 && world <= 256 && n >= 0, "brShardPlanPair: bad world / n");
  const int n_jobs = ids_b ? 2 : 1;
  BR_CHECK_ARG(ids_a && dest_a && sorted_dest_a && order_a && ws_a && inv_a && send_local_a && counts_a, "brShardPlanPair: null pointer (stream a)");
  BR_CHECK_ARG(!ids_b || (dest_b && sorted_dest_b && order_b && ws_b && inv_b && send_local_b && counts_b), "brShardPlanPair: null pointer (stream b)");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    (void)hipMemsetAsync(counts_a, 0, sizeof(int64_t) * world, s);
    if (ids_b) (void)hipMemsetAsync(counts_b, 0, sizeof(int64_t) * world, s);
    return BR_OK;
  }
  const dim3 g((unsigned)ceil_div(n, 256), (unsigned)n_jobs);
  if (id_type == BR_IDS_I32) shard_dest_kernel<int32_t><<<g, 256, 0, s>>>((const int32_t*)ids_a, (const int32_t*)ids_b, (int32_t*)dest_a, (int32_t*)dest_b, n, world);
  else shard_dest_kernel<int64_t><<<g, 256, 0, s>>>((const int64_t*)ids_a, (const int64_t*)ids_b, (int64_t*)dest_a, (int64_t*)dest_b, n, world);
  BR_CHECK_LAUNCH("brShardPlanPair(dest)");
  int rc;
  if (ids_b) rc = brRowIndexBuildPair(dest_a, world, sorted_dest_a, order_a, ws_a, ws_bytes, dest_b, world, sorted_dest_b, order_b, ws_b, ws_bytes, id_type, n, stream);
  else rc = brRowIndexBuild(dest_a, id_type, n, world, sorted_dest_a, order_a, ws_a, ws_bytes, stream);
  if (rc != BR_OK) return rc;
  ShardFinishJobs J;
  J.j[0] = ShardFinishJob{ids_a, sorted_dest_a, order_a, inv_a, send_local_a, counts_a};
  J.j[1] = ids_b ? ShardFinishJob{ids_b, sorted_dest_b, order_b, inv_b, send_local_b, counts_b} : J.j[0];
  const dim3 g2((unsigned)ceil_div(n > world ? n : world, 256), (unsigned)n_jobs);
  if (id_type == BR_IDS_I32) shard_finish_kernel<int32_t><<<g2, 256, 0, s>>>(J, n, world);
  else shard_finish_kernel<int64_t><<<g2, 256, 0, s>>>(J, n, world);
  BR_CHECK_LAUNCH("brShardPlanPair(finish)");
  return BR_OK;
}
