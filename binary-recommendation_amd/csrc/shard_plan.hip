// Requester-side plan of the row-sharded exchange with duplicate ids merged (parallel.py PaddedExchange).
//
// The reference mirrors every table on every worker (src/models/RModel.py:119); here owner(id) = id mod W holds row id div W, and a
// step sends each owner the ids it must serve.  A batch names the same id many times (Zipf: one id on thousands of positions) - the
// owner needs it once, and so does the wire:
//   key[b]  = owner(id_b) * R + id_b div W        (R = rows per owner, rounded up: keys of one owner are contiguous)
//   (key, position) pairs are sorted (the dedup index machinery of sparse_opt.hip: stable, so equal keys keep batch order),
//   the k-th DISTINCT key of owner d gets send slot (d * 2 + stream) * cap + k of the step's ONE all-to-all buffer
//   [owner][stream: user | item][cap], every position of the batch learns its id's slot, and the sorted index stays behind for the
//   backward: the row gradients of an id's positions are summed (ordered, two-level) into that slot before they travel
//   (brSegmentSumToSlotsPair).
// Fixed capacity: `cap` slots per owner and stream, so the three all-to-alls of a step have equal static splits and nothing is read
// back by the host.  Pad slots name the owner's spare row (its gradient slot is zero).  An owner with more than `cap` DISTINCT ids in
// one stream overflows: the surplus ids get NO slot (slot -1: the step reads zeros for them and sends no gradient - nothing collides,
// no row of any table sees a wrong gradient) and BR_ERRFLAG_CAPACITY is raised for the host's next check.  Out-of-range ids likewise
// (BR_ERRFLAG_RANGE).
#include "common.h"

namespace br {

struct DedupJob {
  const void* ids; void* keys;                   // [n]
  const void* skeys; const int32_t* spos;        // the sorted index (written by brRowIndexBuildPair between the two plan launches)
  int32_t* urank;                                // [n]   rank of sorted position j's key among the distinct keys
  int32_t* first;                                // [W+1] urank of each owner's first distinct key ([W]: number of distinct keys)
  int32_t* slot;                                 // [n]   batch position -> physical slot, -1: none
  int64_t total_rows, R;
  int stream;                                    // 0 | 1: the stream's half of each owner's slot block
};
struct DedupJobs { DedupJob j[2]; };

template <typename IdT>
__global__ __launch_bounds__(256) void dedup_key_kernel(DedupJobs jobs, int64_t n, int world, int* __restrict__ err) {
  const DedupJob& jb = jobs.j[blockIdx.y];
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= n) return;
  const int64_t id = (int64_t)((const IdT*)jb.ids)[b];
  int64_t key = (int64_t)world * jb.R;                                   // out of range: behind every owner's keys
  if ((uint64_t)id < (uint64_t)jb.total_rows) key = (id % world) * jb.R + id / world;
  else atomicOr(err, BR_ERRFLAG_RANGE);
  ((IdT*)jb.keys)[b] = (IdT)key;
}

// urank[j] = (number of distinct keys in sorted positions [0, j]) - 1, in two launches of 2048-position blocks: block counts of the heads,
// then every block adds the counts of the blocks in front of it to its own inclusive scan.  (One 1024-thread workgroup scanning a
// stream alone took 118 us at 65 536 positions: 64 dependent loads per thread.)
constexpr int kScanThreads = 256, kScanPer = 8, kScanBlock = kScanThreads * kScanPer;

template <typename IdT>
__device__ __forceinline__ int scan_heads(const IdT* __restrict__ sk, int64_t n, int64_t base, int (&flag)[kScanPer]) {
  int c = 0;
#pragma unroll
  for (int q = 0; q < kScanPer; ++q) {
    const int64_t j = base + q;
    flag[q] = (j < n && (j == 0 || sk[j] != sk[j - 1])) ? 1 : 0;
    c += flag[q];
  }
  return c;
}
__device__ __forceinline__ int block_sum_256(int v, int* red) {      // sum over the 256 threads of a workgroup (red: 4 ints of LDS)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const int t = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return t;
}

template <typename IdT>
__global__ __launch_bounds__(kScanThreads) void dedup_count_kernel(DedupJobs jobs, int64_t n, int32_t* __restrict__ block_cnt, int n_blocks) {
  const DedupJob& jb = jobs.j[blockIdx.y];
  __shared__ int red[4];
  int flag[kScanPer];
  const int c = scan_heads((const IdT*)jb.skeys, n, (int64_t)blockIdx.x * kScanBlock + (int64_t)threadIdx.x * kScanPer, flag);
  const int t = block_sum_256(c, red);
  if (threadIdx.x == 0) block_cnt[blockIdx.y * n_blocks + blockIdx.x] = t;
}

template <typename IdT>
__global__ __launch_bounds__(kScanThreads) void dedup_rank_kernel(DedupJobs jobs, int64_t n, const int32_t* __restrict__ block_cnt, int n_blocks) {
  const DedupJob& jb = jobs.j[blockIdx.y];
  __shared__ int red[4];
  __shared__ int wave_base[4];
  // heads in the blocks in front of this one (n_blocks <= 256 for n <= 524 288; more: strided)
  int before = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += kScanThreads) before += block_cnt[blockIdx.y * n_blocks + b];
  before = block_sum_256(before, red);
  int flag[kScanPer];
  const int64_t base = (int64_t)blockIdx.x * kScanBlock + (int64_t)threadIdx.x * kScanPer;
  const int c = scan_heads((const IdT*)jb.skeys, n, base, flag);
  // exclusive scan of the per-thread counts: inside the wave by shuffles, across the 4 waves through LDS
  int incl = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if ((int)(threadIdx.x & 63) >= off) incl += v;
  }
  if ((threadIdx.x & 63) == 63) wave_base[threadIdx.x >> 6] = incl;
  __syncthreads();
  int run = before + incl - c;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) run += wave_base[w];
#pragma unroll
  for (int q = 0; q < kScanPer; ++q) {
    run += flag[q];
    if (base + q < n) jb.urank[base + q] = run - 1;
  }
}

// first[d] = number of distinct keys in front of owner d's key range (d = 0 .. world; [world] = all valid distinct keys)
template <typename IdT>
__global__ __launch_bounds__(256) void dedup_first_kernel(DedupJobs jobs, int64_t n, int world) {
  const DedupJob& jb = jobs.j[blockIdx.y];
  const IdT* __restrict__ sk = (const IdT*)jb.skeys;
  const int d = (int)(blockIdx.x * 256 + threadIdx.x);
  if (d > world) return;
  const int64_t want = (int64_t)d * jb.R;                                // first sorted position with key >= d * R
  int64_t a = 0, b = n;
  while (a < b) { const int64_t mid = (a + b) >> 1; if ((int64_t)sk[mid] < want) a = mid + 1; else b = mid; }
  jb.first[d] = a == 0 ? 0 : jb.urank[a - 1] + 1;                        // (the key at a, if any, starts a new owner's range: it is a head)
}

template <typename IdT>
__global__ __launch_bounds__(256) void dedup_slot_kernel(DedupJobs jobs, int64_t n, int world, int64_t cap, IdT* __restrict__ send_ids, int* __restrict__ err) {
  const DedupJob& jb = jobs.j[blockIdx.y];
  const IdT* __restrict__ sk = (const IdT*)jb.skeys;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < n) {                                                           // sorted position t -> the slot of its batch position
    const int64_t key = (int64_t)sk[t];
    const int64_t d = key / jb.R;
    int32_t sl = -1;
    if (d < world) {
      const int64_t k = (int64_t)jb.urank[t] - jb.first[d];
      if (k < cap) {
        const int64_t ps = (d * 2 + jb.stream) * cap + k;
        sl = (int32_t)ps;
        if (t == 0 || sk[t - 1] != sk[t]) send_ids[ps] = (IdT)(key - d * jb.R);      // the head publishes the local row id
      } else {
        atomicOr(err, BR_ERRFLAG_CAPACITY);
      }
    }
    jb.slot[jb.spos[t]] = sl;
  }
  if (t < (int64_t)world * cap) {                                        // pad slots: the owner's spare row
    const int64_t d = t / cap, k = t - d * cap;
    const int64_t used = (int64_t)jb.first[d + 1] - jb.first[d];
    if (k >= used) send_ids[(d * 2 + jb.stream) * cap + k] = (IdT)(jb.total_rows > d ? (jb.total_rows - d + world - 1) / world : 0);      // = rows owner d holds
  }
}

// the gradient rows of the pad slots cleared (one 16-B store per thread: whole rows leave as coalesced segments)
__global__ __launch_bounds__(256) void dedup_zero_pads_kernel(DedupJobs jobs, int world, int64_t cap, float* __restrict__ zero_rows, int zero_dim) {
  const DedupJob& jb = jobs.j[blockIdx.y];
  const int q4 = zero_dim >> 2;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)world * cap * q4) return;
  const int64_t t = e / q4;
  const int c = (int)(e - t * q4) << 2;
  const int64_t d = t / cap, k = t - d * cap;
  if (k >= (int64_t)jb.first[d + 1] - jb.first[d])
    *reinterpret_cast<float4*>(zero_rows + ((d * 2 + jb.stream) * cap + k) * zero_dim + c) = make_float4(0.f, 0.f, 0.f, 0.f);
}

}  // namespace br

using namespace br;

extern "C" int brShardDedupPlanPair(const void* ids_a, const void* ids_b, int id_type, int64_t n, int world, int64_t cap, int64_t total_rows_a,
                                    int64_t total_rows_b, void* keys_a, void* keys_b, void* sorted_keys_a, void* sorted_keys_b, int32_t* sorted_pos_a,
                                    int32_t* sorted_pos_b, void* ws_a, void* ws_b, int64_t ws_bytes, int32_t* urank_a, int32_t* urank_b, int32_t* first_a,
                                    int32_t* first_b, void* send_ids, int32_t* slot_a, int32_t* slot_b, float* grad_slots, int zero_dim, int* err_flag,
                                    brStream stream) {
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brShardDedupPlanPair: bad id_type");
  BR_CHECK_ARG(world >= 1 && world <= 256 && n >= 0 && cap >= 1 && total_rows_a >= 1 && total_rows_b >= 1 && err_flag, "brShardDedupPlanPair: bad world / n / cap / rows / flag");
  BR_CHECK_ARG(ids_a && ids_b && keys_a && keys_b && sorted_keys_a && sorted_keys_b && sorted_pos_a && sorted_pos_b && ws_a && ws_b && urank_a && urank_b && first_a &&
                   first_b && send_ids && slot_a && slot_b, "brShardDedupPlanPair: null pointer");
  BR_CHECK_ARG(!grad_slots || (zero_dim >= 4 && zero_dim % 4 == 0 && (reinterpret_cast<uintptr_t>(grad_slots) & 15) == 0),
               "brShardDedupPlanPair: gradient slots need a dim that is a multiple of 4 and 16-byte alignment");
  BR_CHECK_ARG((int64_t)world * 2 * cap < ((int64_t)1 << 31), "brShardDedupPlanPair: world * 2 * cap must stay below 2^31 slots");
  const int64_t Ra = (total_rows_a + world - 1) / world + 1, Rb = (total_rows_b + world - 1) / world + 1;     // (+1: the spare row's index is a valid local id)
  BR_CHECK_ARG(id_type == BR_IDS_I64 || ((int64_t)world * Ra < ((int64_t)1 << 31) - 2 && (int64_t)world * Rb < ((int64_t)1 << 31) - 2),
               "brShardDedupPlanPair: int32 ids: world * rows per owner must stay below 2^31");
  hipStream_t s = (hipStream_t)stream;
  DedupJobs J;
  J.j[0] = DedupJob{ids_a, keys_a, sorted_keys_a, sorted_pos_a, urank_a, first_a, slot_a, total_rows_a, Ra, 0};
  J.j[1] = DedupJob{ids_b, keys_b, sorted_keys_b, sorted_pos_b, urank_b, first_b, slot_b, total_rows_b, Rb, 1};
  if (n > 0) {
    const dim3 g((unsigned)ceil_div(n, 256), 2);
    if (id_type == BR_IDS_I32) dedup_key_kernel<int32_t><<<g, 256, 0, s>>>(J, n, world, err_flag);
    else dedup_key_kernel<int64_t><<<g, 256, 0, s>>>(J, n, world, err_flag);
    BR_CHECK_LAUNCH("brShardDedupPlanPair(keys)");
    const int rc = brRowIndexBuildPair(keys_a, (int64_t)world * Ra + 1, sorted_keys_a, sorted_pos_a, ws_a, ws_bytes, keys_b, (int64_t)world * Rb + 1, sorted_keys_b,
                                       sorted_pos_b, ws_b, ws_bytes, id_type, n, stream);
    if (rc != BR_OK) return rc;
  }
  if (n > 0) {
    // block counts of the scan live in the head of workspace a's second half (the sort is done with it: brRowIndexWorkspaceBytes covers
    // 2 x align256(4 n) bytes, the counts need 8 * ceil(n / 2048) bytes)
    const int n_blocks = (int)ceil_div(n, kScanBlock);
    int32_t* block_cnt = (int32_t*)ws_a;
    BR_CHECK_ARG(ws_bytes >= (int64_t)sizeof(int32_t) * 2 * n_blocks, "brShardDedupPlanPair: workspace too small for the scan");
    const dim3 gs((unsigned)n_blocks, 2);
    if (id_type == BR_IDS_I32) { dedup_count_kernel<int32_t><<<gs, kScanThreads, 0, s>>>(J, n, block_cnt, n_blocks); dedup_rank_kernel<int32_t><<<gs, kScanThreads, 0, s>>>(J, n, block_cnt, n_blocks); }
    else { dedup_count_kernel<int64_t><<<gs, kScanThreads, 0, s>>>(J, n, block_cnt, n_blocks); dedup_rank_kernel<int64_t><<<gs, kScanThreads, 0, s>>>(J, n, block_cnt, n_blocks); }
    BR_CHECK_LAUNCH("brShardDedupPlanPair(scan)");
  }
  const dim3 gf((unsigned)ceil_div(world + 1, 256), 2);
  if (id_type == BR_IDS_I32) dedup_first_kernel<int32_t><<<gf, 256, 0, s>>>(J, n, world);
  else dedup_first_kernel<int64_t><<<gf, 256, 0, s>>>(J, n, world);
  BR_CHECK_LAUNCH("brShardDedupPlanPair(first)");
  const int64_t m = n > (int64_t)world * cap ? n : (int64_t)world * cap;
  const dim3 g2((unsigned)ceil_div(m, 256), 2);
  if (id_type == BR_IDS_I32) dedup_slot_kernel<int32_t><<<g2, 256, 0, s>>>(J, n, world, cap, (int32_t*)send_ids, err_flag);
  else dedup_slot_kernel<int64_t><<<g2, 256, 0, s>>>(J, n, world, cap, (int64_t*)send_ids, err_flag);
  BR_CHECK_LAUNCH("brShardDedupPlanPair(slots)");
  if (grad_slots) {
    const dim3 gz((unsigned)ceil_div((int64_t)world * cap * (zero_dim >> 2), 256), 2);
    dedup_zero_pads_kernel<<<gz, 256, 0, s>>>(J, world, cap, grad_slots, zero_dim);
    BR_CHECK_LAUNCH("brShardDedupPlanPair(pads)");
  }
  return BR_OK;
}
