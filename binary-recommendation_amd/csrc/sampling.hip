// Batch construction on the device: the three negative samplers of the reference as pure functions of
// (seed, output position), so the host leaves the fit loop and the numpy oracle can restate them bit for bit.
//
//   brBootstrapDataset   NeuMFModel.bootstrapDataset (src/models/NeuMFModel.py:102-109): positives + round(negRatio*n) rows
//                        sampled WITH replacement whose item column is permuted (no collision check), labels 1 / 0, shuffled.
//   brBprSampleTriplets  BPR triplets (src/models/BPRModel.py:94-98,111-119).  The reference enumerates every (positive,
//                        non-interacted item) pair of every customer: O(U*I) rows.  Here: `neg_per_pos` negatives per positive,
//                        drawn uniformly from the candidate items and rejected while they are positives of the customer.
//   brNcfNegativeCandidates  Data handling/synthetic.py:208-223,237-256 (generateSyntethic / generateNegativeFeedback): candidate
//                        negatives = the customer column and the product column shuffled independently, round after round; pairs
//                        that are positives are marked invalid.  brSortUniqueKeys64 + brGatherPermutedPairs finish the contract
//                        (distinct pairs, head(size)).
//
// Randomness: uniform draws = Philox4x32-10 (csrc/philox.h) of the counter (index, attempt, stream); permutations = a 4-round
// Feistel network over ceil(log2 M) bits with cycle walking (a keyed bijection of [0, M): O(1) per element, no sort, no state).
// [pandas-sem] the reference is unseeded and uses pandas' sample(): only the distributions are restated, like the initialisers.
#include "common.h"
#include "philox.h"

#include <hipcub/hipcub.hpp>

namespace br {

__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {      // murmur3 finaliser
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}
struct PermKey { uint32_t k[4]; };
__host__ __device__ __forceinline__ PermKey perm_key(uint64_t seed, uint32_t stream) {
  PermKey p;
  for (int r = 0; r < 4; ++r) p.k[r] = mix32((uint32_t)seed + 0x9E3779B9u * (uint32_t)(r + 1)) ^ mix32((uint32_t)(seed >> 32) ^ (stream * 0x85ebca6bu + (uint32_t)r));
  return p;
}
// keyed bijection of [0, M), M >= 1
__host__ __device__ __forceinline__ uint32_t feistel_perm(uint32_t x, uint32_t M, const PermKey& key) {
  if (M <= 1) return 0;
  int b = 1;
  while (b < 32 && (1ull << b) < (uint64_t)M) ++b;
  const int hb = (b + 1) >> 1;
  const uint32_t mask = (1u << hb) - 1u;
  do {
    uint32_t L = x >> hb, R = x & mask;
    for (int r = 0; r < 4; ++r) {
      const uint32_t F = mix32(R ^ key.k[r]) & mask;
      const uint32_t t = L ^ F;
      L = R; R = t;
    }
    x = (L << hb) | R;
  } while (x >= M);
  return x;
}
// uniform integer in [0, n) from one Philox call: floor(u32 * n / 2^32)
__device__ __forceinline__ uint32_t draw_below(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t stream, uint32_t n) {
  const Philox4 d = philox4x32_10(c0, c1, stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
  return (uint32_t)(((uint64_t)d.x * (uint64_t)n) >> 32);
}

template <typename IdT>
__global__ __launch_bounds__(256) void bootstrap_kernel(const IdT* __restrict__ users, const IdT* __restrict__ items, uint32_t n, uint32_t K, uint64_t seed,
                                                         IdT* __restrict__ ou, IdT* __restrict__ oi, float* __restrict__ oy) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n + K) return;
  const uint32_t src = feistel_perm(t, n + K, perm_key(seed, 1u));           // mergeDf.sample(frac=1.)
  if (src < n) { ou[t] = users[src]; oi[t] = items[src]; oy[t] = 1.f; return; }
  const uint32_t j = src - n;
  const uint32_t a = draw_below(seed, j, 0u, 2u, n);                           // negDf = df.sample(frac=negRatio, replace=True)
  const uint32_t b = draw_below(seed, feistel_perm(j, K, perm_key(seed, 3u)), 0u, 2u, n);   // negDf.PRODUCT_ID.sample(frac=1.).values
  ou[t] = users[a]; oi[t] = items[b]; oy[t] = 0.f;
}

// membership of `item` in the sorted positive list of `user` (CSR)
template <typename IdT>
__device__ __forceinline__ bool is_positive(const int64_t* __restrict__ off, const IdT* __restrict__ pos_items, int64_t user, IdT item) {
  int64_t lo = off[user], hi = off[user + 1];
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    const IdT v = pos_items[mid];
    if (v < item) lo = mid + 1; else hi = mid;
  }
  return lo < off[user + 1] && pos_items[lo] == item;
}

template <typename IdT>
__global__ __launch_bounds__(256) void bpr_triplets_kernel(const IdT* __restrict__ users, const IdT* __restrict__ items, uint32_t n, int neg_per_pos,
                                                            const int64_t* __restrict__ pos_off, const IdT* __restrict__ pos_items,
                                                            const IdT* __restrict__ cand, uint32_t n_cand, uint64_t seed, int max_tries,
                                                            IdT* __restrict__ ou, IdT* __restrict__ op, IdT* __restrict__ on) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if ((uint64_t)t >= (uint64_t)n * neg_per_pos) return;
  const uint32_t row = t / (uint32_t)neg_per_pos;
  const IdT u = users[row];
  IdT neg = 0;
  for (int a = 0; a < max_tries; ++a) {
    const uint32_t c = draw_below(seed, t, (uint32_t)a, 4u, n_cand);
    neg = cand ? cand[c] : (IdT)c;
    if (!is_positive(pos_off, pos_items, (int64_t)u, neg)) break;             // the last candidate stands after max_tries
  }
  ou[t] = u; op[t] = items[row]; on[t] = neg;
}

// candidate j = (round j / n, slot j % n): customer column and product column shuffled independently (generateSyntethic);
// key = user * num_items + item, or ~0 when the pair is a positive
template <typename IdT>
__global__ __launch_bounds__(256) void ncf_candidates_kernel(const IdT* __restrict__ users, const IdT* __restrict__ items, uint32_t n, uint64_t n_cand,
                                                              const int64_t* __restrict__ pos_off, const IdT* __restrict__ pos_items, int64_t num_items,
                                                              uint64_t seed, uint64_t* __restrict__ keys) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_cand) return;
  const uint32_t round = (uint32_t)(j / n), slot = (uint32_t)(j % n);
  const IdT u = users[feistel_perm(slot, n, perm_key(seed, 16u + 2u * round))];
  const IdT i = items[feistel_perm(slot, n, perm_key(seed, 17u + 2u * round))];
  keys[j] = is_positive(pos_off, pos_items, (int64_t)u, i) ? ~0ull : (uint64_t)u * (uint64_t)num_items + (uint64_t)i;
}

template <typename IdT>
__global__ __launch_bounds__(256) void gather_permuted_pairs_kernel(const uint64_t* __restrict__ keys, uint32_t n_keys, uint32_t n_out, int64_t num_items,
                                                                     uint64_t seed, IdT* __restrict__ ou, IdT* __restrict__ oi) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_out) return;
  const uint64_t k = keys[feistel_perm(t, n_keys, perm_key(seed, 5u))];
  ou[t] = (IdT)(k / (uint64_t)num_items);
  oi[t] = (IdT)(k % (uint64_t)num_items);
}

}  // namespace br

using namespace br;

extern "C" int brBootstrapDataset(const void* users, const void* items, int id_type, int64_t n, int64_t n_neg, uint64_t seed, void* out_users,
                                  void* out_items, float* out_labels, brStream stream) {
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brBootstrapDataset: bad id_type");
  BR_CHECK_ARG(n >= 1 && n_neg >= 0 && n + n_neg < ((int64_t)1 << 31) && users && items && out_users && out_items && out_labels, "brBootstrapDataset: bad args");
  const unsigned grid = (unsigned)ceil_div(n + n_neg, 256);
  hipStream_t s = (hipStream_t)stream;
  if (id_type == BR_IDS_I32)
    bootstrap_kernel<int32_t><<<grid, 256, 0, s>>>((const int32_t*)users, (const int32_t*)items, (uint32_t)n, (uint32_t)n_neg, seed, (int32_t*)out_users, (int32_t*)out_items, out_labels);
  else
    bootstrap_kernel<int64_t><<<grid, 256, 0, s>>>((const int64_t*)users, (const int64_t*)items, (uint32_t)n, (uint32_t)n_neg, seed, (int64_t*)out_users, (int64_t*)out_items, out_labels);
  BR_CHECK_LAUNCH("brBootstrapDataset");
  return BR_OK;
}

extern "C" int brBprSampleTriplets(const void* users, const void* items, int id_type, int64_t n, int neg_per_pos, const int64_t* pos_off,
                                   const void* pos_items, const void* cand_items, int64_t n_cand, uint64_t seed, int max_tries, void* out_users,
                                   void* out_pos, void* out_neg, brStream stream) {
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brBprSampleTriplets: bad id_type");
  BR_CHECK_ARG(n >= 1 && neg_per_pos >= 1 && n * neg_per_pos < ((int64_t)1 << 31) && n_cand >= 1 && n_cand < ((int64_t)1 << 32) && max_tries >= 1 && users && items &&
               pos_off && pos_items && out_users && out_pos && out_neg, "brBprSampleTriplets: bad args");
  const unsigned grid = (unsigned)ceil_div(n * neg_per_pos, 256);
  hipStream_t s = (hipStream_t)stream;
  if (id_type == BR_IDS_I32)
    bpr_triplets_kernel<int32_t><<<grid, 256, 0, s>>>((const int32_t*)users, (const int32_t*)items, (uint32_t)n, neg_per_pos, pos_off, (const int32_t*)pos_items,
                                                      (const int32_t*)cand_items, (uint32_t)n_cand, seed, max_tries, (int32_t*)out_users, (int32_t*)out_pos, (int32_t*)out_neg);
  else
    bpr_triplets_kernel<int64_t><<<grid, 256, 0, s>>>((const int64_t*)users, (const int64_t*)items, (uint32_t)n, neg_per_pos, pos_off, (const int64_t*)pos_items,
                                                      (const int64_t*)cand_items, (uint32_t)n_cand, seed, max_tries, (int64_t*)out_users, (int64_t*)out_pos, (int64_t*)out_neg);
  BR_CHECK_LAUNCH("brBprSampleTriplets");
  return BR_OK;
}

extern "C" int brNcfNegativeCandidates(const void* users, const void* items, int id_type, int64_t n, int64_t n_cand, const int64_t* pos_off,
                                       const void* pos_items, int64_t num_items, uint64_t seed, uint64_t* keys, brStream stream) {
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brNcfNegativeCandidates: bad id_type");
  BR_CHECK_ARG(n >= 1 && n < ((int64_t)1 << 31) && n_cand >= 1 && num_items >= 1 && users && items && pos_off && pos_items && keys, "brNcfNegativeCandidates: bad args");
  const unsigned grid = (unsigned)ceil_div(n_cand, 256);
  hipStream_t s = (hipStream_t)stream;
  if (id_type == BR_IDS_I32)
    ncf_candidates_kernel<int32_t><<<grid, 256, 0, s>>>((const int32_t*)users, (const int32_t*)items, (uint32_t)n, (uint64_t)n_cand, pos_off, (const int32_t*)pos_items, num_items, seed, keys);
  else
    ncf_candidates_kernel<int64_t><<<grid, 256, 0, s>>>((const int64_t*)users, (const int64_t*)items, (uint32_t)n, (uint64_t)n_cand, pos_off, (const int64_t*)pos_items, num_items, seed, keys);
  BR_CHECK_LAUNCH("brNcfNegativeCandidates");
  return BR_OK;
}

// sort + unique of 64-bit keys (hipcub device primitives; caller scratch).  out_keys: n keys; n_unique: device int64 (the ~0 key of the
// invalid candidates, if any, sorts last and counts as one).
extern "C" int64_t brSortUniqueWorkspaceBytes(int64_t n) {
  if (n <= 0) return 256;
  size_t a = 0, b = 0;
  (void)hipcub::DeviceRadixSort::SortKeys((void*)nullptr, a, (const uint64_t*)nullptr, (uint64_t*)nullptr, (int)n, 0, 64, (hipStream_t)0);
  (void)hipcub::DeviceSelect::Unique((void*)nullptr, b, (const uint64_t*)nullptr, (uint64_t*)nullptr, (int64_t*)nullptr, (int)n, (hipStream_t)0);
  return (int64_t)((a > b ? a : b) + (size_t)n * 8 + 512);
}
extern "C" int brSortUniqueKeys64(const uint64_t* keys, int64_t n, uint64_t* out_keys, int64_t* n_unique, void* workspace, int64_t workspace_bytes,
                                  brStream stream) {
  BR_CHECK_ARG(keys && out_keys && n_unique && workspace && n >= 1 && n < ((int64_t)1 << 31), "brSortUniqueKeys64: bad args");
  BR_CHECK_ARG(workspace_bytes >= brSortUniqueWorkspaceBytes(n), "brSortUniqueKeys64: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  uint64_t* sorted = (uint64_t*)workspace;
  void* tmp = (char*)workspace + ((n * 8 + 255) & ~(int64_t)255);
  size_t tmp_bytes = (size_t)(workspace_bytes - ((n * 8 + 255) & ~(int64_t)255));
  hipError_t e = hipcub::DeviceRadixSort::SortKeys(tmp, tmp_bytes, keys, sorted, (int)n, 0, 64, s);
  if (e == hipSuccess) e = hipcub::DeviceSelect::Unique(tmp, tmp_bytes, sorted, out_keys, n_unique, (int)n, s);
  if (e != hipSuccess) { set_error("brSortUniqueKeys64: %s", hipGetErrorString(e)); return BR_ERR_HIP; }
  return BR_OK;
}

extern "C" int brGatherPermutedPairs(const uint64_t* keys, int64_t n_keys, int64_t n_out, int64_t num_items, uint64_t seed, int id_type, void* out_users,
                                     void* out_items, brStream stream) {
  BR_CHECK_ARG(id_type == BR_IDS_I32 || id_type == BR_IDS_I64, "brGatherPermutedPairs: bad id_type");
  BR_CHECK_ARG(keys && out_users && out_items && n_keys >= 1 && n_keys < ((int64_t)1 << 31) && n_out >= 0 && n_out <= n_keys && num_items >= 1, "brGatherPermutedPairs: bad args");
  if (n_out == 0) return BR_OK;
  const unsigned grid = (unsigned)ceil_div(n_out, 256);
  hipStream_t s = (hipStream_t)stream;
  if (id_type == BR_IDS_I32)
    gather_permuted_pairs_kernel<int32_t><<<grid, 256, 0, s>>>(keys, (uint32_t)n_keys, (uint32_t)n_out, num_items, seed, (int32_t*)out_users, (int32_t*)out_items);
  else
    gather_permuted_pairs_kernel<int64_t><<<grid, 256, 0, s>>>(keys, (uint32_t)n_keys, (uint32_t)n_out, num_items, seed, (int64_t*)out_users, (int64_t*)out_items);
  BR_CHECK_LAUNCH("brGatherPermutedPairs");
  return BR_OK;
}
