// E1/E2 companions on the device: the evaluation functions of src/models/bpr.py and the hit counting of
// trainers/topKmetrics.py, on top of brScoreMatrix / brTopKRows.  The ground truth of every user is a CSR list of COLUMN
// indices into the scored item list, ascending (truth_off[U + 1], truth_idx).
//   brFullAuc     full_auc (bpr.py:230-254): per user sklearn.roc_auc_score(ground truth, scores over ALL items) = the
//                 Mann-Whitney statistic with ties counted one half; users without positives (or with nothing but positives:
//                 roc_auc_score raises there) get NaN and are left out of the mean by the caller (the reference skips the
//                 former, bpr.py:251).
//   brMapAtK      mean_average_precision_k (bpr.py:257-289): AP@k of the top-k list (brTopKRows order: descending, ties keep
//                 the lower item position = Python's stable sorted(..., reverse=True)) / min(len(actual), k).
//   brHitCounts   topKMetrics (topKmetrics.py:74-99): hits of each user's top-k list among its positives (tp per user).
#include "common.h"

namespace br {

__device__ __forceinline__ bool in_sorted(const int32_t* __restrict__ v, int64_t lo, int64_t hi, int32_t x) {
  const int64_t end = hi;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (v[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo < end && v[lo] == x;
}

// one workgroup per user: for every positive p: #items scoring below it + half of the ties, over all items, minus the same over
// the positives -> counts against the negatives only
__global__ __launch_bounds__(256) void full_auc_kernel(const float* __restrict__ scores, int64_t ld, const int64_t* __restrict__ off,
                                                        const int32_t* __restrict__ idx, int64_t n_items, float* __restrict__ auc) {
  __shared__ double red[4];
  const int64_t u = blockIdx.x;
  const int64_t p0 = off[u], p1 = off[u + 1];
  const int64_t P = p1 - p0, N = n_items - P;
  if (P <= 0 || N <= 0) {
    if (threadIdx.x == 0) auc[u] = __builtin_nanf("");
    return;
  }
  const float* row = scores + u * ld;
  double total = 0.0;           // thread-local share of sum_p (below + ties/2) over all items
  for (int64_t p = p0; p < p1; ++p) {
    const float s = row[idx[p]];
    uint32_t below = 0, ties = 0;
    for (int64_t i = threadIdx.x; i < n_items; i += blockDim.x) {
      const float v = row[i];
      below += v < s ? 1u : 0u;
      ties += v == s ? 1u : 0u;
    }
    total += (double)below + 0.5 * (double)ties;
  }
  // the positives' own contribution (P is small): positive q against positive p
  for (int64_t t = threadIdx.x; t < P * P; t += blockDim.x) {
    const float s = row[idx[p0 + t / P]], v = row[idx[p0 + t % P]];
    total -= (v < s ? 1.0 : 0.0) + (v == s ? 0.5 : 0.0);
  }
  total = wave_sum_d(total);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = total;
  __syncthreads();
  if (threadIdx.x == 0) auc[u] = (float)((red[0] + red[1] + red[2] + red[3]) / ((double)P * (double)N));
}

__global__ __launch_bounds__(256) void map_at_k_kernel(const int32_t* __restrict__ topk, int k, const int64_t* __restrict__ off,
                                                        const int32_t* __restrict__ idx, int64_t n_users, float* __restrict__ ap,
                                                        int32_t* __restrict__ hits) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_users) return;
  const int64_t p0 = off[u], p1 = off[u + 1];
  double score = 0.0, nh = 0.0;
  int h = 0;
  for (int i = 0; i < k; ++i) {
    if (in_sorted(idx, p0, p1, topk[u * k + i])) { nh += 1.0; ++h; score += nh / (double)(i + 1); }
  }
  const int64_t P = p1 - p0;
  if (ap) ap[u] = P > 0 ? (float)(score / (double)(P < k ? P : k)) : 0.f;
  if (hits) hits[u] = h;
}

}  // namespace br

using namespace br;

extern "C" int brFullAuc(const float* scores, int64_t ld_scores, const int64_t* truth_off, const int32_t* truth_idx, int64_t n_users,
                         int64_t n_items, float* auc, brStream stream) {
  BR_CHECK_ARG(scores && truth_off && truth_idx && auc && n_users >= 0 && n_items >= 1 && ld_scores >= n_items, "brFullAuc: bad args");
  if (n_users == 0) return BR_OK;
  full_auc_kernel<<<(unsigned)n_users, 256, 0, (hipStream_t)stream>>>(scores, ld_scores, truth_off, truth_idx, n_items, auc);
  BR_CHECK_LAUNCH("brFullAuc");
  return BR_OK;
}

extern "C" int brMapAtK(const int32_t* topk_index, int64_t n_users, int k, const int64_t* truth_off, const int32_t* truth_idx, float* ap,
                        int32_t* hits, brStream stream) {
  BR_CHECK_ARG(topk_index && truth_off && truth_idx && (ap || hits) && n_users >= 0 && k >= 1, "brMapAtK: bad args");
  if (n_users == 0) return BR_OK;
  map_at_k_kernel<<<(unsigned)ceil_div(n_users, 256), 256, 0, (hipStream_t)stream>>>(topk_index, k, truth_off, truth_idx, n_users, ap, hits);
  BR_CHECK_LAUNCH("brMapAtK");
  return BR_OK;
}
