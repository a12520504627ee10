/*
 * binrec.h — C-ABI of libbinrec_hip.so: the MI355X (gfx950) hot path of the
 * NeuMF / BPR / TwoTower trainers of leotimus/binary-recommendation.
 *
 * The reference has NO plugin / FFI / operator interface for this path: every op is a stock
 * Keras layer declared in Python.  Each entry point below therefore cites the
 * reference *call site* (path:line under /root/reference) whose arithmetic it replaces; the
 * Python host in binary-recommendation_amd/ re-exposes the reference's model surface
 * (compileModel / fit / train_step / predict ...) on top of these.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch types.  Every function returns int:
 *    BR_OK (0) or a negative BR_ERR_*; brGetLastError() gives the thread-local message.
 *  - Caller owns ALL memory (device buffers, e.g. torch tensors' data_ptr()); scratch needs are queried
 *    (br*WorkspaceBytes) and passed in.  The library allocates no device memory and does not synchronise on a step's
 *    path.  What it does keep, per process: the thread-local error string; four fork / join events for brNeumfStep.aux_stream
 *    (created on first use: ONE stepping thread per process, as everywhere in this build); the optional measurement probes
 *    (brProbe*, brProbeGraph*: HIP events created by brProbeEnable / brProbeGraphSelect, never in a launch path); and, during a
 *    brNeumfStepRun call, a pointer to that call's step state.  Two entry points synchronise their stream because they copy from
 *    pageable host memory: brStepStateSet and brStepStateInit (set-up / reload, never inside a step).
 *  - Every launch goes to the caller's stream (`brStream` = hipStream_t).
 *  - Tables are row-major fp32 [rows][dim].  ids are int32 or int64 (BR_IDS_*); ids == NULL
 *    means identity (row b of an already-gathered [batch][dim] buffer: the row-sharded
 *    multi-GPU path hands rows over that way).
 *  - Out-of-range ids never fault: the access is skipped (zeros are produced) and *err_flag
 *    (device int, may be NULL) is set to 1.  TF-CPU raises InvalidArgument there [TF-sem];
 *    the Python host turns the flag into an IndexError.
 *  - Dropout masks are a pure function (Philox4x32-10) of (seed, step, site, global row, col):
 *    see csrc/philox.h; restated bit-exactly in oracle/binrec_oracle.py.
 */
#ifndef BINREC_H
#define BINREC_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* brStream; /* hipStream_t */

enum { BR_OK = 0, BR_ERR_ARG = -1, BR_ERR_HIP = -2, BR_ERR_UNSUPPORTED = -3, BR_ERR_WORKSPACE = -4 };
enum { BR_IDS_I32 = 0, BR_IDS_I64 = 1 };
enum { BR_ACT_LINEAR = 0, BR_ACT_SIGMOID = 1, BR_ACT_RELU = 2 };
enum { BR_LOSS_BCE = 0, BR_LOSS_MSE = 1 };
enum { BR_MAX_TABLES = 8, BR_SUM_SLOTS = 64, BR_STAT_REPLICAS = 8, BR_METRIC_SUMS = 8 };

const char* brGetLastError(void);
int brVersion(void);
/* cu_count / arch string of the current device (host-side query, no launch). */
int brDeviceInfo(int* cu_count, int* wave_size, char* arch, int arch_len);

/* ---- G1/G2: Embedding(...)(input) + Flatten ------------------------------------------------
 * NFC_plain.py:115-126, NeuMFModel.py:58-63, BPRModel.py:55-61, twoTower.py:34,36.
 * outs[t][b,:] = tables[t][ids[t][b],:] for t < n_tables, fused in one launch. Bit-exact. */
int brGatherRows(int n_tables, const float* const* tables, const int64_t* table_rows,
                 const void* const* ids, float* const* outs, int dim, int64_t batch,
                 int id_type, int* err_flag, brStream stream);

/* ---- M1: Dot(axes=1) — NFC_plain.py:148, NeuMFModel.py:79, twoTower.py:44,86 ----------------
 * out[b] = sum_d a[b,d]*b[b,d];  backward: da = dout*b, db = dout*a. */
int brRowDot(const float* a, const float* b, float* out, int dim, int64_t batch, brStream stream);
int brRowDotBackward(const float* a, const float* b, const float* dout, float* da, float* db,
                     int dim, int64_t batch, brStream stream);

/* ---- G1+M1+T1 fused for NeuMF: 4 lookups, GMF dot, MLP concat ------------------------------
 * NFC_plain.py:115-126,137,148 (item_first=1) / NeuMFModel.py:58-66,79 (item_first=0).
 * x0[b] = concat(first_mlp[b], second_mlp[b]) (B x 2*dim), dot[b] = user_mf[u]·item_mf[i].
 * ld_user / ld_item = row strides (floats) of the user / item tables: `dim` for four separate
 * tables, 2*dim when a stream's MLP and MF tables are stored interleaved as one [rows][mlp|mf]
 * allocation (then user_mf = user_mlp + dim): one 512-B row instead of two 256-B rows per id. */
int brNeumfEmbedForward(const float* user_mlp, const float* item_mlp, const float* user_mf,
                        const float* item_mf, int64_t ld_user, int64_t ld_item, int64_t user_rows,
                        int64_t item_rows, const void* users, const void* items, int id_type,
                        int dim, int64_t batch, int item_first, float* x0, float* dot,
                        int* err_flag, brStream stream);
/* The same, also leaving the two MF rows of every pair behind (stash_user[b] = user_mf[u_b], stash_item[b] = item_mf[i_b], stride ld_stash):
 * the backward of the GMF dot is then ddot[b] * the partner's stash, formed where it is consumed (brSegmentSumToSlotsPair,
 * brAdamRowsSortedPair hi_scale) instead of by a launch that writes it out per pair. */
int brNeumfEmbedForwardStash(const float* user_mlp, const float* item_mlp, const float* user_mf,
                             const float* item_mf, int64_t ld_user, int64_t ld_item, int64_t user_rows,
                             int64_t item_rows, const void* users, const void* items, int id_type, int dim,
                             int64_t batch, int item_first, float* x0, float* dot, float* stash_user, float* stash_item, int64_t ld_stash,
                             int* err_flag, brStream stream);
/* B1 for the same block: per-pair row gradients (IndexedSlices values, [TF-sem]).
 * dx0: (B x 2*dim) gradient w.r.t. the concat; ddot: (B).
 * g_user_mf[b] = ddot[b]*item_mf[i_b], g_item_mf[b] = ddot[b]*user_mf[u_b];
 * g_*_mlp (optional, both or neither) = the two halves of dx0 in concat order.
 * All outputs have row stride ldg (dim, or 2*dim for fused [mlp|mf] gradient buffers). */
int brNeumfEmbedBackward(const float* user_mf, const float* item_mf, int64_t ld_user,
                         int64_t ld_item, int64_t user_rows, int64_t item_rows, const void* users,
                         const void* items, int id_type, int dim, int64_t batch, int item_first,
                         const float* dx0, const float* ddot, float* g_user_mlp, float* g_item_mlp,
                         float* g_user_mf, float* g_item_mf, int64_t ldg, int out_rows_by_id,
                         brStream stream);   /* out_rows_by_id: pair b's gradients go to rows users[b] / items[b] of the outputs
                                                (row-sharded host: ids = slots of the padded exchange buffers) instead of row b */

/* ---- L3: BPR triplet step — BPRModel.py:49-74,124-144; bpr.py:141-157 -----------------------
 * x = u·p - u·n ; l = 1 - sigmoid(x) ; loss = mean(l).  One fused launch: 3 gathers, 2 dots,
 * loss partials, 3 row gradients.  g_item is (2B x dim): rows [0,B) = d/d pos, [B,2B) = d/d neg
 * (the shared item table's two IndexedSlices concatenated).  loss_sum: device
 * double[BR_SUM_SLOTS], slot (workgroup & 63) += sum l (same-address atomics serialise at
 * ~12 ns each on MI355X; the caller adds the slots).  per_triplet (B) may be NULL.
 * inv_batch = 1/global batch. */
int brBprForwardBackward(const float* user_table, const float* item_table, int64_t user_rows,
                         int64_t item_rows, const void* users, const void* pos, const void* neg,
                         int id_type, int dim, int64_t batch, float inv_batch, float* per_triplet,
                         double* loss_sum, float* g_user, float* g_item, int* err_flag,
                         brStream stream);

/* ---- S1: duplicate-id handling ([TF-sem] _deduplicate_indexed_slices) ----------------------
 * brRowIndexBuild: stable radix sort of (id, batch position) -> sorted_ids, sorted_pos (int32).
 * Equal ids stay in ascending batch position, so every later segment sum runs in exactly the
 * order of a sequential unsorted_segment_sum: bitwise reproducible, no float atomics. */
int64_t brRowIndexWorkspaceBytes(int64_t n, int id_type);
int brRowIndexBuild(const void* ids, int id_type, int64_t n, int64_t id_upper_bound,
                    void* sorted_ids, int32_t* sorted_pos, void* workspace,
                    int64_t workspace_bytes, brStream stream);
/* Both id streams of a step (same n) in shared launches: for n <= 524 288 (64 chunks of 8192) and id bounds < 2^31 the index is
 * built in TWO launches for the pair (chunk sort in LDS + rank-by-binary-search scatter) instead of one device
 * radix sort (~10 launches) per stream.  Same outputs as two brRowIndexBuild calls. */
int brRowIndexBuildPair(const void* ids_a, int64_t upper_a, void* sorted_ids_a, int32_t* sorted_pos_a, void* ws_a, int64_t ws_a_bytes,
                        const void* ids_b, int64_t upper_b, void* sorted_ids_b, int32_t* sorted_pos_b, void* ws_b, int64_t ws_b_bytes,
                        int id_type, int64_t n, brStream stream);
/* Row-sharded exchange planning for one or two equally long id streams (ids_b NULL: one): owner(id) = id mod world.
 * order[j] = batch position of bucket slot j (buckets = positions stably sorted by owner), inv[b] = bucket slot of position b,
 * send_local[j] = id div world in bucket order, counts[d] = rows for owner d (int64[world]).  dest / sorted_dest: scratch of n
 * ids each; ws: brRowIndexWorkspaceBytes(n) bytes each. */
int brShardPlanPair(const void* ids_a, const void* ids_b, int id_type, int64_t n, int world, void* dest_a, void* dest_b,
                    void* sorted_dest_a, void* sorted_dest_b, int32_t* order_a, int32_t* order_b, void* ws_a, void* ws_b,
                    int64_t ws_bytes, int32_t* inv_a, int32_t* inv_b, void* send_local_a, void* send_local_b,
                    int64_t* counts_a, int64_t* counts_b, brStream stream);
/* Fixed-capacity form of that plan: every owner gets exactly `cap` slots per stream (slot d*cap + k = the k-th row for owner d in
 * bucket order), so the three all-to-alls of a step have equal, static splits and no row count crosses to the host.
 * send_pad[world*cap]: local row ids, pad slots hold the owner's SPARE row (local index = the owner's row count; a padded engine
 * allocates one row more per table, its gradient is always 0); slot[n]: batch position -> slot; bpos[world*cap]: slot -> batch
 * position or -1.  More than `cap` rows for one owner sets bit BR_ERRFLAG_CAPACITY in *err_flag (the surplus rows collide in the
 * owner's last slot: the step's result is then wrong and the host must raise). */
enum { BR_ERRFLAG_RANGE = 1, BR_ERRFLAG_CAPACITY = 2 };
int brShardPadPair(const void* sorted_dest_a, const void* sorted_dest_b, const int32_t* order_a, const int32_t* order_b,
                   const void* send_local_a, const void* send_local_b, const int64_t* counts_a, const int64_t* counts_b, int id_type,
                   int64_t n, int world, int64_t cap, int64_t total_rows_a, int64_t total_rows_b, void* send_pad_a, void* send_pad_b,
                   int32_t* slot_a, int32_t* slot_b, int32_t* bpos_a, int32_t* bpos_b, float* zero_a, float* zero_b, int zero_dim,
                   int* err_flag, brStream stream);   /* zero_*: optional [world*cap][zero_dim] buffers whose pad rows are cleared
                                                         (the gradient send slots, written in place by brNeumfEmbedBackward out_rows_by_id) */
/* ---- The same exchange with duplicate ids MERGED and both id streams in ONE all-to-all buffer per phase (parallel.py PaddedExchange) ----
 * Replaces the reference's mirrored variables + all-reduced IndexedSlices (src/models/RModel.py:119, NeuMFModel.py:92-98) for one
 * NeuMF step: owner(id) = id mod world serves each DISTINCT id of the batch once.
 * Requester side, brShardDedupPlanPair: key = owner * R + id div world (R = rows per owner, rounded up, + 1), (key, position) sorted;
 * the k-th distinct key of owner d gets physical slot (d * 2 + stream) * cap + k of the merged buffers [owner][user | item][cap]
 * (ids out: send_ids; rows back and gradients out: the same slot numbering, `dim` floats per slot); slot[b] = the slot of batch
 * position b's id or -1; sorted_keys / sorted_pos stay behind as the dedup index of the backward (brSegmentSumToSlotsPair).  Pad slots
 * name the owner's spare row (index = its row count) and their rows of grad_slots (optional, zero_dim floats each) are cleared.
 * More than cap DISTINCT ids for one owner and stream: the surplus ids get slot -1 (the step reads zeros for them and sends no
 * gradient - no slot is ever written twice, no table row sees another row's gradient) and BR_ERRFLAG_CAPACITY is set; ids outside
 * [0, total_rows): slot -1, BR_ERRFLAG_RANGE.  Scratch: keys / urank [n] each, first [world + 1] int32 each, ws brRowIndexWorkspaceBytes(n). */
int brShardDedupPlanPair(const void* ids_a, const void* ids_b, int id_type, int64_t n, int world, int64_t cap, int64_t total_rows_a,
                         int64_t total_rows_b, void* keys_a, void* keys_b, void* sorted_keys_a, void* sorted_keys_b, int32_t* sorted_pos_a,
                         int32_t* sorted_pos_b, void* ws_a, void* ws_b, int64_t ws_bytes, int32_t* urank_a, int32_t* urank_b, int32_t* first_a,
                         int32_t* first_b, void* send_ids, int32_t* slot_a, int32_t* slot_b, float* grad_slots, int zero_dim, int* err_flag,
                         brStream stream);
/* Backward of that plan: out_slots[slot[p]] = ordered (two-level, see below) sum over the positions p' of p's id of the per-pair row
 * gradient [g0[p'] (split floats, stride ldg0) | hi_scale[p'] * g1[p'] (dim - split floats, stride ldg1)] - for a NeuMF step g0 = the
 * MLP half of dx0, g1 = the PARTNER stream's stashed MF rows, hi_scale = ddot (brNeumfEmbedBackward's products without writing them
 * out per pair).  g1_* NULL: one source (split = dim).  seg_ws_*: brSegmentScratchFloats(n, dim) each, or both NULL. */
int brSegmentSumToSlotsPair(const void* sorted_ids_a, const int32_t* sorted_pos_a, const int32_t* slot_a, const float* g0_a, const float* g1_a,
                            const void* sorted_ids_b, const int32_t* sorted_pos_b, const int32_t* slot_b, const float* g0_b, const float* g1_b,
                            int64_t ldg0, int64_t ldg1, const float* hi_scale, int id_type, int64_t n, int dim, int split, float* out_slots,
                            float* seg_ws_a, float* seg_ws_b, brStream stream);
/* Owner side: the ids arrive as ONE array [source rank][stream][cap]; stream k's logical position t = src * cap + j sits at element
 * (src * 2 + k) * cap + j (seg_len = cap, seg_stride = 2 * cap, seg_off = k * cap).  brRowIndexBuildPairSeg = brRowIndexBuildPair over
 * such arrays, sorted_pos = the PHYSICAL element index (so the optimizer kernels read received gradient rows / served rows of the same
 * layout unchanged); brGatherRows[Deferred]PairSeg: out[p] = row ids[p] of the stream's table at every physical position p of both
 * streams, one launch. */
int brRowIndexBuildPairSeg(const void* ids_a, int64_t upper_a, void* sorted_ids_a, int32_t* sorted_pos_a, void* ws_a, int64_t ws_a_bytes,
                           const void* ids_b, int64_t upper_b, void* sorted_ids_b, int32_t* sorted_pos_b, void* ws_b, int64_t ws_b_bytes,
                           int id_type, int64_t n, int64_t seg_len, int64_t seg_stride, int64_t seg_off_a, int64_t seg_off_b, brStream stream);
/* The same index when every segment is ALREADY sorted ascending - what the fixed-capacity exchange delivers: each requester sends an owner
 * its distinct local rows in key order with the pads (the spare row = the largest id) behind them, so the W segments of a stream are W
 * sorted runs and the index is their merge (one binary search per other run and key, no sort, no workspace).  Output identical to
 * brRowIndexBuildPairSeg.  A segment that is not sorted sets BR_ERRFLAG_RANGE in *err_flag (may be NULL); the index is then wrong.
 * Replaces on the owners of parallel.py what RModel.py:119's MultiWorkerMirroredStrategy does with an all-reduce of IndexedSlices. */
int brRowIndexMergePairSeg(const void* ids_a, int64_t upper_a, void* sorted_ids_a, int32_t* sorted_pos_a, const void* ids_b, int64_t upper_b,
                           void* sorted_ids_b, int32_t* sorted_pos_b, int id_type, int64_t n, int64_t seg_len, int64_t seg_stride, int64_t seg_off_a,
                           int64_t seg_off_b, int* err_flag, brStream stream);
int brGatherRowsDeferredPairSeg(const float* table_a, const float* m_a, const float* v_a, const int32_t* last_a, int64_t rows_a, const float* table_b,
                                const float* m_b, const float* v_b, const int32_t* last_b, int64_t rows_b, const void* ids, float* out, int dim,
                                int id_type, int64_t n, int64_t seg_len, int64_t seg_stride, int64_t seg_off_a, int64_t seg_off_b,
                                const void* step_state, double beta1, double beta2, double eps, int64_t ld_out, int* err_flag, brStream stream);
int brGatherRowsPairSeg(const float* table_a, int64_t rows_a, const float* table_b, int64_t rows_b, const void* ids, float* out, int dim, int id_type,
                        int64_t n, int64_t seg_len, int64_t seg_stride, int64_t seg_off_a, int64_t seg_off_b, int64_t ld_out, int* err_flag,
                        brStream stream);
/* dst[t] = bpos[t] >= 0 ? src[bpos[t]] : 0 (n_slots rows of dim floats; src rows at stride ld): per-pair rows -> padded send slots,
 * for one or two sets of equal shape (set b NULL: one). */
int brRowsToSlotsPair(const float* src_a, const float* src_b, int64_t ld, const int32_t* bpos_a, const int32_t* bpos_b, float* dst_a,
                      float* dst_b, int64_t n_slots, int dim, brStream stream);
/* The ordered duplicate sum and its scratch.  Every kernel that sums a segment (brSegmentSumRows, brAdamRowsSorted*,
 * brAdagradRowsSorted) takes an optional `seg_ws` of brSegmentScratchFloats(n, dim) floats (16-byte aligned):
 *   seg_ws == NULL: the segment head adds its duplicates one by one in ascending batch position = a sequential fp32
 *     unsorted_segment_sum ([TF-sem] _deduplicate_indexed_slices).  The chain is as long as the segment: fine for
 *     near-uniform ids, 2.3 ms for ONE launch on a Zipf(1.05) batch of 65 536 (one id holds thousands of positions).
 *   seg_ws != NULL: two levels, still a fixed order: every 64-aligned block of the SORTED order that continues its
 *     predecessor's id gets the one-by-one sum of its own run (one extra small launch, all blocks in parallel); the head
 *     adds its own positions up to the next block boundary one by one, then one partial per later block:
 *     ((((g_i + g_i+1) + ..) + P_b) + P_b+1) + .., P_b = ((g_64b + g_64b+1) + ..).  Segments that do not cross a
 *     64-aligned boundary are summed exactly as with NULL.  oracle/binrec_oracle.py::ordered_segment_sum restates it. */
int64_t brSegmentScratchFloats(int64_t n, int dim);
/* Materialised dedup: for each segment head h (sorted position), out_rows[h] = ordered sum of
 * row_grads[sorted_pos[j]] over the segment, head_flag[h]=1; non-head rows untouched, flag 0. */
int brSegmentSumRows(const void* sorted_ids, int id_type, const int32_t* sorted_pos, int64_t n,
                     const float* row_grads, int64_t ldg, int dim, float* out_rows,
                     int32_t* head_flag, float* seg_ws, brStream stream);
/* Fast form (order-nondeterministic float atomics): g_table[ids[b],:] += rows[b,:]. */
int brScatterAddRows(float* g_table, int64_t table_rows, const void* ids, int id_type,
                     int64_t n, const float* rows, int dim, int* err_flag, brStream stream);

/* ---- O1: Keras/TF-form Adam — NFC_plain.py:153, NeuMFModel.py:89, BPRModel.py:70 ------------
 * alpha_t = lr*sqrt(1-b2^t)/(1-b1^t) is computed by the caller in double; the launcher rounds
 * alpha_t, b1, 1-b1, b2, 1-b2, eps to fp32 once.  row_grads rows have stride ldg floats (so the
 * two halves of a (B x 2*dim) dx0 buffer serve as the MLP tables' row gradients in place).
 * m = b1*m+(1-b1)*g ; v = b2*v+(1-b2)*g^2 ; theta -= alpha_t*m/(sqrt(v)+eps)   (eps outside).
 * brAdamRowsSorted: touched rows only; g = ordered segment sum (dedup BEFORE the square).
 *   row_grads_hi non-NULL: columns [split, dim) of a row's gradient come from row_grads_hi
 *   (stride ldg_hi), columns [0, split) from row_grads: a fused [mlp|mf] table takes its MLP half
 *   straight from the (B x 2*dim) input gradient and its MF half from the embed backward.
 *   mark (uint8[table_rows]) non-NULL: set mark[row]=1 for touched rows (for the sweep).
 * brAdamDenseSweep: every row NOT marked gets the g=0 update (m,v decay + theta step): with
 *   brAdamRowsSorted before it this is exactly Keras' non-lazy sparse Adam; it clears marks. */
int brAdamRowsSorted(float* table, float* m, float* v, int64_t table_rows, int dim,
                     const void* sorted_ids, int id_type, const int32_t* sorted_pos, int64_t n,
                     const float* row_grads, int64_t ldg, const float* row_grads_hi,
                     int64_t ldg_hi, int split, double alpha_t, double beta1, double beta2,
                     double eps, uint8_t* mark, float* seg_ws, brStream stream);
int brAdamDenseSweep(float* table, float* m, float* v, int64_t table_rows, int dim,
                     double alpha_t, double beta1, double beta2, double eps, uint8_t* mark,
                     brStream stream);
/* Flat dense parameters (MLP weights, biases, BN gamma/beta): plain TF-form Adam. */
int brAdamFlat(float* theta, float* m, float* v, const float* g, int64_t n, double alpha_t,
               double beta1, double beta2, double eps, brStream stream);

/* ---- O2: Keras Adagrad — twoTower.py:209,278-279 ([TF-sem] acc0 = 0.1, eps 1e-7) -------------
 * acc += g^2 ; theta -= lr*g/(sqrt(acc)+eps); sparse apply touches only deduplicated rows. */
int brAdagradRowsSorted(float* table, float* acc, int64_t table_rows, int dim,
                        const void* sorted_ids, int id_type, const int32_t* sorted_pos, int64_t n,
                        const float* row_grads, int64_t ldg, double lr, double eps,
                        float* seg_ws, brStream stream);
int brAdagradFlat(float* theta, float* acc, const float* g, int64_t n, double lr, double eps,
                  brStream stream);

/* ---- T4: Dropout keep-bit planes — NFC_plain.py:138,141,144; NeuMFModel.py:67,71,75 ----------
 * The masks are a pure function (Philox4x32-10, csrc/philox.h) of (seed, step, site, global row, col).  They are
 * materialised once per (step, site) as bit planes and read by the forward and the backward of the layer behind the
 * site: keep[r][w] (uint32, row stride ceil(K/32) words), bit b of word w = element (row0 + r, 32w + b) is KEPT.
 * brDropoutKeepWords(batch, K) = words of one plane.  Up to three sites (widths[i] columns, dropout site id sites[i],
 * plane out[i]) in one launch.  Inside brNeumfStepRun `step` comes from the device step state when there is one. */
int64_t brDropoutKeepWords(int64_t batch, int K);
int brDropoutKeepBits(float drop_p, uint64_t seed, uint32_t step, int64_t row0, int64_t batch, int n_sites,
                      const uint32_t* sites, const int* widths, uint32_t* const* out, brStream stream);

/* BatchNorm finalize folded into the consumer of the normalised activations (brDenseForward / brNeumfTailFused): the
 * consumer's workgroups each turn the producer's column sums into scale / shift in their prologue (a few hundred flops),
 * workgroup 0 also writes scale / shift / mean / rstd (N each, kept for the backward) and updates the moving statistics -
 * the same arithmetic as brBnFinalize without its launch. */
typedef struct brBnFold {
  const double* stats;        /* [BR_STAT_REPLICAS][2N] column sums of y, y^2 */
  double batch_total;         /* rows behind the sums */
  const float* gamma; const float* beta;
  float eps; float momentum;
  float* moving_mean; float* moving_var;            /* may be NULL (both) */
  float* scale; float* shift; float* mean; float* rstd;   /* outputs */
} brBnFold;

/* ---- T1-T4: MLP tower layer, fp32 MFMA (v_mfma_f32_16x16x4_f32) ----------------------------
 * Dense/BatchNormalization/Dropout: NFC_plain.py:138-147, NeuMFModel.py:67-78, twoTower.py:40-41.
 * y = act( T(x)·W + bias ),  T(x)[r,k] = (x[r,k]*in_scale[k] + in_shift[k]) * keep(r,k)/(1-p)
 *   in_scale/in_shift (K) NULL => identity (they carry the previous layer's BatchNorm); in_bn != NULL (then in_scale and
 *   in_shift must be NULL): the BatchNorm is finalized from its column sums in this launch (brBnFold);
 *   drop_p == 0 (keep NULL) => no dropout; else keep = the site's bit plane (brDropoutKeepBits).
 * stats (double[BR_STAT_REPLICAS][2N], may be NULL): += column sums of y and y^2 (BatchNorm batch
 *   statistics), spread over 8 replicas (workgroup % 8) so same-address atomics do not serialise;
 *   consumers (brBnFinalize, brDenseBackward, brBnParamGrads) add the replicas.  The same layout
 *   holds for every BN-sum buffer below (bn_sums, in_bn_sums).
 * x: (B x K) row stride ldx; W: (K x N) row-major; y: (B x N) row stride ldy. */
int brDenseForward(const float* x, int64_t ldx, const float* W, const float* bias, float* y,
                   int64_t ldy, int64_t batch, int K, int N, int act, const float* in_scale,
                   const float* in_shift, const brBnFold* in_bn, float drop_p, const uint32_t* keep,
                   double* stats, brStream stream);
/* BatchNorm bookkeeping from the column sums (tiny): mean, biased var, scale = gamma*rstd,
 * shift = beta - mean*scale, moving stats <- momentum*moving + (1-momentum)*batch [TF-sem].
 * bstats (mean[N], rstd[N]) kept for the backward. */
int brBnFinalize(const double* stats, double batch_total, const float* gamma, const float* beta,
                 float eps, float momentum, float* moving_mean, float* moving_var, float* scale,
                 float* shift, float* mean, float* rstd, int N, brStream stream);
/* Inference-mode affine from the moving statistics. */
int brBnInference(const float* gamma, const float* beta, const float* moving_mean,
                  const float* moving_var, float eps, float* scale, float* shift, int N,
                  brStream stream);
/* Backward of one tower layer.
 *  gy: (B x N) gradient w.r.t. this layer's BatchNorm output h = BN(y) (dropout already
 *      transposed by the consumer's backward), or w.r.t. y itself when no BN follows (out_* NULL).
 *  out BN (after this layer): out_mean/out_rstd/out_gamma (N) and bn_sums (double[2N]:
 *      sum_r gy, sum_r gy*xhat — accumulated by the consumer's backward, see in_bn_sums).
 *      da = gamma*rstd*(gy - mean(gy) - xhat*mean(gy*xhat)); dz = da*act'(y).
 *  in transform (before this layer) as in brDenseForward, plus in_mean/in_rstd (K) when the
 *      input carries a BN: then in_bn_sums (double[2K]) += (sum gx, sum gx*xhat_in).
 *  Outputs: gx (B x K) = (dz·W^T) * keep(r,k)/(1-p): gradient w.r.t. the producer's BN output
 *      (w.r.t. x itself when the input has no BN) — may be NULL for the first layer of a tower
 *      whose input needs no gradient;  dW_slabs: (n_slabs x (K*N + N)) per-workgroup partials
 *      of [dW | db], reduced by brReduceSlabs in a fixed order (bitwise reproducible; no float
 *      atomics).  n_slabs = brDenseBackwardSlabs(). */
int brDenseBackwardSlabs(int64_t batch, int K, int N);
/* One launch (dz, dx and dW together, dz never leaves the CU: csrc/dense_bwd.hip) when K % 4 == N % 4 == 0 and every row is
 * 16-B aligned; other shapes run a dx and a dW kernel with dz handed over through dz_ws: caller scratch of
 * brDenseBackwardWorkspaceFloats() floats (16-B aligned).  keep: the bit plane of the layer's input dropout (brDropoutKeepBits),
 * required exactly when in_drop_p > 0. */
int64_t brDenseBackwardWorkspaceFloats(int64_t batch, int K, int N);
int brDenseBackward(const float* gy, int64_t ldgy, const float* y, int64_t ldy, const float* x,
                    int64_t ldx, const float* W, int64_t batch, int K, int N, int act,
                    const float* out_mean, const float* out_rstd, const float* out_gamma,
                    const double* bn_sums, double batch_total, const float* in_scale,
                    const float* in_shift, const float* in_mean, const float* in_rstd,
                    float in_drop_p, const uint32_t* keep, float* gx, int64_t ldgx, float* dz_ws,
                    float* dW_slabs, int n_slabs, double* in_bn_sums, brStream stream);
/* dgamma = sum gy*xhat, dbeta = sum gy: the BN-backward column sums as fp32 parameter grads. */
int brBnParamGrads(const double* bn_sums, float* dgamma, float* dbeta, int N, brStream stream);
int brBnParamGradsPair(const double* sums_a, float* dgamma_a, float* dbeta_a, int Na, const double* sums_b, float* dgamma_b,
                       float* dbeta_b, int Nb, brStream stream);
int brReduceSlabs(const float* slabs, int n_slabs, int64_t slab_elems, float* out, brStream stream);

/* End of the dense backward in one launch (single-process path): fixed-order reduction of three slab regions into their
 * ranges of the flat gradient (region r: n_slabs[r] slabs of slab_elems[r] floats -> grad[grad_off[r] ...)), the two BatchNorm
 * blocks' dgamma/dbeta from their backward column sums (as brBnParamGrads), then Adam on theta/m/v (as brAdamFlat).
 * The regions and BatchNorm blocks must tile [0, n) exactly. */
/* theta == NULL (then m, v unused): gradients only */
int brDenseFinalize(const float* const* slabs, const int* n_slabs, const int64_t* slab_elems, const int64_t* grad_off,
                    const double* const* bn_sums, const int* bn_n, const int64_t* dgamma_off, const int64_t* dbeta_off,
                    float* theta, float* m, float* v, float* grad, int64_t n, double alpha_t, double beta1, double beta2,
                    double eps, brStream stream);

/* ---- fused tail of the training step: T(a2) -> Dense(n3) -> concat [dot | a3] -> Dense(1) -> sigmoid -> loss and the
 * backward of all of it, one launch (trainers/NFC_plain.py:143-155, src/models/NeuMFModel.py:75-93) ----
 * Same results as brDenseForward(layer 3) + brNeumfHead + brDenseBackward(layer 3) (fp32 sums in a different
 * order).  a2: (B x n2) raw output of layer 2; scale2/shift2/mean2/rstd2 from brBnFinalize, or all NULL with bn2 given
 * (BatchNorm 2 finalized inside this launch, brBnFold); n2 <= 64, n3 <= 16 and 16-B aligned rows run on MFMA (csrc/tail_mfma.hip); keep: the bit plane of
 * the dropout in front of layer 3 (brDropoutKeepBits; NULL with drop_p == 0).
 * Outputs: a3 (B x n3, may be NULL), logit/prob/ddot (B), gh2 (B x n2) = gradient w.r.t. BN2's output,
 * bn_sums (double[BR_STAT_REPLICAS][2*n2]) += (sum gh2, sum gh2*xhat2), sums as brNeumfHead, and one slab per
 * workgroup [dW3 (n2*n3) | db3 (n3) | dW4 (n3+1, concat order) | db4]: n_slabs = brNeumfTailSlabs(batch),
 * brNeumfTailSlabElems(n2, n3) floats each, reduced by ONE brReduceSlabs into the adjacent W3|b3|W4|b4 grads. */
int brNeumfTailSlabs(int64_t batch);
int64_t brNeumfTailSlabElems(int n2, int n3);
int brNeumfTailFused(const float* a2, int64_t lda2, const float* W3, const float* b3, const float* w4, const float* b4,
                     const float* dot, const float* labels, const float* scale2, const float* shift2, const float* mean2,
                     const float* rstd2, const brBnFold* bn2, float drop_p, const uint32_t* keep,
                     int64_t batch, int n2, int n3, int act, int mf_first, int loss, float inv_batch, float* a3,
                     float* logit, float* prob, double* sums, float* ddot, float* gh2, int64_t ldgh2, double* bn_sums,
                     float* slabs, int n_slabs, brStream stream);

/* ---- head: concat [GMF dot | MLP out] -> Dense(1) -> sigmoid -> loss, and its backward --------
 * NFC_plain.py:149-155 (mf_first=1, BCE) / NeuMFModel.py:80-91 (mf_first=0, MSE).
 * a3: (B x N3); dot: (B); w4: (N3+1) in concat order; b4: (1).  Outputs: logit, prob (B);
 * sums (double[BR_SUM_SLOTS][BR_METRIC_SUMS], slot = workgroup & 63) += [loss, sum (p-y)^2, sum |p-y|, #correct@0.5,
 * sum BCE(logit, y), TP, FP, FN at threshold 0.5] - the sums behind the compiled metrics of trainers/NFC_plain.py:155
 * (BinaryCrossentropy, mse, mae, FalseNegatives, FalsePositives, TrueNegatives = n - TP - FP - FN, TruePositives,
 * BinaryAccuracy); the caller adds the slots; when labels && training:
 * da3 (B x N3), ddot (B), head_slabs (n_slabs x (N3+2)) partials of [dW4 | db4].
 * inv_batch = 1/global batch. */
int brHeadSlabs(int64_t batch);
int brNeumfHead(const float* a3, int64_t lda3, const float* dot, const float* labels,
                const float* w4, const float* b4, int64_t batch, int N3, int mf_first, int loss,
                float inv_batch, float* logit, float* prob, double* sums, float* da3,
                int64_t ldda3, float* ddot, float* head_slabs, int n_slabs, brStream stream);

/* ---- L1 stand-alone: sigmoid + BCE-from-logits (twoTower.py:85-87 rdZero path) ---------------
 * sums[0] += sum_b max(z,0) - z*y + log1p(exp(-|z|)); dz = (sigmoid(z)-y)*inv_batch. */
int brBceLogits(const float* z, const float* y, int64_t batch, float inv_batch, float* prob,
                float* dz, double* sums, brStream stream);

/* ---- L4: in-batch softmax (tfrs.tasks.Retrieval, twoTower.py:47,82-83) [TF-sem] --------------
 * S = Q C^T (Bq x Bc), accidental hits (cand_ids[j]==q_pos_ids[i], j != diag) masked, loss =
 * SUM_i [logsumexp_j S_ij - S_i,diag(i)].  Streaming (the Bq x Bc matrix is never stored).
 * Pass 1 (brInBatchSoftmaxLse): row_lse (Bq), loss_sum (double[BR_SUM_SLOTS], slot = workgroup & 63) +=.
 * Pass 2 (brInBatchSoftmaxGrad): dQ = (P - I) C, dC = (P - I)^T Q (each optional).
 * diag_offset: column of C holding query i's positive = i + diag_offset (data-parallel ranks
 * all-gather C; rank r's queries sit at offset r*Bq).
 * ws / ws_bytes: optional scratch (brInBatchSoftmaxWorkspaceBytes; 16-byte aligned).  A workgroup owns 128 rows, so at the
 * reference's batch sizes the row blocks alone do not fill 256 CUs: with a workspace the other axis is split over more
 * workgroups and the partial results (per-row max / sum / diagonal score, dQ / dC slabs) are combined by a second launch in a
 * fixed order (bit-reproducible).  ws = NULL: one workgroup per 128 rows, results written directly. */
int64_t brInBatchSoftmaxWorkspaceBytes(int64_t Bq, int64_t Bc, int dim);
int brInBatchSoftmaxLse(const float* Q, const float* C, const void* q_pos_ids,
                        const void* cand_ids, int id_type, int64_t Bq, int64_t Bc, int dim,
                        int64_t diag_offset, float* row_lse, double* loss_sum, void* ws,
                        int64_t ws_bytes, brStream stream);
/* Pass 1 and the dQ half of pass 2 in ONE sweep (what a training step needs: twoTower.py:99-102): the dQ accumulator is kept relative
 * to the running row maximum (online softmax) and normalised at the end, so the score tiles are formed twice per step instead of
 * three times.  Same outputs as brInBatchSoftmaxLse + brInBatchSoftmaxGrad(dQ) to rounding. */
int brInBatchSoftmaxLseGradQ(const float* Q, const float* C, const void* q_pos_ids,
                             const void* cand_ids, int id_type, int64_t Bq, int64_t Bc, int dim,
                             int64_t diag_offset, float* row_lse, double* loss_sum, float* dQ,
                             void* ws, int64_t ws_bytes, brStream stream);
int brInBatchSoftmaxGrad(const float* Q, const float* C, const void* q_pos_ids,
                         const void* cand_ids, int id_type, int64_t Bq, int64_t Bc, int dim,
                         int64_t diag_offset, const float* row_lse, float* dQ, float* dC,
                         void* ws, int64_t ws_bytes, brStream stream);

/* scores[q][c] = Q[q]·C[c] (n_q x n_c, row stride ld_scores): candidate scoring for top-k
 * (TwoTower BruteForce, twoTower.py:64-69,229-230; bpr_predict, src/models/bpr.py:122-133). */
int brScoreMatrix(const float* Q, const float* C, int64_t n_q, int64_t n_c, int dim, float* scores,
                  int64_t ld_scores, brStream stream);

/* ---- E1: full-catalogue scoring + stable top-k — topKmetrics.py:17-72, twoTower.py:64-69 -----
 * scores (U x I) row-major -> top-k per user, descending, ties keep the LOWER item position
 * (strict '>' in __topk, topKmetrics.py:59,68).  out_scores/out_index: (U x k). */
int brTopKRows(const float* scores, int64_t n_users, int64_t n_items, int k, float* out_scores,
               int32_t* out_index, brStream stream);

/* ---- evaluation of the BPR notebook model and hit counting -------------------
 * Ground truth per user = CSR list of COLUMN indices into the scored item list, ascending: truth_off (n_users + 1), truth_idx.
 * brFullAuc: full_auc (src/models/bpr.py:230-254) = per user sklearn.roc_auc_score(ground truth, scores over all items): the
 *   Mann-Whitney statistic with ties counted one half.  auc[u] = NaN for a user without positives (skipped by the reference,
 *   bpr.py:251) or without negatives; the caller averages the others.
 * brMapAtK: mean_average_precision_k (bpr.py:257-289) on the top-k lists of brTopKRows: ap[u] = sum over hits of
 *   (hits so far / rank) / min(len(actual), k) (0 for a user without positives); hits[u] = hits of the list = the per-user true
 *   positives of topKMetrics (trainers/topKmetrics.py:85-93).  ap or hits may be NULL. */
int brFullAuc(const float* scores, int64_t ld_scores, const int64_t* truth_off, const int32_t* truth_idx, int64_t n_users,
              int64_t n_items, float* auc, brStream stream);
int brMapAtK(const int32_t* topk_index, int64_t n_users, int k, const int64_t* truth_off, const int32_t* truth_idx, float* ap,
             int32_t* hits, brStream stream);

/* ---- batch construction on the device: csrc/sampling.hip ---------------------
 * Pure functions of (seed, output position): Philox4x32-10 draws + 4-round Feistel permutations with cycle walking, restated
 * bit for bit in oracle/binrec_oracle.py.  [pandas-sem] the reference's samplers are unseeded pandas calls: the distributions
 * are restated, not the streams.
 * brBootstrapDataset: NeuMFModel.bootstrapDataset (src/models/NeuMFModel.py:102-109): n positives (label 1) + n_neg rows sampled
 *   with replacement whose item column is permuted (label 0, no collision check), all shuffled.  Outputs have n + n_neg entries.
 * brBprSampleTriplets: for every positive (user, item) row neg_per_pos triplets (user, item, negative): the negative is drawn
 *   uniformly from cand_items (n_cand ids; NULL = 0..n_cand-1) and re-drawn while it is a positive of the user (at most max_tries
 *   draws) - the sampled replacement of the O(U*I) enumeration of src/models/BPRModel.py:111-119.  pos_off (num_users + 1) /
 *   pos_items: the users' positives as CSR with ascending items.
 * brNcfNegativeCandidates + brSortUniqueKeys64 + brGatherPermutedPairs: generateNegativeFeedback (Data handling/synthetic.py:
 *   208-223,237-256): candidates = customer column and product column shuffled independently, round after round (candidate j:
 *   round j / n); keys[j] = user * num_items + item, or ~0 for a positive pair.  Sorted and de-duplicated they are the pool of
 *   distinct negatives (n_unique counts a trailing ~0 if any candidate was invalid); brGatherPermutedPairs takes n_out of the
 *   first n_keys of them in a shuffled order (the reference's head(size)). */
int brBootstrapDataset(const void* users, const void* items, int id_type, int64_t n, int64_t n_neg, uint64_t seed, void* out_users,
                       void* out_items, float* out_labels, brStream stream);
int brBprSampleTriplets(const void* users, const void* items, int id_type, int64_t n, int neg_per_pos, const int64_t* pos_off,
                        const void* pos_items, const void* cand_items, int64_t n_cand, uint64_t seed, int max_tries, void* out_users,
                        void* out_pos, void* out_neg, brStream stream);
int brNcfNegativeCandidates(const void* users, const void* items, int id_type, int64_t n, int64_t n_cand, const int64_t* pos_off,
                            const void* pos_items, int64_t num_items, uint64_t seed, uint64_t* keys, brStream stream);
int64_t brSortUniqueWorkspaceBytes(int64_t n);
int brSortUniqueKeys64(const uint64_t* keys, int64_t n, uint64_t* out_keys, int64_t* n_unique, void* workspace, int64_t workspace_bytes,
                       brStream stream);
int brGatherPermutedPairs(const uint64_t* keys, int64_t n_keys, int64_t n_out, int64_t num_items, uint64_t seed, int id_type, void* out_users,
                          void* out_items, brStream stream);

/* ---- fused step driver: the whole NeuMF training / inference step from ONE host call ----------
 * trainers/NFC_plain.py:165 `model.fit` step, src/models/RModel.py:130.  Pure launch sequencing
 * over the entry points above (no new arithmetic): removes ~35 Python->C transitions per step so
 * the host stays ahead of the GPU.  The caller fills the struct once, then per step updates
 * users/items/labels/batch/step/alpha_t.  `phases` selects segments so a data-parallel host can
 * interleave its collectives (BatchNorm sums, dense-gradient all-reduce, row exchange):
 *   FWD1 zero scratch, [EMBED] embed forward, dense L1 (+ BN1 sums)
 *   FWD2 BN1 finalize, dense L2 (+ BN2 sums)
 *   FWD3 BN2 finalize, dense L3, head (+ loss/metric sums, head grads), backward L3 (+ BN2-bwd sums)
 *   BWD2 backward L2 (+ BN1-bwd sums)        BWD1 backward L1
 *   BNG  gamma/beta grads from the BN-backward sums
 *   OPT_TABLES [EMBED] embed backward   INDEX the dedup indexes   ROWS_USER / SWEEP_USER / ROWS_ITEM / SWEEP_ITEM the table
 *   optimizer launches (separate bits so a host can keep one of them outside a captured hipGraph)
 *   OPT_DENSE Adam on theta
 * EMBED: include the table-side embed forward/backward (single GPU); cleared by the row-sharded
 * host, which runs its own exchange and calls brNeumfEmbedForward/Backward on the received rows.
 * Dense parameter layout (theta/grad/adam_m/adam_v, floats):
 *   [W1 2*dim x n1 | b1 n1 | g1 n1 | be1 n1 | W2 n1 x n2 | b2 n2 | g2 n2 | be2 n2 | W3 n2 x n3 | b3 n3 | W4 n3+1 | b4 1] */
enum {
  BR_PH_FWD1 = 1, BR_PH_FWD2 = 2, BR_PH_FWD3 = 4, BR_PH_BWD2 = 8, BR_PH_BWD1 = 16, BR_PH_BNG = 32,
  BR_PH_OPT_TABLES = 64,      /* [EMBED] embed backward */
  BR_PH_OPT_DENSE = 128, BR_PH_EMBED = 256,
  BR_PH_ROWS_USER = 512, BR_PH_SWEEP_USER = 1024, BR_PH_ROWS_ITEM = 2048, BR_PH_SWEEP_ITEM = 4096,
  BR_PH_INDEX = 8192,         /* build the two dedup indexes in this call: on aux_stream beside the forward when the
                                 call also holds FWD1 (joined before ROWS_* or at the end of the call), else inline */
  BR_PH_ALL = 16383
};
typedef struct brNeumfStep {
  int64_t batch, batch_total, row0, user_rows, item_rows;
  int32_t dim, n1, n2, n3, act, loss, item_first, mf_first, id_type, adam_dense, training, step;
  int32_t bn_local;   /* 1: BatchNorm statistics over THIS process' batch (per-replica BN, what MirroredStrategy does with a
                         plain BatchNormalization [TF-sem]) while the loss is still scaled by 1/batch_total; 0: the column
                         sums are all-reduced by the host and cover batch_total rows */
  int32_t fused_final;   /* 1 (single process): the slab reductions, the BatchNorm parameter gradients and the dense Adam run
                            as ONE launch at BR_PH_OPT_DENSE (brDenseFinalize) instead of inside their phases; `slabs` then
                            holds three regions [tail | layer 2 | layer 1] (brNeumfStepSlabFloats).
                            2 (data-parallel host): the reductions and the BatchNorm parameter gradients as ONE launch at BR_PH_BNG
                            (brDenseFinalize without parameters); the host all-reduces `grad` and runs BR_PH_OPT_DENSE (brAdamFlat) */
  float dropout, bn_eps, bn_momentum, pad0;
  uint64_t seed;
  double alpha_t, beta1, beta2, adam_eps;
  const void* users;
  const void* items;
  const float* labels;                     /* may be NULL in inference */
  float* user_tab; float* user_m; float* user_v;   /* fused [rows][mlp|mf] (row = 2*dim floats) */
  float* item_tab; float* item_m; float* item_v;
  uint8_t* user_mark; uint8_t* item_mark;          /* [rows], adam_dense only */
  float* theta; float* grad; float* adam_m; float* adam_v;
  float* moving;                           /* [mm1 n1 | mv1 n1 | mm2 n2 | mv2 n2] */
  float* x0; float* dot; float* a1; float* a2; float* a3; float* logit; float* prob;    /* a1 / gh1 and a2 / gh2: row stride n rounded up to 4 floats */
  float* da3; float* ddot; float* gh2; float* gh1; float* dx0; float* g_user; float* g_item;
  float* bn;                               /* [scale1|shift1|mean1|rstd1] n1 each, then the same for layer 2 */
  double* dstat;                           /* [stats1 | stats2 | bsum1 | bsum2], each [BR_STAT_REPLICAS][2n]; zeroed in FWD1 */
  double* msums;                           /* [BR_SUM_SLOTS][BR_METRIC_SUMS] accumulated (brNeumfHead) */
  float* slabs; float* hslabs;
  float* dz_ws;                            /* brDenseBackwardWorkspaceFloats(max layer) floats */
  uint32_t* keep_bits;                     /* the three dropout planes of a step back to back: brDropoutKeepWords(batch, 2*dim) +
                                              (batch, n1) + (batch, n2) words (training with dropout > 0) */
  int* err_flag;
  float* u_seg_ws; float* i_seg_ws;        /* brSegmentScratchFloats(max batch, 2*dim) each, or NULL (one-by-one duplicate sums) */
  void* u_sorted_ids; int32_t* u_sorted_pos; void* u_ws; int64_t u_ws_bytes;
  void* i_sorted_ids; int32_t* i_sorted_pos; void* i_ws; int64_t i_ws_bytes;
  double lr;          /* learning rate (only used with step_state) */
  void* step_state;   /* optional device step state (brStepStateBytes()): when non-NULL (and the call holds
                         BR_PH_FWD1|BR_PH_EMBED; a host that runs the lookup itself advances it itself, before the lookup) the step advances it in
                         FWD1 (step += 1, alpha_t = lr*sqrt(1-b2^step)/(1-b1^step)) and every kernel reads the
                         dropout step / Adam alpha from there instead of `step` / `alpha_t` above, so the whole
                         call can be captured once in a hipGraph and replayed */
  int32_t* user_last; int32_t* item_last;   /* [rows], adam_dense == 2 only (deferred dense Adam, see brAdamFlush) */
  void* aux_stream;   /* optional second hipStream_t: the two dedup sorts depend only on the ids, so a call that
                         holds BR_PH_INDEX together with BR_PH_FWD1 runs them here beside the forward/backward and
                         joins them (event) before the Adam-rows kernels; NULL = same stream */
  int64_t keep_rows;  /* rows the three planes of keep_bits are laid out for (plane offsets = brDropoutKeepWords(keep_rows, K));
                         0 = the batch of the call */
  int32_t keep_ready; /* 1: keep_bits already holds THIS step's masks (a previous call prefetched them): FWD1 does not generate them */
  int32_t keep_prefetch;  /* 1 or 2 (training, dropout, single process): the call that holds BR_PH_ROWS_USER also fills keep_bits with the
                         NEXT step's masks (step + 1, same seed / row0, keep_rows rows) beside the Adam-rows kernel (Philox is ALU-bound,
                         the optimizer HBM-bound): 1 = as a launch on aux_stream (needed), joined before the call returns; 2 = as extra
                         workgroups of the paired Adam-rows launch when the call holds BR_PH_ROWS_ITEM too (no fork / join - 4 us less
                         per step inside a hipGraph - but that launch then takes 101 instead of 94 us).  The host sets keep_ready on
                         the next call if that call's step / row0 are the ones prefetched */
} brNeumfStep;
/* step_state: device {uint32 step; float alpha_t; double b1^step, b2^step; float alpha_hist[BR_ALPHA_RING + BR_RING_MIRROR]; replay form},
 * brStepStateBytes() bytes, zero-initialised (or step / alpha_t set by the host after a reload).
 * Advance: step += 1, alpha_t = alpha_hist[step % BR_ALPHA_RING] = lr*sqrt(1-b2^step)/(1-b1^step) (b^step kept as a
 * running double product), and `zero[0..n_zero)` (the step's double scratch) cleared in the same launch. */
enum { BR_ALPHA_RING = 1024, BR_RING_MIRROR = 8 };
int64_t brStepStateBytes(void);
/* How the deferred kernels replay a row's pending g = 0 steps (see "Deferred dense Adam" below).  Written into the step state once,
 * before the first step (host -> device copy, synchronises the stream); a zero-initialised state replays exactly.
 *   BR_REPLAY_EXACT: step by step with the sweep's own fp32 operations (m *= b1; v *= b2; theta -= alpha_j m / (sqrt(v) + eps)):
 *                    tables bit-equal to brAdamDenseSweep.
 *   BR_REPLAY_FAST : the same recurrence in a cheaper form - d_j = sqrt(v_j) + eps advanced as d_j = sqrt(b2) d_{j-1} + eps (1 - sqrt(b2))
 *                    (one v_sqrt per row visit, none per step), 1 / d_j by one Newton step from 1 / d_{j-1} (relative error
 *                    (1 - sqrt(b2))^2 = 2.5e-7: what the hardware v_rcp_f32 of the exact form is allowed), theta replayed over at most
 *                    `trunc` steps of a lag (the steps behind that move it by less than one ulp of the first: m has decayed by b1^trunc),
 *                    and the moments of a lag of L steps as ONE product each: m b1^L, v b2^L.  Agrees with the exact form to ~1e-6 of
 *                    a row's movement (tests/test_gpu_neumf.py); no longer bit-equal to the sweep. */
enum { BR_REPLAY_EXACT = 0, BR_REPLAY_FAST = 1 };
int brStepStateInit(void* step_state, double beta1, double beta2, double eps, int replay_mode, brStream stream);
int brStepStateAdvance(void* step_state, double lr, double beta1, double beta2, double* zero, int64_t n_zero, brStream stream);
/* host -> device: step, alpha_t and the running beta powers for that step (after a reload); synchronises the stream */
int brStepStateSet(void* step_state, uint32_t step, double lr, double beta1, double beta2, brStream stream);
/* one launch that copies a batch (ids, ids, labels) into the static input buffers a captured hipGraph reads */
int brStageBatch(void* dst_users, void* dst_items, float* dst_labels, const void* users, const void* items,
                 const float* labels, int id_type, int64_t n, brStream stream);

/* ---- Deferred dense Adam (brNeumfStep.adam_dense == 2) --------------------------------------------------
 * Keras' Adam applies the g = 0 update to every row of an embedding table on every step ([TF-sem], the
 * reference's trainers/NFC_plain.py:155 optimizer on Embedding variables): a 6-floats-per-element sweep of the
 * table per step.  The deferred form produces the SAME values without the sweep: last[row] = last step the
 * stored (theta, m, v) of that row include; the lookup replays the missing g = 0 steps in registers
 * (brNeumfEmbedForwardDeferred, nothing written), the optimizer replays them and applies this step's
 * gradient for the rows of the batch (brAdamRowsSortedDeferred, last[row] = step), and brAdamFlush brings the
 * whole table to the current step — required before anything else reads the table, and at least every
 * BR_ALPHA_RING - 1 steps (the replay takes each step's alpha_t from the ring in step_state).
 * Same fp32 operations in the same order as brAdamDenseSweep => bit-equal tables after a flush. */
int brNeumfEmbedForwardDeferred(const float* user_tab, const float* user_m, const float* user_v, const int32_t* user_last,
                                const float* item_tab, const float* item_m, const float* item_v, const int32_t* item_last,
                                int64_t user_rows, int64_t item_rows, const void* users, const void* items, int id_type,
                                int dim, int64_t batch, int item_first, const void* step_state, double beta1, double beta2,
                                double eps, float* x0, float* dot, float* stash_user, float* stash_item, int64_t ld_stash,
                                int* err_flag, brStream stream);
/* G1 on one deferred table: out[b] = row ids[b] as of step-1 (replayed in registers, nothing written back). */
int brGatherRowsDeferred(const float* table, const float* m, const float* v, const int32_t* last, int64_t table_rows, int dim,
                         const void* ids, int id_type, int64_t n, const void* step_state, double beta1, double beta2,
                         double eps, float* out, int64_t ld_out, int* err_flag, brStream stream);
/* the same for two tables of one geometry (dim, ld_out) in ONE launch: a row-sharded owner serves its user and its item shard
 * (n_b != 0: ids_b has n_b entries instead of n) */
int brGatherRowsDeferredPair(const float* table_a, const float* m_a, const float* v_a, const int32_t* last_a, int64_t rows_a, const void* ids_a,
                             float* out_a, const float* table_b, const float* m_b, const float* v_b, const int32_t* last_b, int64_t rows_b,
                             const void* ids_b, float* out_b, int dim, int id_type, int64_t n, int64_t n_b, const void* step_state, double beta1,
                             double beta2, double eps, int64_t ld_out, int* err_flag, brStream stream);
/* brGatherRowsDeferredPair AND both dedup indexes of the step (brRowIndexBuild of ids_a over n_a, of ids_b over n_b) in two launches on
 * one stream: the chunk sorts of the two id streams run as workgroups of the gather's own launch, the chunk-rank launch follows.
 * advance != 0: the step state is advanced (brStepStateAdvance with lr) by that second launch - the gather then computes the step as
 * step + 1 - so a BPR step (src/models/BPRModel.py:49-74: user gather of B rows, [pos | neg] gather of 2 B rows of the shared item
 * table) needs no launch of its own for it and no side stream for its sorts.  Rows of 64 / 128 / 256 floats, n <= 524 288 per stream. */
int brGatherRowsDeferredPairWithIndex(const float* table_a, const float* m_a, const float* v_a, const int32_t* last_a, int64_t rows_a, const void* ids_a,
                                      float* out_a, void* sorted_ids_a, int32_t* sorted_pos_a, void* ws_a, int64_t ws_a_bytes, const float* table_b,
                                      const float* m_b, const float* v_b, const int32_t* last_b, int64_t rows_b, const void* ids_b, float* out_b,
                                      void* sorted_ids_b, int32_t* sorted_pos_b, void* ws_b, int64_t ws_b_bytes, int dim, int id_type, int64_t n_a, int64_t n_b,
                                      void* step_state, int advance, double lr, double beta1, double beta2, double eps, int64_t ld_out, int* err_flag,
                                      brStream stream);
/* B1 of the GMF dot on the MF rows the deferred forward stashed, in place: (u_b, i_b) -> (ddot_b i_b, ddot_b u_b). */
int brMfGradInplace(float* stash_user, float* stash_item, int64_t ld, const float* ddot, int64_t batch, int dim, brStream stream);
int brAdamRowsSortedDeferred(float* table, float* m, float* v, int32_t* last, int64_t table_rows, int dim,
                             const void* sorted_ids, int id_type, const int32_t* sorted_pos, int64_t n,
                             const float* row_grads, int64_t ldg, const float* row_grads_hi, int64_t ldg_hi, int split,
                             const void* step_state, double beta1, double beta2, double eps, float* seg_ws, brStream stream);
/* Same, handed the rows as the step's lookup replayed them: replayed_rows + pos * ld_replayed = row sorted_ids[.] of position pos as
 * brGatherRowsDeferred wrote it this step (bpr.py: the gathered rows; a row-sharded owner: the rows it served).  The optimizer then
 * replays only the moments (two multiplies per lagging step instead of the lookup's sqrt / rcp chain on theta): same bits, since the
 * lookup ran the same replay on the same stored row.  Replaces Keras' sparse apply of an Embedding variable (NFC_plain.py:155,
 * BPRModel.py:52) on the touched rows. */
int brAdamRowsSortedDeferredReplayed(float* table, float* m, float* v, int32_t* last, int64_t table_rows, int dim,
                                     const void* sorted_ids, int id_type, const int32_t* sorted_pos, int64_t n,
                                     const float* row_grads, int64_t ldg, const float* row_grads_hi, int64_t ldg_hi, int split,
                                     const float* replayed_rows, int64_t ld_replayed, const void* step_state, double beta1,
                                     double beta2, double eps, float* seg_ws, brStream stream);
/* The user and the item table of one NeuMF step in ONE launch (same dim, n, split; each alone leaves HBM half idle).
 * last_* non-NULL (both): deferred mode (step_state required, alpha_t ignored); else brAdamRowsSorted semantics with marks.
 * hi_scale (n floats, may be NULL): the grads_hi rows of BOTH tables are multiplied by hi_scale[position] as they are read.  The
 * deferred step passes the stashed partner MF rows as grads_hi (user table: the item rows, item table: the user rows) and
 * ddot here: g_user_mf[b] = ddot[b]*item_mf[i_b] (brNeumfEmbedBackward) without the launch that used to write it out. */
int brAdamRowsSortedPair(float* table_a, float* m_a, float* v_a, int64_t rows_a, const void* sorted_ids_a, const int32_t* sorted_pos_a,
                         const float* grads_a, int64_t ldg_a, const float* grads_hi_a, int64_t ldg_hi_a, uint8_t* mark_a, int32_t* last_a,
                         float* table_b, float* m_b, float* v_b, int64_t rows_b, const void* sorted_ids_b, const int32_t* sorted_pos_b,
                         const float* grads_b, int64_t ldg_b, const float* grads_hi_b, int64_t ldg_hi_b, uint8_t* mark_b, int32_t* last_b,
                         int dim, int id_type, int64_t n, int split, const float* hi_scale, const void* step_state, double alpha_t,
                         double beta1, double beta2, double eps, float* seg_ws_a, float* seg_ws_b, brStream stream);
/* brAdamRowsSortedPair on deferred tables, handed the rows as the step's gathers replayed them (see brAdamRowsSortedDeferredReplayed):
 * replayed_x + pos * ld_replayed = the full row (dim floats) of position pos of table x.  A row-sharded owner updates its user and its
 * item shard with this one launch (both streams have the same number of slots under the fixed-capacity exchange).  n_b != 0: table b has
 * n_b positions instead of n (bpr.py: B user positions, 2B [pos | neg] item positions); grads_hi_* both NULL: one gradient source per table. */
int brAdamRowsSortedPairReplayed(float* table_a, float* m_a, float* v_a, int64_t rows_a, const void* sorted_ids_a, const int32_t* sorted_pos_a,
                                 const float* grads_a, int64_t ldg_a, const float* grads_hi_a, int64_t ldg_hi_a, int32_t* last_a,
                                 const float* replayed_a,
                                 float* table_b, float* m_b, float* v_b, int64_t rows_b, const void* sorted_ids_b, const int32_t* sorted_pos_b,
                                 const float* grads_b, int64_t ldg_b, const float* grads_hi_b, int64_t ldg_hi_b, int32_t* last_b,
                                 const float* replayed_b, int64_t ld_replayed,
                                 int dim, int id_type, int64_t n, int64_t n_b, int split, const void* step_state,
                                 double beta1, double beta2, double eps, float* seg_ws_a, float* seg_ws_b, brStream stream);
int brAdamFlush(float* table, float* m, float* v, int32_t* last, int64_t table_rows, int dim, const void* step_state,
                double beta1, double beta2, double eps, brStream stream);
int64_t brNeumfStepSizeof(void);
/* floats of brNeumfStep.slabs for batches up to `batch`: the three slab regions [tail | layer 2 | layer 1] back to back */
int64_t brNeumfStepSlabFloats(int64_t batch, int dim, int n1, int n2, int n3);
int brNeumfStepRun(const brNeumfStep* s, uint32_t phases, brStream stream);

/* Optional launch probe for measurement (bench.py roofline leg): HIP events on the launch stream
 * around every inner launch of brNeumfStepRun, tagged BR_TAG_*.  brProbeEnable(capacity) creates
 * the events (capacity 0 disables and frees them); after synchronising the stream the host reads
 * record i with brProbeRead (elapsed ms of that launch).  Process-global, not thread-safe. */
enum {
  BR_TAG_EMBED_FWD = 1, BR_TAG_FWD_L1 = 2, BR_TAG_FWD_L2 = 3, BR_TAG_FWD_L3 = 4, BR_TAG_HEAD = 5,
  BR_TAG_BWD_L3 = 6, BR_TAG_BWD_L2 = 7, BR_TAG_BWD_L1 = 8, BR_TAG_EMBED_BWD = 9,
  BR_TAG_INDEX_USER = 10, BR_TAG_INDEX_ITEM = 11, BR_TAG_ADAM_ROWS_USER = 12, BR_TAG_SWEEP_USER = 13,
  BR_TAG_ADAM_ROWS_ITEM = 14, BR_TAG_SWEEP_ITEM = 15, BR_TAG_ADAM_FLAT = 16, BR_TAG_REDUCE = 17,
  BR_TAG_SMALL = 18, BR_TAG_KEEP_BITS = 19, BR_TAG_STEP_STATE = 20,
  BR_TAG_SEG_PARTIALS = 21,   /* first kernel of an ADAM_ROWS_* call */
  BR_TAG_INDEX_SORT = 22      /* first kernel of an INDEX_* call (chunk sort); the rank / merge kernel keeps INDEX_* */
};
int brProbeEnable(int capacity);
int brProbeCount(void);
int brProbeRead(int i, int* tag, float* ms);
/* The same measurement INSIDE a replayed hipGraph, for one tag.  A timed event recorded into a stream capture cannot be read
 * (hipEventElapsedTime: invalid resource handle, ROCm 7.2), an event-record NODE can: while brNeumfStepRun is being captured
 * (hipStreamBeginCapture / torch.cuda.graph), the launch tagged `tag` gets such a node in front and behind
 * (hipStreamGetCaptureInfo_v2 + hipGraphAddEventRecordNode + hipStreamUpdateCaptureDependencies).
 *   brProbeGraphSelect(tag)        before the capture (tag >= 0 forgets earlier nodes; tag < 0: later captures carry none)
 *   brProbeGraphNodes()            after it: record nodes placed (0: that launch was not in the capture)
 *   brProbeGraphEnable(capacity)   event pairs for `capacity` replays (0 frees them)
 *   brProbeGraphArm(exec, slot)    before a replay of the instantiated graph `exec` (hipGraphExec_t): its nodes record into pair `slot`
 *   brProbeGraphRead(slot, &ms)    after synchronising: elapsed ms of the launch in that replay
 * tools/diag/graph_events.cpp is the stand-alone check of the mechanism.  Process-global, not thread-safe. */
int brProbeGraphSelect(int tag);
int brProbeGraphNodes(void);
int brProbeGraphEnable(int capacity);
int brProbeGraphArm(void* graph_exec, int slot);
int brProbeGraphRead(int slot, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* BINREC_H */
