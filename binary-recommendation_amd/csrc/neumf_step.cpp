// brNeumfStepRun: the NeuMF step as one host call (launch sequencing only; see include/binrec.h).
#include "common.h"
#include "dense.h"
#include "finalize.h"
#include "lookup_wave.h"

#include <vector>

// Optional launch probe (bench.py's roofline leg): HIP events recorded on the launch stream around
// each inner call of the step driver.  Events are created in brProbeEnable (never in the launch
// path); brProbeRead is called by the host after it synchronised the stream.
namespace {
struct Probe {
  std::vector<hipEvent_t> ev;   // 2 per record
  std::vector<int> tag;
  int n = 0, cap = 0;
  bool open = false;
};
Probe g_probe;
// One tag can also be bracketed inside a captured hipGraph: event-record NODES added to the graph under capture (see include/binrec.h
// brProbeGraph*).  The nodes record into a placeholder pair until brProbeGraphArm points them at a replay's own pair.
struct GraphProbe {
  int sel = -1;                         // the tag to bracket in captures
  hipEvent_t ph0 = nullptr, ph1 = nullptr;
  std::vector<hipGraphNode_t> begins;   // record nodes in front of the launch (a split call re-records: the last one counts)
  std::vector<hipGraphNode_t> ends;
  bool open = false;
  bool pending = false;                 // selected call under capture whose measured kernel is not its first launch (br::probe_mark)
  std::vector<hipEvent_t> ev;           // 2 per slot
};
GraphProbe g_gp;
// calls that launch a helper kernel of another tag first and tell where the measured launch starts (br::probe_mark)
inline bool tag_marks(int tag) { return tag == BR_TAG_ADAM_ROWS_USER || tag == BR_TAG_ADAM_ROWS_ITEM; }
inline bool capturing(hipStream_t s) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
}
// an event-record node behind everything the capturing stream has issued so far; later work of the stream depends on it
inline bool capture_record(hipStream_t s, hipEvent_t ev, std::vector<hipGraphNode_t>& out) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t g = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t nd = 0;
  if (hipStreamGetCaptureInfo_v2(s, &st, &id, &g, &deps, &nd) != hipSuccess || st != hipStreamCaptureStatusActive || !g) return false;
  hipGraphNode_t node = nullptr;
  if (hipGraphAddEventRecordNode(&node, g, deps, nd, ev) != hipSuccess) return false;
  if (hipStreamUpdateCaptureDependencies(s, &node, 1, hipStreamSetCaptureDependencies) != hipSuccess) return false;
  out.push_back(node);
  return true;
}
inline void probe_begin(int tag, hipStream_t s) {
  g_probe.open = false;
  g_gp.open = g_gp.pending = false;
  if (capturing(s)) {
    // timed events cannot be recorded into a stream capture (hipErrorInvalidHandle on ROCm 7.2): the selected tag gets record nodes
    if (tag == g_gp.sel && g_gp.ph0) {
      if (tag_marks(tag)) g_gp.pending = true;
      else g_gp.open = capture_record(s, g_gp.ph0, g_gp.begins);
    }
    return;
  }
  if (g_probe.n < g_probe.cap) {
    g_probe.tag[g_probe.n] = tag;
    (void)hipEventRecord(g_probe.ev[2 * g_probe.n], s);
    g_probe.open = true;
  }
}
inline void probe_end(hipStream_t s) {
  g_gp.pending = false;
  if (g_gp.open) {
    g_gp.open = false;
    (void)capture_record(s, g_gp.ph1, g_gp.ends);
  }
  if (g_probe.open) {
    g_probe.open = false;
    (void)hipEventRecord(g_probe.ev[2 * g_probe.n + 1], s);
    ++g_probe.n;
  }
}
}  // namespace

// the measured launch of a tag_marks() call comes next (every record node is a barrier in the replayed graph: only two of them)
void br::probe_mark(hipStream_t s) {
  if (g_gp.pending) {
    g_gp.pending = false;
    g_gp.open = capture_record(s, g_gp.ph0, g_gp.begins);
  }
}

extern "C" int brProbeGraphSelect(int tag) {
  g_gp.sel = tag; g_gp.open = g_gp.pending = false;
  if (tag >= 0) { g_gp.begins.clear(); g_gp.ends.clear(); }      // (< 0: later captures carry no nodes, the placed ones stay armable)
  if (tag >= 0 && !g_gp.ph0) {
    if (hipEventCreate(&g_gp.ph0) != hipSuccess || hipEventCreate(&g_gp.ph1) != hipSuccess) { br::set_error("brProbeGraphSelect: hipEventCreate failed"); return BR_ERR_HIP; }
  }
  return BR_OK;
}
extern "C" int brProbeGraphNodes(void) { return g_gp.ends.empty() ? 0 : (int)(g_gp.begins.size() + g_gp.ends.size()); }
extern "C" int brProbeGraphEnable(int capacity) {
  for (hipEvent_t e : g_gp.ev) (void)hipEventDestroy(e);
  g_gp.ev.clear();
  if (capacity <= 0) return BR_OK;
  g_gp.ev.resize(2 * (size_t)capacity);
  for (auto& e : g_gp.ev)
    if (hipEventCreate(&e) != hipSuccess) { br::set_error("brProbeGraphEnable: hipEventCreate failed"); return BR_ERR_HIP; }
  return BR_OK;
}
extern "C" int brProbeGraphArm(void* graph_exec, int slot) {
  BR_CHECK_ARG(graph_exec && slot >= 0 && 2 * (size_t)slot + 1 < g_gp.ev.size() && !g_gp.ends.empty(), "brProbeGraphArm: no record nodes / bad slot");
  hipGraphExec_t ex = (hipGraphExec_t)graph_exec;
  for (hipGraphNode_t nd : g_gp.begins)
    if (hipGraphExecEventRecordNodeSetEvent(ex, nd, g_gp.ev[2 * slot]) != hipSuccess) { br::set_error("brProbeGraphArm: hipGraphExecEventRecordNodeSetEvent failed"); return BR_ERR_HIP; }
  for (hipGraphNode_t nd : g_gp.ends)
    if (hipGraphExecEventRecordNodeSetEvent(ex, nd, g_gp.ev[2 * slot + 1]) != hipSuccess) { br::set_error("brProbeGraphArm: hipGraphExecEventRecordNodeSetEvent failed"); return BR_ERR_HIP; }
  return BR_OK;
}
extern "C" int brProbeGraphRead(int slot, float* ms) {
  BR_CHECK_ARG(ms && slot >= 0 && 2 * (size_t)slot + 1 < g_gp.ev.size(), "brProbeGraphRead: bad slot");
  if (hipEventElapsedTime(ms, g_gp.ev[2 * slot], g_gp.ev[2 * slot + 1]) != hipSuccess) {
    br::set_error("brProbeGraphRead: events not complete (synchronise the stream first)");
    return BR_ERR_HIP;
  }
  return BR_OK;
}

// a call that launches two kernels splits its record: what ran so far is retagged `first_tag`, the rest keeps the call's tag
void br::probe_split(int first_tag, hipStream_t s) {
  if (g_gp.open) {        // graph-resident probe of a two-kernel call's FIRST launch (the lookup of the fused lookup + rank call): it ends here
    g_gp.open = false;
    (void)capture_record(s, g_gp.ph1, g_gp.ends);
  }
  if (!g_probe.open) return;
  const int tag = g_probe.tag[g_probe.n];
  g_probe.tag[g_probe.n] = first_tag;
  probe_end(s);
  probe_begin(tag, s);
}

extern "C" int brProbeEnable(int capacity) {
  for (hipEvent_t e : g_probe.ev) (void)hipEventDestroy(e);
  g_probe.ev.clear(); g_probe.tag.clear(); g_probe.n = 0; g_probe.cap = 0;
  if (capacity <= 0) return BR_OK;
  g_probe.ev.resize(2 * (size_t)capacity);
  g_probe.tag.resize((size_t)capacity);
  for (auto& e : g_probe.ev)
    if (hipEventCreate(&e) != hipSuccess) { br::set_error("brProbeEnable: hipEventCreate failed"); return BR_ERR_HIP; }
  g_probe.cap = capacity;
  return BR_OK;
}
extern "C" int brProbeCount(void) { return g_probe.n; }
extern "C" int brProbeRead(int i, int* tag, float* ms) {
  BR_CHECK_ARG(i >= 0 && i < g_probe.n && tag && ms, "brProbeRead: bad index");
  *tag = g_probe.tag[i];
  if (hipEventElapsedTime(ms, g_probe.ev[2 * i], g_probe.ev[2 * i + 1]) != hipSuccess) {
    br::set_error("brProbeRead: events not complete (synchronise the stream first)");
    return BR_ERR_HIP;
  }
  return BR_OK;
}

#define RUN(tag, call)             \
  do {                             \
    probe_begin(tag, hs);          \
    int rc_ = (call);              \
    probe_end(hs);                 \
    if (rc_ != BR_OK) return rc_;  \
  } while (0)

// fork/join events of the optional aux stream (created once, on first use)
namespace {
hipEvent_t g_fork = nullptr, g_join = nullptr, g_kfork = nullptr, g_kjoin = nullptr;
bool ensure_events() {
  if (g_fork) return true;
  return hipEventCreateWithFlags(&g_fork, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&g_join, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&g_kfork, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&g_kjoin, hipEventDisableTiming) == hipSuccess;
}
}  // namespace

extern "C" int64_t brNeumfStepSizeof(void) { return (int64_t)sizeof(brNeumfStep); }

// slab regions of a step: [tail | layer 2 | layer 1], sized for the batch at hand (slab counts grow with the batch)
namespace {
struct SlabPlan { int ns_t, ns2, ns1; int64_t el_t, el2, el1, off_t, off2, off1, total; };
inline SlabPlan slab_plan(int64_t B, int D, int n1, int n2, int n3) {
  SlabPlan p;
  p.ns_t = brNeumfTailSlabs(B); p.el_t = brNeumfTailSlabElems(n2, n3);
  p.ns2 = brDenseBackwardSlabs(B, n1, n2); p.el2 = (int64_t)n1 * n2 + n2;
  p.ns1 = brDenseBackwardSlabs(B, 2 * D, n1); p.el1 = (int64_t)2 * D * n1 + n1;
  p.off_t = 0; p.off2 = p.off_t + p.ns_t * p.el_t; p.off1 = p.off2 + p.ns2 * p.el2;
  p.total = p.off1 + p.ns1 * p.el1;
  return p;
}
}  // namespace
extern "C" int64_t brNeumfStepSlabFloats(int64_t batch, int dim, int n1, int n2, int n3) {
  return slab_plan(batch > 0 ? batch : 1, dim, n1, n2, n3).total;
}


static int step_run_impl(const brNeumfStep* s, uint32_t ph, brStream stream);

extern "C" int brNeumfStepRun(const brNeumfStep* s, uint32_t ph, brStream stream) {
  BR_CHECK_ARG(s != nullptr, "brNeumfStepRun: null struct");
  br::set_current_step_state(reinterpret_cast<const br::StepStateDev*>(s->step_state));
  const int rc = step_run_impl(s, ph, stream);
  br::set_current_step_state(nullptr);
  return rc;
}

static int step_run_impl(const brNeumfStep* s, uint32_t ph, brStream stream) {
  const int64_t B = s->batch;
  if (B == 0) return BR_OK;
  hipStream_t hs = (hipStream_t)stream;
  const int D = s->dim, n1 = s->n1, n2 = s->n2, n3 = s->n3;
  BR_CHECK_ARG(B > 0 && D >= 1 && n1 >= 1 && n2 >= 1 && n3 >= 1 && n3 <= 32, "brNeumfStepRun: bad geometry");
  const bool train = s->training != 0;
  const bool deferred = s->adam_dense == 2;
  const bool fused_final = train && s->fused_final == 1;     // reductions / BN grads / dense Adam in one launch at OPT_DENSE
  const bool fused_grads = train && s->fused_final == 2;     // reductions / BN grads in one launch at BNG (the host all-reduces before its Adam)
  const SlabPlan sp = slab_plan(B, D, n1, n2, n3);
  float *slabs_t = s->slabs + sp.off_t, *slabs2 = s->slabs + sp.off2, *slabs1 = s->slabs + sp.off1;
  BR_CHECK_ARG(!deferred || (s->step_state && s->user_last && s->item_last), "brNeumfStepRun: deferred Adam needs step_state and the last[] arrays");
  const int l1 = (n1 + 3) & ~3, l2 = (n2 + 3) & ~3;      // row strides of a1 / gh1 and a2 / gh2 (padded to 16 B: include/binrec.h)
  const float p = train ? s->dropout : 0.f;
  const int64_t krows = s->keep_rows > 0 ? s->keep_rows : B;   // rows the keep-bit planes are laid out for
  const double bt = (double)(s->batch_total > 0 ? s->batch_total : B);
  const float inv_b = (float)(1.0 / bt);
  const double bt_bn = s->bn_local ? (double)B : bt;       // rows behind the BatchNorm sums
  // dense parameter layout
  float* th = s->theta;
  float* gr = s->grad;
  int64_t o = 0;
  const int64_t oW1 = o; o += (int64_t)2 * D * n1;
  const int64_t ob1 = o; o += n1;
  const int64_t og1 = o; o += n1;
  const int64_t obe1 = o; o += n1;
  const int64_t oW2 = o; o += (int64_t)n1 * n2;
  const int64_t ob2 = o; o += n2;
  const int64_t og2 = o; o += n2;
  const int64_t obe2 = o; o += n2;
  const int64_t oW3 = o; o += (int64_t)n2 * n3;
  const int64_t ob3 = o; o += n3;
  const int64_t oW4 = o; o += n3 + 1;
  const int64_t ob4 = o; o += 1;
  const int64_t n_dense = o;
  float* bn = s->bn;
  float *scale1 = bn, *shift1 = bn + n1, *mean1 = bn + 2 * n1, *rstd1 = bn + 3 * n1;
  float *scale2 = bn + 4 * n1, *shift2 = scale2 + n2, *mean2 = scale2 + 2 * n2, *rstd2 = scale2 + 3 * n2;
  constexpr int R = BR_STAT_REPLICAS;   // every BN-sum buffer is [R][2N]
  double *stats1 = s->dstat, *stats2 = stats1 + R * 2 * n1, *bsum1 = stats2 + R * 2 * n2, *bsum2 = bsum1 + R * 2 * n1;
  float *mm1 = s->moving, *mv1 = mm1 + n1, *mm2 = mv1 + n1, *mv2 = mm2 + n2;
  const int uoff = s->item_first ? D : 0, ioff = s->item_first ? 0 : D;
  // dedup sorts on the aux stream: forked after the embedding lookup, joined before the Adam-rows kernels.
  // (Forked at the very top, a hipGraph replay ran the first sort BEFORE the step's first main-stream node
  // instead of beside it - rocprofv3 timeline, ROCm 7.2 - which put ~70 us of launch-bound sort passes on the
  // critical path; behind the lookup they overlap the MLP.)
  const bool build_index = train && (ph & BR_PH_INDEX);
  // the chunk sorts inside the lookup's own launch (no fork / join): the single-GPU deferred step at the benchmarked shapes
  const bool fused_index = build_index && deferred && (ph & BR_PH_FWD1) && (ph & BR_PH_EMBED) &&
                           br::lookup_with_index_supported(D, B, s->user_rows, s->item_rows, 2 * D, s->x0, s->g_user + D, s->g_item + D);
  const bool aux_index = build_index && !fused_index && s->aux_stream && (ph & BR_PH_FWD1);
  bool joined = !aux_index;
  const int64_t n_dstat = (int64_t)BR_STAT_REPLICAS * (4 * n1 + 4 * n2);
  // (a host that runs the embedding exchange itself - no EMBED bit - advanced the state before its lookup)
  // fused_index and no keep-bit launch at the top of the step (planes prefetched, or no dropout): the chunk-rank launch advances the
  // state behind the lookup (which then computes its step as ss->step + 1) - one launch less in the chain
  const bool need_advance = (ph & BR_PH_FWD1) && (ph & BR_PH_EMBED) && train && s->step_state;
  const bool defer_advance = need_advance && fused_index && (s->dropout <= 0.f || s->keep_ready);
  if (need_advance && !defer_advance)
    RUN(BR_TAG_STEP_STATE, brStepStateAdvance(s->step_state, s->lr, s->beta1, s->beta2, s->dstat, n_dstat, stream));
  // dropout keep-bit planes of the three sites [2D | n1 | n2], filled once per step (after the step counter advanced)
  uint32_t *keep0 = nullptr, *keep1 = nullptr, *keep2 = nullptr;
  if (p > 0.f) {
    BR_CHECK_ARG(s->keep_bits != nullptr, "brNeumfStepRun: dropout needs brNeumfStep.keep_bits");
    BR_CHECK_ARG(s->keep_rows == 0 || s->keep_rows >= B, "brNeumfStepRun: keep_rows < batch");
    keep0 = s->keep_bits; keep1 = keep0 + brDropoutKeepWords(krows, 2 * D); keep2 = keep1 + brDropoutKeepWords(krows, n1);
  }
  // dedup sorts on the aux stream, forked right behind the step-counter launch: they depend only on the ids and then run beside the
  // embedding lookup (an HBM-latency-bound gather without LDS).  Beside the first dense layer they cost it ~20 us: its one-wave grid
  // had to wait for the 16 CUs the sort workgroups (64 KB of LDS each) were holding.
  // (capturing the lookup AHEAD of the aux launches was tried: a replayed graph starts a fork's nodes ~6 us apart in capture order, so
  //  the lookup started 7 us earlier - but the sort's 16 fat workgroups then queue behind the lookup's grid, finish beside the first
  //  dense layer and hold the 16 CUs its one-generation grid needs: 29 -> 49 us there.  Sorts first, lookup flowing around them.)
  if ((ph & BR_PH_FWD1) && aux_index) {
    if (!ensure_events()) { br::set_error("brNeumfStepRun: hipEventCreate failed"); return BR_ERR_HIP; }
    hipStream_t as = (hipStream_t)s->aux_stream;
    (void)hipEventRecord(g_fork, hs);                 // the previous step's readers of the index buffers are done
    (void)hipStreamWaitEvent(as, g_fork, 0);
    const int rc = brRowIndexBuildPair(s->users, s->user_rows, s->u_sorted_ids, s->u_sorted_pos, s->u_ws, s->u_ws_bytes,
                                       s->items, s->item_rows, s->i_sorted_ids, s->i_sorted_pos, s->i_ws, s->i_ws_bytes, s->id_type, B, s->aux_stream);
    if (rc != BR_OK) return rc;
    (void)hipEventRecord(g_join, as);
  }
  if ((ph & BR_PH_FWD1) && p > 0.f && !(train && s->keep_ready)) {
    const uint32_t sites[3] = {0, 1, 2};
    const int widths[3] = {2 * D, n1, n2};
    uint32_t* const outs[3] = {keep0, keep1, keep2};
    RUN(BR_TAG_KEEP_BITS, brDropoutKeepBits(p, s->seed, (uint32_t)s->step, s->row0, B, 3, sites, widths, outs, stream));
  }
  if (ph & BR_PH_FWD1) {
    if (train && !s->step_state) {
      hipError_t e = hipMemsetAsync(s->dstat, 0, sizeof(double) * (size_t)(BR_STAT_REPLICAS * (4 * n1 + 4 * n2)), hs);
      if (e != hipSuccess) { br::set_error("brNeumfStepRun: memset: %s", hipGetErrorString(e)); return BR_ERR_HIP; }
    }
    if (fused_index) {
      br::AdamHp lh = br::make_hp(0.0, s->beta1, s->beta2, s->adam_eps);
      const br::LookupArgs la{s->user_tab, s->user_m, s->user_v, s->user_last, s->item_tab, s->item_m, s->item_v, s->item_last, s->user_rows, s->item_rows,
                              s->users, s->items, B, s->item_first, reinterpret_cast<const br::StepStateDev*>(s->step_state), lh, s->x0, s->dot,
                              s->g_user + D, s->g_item + D, 2 * D, s->err_flag};
      const br::IndexPairArgs ix{s->u_sorted_ids, s->u_sorted_pos, s->u_ws, s->u_ws_bytes, s->i_sorted_ids, s->i_sorted_pos, s->i_ws, s->i_ws_bytes};
      br::LookupArgs la2 = la;
      la2.spec = br::lookup_spec(s->user_rows, s->item_rows, B);
      br::StepAdvance adv;
      if (defer_advance) {
        la2.step_add = 1;
        adv.st = reinterpret_cast<br::StepStateDev*>(s->step_state); adv.lr = s->lr; adv.b1 = s->beta1; adv.b2 = s->beta2; adv.zero = s->dstat; adv.n_zero = n_dstat;
      }
      // (its first launch is retagged EMBED_FWD by probe_split; under capture with EMBED_FWD selected the record nodes bracket that launch)
      const int ltag = (g_gp.sel == BR_TAG_EMBED_FWD && capturing(hs)) ? BR_TAG_EMBED_FWD : BR_TAG_INDEX_USER;
      RUN(ltag, br::lookup_with_index(la2, D, s->id_type, ix, stream, defer_advance ? &adv : nullptr));
    } else if ((ph & BR_PH_EMBED) && deferred && train)
      RUN(BR_TAG_EMBED_FWD, brNeumfEmbedForwardDeferred(s->user_tab, s->user_m, s->user_v, s->user_last, s->item_tab, s->item_m, s->item_v, s->item_last,
                              s->user_rows, s->item_rows, s->users, s->items, s->id_type, D, B, s->item_first, s->step_state, s->beta1, s->beta2,
                              s->adam_eps, s->x0, s->dot, s->g_user + D, s->g_item + D, 2 * D, s->err_flag, stream));
    else if (ph & BR_PH_EMBED)   /* deferred + inference: the host flushed the tables (brAdamFlush) first */
      RUN(BR_TAG_EMBED_FWD, brNeumfEmbedForward(s->user_tab, s->item_tab, s->user_tab + D, s->item_tab + D, 2 * D, 2 * D, s->user_rows, s->item_rows,
                              s->users, s->items, s->id_type, D, B, s->item_first, s->x0, s->dot, s->err_flag, stream));
    RUN(BR_TAG_FWD_L1, brDenseForward(s->x0, 2 * D, th + oW1, th + ob1, s->a1, l1, B, 2 * D, n1, s->act, nullptr, nullptr, nullptr, p, keep0,
                       train ? stats1 : nullptr, stream));
  }
  if (ph & BR_PH_FWD2) {
    // training: BatchNorm 1 is finalized inside the layer-2 launch (brBnFold), BatchNorm 2 inside the tail launch
    brBnFold f1{stats1, bt_bn, th + og1, th + obe1, s->bn_eps, s->bn_momentum, mm1, mv1, scale1, shift1, mean1, rstd1};
    if (!train)
      RUN(BR_TAG_SMALL, brBnInference(th + og1, th + obe1, mm1, mv1, s->bn_eps, scale1, shift1, n1, stream));
    RUN(BR_TAG_FWD_L2, brDenseForward(s->a1, l1, th + oW2, th + ob2, s->a2, l2, B, n1, n2, s->act, train ? nullptr : scale1, train ? nullptr : shift1,
                       train ? &f1 : nullptr, p, keep1, train ? stats2 : nullptr, stream));
  }
  if (ph & BR_PH_FWD3) {
    brBnFold f2{stats2, bt_bn, th + og2, th + obe2, s->bn_eps, s->bn_momentum, mm2, mv2, scale2, shift2, mean2, rstd2};
    if (!train)
      RUN(BR_TAG_SMALL, brBnInference(th + og2, th + obe2, mm2, mv2, s->bn_eps, scale2, shift2, n2, stream));
    if (!train)
      RUN(BR_TAG_FWD_L3, brDenseForward(s->a2, l2, th + oW3, th + ob3, s->a3, n3, B, n2, n3, s->act, scale2, shift2, nullptr, 0.f, nullptr, nullptr, stream));
    if (train) {
      // L3 forward, head, loss and their backward in one launch; W3|b3|W4|b4 are adjacent in theta / grad
      const int nst = sp.ns_t;
      RUN(BR_TAG_HEAD, brNeumfTailFused(s->a2, l2, th + oW3, th + ob3, th + oW4, th + ob4, s->dot, s->labels, nullptr, nullptr, nullptr, nullptr, &f2, p, keep2,
                      B, n2, n3, s->act, s->mf_first, s->loss, inv_b, s->a3, s->logit, s->prob, s->msums, s->ddot,
                      s->gh2, l2, bsum2, slabs_t, nst, stream));
      if (!fused_final && !fused_grads) RUN(BR_TAG_REDUCE, brReduceSlabs(slabs_t, nst, sp.el_t, gr + oW3, stream));
    } else {
      RUN(BR_TAG_HEAD, brNeumfHead(s->a3, n3, s->dot, s->labels, th + oW4, th + ob4, B, n3, s->mf_first, s->loss, inv_b, s->logit, s->prob,
                      s->labels ? s->msums : nullptr, nullptr, 0, nullptr, nullptr, 0, stream));
    }
  }
  if (!train) return BR_OK;
  if (ph & BR_PH_BWD2) {
    const int ns2 = sp.ns2;
    RUN(BR_TAG_BWD_L2, brDenseBackward(s->gh2, l2, s->a2, l2, s->a1, l1, th + oW2, B, n1, n2, s->act, mean2, rstd2, th + og2, bsum2, bt_bn, scale1, shift1,
                        mean1, rstd1, p, keep1, s->gh1, l1, s->dz_ws, slabs2, ns2, bsum1, stream));
    if (!fused_final && !fused_grads) RUN(BR_TAG_REDUCE, brReduceSlabs(slabs2, ns2, sp.el2, gr + oW2, stream));
  }
  if (ph & BR_PH_BWD1) {
    const int ns1 = sp.ns1;
    RUN(BR_TAG_BWD_L1, brDenseBackward(s->gh1, l1, s->a1, l1, s->x0, 2 * D, th + oW1, B, 2 * D, n1, s->act, mean1, rstd1, th + og1, bsum1, bt_bn, nullptr,
                        nullptr, nullptr, nullptr, p, keep0, s->dx0, 2 * D, s->dz_ws, slabs1, ns1, nullptr, stream));
    if (!fused_final && !fused_grads) RUN(BR_TAG_REDUCE, brReduceSlabs(slabs1, ns1, sp.el1, gr + oW1, stream));
  }
  if ((ph & BR_PH_BNG) && fused_grads) {
    const float* const rs[3] = {slabs1, slabs2, slabs_t};
    const int rn[3] = {sp.ns1, sp.ns2, sp.ns_t};
    const int64_t re[3] = {sp.el1, sp.el2, sp.el_t}, ro[3] = {oW1, oW2, oW3};
    const double* const bs[2] = {bsum1, bsum2};
    const int bn_n[2] = {n1, n2};
    const int64_t bg[2] = {og1, og2}, bb[2] = {obe1, obe2};
    RUN(BR_TAG_REDUCE, brDenseFinalize(rs, rn, re, ro, bs, bn_n, bg, bb, nullptr, nullptr, nullptr, gr, n_dense, 0.0, s->beta1, s->beta2, s->adam_eps, stream));
  } else if ((ph & BR_PH_BNG) && !fused_final) {
    RUN(BR_TAG_SMALL, brBnParamGradsPair(bsum2, gr + og2, gr + obe2, n2, bsum1, gr + og1, gr + obe1, n1, stream));
  }
  if (ph & BR_PH_OPT_TABLES) {
    // deferred: the forward stashed the MF rows; their gradients ddot[b] * partner row are formed by the optimizer launch as it
    // reads them (brAdamRowsSortedPair hi_scale) - unless the two tables are updated by separate calls
    const bool both = (ph & BR_PH_ROWS_USER) && (ph & BR_PH_ROWS_ITEM);
    if ((ph & BR_PH_EMBED) && deferred && !both)
      RUN(BR_TAG_EMBED_BWD, brMfGradInplace(s->g_user + D, s->g_item + D, 2 * D, s->ddot, B, D, stream));
    else if ((ph & BR_PH_EMBED) && !deferred)
      RUN(BR_TAG_EMBED_BWD, brNeumfEmbedBackward(s->user_tab + D, s->item_tab + D, 2 * D, 2 * D, s->user_rows, s->item_rows, s->users, s->items, s->id_type, D,
                               B, s->item_first, nullptr, s->ddot, nullptr, nullptr, s->g_user + D, s->g_item + D, 2 * D, 0, stream));
  }
  if (build_index && !aux_index && !fused_index) {
    RUN(BR_TAG_INDEX_USER, brRowIndexBuildPair(s->users, s->user_rows, s->u_sorted_ids, s->u_sorted_pos, s->u_ws, s->u_ws_bytes,
                                               s->items, s->item_rows, s->i_sorted_ids, s->i_sorted_pos, s->i_ws, s->i_ws_bytes, s->id_type, B, stream));
  }
  if (!joined && (ph & (BR_PH_ROWS_USER | BR_PH_ROWS_ITEM))) {
    (void)hipStreamWaitEvent(hs, g_join, 0);
    joined = true;
  }
  // next step's dropout planes beside the Adam-rows kernel (Philox is ALU-bound, the optimizer HBM-bound; the planes' last readers -
  // this step's backward - are behind us on the launch stream): inside the pair launch's own grid when there is one, else as a
  // launch on the aux stream (the dedup sorts there were joined above)
  const bool prefetch = train && p > 0.f && s->keep_prefetch && (ph & BR_PH_ROWS_USER);
  const bool prefetch_fused = prefetch && s->keep_prefetch == 2 && (ph & BR_PH_ROWS_ITEM);
  const bool prefetch_aux = prefetch && !prefetch_fused && s->aux_stream;
  br::KeepArgs next_keep;
  bool final_on_aux = false;
  if (prefetch_fused || prefetch_aux) {
    const uint32_t sites[3] = {0, 1, 2};
    const int widths[3] = {2 * D, n1, n2};
    uint32_t* const outs[3] = {keep0, keep1, keep2};
    if (prefetch_fused) {
      const int rc = br::make_keep_args(next_keep, p, s->seed, (uint32_t)s->step, 1, s->row0, krows, 3, sites, widths, outs);
      if (rc != BR_OK) return rc;
    } else {
      if (!ensure_events()) { br::set_error("brNeumfStepRun: hipEventCreate failed"); return BR_ERR_HIP; }
      hipStream_t as = (hipStream_t)s->aux_stream;
      (void)hipEventRecord(g_kfork, hs);
      (void)hipStreamWaitEvent(as, g_kfork, 0);
      int rc = br::dropout_keep_bits_ahead(p, s->seed, (uint32_t)s->step, 1, s->row0, krows, 3, sites, widths, outs, s->aux_stream);
      if (rc != BR_OK) return rc;
      if ((ph & BR_PH_OPT_DENSE) && fused_final) {
        // the dense finalize (slab reductions, BatchNorm parameter gradients, Adam on the flat vector) only needs the backward, which
        // is behind the fork: it rides the same aux branch beside the Adam-rows kernel instead of waiting for it
        const float* const rs[3] = {slabs1, slabs2, slabs_t};
        const int rn[3] = {sp.ns1, sp.ns2, sp.ns_t};
        const int64_t re[3] = {sp.el1, sp.el2, sp.el_t}, ro[3] = {oW1, oW2, oW3};
        const double* const bs[2] = {bsum1, bsum2};
        const int bn_n[2] = {n1, n2};
        const int64_t bg[2] = {og1, og2}, bb[2] = {obe1, obe2};
        rc = brDenseFinalize(rs, rn, re, ro, bs, bn_n, bg, bb, th, s->adam_m, s->adam_v, gr, n_dense, s->alpha_t, s->beta1, s->beta2, s->adam_eps, s->aux_stream);
        if (rc != BR_OK) return rc;
        final_on_aux = true;
      }
      (void)hipEventRecord(g_kjoin, as);
    }
  }
  {
    uint8_t* um = s->adam_dense == 1 ? s->user_mark : nullptr;
    uint8_t* im = s->adam_dense == 1 ? s->item_mark : nullptr;
    const bool both_rows = (ph & BR_PH_ROWS_USER) && (ph & BR_PH_ROWS_ITEM);
    if (both_rows) {
      // the two fused tables in one launch (they share dim, batch, split): the sweeps of adam_dense == 1 follow.
      // deferred + EMBED: g_user / g_item still hold the stashed MF rows, so each table reads the PARTNER's stash times ddot
      const bool fuse_mf = deferred && (ph & BR_PH_EMBED) && (ph & BR_PH_OPT_TABLES);
      const br::AdamPairCall pc{
          s->user_tab, s->user_m, s->user_v, s->user_rows, s->u_sorted_ids, s->u_sorted_pos, s->dx0 + uoff, 2 * D, (fuse_mf ? s->g_item : s->g_user) + D, 2 * D, um,
          deferred ? s->user_last : nullptr,
          s->item_tab, s->item_m, s->item_v, s->item_rows, s->i_sorted_ids, s->i_sorted_pos, s->dx0 + ioff, 2 * D, (fuse_mf ? s->g_user : s->g_item) + D, 2 * D, im,
          deferred ? s->item_last : nullptr, 2 * D, s->id_type, B, D, fuse_mf ? s->ddot : nullptr, s->step_state, s->alpha_t, s->beta1, s->beta2, s->adam_eps,
          s->u_seg_ws, s->i_seg_ws,
          // the rows as this step's lookup replayed them: MLP halves in x0, MF halves in the stashes (both untouched since)
          fuse_mf ? s->x0 + uoff : nullptr, fuse_mf ? s->g_user + D : nullptr, fuse_mf ? s->x0 + ioff : nullptr, fuse_mf ? s->g_item + D : nullptr, 2 * D};
      // keep_prefetch == 2: the next step's planes and the dense finalize ride in this launch's grid (no fork / join in the step)
      br::FinalArgs fin;
      bool fin_ok = false, fin_done = false;
      if (prefetch_fused && (ph & BR_PH_OPT_DENSE) && fused_final) {
        const float* const rs[3] = {slabs1, slabs2, slabs_t};
        const int rn[3] = {sp.ns1, sp.ns2, sp.ns_t};
        const int64_t re[3] = {sp.el1, sp.el2, sp.el_t}, ro[3] = {oW1, oW2, oW3};
        const double* const bs[2] = {bsum1, bsum2};
        const int bn_n[2] = {n1, n2};
        const int64_t bg[2] = {og1, og2}, bb[2] = {obe1, obe2};
        const int rc = br::make_final_args(fin, rs, rn, re, ro, bs, bn_n, bg, bb, th, s->adam_m, s->adam_v, gr, n_dense);
        if (rc != BR_OK) return rc;
        fin_ok = true;
      }
      RUN(BR_TAG_ADAM_ROWS_USER, br::adam_rows_pair_keep(pc, prefetch_fused ? &next_keep : nullptr, stream, fin_ok ? &fin : nullptr, &fin_done));
      final_on_aux = final_on_aux || fin_done;
    } else if ((ph & BR_PH_ROWS_USER) && deferred) {
      RUN(BR_TAG_ADAM_ROWS_USER, brAdamRowsSortedDeferred(s->user_tab, s->user_m, s->user_v, s->user_last, s->user_rows, 2 * D, s->u_sorted_ids, s->id_type,
                           s->u_sorted_pos, B, s->dx0 + uoff, 2 * D, s->g_user + D, 2 * D, D, s->step_state, s->beta1, s->beta2, s->adam_eps, s->u_seg_ws, stream));
    } else if (ph & BR_PH_ROWS_USER) {
      RUN(BR_TAG_ADAM_ROWS_USER, brAdamRowsSorted(s->user_tab, s->user_m, s->user_v, s->user_rows, 2 * D, s->u_sorted_ids, s->id_type, s->u_sorted_pos, B,
                           s->dx0 + uoff, 2 * D, s->g_user + D, 2 * D, D, s->alpha_t, s->beta1, s->beta2, s->adam_eps, um, s->u_seg_ws, stream));
    }
    if ((ph & BR_PH_SWEEP_USER) && s->adam_dense == 1)
      RUN(BR_TAG_SWEEP_USER, brAdamDenseSweep(s->user_tab, s->user_m, s->user_v, s->user_rows, 2 * D, s->alpha_t, s->beta1, s->beta2, s->adam_eps, um, stream));
    if (both_rows) {
    } else if ((ph & BR_PH_ROWS_ITEM) && deferred) {
      RUN(BR_TAG_ADAM_ROWS_ITEM, brAdamRowsSortedDeferred(s->item_tab, s->item_m, s->item_v, s->item_last, s->item_rows, 2 * D, s->i_sorted_ids, s->id_type,
                           s->i_sorted_pos, B, s->dx0 + ioff, 2 * D, s->g_item + D, 2 * D, D, s->step_state, s->beta1, s->beta2, s->adam_eps, s->i_seg_ws, stream));
    } else if (ph & BR_PH_ROWS_ITEM) {
      RUN(BR_TAG_ADAM_ROWS_ITEM, brAdamRowsSorted(s->item_tab, s->item_m, s->item_v, s->item_rows, 2 * D, s->i_sorted_ids, s->id_type, s->i_sorted_pos, B,
                           s->dx0 + ioff, 2 * D, s->g_item + D, 2 * D, D, s->alpha_t, s->beta1, s->beta2, s->adam_eps, im, s->i_seg_ws, stream));
    }
    if ((ph & BR_PH_SWEEP_ITEM) && s->adam_dense == 1)
      RUN(BR_TAG_SWEEP_ITEM, brAdamDenseSweep(s->item_tab, s->item_m, s->item_v, s->item_rows, 2 * D, s->alpha_t, s->beta1, s->beta2, s->adam_eps, im, stream));
  }
  if (prefetch_aux) (void)hipStreamWaitEvent(hs, g_kjoin, 0);
  if ((ph & BR_PH_OPT_DENSE) && fused_final && !final_on_aux) {
    const float* const rs[3] = {slabs1, slabs2, slabs_t};
    const int rn[3] = {sp.ns1, sp.ns2, sp.ns_t};
    const int64_t re[3] = {sp.el1, sp.el2, sp.el_t}, ro[3] = {oW1, oW2, oW3};
    const double* const bs[2] = {bsum1, bsum2};
    const int bn_n[2] = {n1, n2};
    const int64_t bg[2] = {og1, og2}, bb[2] = {obe1, obe2};
    RUN(BR_TAG_ADAM_FLAT, brDenseFinalize(rs, rn, re, ro, bs, bn_n, bg, bb, th, s->adam_m, s->adam_v, gr, n_dense, s->alpha_t, s->beta1, s->beta2,
                        s->adam_eps, stream));
  } else if ((ph & BR_PH_OPT_DENSE) && !final_on_aux) {
    RUN(BR_TAG_ADAM_FLAT, brAdamFlat(th, s->adam_m, s->adam_v, gr, n_dense, s->alpha_t, s->beta1, s->beta2, s->adam_eps, stream));
  }
  if (!joined) (void)hipStreamWaitEvent(hs, g_join, 0);   // the consumers run in a later call: join the sorts here
  return BR_OK;
}
