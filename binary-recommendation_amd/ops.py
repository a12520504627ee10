"""Thin torch-tensor wrappers over the C-ABI (include/binrec.h).

PyTorch is plumbing here: device memory (`tensor.data_ptr()`), the current HIP stream and,
for the multi-GPU path, `torch.distributed` (RCCL).  All arithmetic happens in
libbinrec_hip.so; nothing in this module computes on the CPU or with torch ops.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import check

I32, I64 = 0, 1
ACT = {"linear": 0, "sigmoid": 1, "relu": 2}
LOSS = {"bce": 0, "mse": 1}
STAT_REPLICAS = 8   # BR_STAT_REPLICAS: every BatchNorm column-sum buffer is double[8][2N]


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_DEV_INDEX = None


def _stream() -> int:
    """hipStream_t of torch's current stream on this process' device (one process per GPU).  The raw getter costs ~1 us;
    building a torch.cuda.Stream object per launch was 0.1 ms of a 0.75 ms row-sharded step."""
    global _DEV_INDEX
    if _RAW_STREAM is None:
        return torch.cuda.current_stream().cuda_stream
    if _DEV_INDEX is None:
        _DEV_INDEX = torch.cuda.current_device()
    return _RAW_STREAM(_DEV_INDEX)


def _p(t):
    return 0 if t is None else t.data_ptr()


def _f32(t: torch.Tensor, name: str):
    if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
        raise TypeError(f"{name}: expected a contiguous float32 device tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")
    return t


def _ids(t, name: str):
    if t is None:
        return None, I32
    if not t.is_cuda or not t.is_contiguous() or t.dtype not in (torch.int32, torch.int64):
        raise TypeError(f"{name}: expected contiguous int32/int64 device ids, got {t.dtype} {t.device}")
    return t, (I64 if t.dtype == torch.int64 else I32)


def _same_id_type(*types):
    ts = {t for t in types}
    if len(ts) != 1:
        raise TypeError("all id tensors of one call must share a dtype (int32 or int64)")
    return ts.pop()


def new_err_flag(device) -> torch.Tensor:
    return torch.zeros(1, dtype=torch.int32, device=device)


def raise_if_flag(flag: torch.Tensor, what="embedding id"):
    """Host sync. Turns the device flag into the IndexError TF-CPU would raise [TF-sem]."""
    v = int(flag.item())
    if v != 0:
        flag.zero_()
        if v & 2:      # BR_ERRFLAG_CAPACITY (brShardDedupPlanPair)
            raise RuntimeError("row-sharded exchange: more DISTINCT ids for one owner than the fixed per-peer capacity in a step (the surplus ids "
                               "were dropped from that step: zeros in the forward, no gradient); raise exchange_capacity or use exchange='exact'")
        raise IndexError(f"{what} out of range")


# ------------------------------------------------------------------------------ G1 / M1
def gather_rows(tables, ids, outs=None, err_flag=None):
    """outs[t] = tables[t][ids[t]] for all t in ONE launch (Embedding lookups)."""
    n = len(tables)
    dim = tables[0].shape[1]
    idt = [_ids(i, "ids") for i in ids]
    id_type = _same_id_type(*[t for _, t in idt])
    batch = ids[0].shape[0] if ids[0] is not None else outs[0].shape[0]
    if outs is None:
        outs = [torch.empty((batch, dim), dtype=torch.float32, device=tables[0].device) for _ in range(n)]
    for t in tables:
        _f32(t, "table")
        if t.shape[1] != dim:
            raise ValueError("gather_rows: all tables must share dim")
    TP = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tables])
    IP = (ctypes.c_void_p * n)(*[_p(i) for i, _ in idt])
    OP = (ctypes.c_void_p * n)(*[_f32(o, "out").data_ptr() for o in outs])
    RW = (ctypes.c_int64 * n)(*[t.shape[0] for t in tables])
    check(_lib.load().brGatherRows(n, TP, RW, IP, OP, dim, batch, id_type, _p(err_flag), _stream()), "brGatherRows")
    return outs


def gather_rows_deferred(table, m, v, last, ids, step_state, beta1=0.9, beta2=0.999, eps=1e-7, out=None, err_flag=None):
    """Lookup on a deferred-Adam table (include/binrec.h "Deferred dense Adam"): rows as of the previous step."""
    t, ty = _ids(ids, "ids")
    n, dim = t.shape[0], table.shape[1]
    if out is None:
        out = torch.empty((n, dim), dtype=torch.float32, device=table.device)
    check(_lib.load().brGatherRowsDeferred(_f32(table, "table").data_ptr(), m.data_ptr(), v.data_ptr(), last.data_ptr(), table.shape[0], dim,
                                           t.data_ptr(), ty, n, step_state.data_ptr(), beta1, beta2, eps, out.data_ptr(), out.stride(0),
                                           _p(err_flag), _stream()), "brGatherRowsDeferred")
    return out


def gather_rows_deferred_pair(tab_a, m_a, v_a, last_a, ids_a, out_a, tab_b, m_b, v_b, last_b, ids_b, out_b, step_state, beta1=0.9, beta2=0.999, eps=1e-7,
                              err_flag=None):
    """gather_rows_deferred on two tables of one geometry in one launch (brGatherRowsDeferredPair)."""
    ta, ty = _ids(ids_a, "ids_a"); tb, tyb = _ids(ids_b, "ids_b")
    n, nb, dim = ta.shape[0], tb.shape[0], tab_a.shape[1]
    if ty != tyb or tab_b.shape[1] != dim or out_a.stride(0) != out_b.stride(0) or out_a.shape[0] < n or out_b.shape[0] < nb:
        raise ValueError("gather_rows_deferred_pair: the two gathers must share id type, dim and output stride")
    check(_lib.load().brGatherRowsDeferredPair(_f32(tab_a, "table_a").data_ptr(), m_a.data_ptr(), v_a.data_ptr(), last_a.data_ptr(), tab_a.shape[0], ta.data_ptr(),
                                               _f32(out_a, "out_a").data_ptr(), _f32(tab_b, "table_b").data_ptr(), m_b.data_ptr(), v_b.data_ptr(), last_b.data_ptr(),
                                               tab_b.shape[0], tb.data_ptr(), _f32(out_b, "out_b").data_ptr(), dim, ty, n, nb, step_state.data_ptr(), beta1, beta2, eps,
                                               out_a.stride(0), _p(err_flag), _stream()), "brGatherRowsDeferredPair")
    return out_a, out_b


def gather_rows_deferred_pair_with_index(tab_a, m_a, v_a, last_a, ids_a, out_a, idx_a: "RowIndex", tab_b, m_b, v_b, last_b, ids_b, out_b, idx_b: "RowIndex",
                                         step_state, lr, beta1=0.9, beta2=0.999, eps=1e-7, advance=True, err_flag=None):
    """gather_rows_deferred_pair + both dedup indexes (+ the step-state advance) in two launches on one stream
    (brGatherRowsDeferredPairWithIndex): the chunk sorts ride in the gather's launch."""
    ta, ty = _ids(ids_a, "ids_a"); tb, tyb = _ids(ids_b, "ids_b")
    na, nb, dim = ta.shape[0], tb.shape[0], tab_a.shape[1]
    if ty != tyb or ty != idx_a.id_type or ty != idx_b.id_type or tab_b.shape[1] != dim or out_a.stride(0) != out_b.stride(0) or na > idx_a.capacity or nb > idx_b.capacity:
        raise ValueError("gather_rows_deferred_pair_with_index: id type / dim / stride / capacity mismatch")
    idx_a.n, idx_b.n = na, nb
    check(_lib.load().brGatherRowsDeferredPairWithIndex(
        _f32(tab_a, "table_a").data_ptr(), m_a.data_ptr(), v_a.data_ptr(), last_a.data_ptr(), tab_a.shape[0], ta.data_ptr(), _f32(out_a, "out_a").data_ptr(),
        idx_a.sorted_ids.data_ptr(), idx_a.sorted_pos.data_ptr(), idx_a.ws.data_ptr(), idx_a.ws_bytes,
        _f32(tab_b, "table_b").data_ptr(), m_b.data_ptr(), v_b.data_ptr(), last_b.data_ptr(), tab_b.shape[0], tb.data_ptr(), _f32(out_b, "out_b").data_ptr(),
        idx_b.sorted_ids.data_ptr(), idx_b.sorted_pos.data_ptr(), idx_b.ws.data_ptr(), idx_b.ws_bytes,
        dim, ty, na, nb, step_state.data_ptr(), 1 if advance else 0, float(lr), beta1, beta2, eps, out_a.stride(0), _p(err_flag), _stream()),
        "brGatherRowsDeferredPairWithIndex")
    return out_a, out_b


def adam_rows_sorted_deferred(table, m, v, last, index, row_grads, ldg, step_state, beta1=0.9, beta2=0.999, eps=1e-7,
                              row_grads_hi=None, ldg_hi=0, split=0, replayed=None):
    """replayed: (n, >= dim) rows as this step's gather_rows_deferred wrote them, aligned with the positions the index was built
    on - the optimizer then replays the moments only (brAdamRowsSortedDeferredReplayed)."""
    if replayed is not None:
        check(_lib.load().brAdamRowsSortedDeferredReplayed(table.data_ptr(), m.data_ptr(), v.data_ptr(), last.data_ptr(), table.shape[0], table.shape[1],
                                                           index.sorted_ids.data_ptr(), index.id_type, index.sorted_pos.data_ptr(), index.n,
                                                           row_grads.data_ptr(), ldg, _p(row_grads_hi), ldg_hi, split, _f32(replayed, "replayed").data_ptr(),
                                                           replayed.stride(0), step_state.data_ptr(), beta1, beta2, eps,
                                                           index.seg_ws(table.shape[1]).data_ptr(), _stream()), "brAdamRowsSortedDeferredReplayed")
        return
    check(_lib.load().brAdamRowsSortedDeferred(table.data_ptr(), m.data_ptr(), v.data_ptr(), last.data_ptr(), table.shape[0], table.shape[1],
                                               index.sorted_ids.data_ptr(), index.id_type, index.sorted_pos.data_ptr(), index.n,
                                               row_grads.data_ptr(), ldg, _p(row_grads_hi), ldg_hi, split, step_state.data_ptr(),
                                               beta1, beta2, eps, index.seg_ws(table.shape[1]).data_ptr(), _stream()), "brAdamRowsSortedDeferred")


def adam_rows_sorted_deferred_pair_replayed(tab_a, m_a, v_a, last_a, idx_a, g_a, rep_a, tab_b, m_b, v_b, last_b, idx_b, g_b, rep_b, split, step_state,
                                            beta1=0.9, beta2=0.999, eps=1e-7):
    """two deferred tables of one row width in ONE launch (brAdamRowsSortedPairReplayed): g_x = (n_x, dim) row gradients by position
    (split > 0: [0, split) | [split, dim) read as two halves of the same rows; split = 0: one source), rep_x = the rows as this step's
    deferred gather wrote them, same positions.  The two indexes may differ in length (one wave per row shapes: dim 64 / 128 / 256)."""
    dim = tab_a.shape[1]
    if tab_b.shape[1] != dim or g_a.stride(0) != g_b.stride(0) or rep_a.stride(0) != rep_b.stride(0) or idx_a.id_type != idx_b.id_type:
        raise ValueError("adam_rows_sorted_deferred_pair_replayed: the two tables must share dim, id type and strides")
    ld, ldr = g_a.stride(0), rep_a.stride(0)
    hi_a = g_a.data_ptr() + 4 * split if split else None
    hi_b = g_b.data_ptr() + 4 * split if split else None
    check(_lib.load().brAdamRowsSortedPairReplayed(
        tab_a.data_ptr(), m_a.data_ptr(), v_a.data_ptr(), tab_a.shape[0], idx_a.sorted_ids.data_ptr(), idx_a.sorted_pos.data_ptr(),
        g_a.data_ptr(), ld, hi_a, ld, last_a.data_ptr(), _f32(rep_a, "rep_a").data_ptr(),
        tab_b.data_ptr(), m_b.data_ptr(), v_b.data_ptr(), tab_b.shape[0], idx_b.sorted_ids.data_ptr(), idx_b.sorted_pos.data_ptr(),
        g_b.data_ptr(), ld, hi_b, ld, last_b.data_ptr(), _f32(rep_b, "rep_b").data_ptr(), ldr,
        dim, idx_a.id_type, idx_a.n, idx_b.n if idx_b.n != idx_a.n else 0, split if split else dim, step_state.data_ptr(), beta1, beta2, eps,
        idx_a.seg_ws(dim).data_ptr(), idx_b.seg_ws(dim).data_ptr(), _stream()), "brAdamRowsSortedPairReplayed")


def row_dot(a, b, out=None):
    _f32(a, "a"); _f32(b, "b")
    if out is None:
        out = torch.empty(a.shape[0], dtype=torch.float32, device=a.device)
    check(_lib.load().brRowDot(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.shape[1], a.shape[0], _stream()), "brRowDot")
    return out


def row_dot_backward(a, b, dout, da=None, db=None):
    da = torch.empty_like(a) if da is None else da
    db = torch.empty_like(b) if db is None else db
    check(_lib.load().brRowDotBackward(_f32(a, "a").data_ptr(), _f32(b, "b").data_ptr(), _f32(dout, "dout").data_ptr(),
                                       da.data_ptr(), db.data_ptr(), a.shape[1], a.shape[0], _stream()), "brRowDotBackward")
    return da, db


def neumf_embed_forward(user_mlp, item_mlp, user_mf, item_mf, users, items, item_first, x0, dot, err_flag=None, stash=None):
    """Tables may be separate (row stride = dim) or column views of fused [rows][mlp|mf] allocations.
    stash = (stash_user, stash_item): (B, dim) views of one row stride that receive the two MF rows of every pair (brNeumfEmbedForwardStash)."""
    u, ut = _ids(users, "users"); i, it = _ids(items, "items")
    id_type = _same_id_type(ut, it)
    dim = user_mlp.shape[1]
    batch = x0.shape[0]
    if user_mlp.stride(0) != user_mf.stride(0) or item_mlp.stride(0) != item_mf.stride(0):
        raise ValueError("mlp/mf tables of one stream must share a row stride")
    su, si = stash if stash is not None else (None, None)
    if su is not None and (su.stride(0) != si.stride(0) or su.shape[0] < batch or si.shape[0] < batch):
        raise ValueError("stashes must share a row stride and hold the batch")
    check(_lib.load().brNeumfEmbedForwardStash(user_mlp.data_ptr(), item_mlp.data_ptr(), user_mf.data_ptr(), item_mf.data_ptr(),
                                               user_mlp.stride(0), item_mlp.stride(0), user_mlp.shape[0], item_mlp.shape[0],
                                               _p(u), _p(i), id_type, dim, batch, int(item_first), _f32(x0, "x0").data_ptr(),
                                               _f32(dot, "dot").data_ptr(), _p(su), _p(si), su.stride(0) if su is not None else 0, _p(err_flag), _stream()),
          "brNeumfEmbedForwardStash")


def neumf_embed_backward(user_mf, item_mf, users, items, item_first, dx0, ddot, g_user_mf, g_item_mf,
                         g_user_mlp=None, g_item_mlp=None, out_rows_by_id=False):
    """g_* may be (B, dim) buffers or column views of fused (B, 2*dim) buffers (shared row stride)."""
    u, ut = _ids(users, "users"); i, it = _ids(items, "items")
    id_type = _same_id_type(ut, it)
    dim = user_mf.shape[1]
    batch = ddot.shape[0]
    ldg = g_user_mf.stride(0)
    for t in (g_item_mf, g_user_mlp, g_item_mlp):
        if t is not None and t.stride(0) != ldg:
            raise ValueError("row-gradient outputs must share a row stride")
    check(_lib.load().brNeumfEmbedBackward(user_mf.data_ptr(), item_mf.data_ptr(), user_mf.stride(0), item_mf.stride(0),
                                           user_mf.shape[0], item_mf.shape[0], _p(u), _p(i), id_type, dim, batch,
                                           int(item_first), _p(dx0), _f32(ddot, "ddot").data_ptr(), _p(g_user_mlp),
                                           _p(g_item_mlp), g_user_mf.data_ptr(), g_item_mf.data_ptr(), ldg, 1 if out_rows_by_id else 0, _stream()),
          "brNeumfEmbedBackward")


# ------------------------------------------------------------------------------ L3 BPR
def bpr_forward_backward(user_table, item_table, users, pos, neg, inv_batch, loss_sum, g_user, g_item,
                         per_triplet=None, err_flag=None):
    u, ut = _ids(users, "users"); p, pt = _ids(pos, "pos"); n, nt = _ids(neg, "neg")
    id_type = _same_id_type(ut, pt, nt)
    check(_lib.load().brBprForwardBackward(_f32(user_table, "user_table").data_ptr(), _f32(item_table, "item_table").data_ptr(),
                                           user_table.shape[0], item_table.shape[0], u.data_ptr(), p.data_ptr(), n.data_ptr(),
                                           id_type, user_table.shape[1], u.shape[0], float(inv_batch), _p(per_triplet),
                                           loss_sum.data_ptr(), _f32(g_user, "g_user").data_ptr(),
                                           _f32(g_item, "g_item").data_ptr(), _p(err_flag), _stream()), "brBprForwardBackward")


# ------------------------------------------------------------------------------ S1 index
class RowIndex:
    """Sorted (id, batch position) index of one id stream, reusable by every table fed by it."""

    def __init__(self, capacity: int, id_dtype: torch.dtype, device):
        self.capacity = capacity
        self.id_dtype = id_dtype
        self.id_type = I64 if id_dtype == torch.int64 else I32
        self.sorted_ids = torch.empty(capacity, dtype=id_dtype, device=device)
        self.sorted_pos = torch.empty(capacity, dtype=torch.int32, device=device)
        self.ws_bytes = int(_lib.load().brRowIndexWorkspaceBytes(capacity, self.id_type))
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device)
        self.n = 0
        self._seg = {}

    def seg_ws(self, dim: int):
        """scratch for the two-level ordered duplicate sum (brSegmentScratchFloats): hot ids cost a chain of
        capacity/64 + 64 dependent adds instead of one as long as the segment"""
        t = self._seg.get(dim)
        if t is None:
            t = self._seg[dim] = torch.empty(int(_lib.load().brSegmentScratchFloats(self.capacity, dim)), dtype=torch.float32,
                                             device=self.sorted_ids.device)
        return t

    def build(self, ids: torch.Tensor, id_upper_bound: int = 0):
        t, ty = _ids(ids, "ids")
        if ty != self.id_type or t.shape[0] > self.capacity:
            raise ValueError("RowIndex.build: dtype/capacity mismatch")
        self.n = t.shape[0]
        check(_lib.load().brRowIndexBuild(t.data_ptr(), ty, self.n, int(id_upper_bound), self.sorted_ids.data_ptr(),
                                          self.sorted_pos.data_ptr(), self.ws.data_ptr(), self.ws_bytes, _stream()),
              "brRowIndexBuild")
        return self


def row_index_build_pair(idx_a: RowIndex, ids_a, upper_a: int, idx_b: RowIndex, ids_b, upper_b: int):
    """both dedup indexes of a step in shared launches (brRowIndexBuildPair); ids of equal length and type."""
    ta, ty = _ids(ids_a, "ids_a"); tb, tyb = _ids(ids_b, "ids_b")
    n = ta.shape[0]
    if ty != tyb or tb.shape[0] != n or ty != idx_a.id_type or ty != idx_b.id_type or n > min(idx_a.capacity, idx_b.capacity):
        raise ValueError("row_index_build_pair: dtype / length / capacity mismatch")
    idx_a.n = idx_b.n = n
    check(_lib.load().brRowIndexBuildPair(ta.data_ptr(), int(upper_a), idx_a.sorted_ids.data_ptr(), idx_a.sorted_pos.data_ptr(), idx_a.ws.data_ptr(), idx_a.ws_bytes,
                                          tb.data_ptr(), int(upper_b), idx_b.sorted_ids.data_ptr(), idx_b.sorted_pos.data_ptr(), idx_b.ws.data_ptr(), idx_b.ws_bytes,
                                          ty, n, _stream()), "brRowIndexBuildPair")


def row_index_build_pair_seg(idx_a: RowIndex, upper_a: int, idx_b: RowIndex, upper_b: int, ids, n: int, seg):
    """both owner-side dedup indexes over the merged id array of the row-sharded exchange ([source][stream][cap]; seg = (seg_len,
    seg_stride, offset a, offset b)): n logical positions per stream, sorted_pos = physical element index (brRowIndexBuildPairSeg)."""
    t, ty = _ids(ids, "ids")
    if ty != idx_a.id_type or ty != idx_b.id_type or n > min(idx_a.capacity, idx_b.capacity):
        raise ValueError("row_index_build_pair_seg: dtype / capacity mismatch")
    idx_a.n = idx_b.n = n
    check(_lib.load().brRowIndexBuildPairSeg(t.data_ptr(), int(upper_a), idx_a.sorted_ids.data_ptr(), idx_a.sorted_pos.data_ptr(), idx_a.ws.data_ptr(), idx_a.ws_bytes,
                                             t.data_ptr(), int(upper_b), idx_b.sorted_ids.data_ptr(), idx_b.sorted_pos.data_ptr(), idx_b.ws.data_ptr(), idx_b.ws_bytes,
                                             ty, n, *[int(v) for v in seg], _stream()), "brRowIndexBuildPairSeg")


def row_index_merge_pair_seg(idx_a: RowIndex, upper_a: int, idx_b: RowIndex, upper_b: int, ids, n: int, seg, err_flag=None):
    """the same two indexes when every segment of `ids` is already sorted ascending (the fixed-capacity exchange delivers each requester's
    distinct rows in key order, pads last): a merge of the per-source runs, no sort (brRowIndexMergePairSeg).  An unsorted segment sets
    BR_ERRFLAG_RANGE in err_flag."""
    t, ty = _ids(ids, "ids")
    if ty != idx_a.id_type or ty != idx_b.id_type or n > min(idx_a.capacity, idx_b.capacity):
        raise ValueError("row_index_merge_pair_seg: dtype / capacity mismatch")
    idx_a.n = idx_b.n = n
    check(_lib.load().brRowIndexMergePairSeg(t.data_ptr(), int(upper_a), idx_a.sorted_ids.data_ptr(), idx_a.sorted_pos.data_ptr(),
                                             t.data_ptr(), int(upper_b), idx_b.sorted_ids.data_ptr(), idx_b.sorted_pos.data_ptr(),
                                             ty, n, *[int(v) for v in seg], _p(err_flag), _stream()), "brRowIndexMergePairSeg")


def gather_rows_deferred_pair_seg(tab_a, m_a, v_a, last_a, tab_b, m_b, v_b, last_b, ids, out, n, seg, step_state, beta1=0.9, beta2=0.999, eps=1e-7, err_flag=None):
    """owner-side lookup of both streams of the merged exchange buffer on deferred tables, one launch: out[p] = row ids[p] of the stream's
    table (as of step - 1) at every physical slot p."""
    t, ty = _ids(ids, "ids")
    dim = tab_a.shape[1]
    check(_lib.load().brGatherRowsDeferredPairSeg(_f32(tab_a, "table_a").data_ptr(), m_a.data_ptr(), v_a.data_ptr(), last_a.data_ptr(), tab_a.shape[0],
                                                  _f32(tab_b, "table_b").data_ptr(), m_b.data_ptr(), v_b.data_ptr(), last_b.data_ptr(), tab_b.shape[0], t.data_ptr(),
                                                  _f32(out, "out").data_ptr(), dim, ty, n, *[int(v) for v in seg], step_state.data_ptr(), beta1, beta2, eps,
                                                  out.stride(0), _p(err_flag), _stream()), "brGatherRowsDeferredPairSeg")
    return out


def gather_rows_pair_seg(tab_a, tab_b, ids, out, n, seg, err_flag=None):
    """the same on plain tables (brGatherRowsPairSeg)."""
    t, ty = _ids(ids, "ids")
    check(_lib.load().brGatherRowsPairSeg(_f32(tab_a, "table_a").data_ptr(), tab_a.shape[0], _f32(tab_b, "table_b").data_ptr(), tab_b.shape[0], t.data_ptr(),
                                          _f32(out, "out").data_ptr(), tab_a.shape[1], ty, n, *[int(v) for v in seg], out.stride(0), _p(err_flag), _stream()),
          "brGatherRowsPairSeg")
    return out


def segment_sum_rows(index: RowIndex, row_grads, dim=None, ldg=None, out=None, head_flag=None, two_level=True):
    dim = row_grads.shape[1] if dim is None else dim
    ldg = row_grads.stride(0) if ldg is None else ldg
    if out is None:
        out = torch.zeros((index.n, dim), dtype=torch.float32, device=row_grads.device)
    if head_flag is None:
        head_flag = torch.empty(index.n, dtype=torch.int32, device=row_grads.device)
    check(_lib.load().brSegmentSumRows(index.sorted_ids.data_ptr(), index.id_type, index.sorted_pos.data_ptr(), index.n,
                                       row_grads.data_ptr(), ldg, dim, out.data_ptr(), head_flag.data_ptr(),
                                       index.seg_ws(dim).data_ptr() if two_level else None, _stream()),
          "brSegmentSumRows")
    return out, head_flag


def scatter_add_rows(g_table, ids, rows, err_flag=None):
    t, ty = _ids(ids, "ids")
    check(_lib.load().brScatterAddRows(_f32(g_table, "g_table").data_ptr(), g_table.shape[0], t.data_ptr(), ty, t.shape[0],
                                       _f32(rows, "rows").data_ptr(), g_table.shape[1], _p(err_flag), _stream()),
          "brScatterAddRows")


# ------------------------------------------------------------------------------ O1 / O2
def adam_rows_sorted(table, m, v, index: RowIndex, row_grads, ldg, alpha_t, beta1=0.9, beta2=0.999, eps=1e-7, mark=None,
                     row_grads_hi=None, ldg_hi=0, split=0):
    check(_lib.load().brAdamRowsSorted(_f32(table, "table").data_ptr(), _f32(m, "m").data_ptr(), _f32(v, "v").data_ptr(),
                                       table.shape[0], table.shape[1], index.sorted_ids.data_ptr(), index.id_type,
                                       index.sorted_pos.data_ptr(), index.n, row_grads.data_ptr(), int(ldg), _p(row_grads_hi),
                                       int(ldg_hi), int(split), float(alpha_t), float(beta1), float(beta2), float(eps),
                                       _p(mark), index.seg_ws(table.shape[1]).data_ptr(), _stream()), "brAdamRowsSorted")


def adam_dense_sweep(table, m, v, alpha_t, beta1=0.9, beta2=0.999, eps=1e-7, mark=None):
    check(_lib.load().brAdamDenseSweep(_f32(table, "table").data_ptr(), _f32(m, "m").data_ptr(), _f32(v, "v").data_ptr(),
                                       table.shape[0], table.shape[1], float(alpha_t), float(beta1), float(beta2), float(eps),
                                       _p(mark), _stream()), "brAdamDenseSweep")


def adam_flat(theta, m, v, g, alpha_t, beta1=0.9, beta2=0.999, eps=1e-7):
    check(_lib.load().brAdamFlat(_f32(theta, "theta").data_ptr(), _f32(m, "m").data_ptr(), _f32(v, "v").data_ptr(),
                                 _f32(g, "g").data_ptr(), theta.numel(), float(alpha_t), float(beta1), float(beta2), float(eps),
                                 _stream()), "brAdamFlat")


def adagrad_rows_sorted(table, acc, index: RowIndex, row_grads, ldg, lr, eps=1e-7):
    check(_lib.load().brAdagradRowsSorted(_f32(table, "table").data_ptr(), _f32(acc, "acc").data_ptr(), table.shape[0],
                                          table.shape[1], index.sorted_ids.data_ptr(), index.id_type,
                                          index.sorted_pos.data_ptr(), index.n, row_grads.data_ptr(), int(ldg), float(lr),
                                          float(eps), index.seg_ws(table.shape[1]).data_ptr(), _stream()), "brAdagradRowsSorted")


def adagrad_flat(theta, acc, g, lr, eps=1e-7):
    check(_lib.load().brAdagradFlat(_f32(theta, "theta").data_ptr(), _f32(acc, "acc").data_ptr(), _f32(g, "g").data_ptr(),
                                    theta.numel(), float(lr), float(eps), _stream()), "brAdagradFlat")


REPLAY = {"exact": 0, "fast": 1}     # BR_REPLAY_EXACT / BR_REPLAY_FAST


def new_step_state(device, beta1=0.9, beta2=0.999, eps=1e-7, replay="fast") -> torch.Tensor:
    """The device step state of a deferred-Adam engine (brStepStateBytes): zeroed, with the replay form written into it
    (brStepStateInit: "fast" = the cheaper recurrence, "exact" = the sweep's own operations, bit-equal tables)."""
    st = torch.zeros(int(_lib.load().brStepStateBytes()) // 4, dtype=torch.int32, device=device)
    check(_lib.load().brStepStateInit(st.data_ptr(), float(beta1), float(beta2), float(eps), REPLAY[replay], _stream()), "brStepStateInit")
    return st


def adam_alpha(lr: float, t: int, beta1=0.9, beta2=0.999) -> float:
    """[TF-sem] alpha_t = lr*sqrt(1-b2^t)/(1-b1^t) in double on the host."""
    return lr * (1.0 - beta2 ** t) ** 0.5 / (1.0 - beta1 ** t)


# ------------------------------------------------------------------------------ T1-T4 MLP tower
def dropout_keep_words(batch, K) -> int:
    return int(_lib.load().brDropoutKeepWords(batch, K))


def dropout_keep_bits(drop_p, seed, step, row0, batch, sites, widths, outs=None):
    """T4: the Philox keep-bit planes of up to three dropout sites in one launch (uint32 [batch][ceil(K/32)] each)."""
    n = len(sites)
    if outs is None:
        dev = torch.device("cuda", torch.cuda.current_device())
        outs = [torch.empty(dropout_keep_words(batch, k), dtype=torch.int32, device=dev) for k in widths]
    S = (ctypes.c_uint32 * n)(*[int(s) for s in sites])
    Wd = (ctypes.c_int * n)(*[int(k) for k in widths])
    OP = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
    check(_lib.load().brDropoutKeepBits(float(drop_p), int(seed), int(step), int(row0), int(batch), n, S, Wd, OP, _stream()), "brDropoutKeepBits")
    return outs


def dense_forward(x, W, bias, y, act, in_scale=None, in_shift=None, drop_p=0.0, seed=0, step=0, site=0, row0=0,
                  stats=None, batch=None, keep=None):
    """keep: the site's bit plane (dropout_keep_bits); built here from (seed, step, site, row0) when drop_p > 0 and it is None."""
    B = x.shape[0] if batch is None else batch
    K, N = W.shape
    if drop_p > 0 and keep is None and B > 0:
        keep = dropout_keep_bits(drop_p, seed, step, row0, B, [site], [K])[0]
    check(_lib.load().brDenseForward(x.data_ptr(), x.stride(0), _f32(W, "W").data_ptr(), _p(bias), y.data_ptr(), y.stride(0),
                                     B, K, N, ACT[act], _p(in_scale), _p(in_shift), None, float(drop_p), _p(keep) if drop_p > 0 else 0,
                                     _p(stats), _stream()), "brDenseForward")


def bn_finalize(stats, batch_total, gamma, beta, eps, momentum, moving_mean, moving_var, scale, shift, mean, rstd):
    check(_lib.load().brBnFinalize(stats.data_ptr(), float(batch_total), gamma.data_ptr(), beta.data_ptr(), float(eps),
                                   float(momentum), _p(moving_mean), _p(moving_var), scale.data_ptr(), shift.data_ptr(),
                                   mean.data_ptr(), rstd.data_ptr(), gamma.numel(), _stream()), "brBnFinalize")


def bn_inference(gamma, beta, moving_mean, moving_var, eps, scale, shift):
    check(_lib.load().brBnInference(gamma.data_ptr(), beta.data_ptr(), moving_mean.data_ptr(), moving_var.data_ptr(),
                                    float(eps), scale.data_ptr(), shift.data_ptr(), gamma.numel(), _stream()), "brBnInference")


def dense_backward_slabs(batch, K, N) -> int:
    return int(_lib.load().brDenseBackwardSlabs(batch, K, N))


def dense_backward_ws_floats(batch, K, N) -> int:
    return int(_lib.load().brDenseBackwardWorkspaceFloats(batch, K, N))


def dense_backward(gy, y, x, W, act, slabs, n_slabs, gx=None, out_bn=None, bn_sums=None, batch_total=None,
                   in_scale=None, in_shift=None, in_bn=None, in_drop_p=0.0, in_site=0, seed=0, step=0, row0=0,
                   in_bn_sums=None, batch=None, dz_ws=None, keep=None):
    """out_bn = (mean, rstd, gamma) of the BN after this layer; in_bn = (mean, rstd) of the BN before it.
    keep: bit plane of the input dropout; built from (seed, step, in_site, row0) when in_drop_p > 0 and it is None."""
    B = gy.shape[0] if batch is None else batch
    K, N = W.shape
    om, ors, og = out_bn if out_bn is not None else (None, None, None)
    im, irs = in_bn if in_bn is not None else (None, None)
    if dz_ws is None:
        dz_ws = torch.empty(dense_backward_ws_floats(B, K, N), dtype=torch.float32, device=gy.device)
    if in_drop_p > 0 and keep is None and B > 0:
        keep = dropout_keep_bits(in_drop_p, seed, step, row0, B, [in_site], [K])[0]
    check(_lib.load().brDenseBackward(gy.data_ptr(), gy.stride(0), y.data_ptr(), y.stride(0), x.data_ptr(), x.stride(0),
                                      W.data_ptr(), B, K, N, ACT[act], _p(om), _p(ors), _p(og), _p(bn_sums),
                                      float(batch_total if batch_total is not None else B), _p(in_scale), _p(in_shift),
                                      _p(im), _p(irs), float(in_drop_p), _p(keep) if in_drop_p > 0 else 0,
                                      _p(gx), gx.stride(0) if gx is not None else 0, dz_ws.data_ptr(), slabs.data_ptr(), int(n_slabs),
                                      _p(in_bn_sums), _stream()), "brDenseBackward")


def reduce_slabs(slabs, n_slabs, slab_elems, out):
    check(_lib.load().brReduceSlabs(slabs.data_ptr(), int(n_slabs), int(slab_elems), out.data_ptr(), _stream()), "brReduceSlabs")


def bn_param_grads(bn_sums, dgamma, dbeta):
    check(_lib.load().brBnParamGrads(bn_sums.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), dgamma.numel(), _stream()),
          "brBnParamGrads")


def head_slabs(batch) -> int:
    return int(_lib.load().brHeadSlabs(batch))


def neumf_head(a3, dot, labels, w4, b4, mf_first, loss, inv_batch, logit=None, prob=None, sums=None, da3=None, ddot=None,
               slabs=None, n_slabs=0, batch=None):
    B = dot.shape[0] if batch is None else batch
    check(_lib.load().brNeumfHead(a3.data_ptr(), a3.stride(0), dot.data_ptr(), _p(labels), w4.data_ptr(), b4.data_ptr(), B,
                                  a3.shape[1], int(mf_first), LOSS[loss], float(inv_batch), _p(logit), _p(prob), _p(sums),
                                  _p(da3), da3.stride(0) if da3 is not None else 0, _p(ddot), _p(slabs), int(n_slabs),
                                  _stream()), "brNeumfHead")


def bce_logits(z, y, inv_batch, prob=None, dz=None, sums=None):
    check(_lib.load().brBceLogits(z.data_ptr(), y.data_ptr(), z.shape[0], float(inv_batch), _p(prob), _p(dz), _p(sums),
                                  _stream()), "brBceLogits")


# ------------------------------------------------------------------------------ L4 in-batch softmax / E1 scoring + top-k
SUM_SLOTS = 64   # BR_SUM_SLOTS
METRIC_SUMS = 8  # BR_METRIC_SUMS: [loss, se, ae, correct, bce, tp, fp, fn]


_softmax_ws = {}


def _softmax_workspace(Q, C):
    """scratch of the split in-batch softmax sweeps, one grow-only buffer per (device, stream)."""
    need = int(_lib.load().brInBatchSoftmaxWorkspaceBytes(Q.shape[0], C.shape[0], Q.shape[1]))
    key = (Q.device, _stream())
    buf = _softmax_ws.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.empty(max(need, 16), dtype=torch.uint8, device=Q.device)
        _softmax_ws[key] = buf
    return buf


def inbatch_softmax_lse(Q, C, q_pos_ids, cand_ids, diag_offset, row_lse, loss_sum):
    qi, qt = _ids(q_pos_ids, "q_pos_ids"); ci, ct = _ids(cand_ids, "cand_ids")
    id_type = _same_id_type(qt, ct) if qi is not None else I32
    ws = _softmax_workspace(Q, C)
    check(_lib.load().brInBatchSoftmaxLse(_f32(Q, "Q").data_ptr(), _f32(C, "C").data_ptr(), _p(qi), _p(ci), id_type, Q.shape[0], C.shape[0],
                                          Q.shape[1], int(diag_offset), _f32(row_lse, "row_lse").data_ptr(), loss_sum.data_ptr(),
                                          ws.data_ptr(), ws.numel(), _stream()), "brInBatchSoftmaxLse")


def inbatch_softmax_lse_grad_q(Q, C, q_pos_ids, cand_ids, diag_offset, row_lse, loss_sum, dQ):
    """lse + loss + dQ in one sweep (online softmax): the training step's first softmax pass."""
    qi, qt = _ids(q_pos_ids, "q_pos_ids"); ci, ct = _ids(cand_ids, "cand_ids")
    id_type = _same_id_type(qt, ct) if qi is not None else I32
    ws = _softmax_workspace(Q, C)
    check(_lib.load().brInBatchSoftmaxLseGradQ(_f32(Q, "Q").data_ptr(), _f32(C, "C").data_ptr(), _p(qi), _p(ci), id_type, Q.shape[0], C.shape[0],
                                               Q.shape[1], int(diag_offset), _f32(row_lse, "row_lse").data_ptr(), loss_sum.data_ptr(),
                                               _f32(dQ, "dQ").data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "brInBatchSoftmaxLseGradQ")


def inbatch_softmax_grad(Q, C, q_pos_ids, cand_ids, diag_offset, row_lse, dQ=None, dC=None):
    qi, qt = _ids(q_pos_ids, "q_pos_ids"); ci, ct = _ids(cand_ids, "cand_ids")
    id_type = _same_id_type(qt, ct) if qi is not None else I32
    ws = _softmax_workspace(Q, C)
    check(_lib.load().brInBatchSoftmaxGrad(_f32(Q, "Q").data_ptr(), _f32(C, "C").data_ptr(), _p(qi), _p(ci), id_type, Q.shape[0], C.shape[0],
                                           Q.shape[1], int(diag_offset), row_lse.data_ptr(), _p(dQ), _p(dC), ws.data_ptr(), ws.numel(), _stream()),
          "brInBatchSoftmaxGrad")


def score_matrix(Q, C, out=None):
    if out is None:
        out = torch.empty(Q.shape[0], C.shape[0], dtype=torch.float32, device=Q.device)
    check(_lib.load().brScoreMatrix(_f32(Q, "Q").data_ptr(), _f32(C, "C").data_ptr(), Q.shape[0], C.shape[0], Q.shape[1], out.data_ptr(),
                                    out.stride(0), _stream()), "brScoreMatrix")
    return out


def topk_rows(scores, k):
    """stable top-k per row: descending, ties keep the lower column (topKmetrics.py:59,68)."""
    U, I = scores.shape
    os_ = torch.empty(U, k, dtype=torch.float32, device=scores.device)
    oi = torch.empty(U, k, dtype=torch.int32, device=scores.device)
    check(_lib.load().brTopKRows(_f32(scores, "scores").data_ptr(), U, I, int(k), os_.data_ptr(), oi.data_ptr(), _stream()), "brTopKRows")
    return os_, oi


# ------------------------------------------------------------------------------ 8f-1 evaluation: full AUC, MAP@k, hit counts
def truth_csr(n_users: int, user_rows, item_cols, device):
    """Ground truth of `n_users` rows as the CSR the eval kernels take: (offsets int64 (n_users + 1), column indices int32 ascending
    per user).  user_rows / item_cols: equal-length integer sequences (row index into the scored users, column index into the
    scored items); duplicates are dropped.  Host-side index plumbing (numpy), done once per evaluation."""
    import numpy as np
    u = np.asarray(user_rows, dtype=np.int64); c = np.asarray(item_cols, dtype=np.int64)
    key = np.unique(u * (int(c.max()) + 1 if c.size else 1) + c) if u.size else np.empty(0, np.int64)
    m = int(c.max()) + 1 if c.size else 1
    uu, cc = key // m, key % m
    off = np.zeros(n_users + 1, dtype=np.int64)
    np.add.at(off, uu + 1, 1)
    off = np.cumsum(off)
    return torch.from_numpy(off).to(device), torch.from_numpy(cc.astype(np.int32)).to(device)


def full_auc(scores, truth_off, truth_idx):
    """per-user roc_auc_score over all scored items (src/models/bpr.py:230-254); NaN where undefined."""
    U, I = scores.shape
    out = torch.empty(U, dtype=torch.float32, device=scores.device)
    check(_lib.load().brFullAuc(_f32(scores, "scores").data_ptr(), scores.stride(0), truth_off.data_ptr(), truth_idx.data_ptr(), U, I,
                                out.data_ptr(), _stream()), "brFullAuc")
    return out


def map_at_k(topk_index, truth_off, truth_idx, want_ap=True, want_hits=True):
    """(AP@k per user, hits per user) of top-k index lists (src/models/bpr.py:257-289; trainers/topKmetrics.py:85-93)."""
    U, k = topk_index.shape
    ap = torch.empty(U, dtype=torch.float32, device=topk_index.device) if want_ap else None
    hits = torch.empty(U, dtype=torch.int32, device=topk_index.device) if want_hits else None
    check(_lib.load().brMapAtK(topk_index.contiguous().data_ptr(), U, k, truth_off.data_ptr(), truth_idx.data_ptr(), _p(ap), _p(hits), _stream()),
          "brMapAtK")
    return ap, hits


# ------------------------------------------------------------------------------ 8f-2 batch construction on the device
def positives_csr(users, items, num_users: int, device):
    """The users' positives as CSR (offsets int64 (num_users + 1), items ascending per user, dtype of `items`): the membership
    structure of the rejection samplers.  Built once per dataset on the host (numpy lexsort: index plumbing, not the hot path)."""
    import numpy as np
    u = np.asarray(users.cpu() if torch.is_tensor(users) else users).astype(np.int64)
    i = np.asarray(items.cpu() if torch.is_tensor(items) else items).astype(np.int64)
    order = np.lexsort((i, u))
    u, i = u[order], i[order]
    keep = np.ones(len(u), bool)
    keep[1:] = (u[1:] != u[:-1]) | (i[1:] != i[:-1])
    u, i = u[keep], i[keep]
    off = np.zeros(num_users + 1, dtype=np.int64)
    np.add.at(off, u + 1, 1)
    dt = items.dtype if torch.is_tensor(items) else torch.int32
    return torch.from_numpy(np.cumsum(off)).to(device), torch.from_numpy(i).to(device).to(dt)


def bootstrap_dataset(users, items, n_neg: int, seed: int):
    """NeuMFModel.bootstrapDataset on the device -> (users, items, labels) of n + n_neg shuffled samples."""
    u, ut = _ids(users, "users"); i, it = _ids(items, "items")
    ty = _same_id_type(ut, it)
    n = u.shape[0]
    ou, oi = torch.empty(n + n_neg, dtype=u.dtype, device=u.device), torch.empty(n + n_neg, dtype=u.dtype, device=u.device)
    oy = torch.empty(n + n_neg, dtype=torch.float32, device=u.device)
    check(_lib.load().brBootstrapDataset(u.data_ptr(), i.data_ptr(), ty, n, int(n_neg), int(seed), ou.data_ptr(), oi.data_ptr(), oy.data_ptr(), _stream()),
          "brBootstrapDataset")
    return ou, oi, oy


def bpr_sample_triplets(users, items, pos_off, pos_items, neg_per_pos: int, seed: int, n_cand: int, cand_items=None, max_tries: int = 16):
    """Sampled BPR triplets -> (users, positives, negatives), n * neg_per_pos each."""
    u, ut = _ids(users, "users"); i, it = _ids(items, "items")
    ty = _same_id_type(ut, it)
    n = u.shape[0]
    T = n * int(neg_per_pos)
    ou, op, on = (torch.empty(T, dtype=u.dtype, device=u.device) for _ in range(3))
    check(_lib.load().brBprSampleTriplets(u.data_ptr(), i.data_ptr(), ty, n, int(neg_per_pos), pos_off.data_ptr(), pos_items.data_ptr(), _p(cand_items),
                                          int(n_cand), int(seed), int(max_tries), ou.data_ptr(), op.data_ptr(), on.data_ptr(), _stream()),
          "brBprSampleTriplets")
    return ou, op, on


def ncf_negatives(users, items, pos_off, pos_items, num_items: int, size: int, seed: int, oversample: float = 1.3, max_rounds: int = 8):
    """generateNegativeFeedback on the device: `size` DISTINCT (user, item) pairs outside the positives, users and items drawn by
    shuffling the two columns independently.  One host sync per round (the count of distinct valid candidates)."""
    u, ut = _ids(users, "users"); i, it = _ids(items, "items")
    ty = _same_id_type(ut, it)
    n, dev, lib = u.shape[0], u.device, _lib.load()
    n_cand = int(size * oversample) + n
    for _ in range(max_rounds):
        keys = torch.empty(n_cand, dtype=torch.int64, device=dev)
        check(lib.brNcfNegativeCandidates(u.data_ptr(), i.data_ptr(), ty, n, n_cand, pos_off.data_ptr(), pos_items.data_ptr(), int(num_items), int(seed),
                                          keys.data_ptr(), _stream()), "brNcfNegativeCandidates")
        wsb = int(lib.brSortUniqueWorkspaceBytes(n_cand))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        uniq = torch.empty(n_cand, dtype=torch.int64, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        check(lib.brSortUniqueKeys64(keys.data_ptr(), n_cand, uniq.data_ptr(), cnt.data_ptr(), ws.data_ptr(), wsb, _stream()), "brSortUniqueKeys64")
        n_u = int(cnt.item())
        if n_u and int(uniq[n_u - 1].item()) == -1:      # the ~0 key of the rejected candidates sorts last
            n_u -= 1
        if n_u >= size:
            ou, oi = torch.empty(size, dtype=u.dtype, device=dev), torch.empty(size, dtype=u.dtype, device=dev)
            check(lib.brGatherPermutedPairs(uniq.data_ptr(), n_u, size, int(num_items), int(seed), ty, ou.data_ptr(), oi.data_ptr(), _stream()),
                  "brGatherPermutedPairs")
            return ou, oi
        n_cand = int(n_cand * 1.6) + n        # more rounds of column shuffles
    raise RuntimeError("ncf_negatives: not enough distinct negatives (is the interaction matrix nearly full?)")
