"""Row-sharded embedding tables over the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference only mirrors every variable on every worker (MultiWorkerMirroredStrategy,
src/models/RModel.py:119: all-reduce of dense AND whole-table gradients).  MI355X-first
(SURVEY.md §8e): each table is row-sharded, owner(r) = r mod W, local row = r div W (balances
Zipf heads); per step and id stream
   all-to-all #1  ids      requester -> owner   (B/W * 4 B per stream: latency-bound)
   all-to-all #2  rows     owner -> requester   (direct xGMI links, all 7 peers concurrently)
   all-to-all #3  row grads requester -> owner  (same volume), owner dedups + runs Adam locally
and one small all-reduce of the flat dense gradient (~75 KB) plus the BatchNorm column sums.

`ShardExchange` is backend-agnostic index plumbing (torch index ops + torch.distributed); the
row gather / optimizer compute is injected, so the product path passes the HIP ops and the
CPU tests (gloo, world_size 2) pass the oracle.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


class DistCtx:
    """Thin wrapper over a torch.distributed process group ("nccl" == RCCL on ROCm)."""

    def __init__(self, group=None, force_collectives: bool = False):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        # a 1-rank group normally short-cuts every collective to a local copy; force_collectives issues them anyway
        # (rehearsal of the RCCL call paths on a one-GPU box: tests/test_gpu_nccl_world1.py, bench.py BR_BENCH_FORCE_SHARDED)
        self.local = self.world == 1 and not force_collectives

    def _stage(self, t):
        # gloo has no device-side collectives for every op: stage through the host there
        return self.backend == "gloo" and t.is_cuda

    def all_reduce_sum(self, t: torch.Tensor):
        if self.local:
            return t
        if self._stage(t):
            c = t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_to_all(self, out: torch.Tensor, inp: torch.Tensor, out_splits, in_splits):
        """Row-wise all_to_all_single (splits count rows of dim 0)."""
        if self.local:
            out.copy_(inp)
            return out
        if self._stage(inp):
            co, ci = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
            dist.all_to_all_single(co, ci, list(out_splits), list(in_splits), group=self.group)
            out.copy_(co)
        else:
            dist.all_to_all_single(out, inp.contiguous(), list(out_splits), list(in_splits), group=self.group)
        return out

    def all_to_all_equal(self, out: torch.Tensor, inp: torch.Tensor):
        """all_to_all_single with equal, static splits (dim 0 of both is a multiple of the world size): no split lists, no counts."""
        if self.local:
            out.copy_(inp)
        elif self._stage(inp):
            co = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(co, inp.cpu(), group=self.group)
            out.copy_(co)
        else:
            dist.all_to_all_single(out, inp, group=self.group)
        return out

    def all_gather_rows(self, inp: torch.Tensor) -> torch.Tensor:
        out = torch.empty((self.world * inp.shape[0],) + tuple(inp.shape[1:]), dtype=inp.dtype, device=inp.device)
        if self.local:
            out.copy_(inp)
        elif self._stage(inp):
            co = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(co, inp.cpu(), group=self.group)
            out.copy_(co)
        else:
            dist.all_gather_into_tensor(out, inp.contiguous(), group=self.group)
        return out

    def barrier(self):
        if not self.local:
            dist.barrier(group=self.group)


def shard_rows(total_rows: int, rank: int, world: int) -> int:
    """rows r with r % world == rank."""
    return (total_rows - rank + world - 1) // world if total_rows > rank else 0


class ShardExchange:
    """One id stream of one step: who owns what, in which order rows travel.

    bucket order = batch positions stably sorted by owner; `order[j]` is the batch position of
    bucket slot j, `inv[b]` the bucket slot of batch position b.
    """

    def __init__(self, ctx: DistCtx):
        self.ctx = ctx

    def plan(self, ids: torch.Tensor):
        if ids.is_cuda:
            return ShardExchange.plan_pair(self, ids)
        # host tensors (CPU unit tests of the exchange logic): the same plan with torch ops
        W = self.ctx.world
        dest = (ids % W).to(torch.int64)
        _, order = torch.sort(dest, stable=True)
        # rows per destination; NOT torch.bincount: it sizes its output from a device max => a hidden host sync per call
        counts = (dest.unsqueeze(1) == torch.arange(W, device=dest.device).unsqueeze(0)).sum(0)
        self.order = order
        self.inv = torch.empty_like(order)
        self.inv[order] = torch.arange(order.numel(), device=order.device, dtype=order.dtype)
        self.send_local = torch.div(ids, W, rounding_mode="floor")[order].contiguous()
        self.send_counts_t = counts
        return self

    def _scratch(self, n, ids):
        """device scratch of the planning kernels, grown on demand"""
        if getattr(self, "_cap", -1) < n or self._dtype != ids.dtype:
            from . import _lib
            cap = int(n * 1.25) + 64
            dev, W = ids.device, self.ctx.world
            e = lambda dt: torch.empty(cap, dtype=dt, device=dev)
            self._dest, self._sdest, self._send = e(ids.dtype), e(ids.dtype), e(ids.dtype)
            self._order, self._inv = e(torch.int32), e(torch.int32)
            self._counts = torch.zeros(W, dtype=torch.int64, device=dev)
            self._ws_bytes = int(_lib.load().brRowIndexWorkspaceBytes(cap, 1 if ids.dtype == torch.int64 else 0))
            self._ws = torch.empty(self._ws_bytes, dtype=torch.uint8, device=dev)
            self._cap, self._dtype = cap, ids.dtype

    @staticmethod
    def plan_pair(xa: "ShardExchange", ids_a: torch.Tensor, xb: "ShardExchange" = None, ids_b: torch.Tensor = None):
        """Plan one stream, or two equally long ones in shared launches (brShardPlanPair: 4 launches instead of ~10 torch ops
        per stream)."""
        from . import _lib, ops
        n, W = ids_a.shape[0], xa.ctx.world
        if xb is not None and ids_b.shape[0] != n:
            xa.plan(ids_a); xb.plan(ids_b)
            return xa, xb
        for x, ids in ((xa, ids_a),) + (((xb, ids_b),) if xb is not None else ()):
            if not ids.is_contiguous() or ids.dtype not in (torch.int32, torch.int64):
                raise TypeError("ids must be contiguous int32 / int64 device tensors")
            x._scratch(n, ids)
        P = lambda t: t.data_ptr()
        b = xb if xb is not None else xa
        wsb = min(xa._ws_bytes, b._ws_bytes)
        _lib.check(_lib.load().brShardPlanPair(P(ids_a), P(ids_b) if xb is not None else None, ops.I64 if ids_a.dtype == torch.int64 else ops.I32, n, W,
                                               P(xa._dest), P(b._dest), P(xa._sdest), P(b._sdest), P(xa._order), P(b._order), P(xa._ws), P(b._ws), wsb,
                                               P(xa._inv), P(b._inv), P(xa._send), P(b._send), P(xa._counts), P(b._counts), ops._stream()),
                   "brShardPlanPair")
        for x in (xa,) + ((xb,) if xb is not None else ()):
            x.order, x.inv, x.send_local, x.send_counts_t = x._order[:n], x._inv[:n], x._send[:n], x._counts
        return (xa, xb) if xb is not None else xa

    def exchange_counts(self, *others: "ShardExchange"):
        """ONE host sync for all streams of the step: send/recv row counts per peer."""
        ctx, W = self.ctx, self.ctx.world
        plans = (self,) + others
        S = len(plans)
        sc = torch.stack([p.send_counts_t for p in plans]).to(torch.int64)          # (S, W)
        if not ctx.local:
            # peer-major layout (inp[d*S + s] = sc[s][d]) so ONE all_to_all moves every stream's count
            out = torch.empty(W * S, dtype=torch.int64, device=sc.device)
            ctx.all_to_all(out, sc.t().contiguous().view(-1), [S] * W, [S] * W)
            rc = out.view(W, S).t().contiguous()                                    # rc[s][src]
        else:
            rc = sc.clone()
        sc_h, rc_h = sc.cpu().tolist(), rc.cpu().tolist()
        for p, s, r in zip(plans, sc_h, rc_h):
            p.send_counts, p.recv_counts = s, r
            p.n_recv = int(sum(r))
        return self

    def send_ids(self) -> torch.Tensor:
        """all-to-all #1: local row ids to their owners -> ids this rank must serve."""
        out = torch.empty(self.n_recv, dtype=self.send_local.dtype, device=self.send_local.device)
        self.recv_local = self.ctx.all_to_all(out, self.send_local, self.recv_counts, self.send_counts)
        return self.recv_local

    def return_rows(self, rows: torch.Tensor) -> torch.Tensor:
        """all-to-all #2: rows served (n_recv, D) -> rows requested, in bucket order (B, D)."""
        out = torch.empty((self.order.numel(), rows.shape[1]), dtype=rows.dtype, device=rows.device)
        return self.ctx.all_to_all(out, rows, self.send_counts, self.recv_counts)

    def send_row_grads(self, grads_bucket_order: torch.Tensor) -> torch.Tensor:
        """all-to-all #3: per-pair row gradients (bucket order) -> owners, aligned with recv_local."""
        out = torch.empty((self.n_recv, grads_bucket_order.shape[1]), dtype=grads_bucket_order.dtype,
                          device=grads_bucket_order.device)
        return self.ctx.all_to_all(out, grads_bucket_order, self.recv_counts, self.send_counts)


class PaddedExchange:
    """Both id streams of a NeuMF step through ONE all-to-all per phase, duplicate ids merged, a FIXED number of slots per peer
    (brShardDedupPlanPair): ids out, rows back, row gradients out = 3 collectives per step (+ the dense all-reduce), every split equal
    and static, every buffer allocated once, nothing about the step read back by the host - the launches and collectives of a step
    are enqueued without a sync and can be captured into a hipGraph (the exact exchange above needs the per-peer row counts on the
    host before it can post its first all-to-all).
    Layout of every exchanged buffer: [peer][stream: user | item][cap] slots (ids: one id per slot; rows / gradients: `dim` floats
    per slot).  An owner serves each DISTINCT id of a batch once (a Zipf batch with one id on thousands of positions costs one slot);
    the requester sums the row gradients of an id's positions (ordered, two-level) into its slot before they travel.  Pad slots
    carry the owner's spare table row (gradient 0).  More than `cap` distinct ids of one stream for one owner: the surplus ids get
    no slot - the step reads zeros for them and sends no gradient, nothing collides - and a device flag that the engine's
    check_ids() turns into an error (cap = 1.25 x batch / world: uniform ids at batch 65 536 / 8 ranks sit 20 sigma below it, skewed
    ids further still; only a batch whose DISTINCT ids crowd onto one owner can overflow)."""

    def __init__(self, ctx: DistCtx, max_batch: int, id_dtype, device, total_rows, dim: int, factor: float = 1.25):
        from . import _lib, ops
        W = ctx.world
        self.ctx, self.W, self.dim, self.id_dtype, self.total_rows = ctx, W, int(dim), id_dtype, tuple(int(r) for r in total_rows)
        cap = max_batch if W == 1 else min(max_batch, (int(max_batch / W * factor) + 64) // 64 * 64)
        self.cap, self.n_slots = cap, W * cap                 # slots per stream; the merged buffers hold 2 * n_slots
        self.id_type = ops.I64 if id_dtype == torch.int64 else ops.I32
        e = lambda n, dt: torch.empty(n, dtype=dt, device=device)
        B, S = max_batch, 2 * self.n_slots
        self.keys, self.skeys = [e(B, id_dtype) for _ in range(2)], [e(B, id_dtype) for _ in range(2)]
        self.spos, self.urank, self.slot = ([e(B, torch.int32) for _ in range(2)] for _ in range(3))
        self.first = [torch.zeros(W + 1, dtype=torch.int32, device=device) for _ in range(2)]
        self.ws_bytes = int(_lib.load().brRowIndexWorkspaceBytes(B, self.id_type))
        self.ws = [e(self.ws_bytes, torch.uint8) for _ in range(2)]
        self.seg_ws = [e(int(_lib.load().brSegmentScratchFloats(B, self.dim)), torch.float32) for _ in range(2)]
        self.send_ids_buf, self.recv_ids = e(S, id_dtype), e(S, id_dtype)
        f = lambda: torch.empty(S, self.dim, dtype=torch.float32, device=device)
        self.served, self.rows, self.gpad, self.grecv = f(), f(), f(), f()
        self.seg = (cap, 2 * cap, 0, cap)                     # (seg_len, seg_stride, offset of the user half, of the item half): common.h seg_phys

    def plan(self, ids_a, ids_b, err_flag):
        from . import _lib, ops
        n = ids_a.shape[0]
        if ids_b.shape[0] != n or n > self.keys[0].shape[0]:
            raise ValueError("PaddedExchange.plan: both streams must have the same length <= max_batch")
        P = lambda t: t.data_ptr()
        _lib.check(_lib.load().brShardDedupPlanPair(
            P(ids_a), P(ids_b), self.id_type, n, self.W, self.cap, self.total_rows[0], self.total_rows[1], P(self.keys[0]), P(self.keys[1]),
            P(self.skeys[0]), P(self.skeys[1]), P(self.spos[0]), P(self.spos[1]), P(self.ws[0]), P(self.ws[1]), self.ws_bytes, P(self.urank[0]), P(self.urank[1]),
            P(self.first[0]), P(self.first[1]), P(self.send_ids_buf), P(self.slot[0]), P(self.slot[1]), P(self.gpad), self.dim, P(err_flag), ops._stream()),
            "brShardDedupPlanPair")
        self.n = n
        return self

    def send_ids(self):
        """all-to-all #1: every owner's distinct local row ids of both streams -> the ids this rank serves ([source][stream][cap])."""
        return self.ctx.all_to_all_equal(self.recv_ids, self.send_ids_buf)

    def return_rows(self):
        """all-to-all #2: self.served (filled by the owner-side gather, slot for slot) -> self.rows in the requester's slot numbering."""
        return self.ctx.all_to_all_equal(self.rows, self.served)

    def send_row_grads(self, g0_a, g1_a, g0_b, g1_b, ldg, hi_scale, split):
        """the per-id sums of the step's row gradients into the send slots (brSegmentSumToSlotsPair: columns [0, split) from g0, the
        rest hi_scale[position] * g1 - the NeuMF step passes the MLP halves of dx0, the partner streams' stashed MF rows and ddot),
        then all-to-all #3 -> self.grecv on the owners, slot for slot with recv_ids."""
        from . import _lib, ops
        P = lambda t: 0 if t is None else t.data_ptr()
        _lib.check(_lib.load().brSegmentSumToSlotsPair(P(self.skeys[0]), P(self.spos[0]), P(self.slot[0]), P(g0_a), P(g1_a), P(self.skeys[1]), P(self.spos[1]),
                                                       P(self.slot[1]), P(g0_b), P(g1_b), ldg, ldg, P(hi_scale), self.id_type, self.n, self.dim, split, P(self.gpad),
                                                       P(self.seg_ws[0]), P(self.seg_ws[1]), ops._stream()), "brSegmentSumToSlotsPair")
        return self.ctx.all_to_all_equal(self.grecv, self.gpad)


# ---------------------------------------------------------------------------------------------------------------------
# Checkpoint / restore of row-sharded engines.  The reference writes ONE SavedModel: the chief to
# checkpoints/<model>/cp, every other worker to a temporary cp/workertemp_<id> that it deletes again, because all its variables
# are mirrored (src/models/RModel.py:139,175-196).  Here every rank OWNS different table rows, so every rank writes its shard:
#   <path>.shard<rank>-of-<world>.pt   state_dict() of the rank's engine (row r of a table = global row r * world + rank)
#   <path>.meta.json                   written by rank 0: world size, names of the row-sharded entries
# Restore accepts any world size: with the same one a rank reads its own file, otherwise every saved shard is read and the
# global rows are dealt out again (owner = row mod new world; also world 1 = the single-GPU engines).
# ---------------------------------------------------------------------------------------------------------------------
def _shard_file(path, rank, world):
    return f"{path}.shard{rank:03d}-of-{world:03d}.pt"


def save_sharded(engine, path, ctx: "DistCtx", sharded_keys):
    """sharded_keys: state_dict entries whose dim 0 is row-sharded (tables and their optimizer slots)."""
    import json
    import os
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    sd = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in engine.state_dict().items()}
    torch.save(sd, _shard_file(path, ctx.rank, ctx.world))
    if ctx.rank == 0:
        with open(path + ".meta.json", "w") as f:
            json.dump({"world": ctx.world, "sharded_keys": list(sharded_keys)}, f)
    ctx.barrier()


def load_sharded(engine, path, rank: int, world: int, global_rows: dict):
    """global_rows: {sharded key: rows of the GLOBAL table}.  Works for any saved world size."""
    import json
    with open(path + ".meta.json") as f:
        meta = json.load(f)
    W0, keys = int(meta["world"]), set(meta["sharded_keys"])

    def fit(sd):
        """row-sharded entries cut / zero-padded to the engine's row count (an engine with the padded exchange has a spare row)"""
        have = engine.state_dict()
        for k in keys:
            want = have[k].shape[0]
            if sd[k].shape[0] > want:
                sd[k] = sd[k][:want]
            elif sd[k].shape[0] < want:
                sd[k] = torch.cat([sd[k], torch.zeros((want - sd[k].shape[0],) + tuple(sd[k].shape[1:]), dtype=sd[k].dtype)])
        return sd

    if W0 == world:
        engine.load_state_dict(fit(torch.load(_shard_file(path, rank, world), map_location="cpu", weights_only=True)))
        return
    shards = [torch.load(_shard_file(path, r, W0), map_location="cpu", weights_only=True) for r in range(W0)]
    sd = {}
    for k, v in shards[0].items():
        if k not in keys:
            sd[k] = v                                     # replicated (dense parameters, counters)
            continue
        rows = int(global_rows[k])
        full = torch.zeros((rows,) + tuple(v.shape[1:]), dtype=v.dtype)
        for r in range(W0):
            n = shard_rows(rows, r, W0)
            if n:
                full[r::W0] = shards[r][k][:n]
        mine = full[rank::world]
        sd[k] = mine if mine.shape[0] else torch.zeros((1,) + tuple(v.shape[1:]), dtype=v.dtype)
    engine.load_state_dict(fit(sd))


def make_sharded_engine(base_cls):
    """ShardedNeuMFEngine = NeuMFEngine with its embed / table-optimizer hooks replaced by the
    row-sharded exchange.  (Factory so this module stays importable without the HIP library.)"""
    from . import ops
    from .neumf import TABLES

    class ShardedNeuMFEngine(base_cls):
        sharded = True

        def __init__(self, cfg, num_user_rows, num_item_rows, device, max_batch, ctx: DistCtx, **kw):
            self.ctx = ctx
            self.full_tables = kw.pop("full_tables", None)   # tests: {name: full (rows, D) tensor} to slice
            # "padded" (default): fixed per-peer capacity, no host sync in the step (PaddedExchange); "exact": row counts over the host
            self.exchange = kw.pop("exchange", "padded")
            self.exchange_capacity = float(kw.pop("exchange_capacity", 1.25))
            if self.exchange not in ("padded", "exact"):
                raise ValueError("exchange must be 'padded' or 'exact'")
            super().__init__(cfg, num_user_rows, num_item_rows, device, max_batch, dist=ctx, **kw)
            self.xu, self.xi = ShardExchange(ctx), ShardExchange(ctx)
            if self.exchange == "padded":
                self.px = PaddedExchange(ctx, max_batch, self.id_dtype, self.device, (num_user_rows, num_item_rows), 2 * cfg.dim, self.exchange_capacity)
                self._grow_index(self.px.n_slots)

        def owned_rows(self, name):
            total = self.num_user_rows if name.startswith("user") else self.num_item_rows
            return shard_rows(total, self.ctx.rank, self.ctx.world)

        def local_rows(self, name):
            """rows of this rank's table tensors: the rows it owns (+ the spare row the pad slots of the padded exchange point at)"""
            return max(1, self.owned_rows(name)) + (1 if self.exchange == "padded" else 0)

        def _init_tables(self, g, init_seed):
            if self.full_tables is not None:     # tests: slice the rows this rank owns out of global tables
                D = self.cfg.dim
                self.fused = {}
                for stream in ("user", "item"):
                    full = torch.cat([self.full_tables[stream + "_mlp"], self.full_tables[stream + "_mf"]], dim=1)
                    t = full[self.ctx.rank::self.ctx.world].contiguous().to(self.device)
                    t = t if t.shape[0] else torch.zeros(1, 2 * D, device=self.device)
                    if self.exchange == "padded":
                        t = torch.cat([t, torch.zeros(1, 2 * D, device=self.device)])
                    self.fused[stream] = t
                self._make_views()
                return
            super()._init_tables(g, init_seed + 104729 * self.ctx.rank)

        def _alloc_sparse(self, B):
            # an owner can receive more than B ids in a step; indexes are (re)grown on demand
            self._idx_cap = 0
            self._grow_index(2 * B)
            if self.deferred:
                self.last = {k: torch.zeros(self.local_rows(k + "_mf"), dtype=torch.int32, device=self.device) for k in ("user", "item")}
            elif self.cfg.optimizer == "adam_dense":
                self.user_mark = torch.zeros(self.local_rows("user_mf"), dtype=torch.uint8, device=self.device)
                self.item_mark = torch.zeros(self.local_rows("item_mf"), dtype=torch.uint8, device=self.device)

        def _grow_index(self, n):
            if n > self._idx_cap:
                self._idx_cap = int(n * 1.25) + 64
                self.user_index = ops.RowIndex(self._idx_cap, self.id_dtype, self.device)
                self.item_index = ops.RowIndex(self._idx_cap, self.id_dtype, self.device)
                if hasattr(self, "step_struct"):
                    self._bind_indexes(self.step_struct)

        def _embed_forward(self, users, items, B):
            D = self.cfg.dim
            if self.exchange == "padded":
                x, cfg = self.px, self.cfg
                x.plan(users, items, self.err)
                rid = x.send_ids()                                 # all-to-all #1 (both streams)
                S, (sl, ss_, ou_, oi_) = x.n_slots, x.seg
                # the owner-side dedup indexes depend only on the ids just received: both in shared launches on a side stream,
                # beside the lookup / MLP, joined before the Adam-rows kernels
                main = torch.cuda.current_stream(self.device)
                if getattr(self, "_side", None) is None:
                    self._side, self._ev_ids, self._ev_index = torch.cuda.Stream(device=self.device), torch.cuda.Event(), torch.cuda.Event()
                self._ev_ids.record(main)
                self._side.wait_event(self._ev_ids)
                with torch.cuda.stream(self._side):
                    # every source's segment arrives sorted (distinct local rows in key order, pads = the spare row last): the index is a
                    # merge of W sorted runs, no sort (BR_MERGE_INDEX=0: the chunk-sort build, A/B)
                    if os.environ.get("BR_MERGE_INDEX", "1") != "0":
                        ops.row_index_merge_pair_seg(self.user_index, self.local_rows("user_mf"), self.item_index, self.local_rows("item_mf"), rid, S, x.seg, self.err)
                    else:
                        ops.row_index_build_pair_seg(self.user_index, self.local_rows("user_mf"), self.item_index, self.local_rows("item_mf"), rid, S, x.seg)
                    self._ev_index.record(self._side)
                # owner-side G1 on the fused [mlp | mf] rows (512 B at dim 64): both shards, one launch, straight into the slots
                if self.deferred and 2 * D in (64, 128, 256):
                    ops.gather_rows_deferred_pair_seg(self.fused["user"], self.fused_m["user"], self.fused_v["user"], self.last["user"],
                                                      self.fused["item"], self.fused_m["item"], self.fused_v["item"], self.last["item"], rid, x.served, S, x.seg,
                                                      self.step_state, cfg.beta1, cfg.beta2, cfg.adam_eps, err_flag=self.err)
                elif self.deferred:      # other row widths: the row-group gather per stream, through contiguous copies of the stream's slots
                    W, cap = x.W, x.cap
                    for k, stream in enumerate(("user", "item")):
                        ids_k = rid.view(W, 2, cap)[:, k].contiguous().view(-1)
                        rows_k = self._serve_rows(stream, ids_k)
                        x.served.view(W, 2, cap, 2 * D)[:, k].copy_(rows_k.view(W, cap, 2 * D))
                else:
                    ops.gather_rows_pair_seg(self.fused["user"], self.fused["item"], rid, x.served, S, x.seg, err_flag=self.err)
                rows = x.return_rows()                             # all-to-all #2 (both streams)
                self.pos_u, self.pos_i = x.slot[0][:B], x.slot[1][:B]
                if self.id_dtype != torch.int32:
                    self.pos_u, self.pos_i = self.pos_u.to(self.id_dtype), self.pos_i.to(self.id_dtype)
                # requester side: the fused embed kernel with the received slots as its "tables" (ids = slots; an id without a slot reads
                # zeros - flagged by the plan, not here); the MF rows of every pair are stashed by position for the backward
                ops.neumf_embed_forward(rows[:, :D], rows[:, :D], rows[:, D:], rows[:, D:], self.pos_u, self.pos_i, self.cfg.item_first,
                                        self.x0[:B], self.dot[:B], None, stash=(self.g_user[:B, D:], self.g_item[:B, D:]))
                return
            xu, xi = ShardExchange.plan_pair(self.xu, users, self.xi, items)
            xu.exchange_counts(xi)                     # the step's one host sync (variable split sizes)
            ru, ri = xu.send_ids(), xi.send_ids()      # all-to-all #1
            empty = torch.empty(0, 2 * D, device=self.device)
            # owner-side G1 on the fused [mlp | mf] rows (512 B at dim 64)
            gu = self._serve_rows("user", ru) if ru.numel() else empty
            gi = self._serve_rows("item", ri) if ri.numel() else empty
            self._served = {"user": gu, "item": gi}    # kept for the optimizer: this rank's rows replayed to step t-1, by received slot
            self.r_user, self.r_item = xu.return_rows(gu), xi.return_rows(gi)      # all-to-all #2
            self.pos_u = xu.inv.to(self.id_dtype)
            self.pos_i = xi.inv.to(self.id_dtype)
            # requester-side: same fused embed kernel, "tables" = received rows, ids = bucket slots
            ops.neumf_embed_forward(self.r_user[:, :D], self.r_item[:, :D], self.r_user[:, D:], self.r_item[:, D:], self.pos_u,
                                    self.pos_i, self.cfg.item_first, self.x0[:B], self.dot[:B], self.err)

        # ------------------------------------------------------------------ hipGraph replay of the sharded step
        def enable_graph(self, batch: int | None = None):
            """Replay the row-sharded step - its ~25 launches AND its four RCCL collectives - as ONE hipGraph per step for local batches
            of exactly `batch` pairs.  Needs the fixed-capacity exchange (no host sync in the step) and deferred tables (every per-step
            scalar in the device step state).  The first such step runs eagerly on the static input buffers (every kernel and every
            collective once outside a capture), then the step's body is captured; a runtime / backend that refuses the capture (gloo
            stages its collectives through the host) keeps the eager sequence - `graph_active` tells which."""
            if self.exchange != "padded" or not self.deferred:
                raise ValueError("graph replay of the sharded step needs exchange='padded' and optimizer='adam_dense' with dense_impl='deferred'")
            B = self.max_batch if batch is None else int(batch)
            if not 0 < B <= self.max_batch:
                raise ValueError("graph batch must be in (0, max_batch]")
            dev = self.device
            self.in_users = torch.zeros(B, dtype=self.id_dtype, device=dev)
            self.in_items = torch.zeros(B, dtype=self.id_dtype, device=dev)
            self.in_labels = torch.zeros(B, dtype=torch.float32, device=dev)
            self._sgraph = {"batch": B, "graph": None, "key": None, "refused": None}

        def disable_graph(self):
            self._sgraph = None

        @property
        def graph_active(self) -> bool:
            g = getattr(self, "_sgraph", None)
            return bool(g and g["graph"] is not None)

        def train_step(self, users, items, labels, row0: int = 0, batch_total: int | None = None):
            g = getattr(self, "_sgraph", None)
            B = users.shape[0]
            self._moving_dirty = True            # (per-replica BatchNorm: this rank's moving statistics move apart from the others')
            if g is None or B != g["batch"]:
                return super().train_step(users, items, labels, row0=row0, batch_total=batch_total)
            from . import _lib
            bt = B if batch_total is None else batch_total
            self._check_batch(users, items, labels)
            _lib.check(_lib.load().brStageBatch(self.in_users.data_ptr(), self.in_items.data_ptr(), self.in_labels.data_ptr(), users.data_ptr(), items.data_ptr(),
                                                labels.data_ptr(), self.step_struct.id_type, B, ops._stream()), "brStageBatch")
            if g["graph"] is not None and g["key"] == (row0, bt):
                if self.t + 1 - self._flush_t >= self.ALPHA_RING - 8:
                    self.flush()
                self._stale = True
                self.t += 1
                g["graph"].replay()
                return
            super().train_step(self.in_users, self.in_items, self.in_labels, row0=row0, batch_total=bt)      # eager: also the warm-up of the capture
            if g["graph"] is None and g["refused"] is None:
                self._capture_step(g, B, row0, bt)

        def _capture_step(self, g, B, row0, bt):
            ctx = self.ctx
            if not ctx.local and ctx.backend != "nccl":
                g["refused"] = f"backend {ctx.backend}: collectives are staged through the host"
                return
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            try:
                self._set_batch(self.in_users, self.in_items, self.in_labels, B, True, row0, bt)
                with torch.cuda.graph(graph):
                    self._dist_step(self.in_users, self.in_items, B)
                g["graph"], g["key"] = graph, (row0, bt)
            except Exception as exc:  # noqa: BLE001 - the runtime refused (a collective or a launch that cannot be captured): eager from here on
                g["refused"] = f"{type(exc).__name__}: {exc}"
                try:
                    torch.cuda.synchronize(self.device)
                except Exception:  # noqa: BLE001
                    pass

        SHARDED_KEYS = ("table.user", "table.user.m", "table.user.v", "table.item", "table.item.m", "table.item.v")

        def save_sharded(self, path):
            """model.save for the row-sharded engine (every rank writes its shard; see save_sharded above)."""
            save_sharded(self, path, self.ctx, self.SHARDED_KEYS)

        def load_sharded(self, path):
            rows = {k: (self.num_user_rows if ".user" in k else self.num_item_rows) for k in self.SHARDED_KEYS}
            load_sharded(self, path, self.ctx.rank, self.ctx.world, rows)

        def state_dict(self):
            """Per-replica BatchNorm (sync_bn=False) updates the moving statistics from each rank's own batch: reconcile them first
            (MirroredStrategy keeps them as ON_READ variables aggregated by MEAN [TF-sem]), so every rank saves / predicts the same."""
            self.sync_moving_stats()
            return super().state_dict()

        def sync_moving_stats(self):
            """a COLLECTIVE when per-replica BatchNorm has trained since the last call (every rank must then reach it: state_dict(),
            save_sharded(), predict() and evaluate_batch() of a sharded engine are collectives anyway); a no-op otherwise, so repeated
            reads between two training steps cost nothing and cannot hang a lone reader."""
            if self.ctx.world > 1 and not self.cfg.sync_bn and getattr(self, "_moving_dirty", False):
                self.ctx.all_reduce_sum(self.moving_buf)
                self.moving_buf.div_(self.ctx.world)
            self._moving_dirty = False

        def _infer(self, users, items, labels, n):
            self.sync_moving_stats()
            return super()._infer(users, items, labels, n)

        def _serve_rows(self, stream, local_ids, out=None):
            """owner side of the lookup: rows of this rank's shard for the ids its peers asked for."""
            if self.deferred:      # rows as of the previous step, replayed in registers (binrec.h "Deferred dense Adam")
                cfg = self.cfg
                return ops.gather_rows_deferred(self.fused[stream], self.fused_m[stream], self.fused_v[stream], self.last[stream], local_ids,
                                                self.step_state, cfg.beta1, cfg.beta2, cfg.adam_eps, out=out, err_flag=self.err)
            return ops.gather_rows([self.fused[stream]], [local_ids], None if out is None else [out], err_flag=self.err)[0]

        def _embed_backward_apply(self, users, items, B):
            cfg, D = self.cfg, self.cfg.dim
            if self.exchange == "padded":
                # per-id sums of the row gradients [mlp half of dx0 | ddot * the partner's stashed MF row] into the send slots
                x = self.px
                uo, io = (D, 0) if cfg.item_first else (0, D)
                dx0 = self.dx0[:B]
                og = x.send_row_grads(dx0[:, uo:], self.g_item[:B, D:], dx0[:, io:], self.g_user[:B, D:], 2 * D, self.ddot[:B], D)      # + all-to-all #3
                torch.cuda.current_stream(self.device).wait_event(self._ev_index)
                # x.served still holds the rows this rank served for these very slots = its rows replayed to step t-1; the indexes carry
                # physical slot numbers, so both tables read the one received buffer
                self._adam_tables({"user": (og, 2 * D, None, 0), "item": (og, 2 * D, None, 0)},
                                  replayed={"user": x.served, "item": x.served} if self.deferred else None)
                return
            gu, gi = self.g_user[:B], self.g_item[:B]
            # fused per-pair row gradients [mlp | mf] (the MLP halves are copied out of dx0 here)
            ops.neumf_embed_backward(self.r_user[:, D:], self.r_item[:, D:], self.pos_u, self.pos_i, cfg.item_first,
                                     self.dx0[:B], self.ddot[:B], gu[:, D:], gi[:, D:], gu[:, :D], gi[:, :D])
            xu, xi = self.xu, self.xi
            # batch order -> bucket order, then all-to-all #3 to the owners
            bu = ops.gather_rows([gu], [xu.order.to(self.id_dtype)])[0]
            bi = ops.gather_rows([gi], [xi.order.to(self.id_dtype)])[0]
            ou, oi = xu.send_row_grads(bu), xi.send_row_grads(bi)
            self._grow_index(max(xu.n_recv, xi.n_recv))
            self.user_index.build(xu.recv_local, self.local_rows("user_mf"))
            self.item_index.build(xi.recv_local, self.local_rows("item_mf"))
            rep = getattr(self, "_served", None) if self.deferred else None
            if rep is not None and (rep["user"].shape[0] != xu.n_recv or rep["item"].shape[0] != xi.n_recv or not xu.n_recv or not xi.n_recv):
                rep = None
            self._adam_tables({"user": (ou, 2 * D, None, 0), "item": (oi, 2 * D, None, 0)}, replayed=rep)
            self._served = None

    return ShardedNeuMFEngine


def make_sharded_two_tower(base_cls):
    """ShardedTwoTowerEngine = trainers/twoTower.py on W GPUs (BASELINE config 4): embedding tables
    row-sharded (owner = id mod W, all-to-all ids / rows / row grads as for NeuMF), towers replicated
    (dense gradients all-reduced), and GLOBAL in-batch negatives: every rank scores its B queries against
    the all-gathered W*B candidates (streaming LSE, never materialised), and its B candidates against
    the all-gathered queries for dC — no reduce-scatter needed."""
    from . import ops

    class ShardedTwoTowerEngine(base_cls):
        def __init__(self, embed_dim, nbr_item, nbr_user, semb, device, max_batch, ctx: DistCtx, full_tables=None, **kw):
            self.ctx, self._full_tables = ctx, full_tables
            self.user_rows_global, self.item_rows_global = nbr_user + 2, nbr_item + 2
            super().__init__(embed_dim, nbr_item, nbr_user, semb, device, max_batch, **kw)     # tables and slots: this rank's shard only
            self.dist = ctx
            self._full_tables = None
            self.xu, self.xi = ShardExchange(ctx), ShardExchange(ctx)
            cap = int(2.5 * self.max_batch) + 64
            self.user_index, self.item_index = ops.RowIndex(cap, self.id_dtype, self.device), ops.RowIndex(cap, self.id_dtype, self.device)
            self._idx_cap = cap

        def _init_tables(self, user_rows, item_rows, g):
            """only the rows this rank owns (r mod W == rank): a slice of `full_tables` (tests), else drawn with a rank-offset
            seed - peak memory per GPU is the shard, not the replicated model"""
            r, W = self.ctx.rank, self.ctx.world
            gl = torch.Generator(device="cpu").manual_seed(int(g.initial_seed()) + 104729 * (r + 1))
            for name, rows in (("user_emb", user_rows), ("item_emb", item_rows)):
                if self._full_tables is not None:
                    shard = self._full_tables[name][r::W].contiguous().to(self.device)
                else:
                    shard = (torch.rand(shard_rows(rows, r, W), self.E, generator=gl) * 0.1 - 0.05).to(self.device)
                setattr(self, name, shard if shard.shape[0] else torch.zeros(1, self.E, device=self.device))

        SHARDED_KEYS = ("user_emb", "item_emb", "user_acc", "item_acc", "user_v", "item_v")

        def _start_indexes(self, users, items):
            pass                                    # the owners index the ids they receive (_apply_tables)

        def save_sharded(self, path):
            save_sharded(self, path, self.ctx, [k for k in self.SHARDED_KEYS if k in self.state_dict()])

        def load_sharded(self, path):
            rows = {k: (self.user_rows_global if k.startswith("user") else self.item_rows_global) for k in self.SHARDED_KEYS}
            load_sharded(self, path, self.ctx.rank, self.ctx.world, rows)

        def _lookup(self, users, items, B):
            xu, xi = ShardExchange.plan_pair(self.xu, users, self.xi, items)
            xu.exchange_counts(xi)
            ru, ri = xu.send_ids(), xi.send_ids()
            E = self.E
            empty = torch.empty(0, E, device=self.device)
            gu = ops.gather_rows([self.user_emb], [ru], err_flag=self.err)[0] if ru.numel() else empty
            gi = ops.gather_rows([self.item_emb], [ri], err_flag=self.err)[0] if ri.numel() else empty
            bu, bi = xu.return_rows(gu), xi.return_rows(gi)                      # bucket order
            ops.gather_rows([bu, bi], [xu.inv.to(self.id_dtype), xi.inv.to(self.id_dtype)], [self.eu[:B], self.ei[:B]])

        def _softmax(self, q, c, items, B, dq, dc):
            ctx = self.ctx
            off = ctx.rank * B
            c_all, ids_all = ctx.all_gather_rows(c.contiguous()), ctx.all_gather_rows(items.contiguous())
            if dq is None:
                ops.inbatch_softmax_lse(q, c_all, items, ids_all, off, self.lse[:B], self.loss_slots)
                return
            ops.inbatch_softmax_lse_grad_q(q, c_all, items, ids_all, off, self.lse[:B], self.loss_slots, dq)     # lse + loss + dQ in one sweep
            q_all, lse_all = ctx.all_gather_rows(q.contiguous()), ctx.all_gather_rows(self.lse[:B].contiguous())
            # dC of MY candidates against ALL queries: query g's positive is my candidate g - rank*B
            ops.inbatch_softmax_grad(q_all, c.contiguous(), ids_all, items, -off, lse_all, None, dc)

        def _apply_tables(self, users, items, B):
            xu, xi = self.xu, self.xi
            bu = ops.gather_rows([self.deu[:B]], [xu.order.to(self.id_dtype)])[0]
            bi = ops.gather_rows([self.dei[:B]], [xi.order.to(self.id_dtype)])[0]
            ou, oi = xu.send_row_grads(bu), xi.send_row_grads(bi)
            if max(xu.n_recv, xi.n_recv) > self._idx_cap:
                self._idx_cap = int(1.25 * max(xu.n_recv, xi.n_recv)) + 64
                self.user_index, self.item_index = ops.RowIndex(self._idx_cap, self.id_dtype, self.device), ops.RowIndex(self._idx_cap, self.id_dtype, self.device)
            self.user_index.build(xu.recv_local, self.user_emb.shape[0])
            self.item_index.build(xi.recv_local, self.item_emb.shape[0])
            self._opt_rows(self.user_emb, self.user_acc, getattr(self, "user_v", None), self.user_index, ou)
            self._opt_rows(self.item_emb, self.item_acc, getattr(self, "item_v", None), self.item_index, oi)

    return ShardedTwoTowerEngine


def make_sharded_bpr(base_cls):
    """ShardedBPREngine = src/models/BPRModel.py on W GPUs: the user and the (shared) item table are row-sharded
    (owner = id mod W).  Per step the three id streams travel as two exchanges - users (B ids) and [pos | neg] (2B ids):
    ids -> owners, owner-side gather, rows back; the fused triplet kernel then runs on the received rows (positions as
    ids), and the per-triplet row gradients go back to the owners, which dedup (one sort per table over the
    contributions of all ranks) and apply Adam.  The loss is the mean over the GLOBAL batch (batch_total)."""
    from . import ops

    class ShardedBPREngine(base_cls):
        def __init__(self, num_users, num_items, num_factor, device, max_batch, ctx: DistCtx, full_tables=None, **kw):
            self.ctx, self._full_tables = ctx, full_tables
            self.num_users_global, self.num_items_global = int(num_users), int(num_items)
            super().__init__(num_users, num_items, num_factor, device, max_batch, **kw)      # tables, slots, marks: this rank's shard only
            self._full_tables = None
            self.xu, self.xi = ShardExchange(ctx), ShardExchange(ctx)
            B = self.max_batch
            self._idx_cap = int(5 * B) + 64
            self.user_index, self.item_index = ops.RowIndex(self._idx_cap, self.id_dtype, self.device), ops.RowIndex(self._idx_cap, self.id_dtype, self.device)
            self.pos_b = torch.arange(2 * B, device=self.device).to(self.id_dtype)

        def _init_tables(self, num_users, num_items, init_seed):
            """only the rows this rank owns: a slice of `full_tables` (tests), else drawn with a rank-offset seed"""
            r, W = self.ctx.rank, self.ctx.world
            g = torch.Generator(device="cpu").manual_seed(init_seed + 104729 * (r + 1))
            for name, rows in (("user", num_users), ("item", num_items)):
                if self._full_tables is not None:
                    shard = self._full_tables[name][r::W].contiguous().to(self.device)
                else:
                    shard = (torch.rand(shard_rows(rows, r, W), self.dim, generator=g) * 0.1 - 0.05).to(self.device)
                setattr(self, name, shard if shard.shape[0] else torch.zeros(1, self.dim, device=self.device))

        def train_step(self, users, pos, neg, batch_total: int | None = None):
            B = users.shape[0]
            if B > self.max_batch:
                raise ValueError("batch exceeds max_batch")
            ctx, D = self.ctx, self.dim
            if self.deferred:
                self._advance()                                      # device step counter / alpha ring (the owner-side replay reads them)
            self.t += 1
            bt = B * ctx.world if batch_total is None else batch_total
            ids2 = self.item_ids[:2 * B]
            ids2[:B].copy_(pos)
            ids2[B:].copy_(neg)
            xu, xi = self.xu.plan(users), self.xi.plan(ids2)
            xu.exchange_counts(xi)                                   # the step's one host sync
            ru, ri = xu.send_ids(), xi.send_ids()
            empty = torch.empty(0, D, device=self.device)
            hp = (self.BETA1, self.BETA2, self.EPS)

            def serve(tab, m, v, last, ids):
                """owner side of the lookup; deferred: rows as of the previous step, replayed in registers"""
                if not ids.numel():
                    return empty
                if self.deferred:
                    return ops.gather_rows_deferred(tab, m, v, last, ids, self.step_state, *hp, err_flag=self.err)
                return ops.gather_rows([tab], [ids], err_flag=self.err)[0]

            gu = serve(self._user, self.user_m, self.user_v, getattr(self, "user_last", None), ru)
            gi = serve(self._item, self.item_m, self.item_v, getattr(self, "item_last", None), ri)
            bu, bi = xu.return_rows(gu), xi.return_rows(gi)          # bucket order
            if B == 0:
                eu, ei = empty, empty
            else:
                eu = ops.gather_rows([bu], [xu.inv.to(self.id_dtype)])[0]            # batch order (B x D)
                ei = ops.gather_rows([bi], [xi.inv.to(self.id_dtype)])[0]            # [pos rows | neg rows] (2B x D)
                ar = self.pos_b[:B]
                ops.bpr_forward_backward(eu, ei, ar, ar, ar + B, 1.0 / bt, self.loss_slots, self.g_user[:B], self.g_item[:2 * B],
                                         self.per_triplet[:B], self.err)
            # row gradients: batch order -> bucket order -> owners
            gub = ops.gather_rows([self.g_user[:B]], [xu.order.to(self.id_dtype)])[0] if B else empty
            gib = ops.gather_rows([self.g_item[:2 * B]], [xi.order.to(self.id_dtype)])[0] if B else empty
            ou, oi = xu.send_row_grads(gub), xi.send_row_grads(gib)
            if max(xu.n_recv, xi.n_recv) > self._idx_cap:
                self._idx_cap = int(1.25 * max(xu.n_recv, xi.n_recv)) + 64
                self.user_index, self.item_index = ops.RowIndex(self._idx_cap, self.id_dtype, self.device), ops.RowIndex(self._idx_cap, self.id_dtype, self.device)
            a = ops.adam_alpha(self.lr, self.t)
            dense = self.optimizer == "adam_dense" and not self.deferred     # per-step sweep of the untouched rows
            for name, idx, ex, og, served in (("user", self.user_index, xu, ou, gu), ("item", self.item_index, xi, oi, gi)):
                tab, m, v = getattr(self, "_" + name), getattr(self, name + "_m"), getattr(self, name + "_v")
                mark = getattr(self, name + "_mark", None)
                if ex.n_recv:
                    idx.build(ex.recv_local, tab.shape[0])
                    if self.deferred:
                        # (the rows this rank served for these positions = its rows replayed to step t-1)
                        ops.adam_rows_sorted_deferred(tab, m, v, getattr(self, name + "_last"), idx, og, D, self.step_state, *hp,
                                                      replayed=served if served.shape[0] == ex.n_recv else None)
                    else:
                        ops.adam_rows_sorted(tab, m, v, idx, og, D, a, mark=mark if dense else None)
                if dense:
                    ops.adam_dense_sweep(tab, m, v, a, mark=mark)
            self.n_seen += B

        def _gather_global(self, name, ids):
            """rows `ids` (global ids, any owner) of the row-sharded table `name` -> (len(ids), dim) on this rank."""
            x = ShardExchange(self.ctx).plan(ids)
            x.exchange_counts()
            served = x.send_ids()
            empty = torch.empty(0, self.dim, device=self.device)
            rows = ops.gather_rows([getattr(self, name)], [served], err_flag=self.err)[0] if served.numel() else empty
            back = x.return_rows(rows)                                               # bucket order
            return ops.gather_rows([back], [x.inv.to(self.id_dtype)])[0] if ids.numel() else empty

        def predict_scores(self, user_ids, item_ids=None):
            """bpr_predict (src/models/bpr.py:122-133) on row-sharded tables: the user vectors (and the item rows, all of them when
            item_ids is None) come through the same id -> owner exchange as the training step; every rank must call this (it is a
            collective) and gets the scores of ITS user_ids."""
            u = self._gather_global("user", user_ids)
            if item_ids is None:
                item_ids = torch.arange(self.num_items_global, device=self.device).to(self.id_dtype)
            it = self._gather_global("item", item_ids)
            return ops.score_matrix(u, it)

        SHARDED_KEYS = ("user", "item", "user_m", "user_v", "item_m", "item_v")

        def save_sharded(self, path):
            save_sharded(self, path, self.ctx, self.SHARDED_KEYS)

        def load_sharded(self, path):
            rows = {k: (self.num_users_global if k.startswith("user") else self.num_items_global) for k in self.SHARDED_KEYS}
            load_sharded(self, path, self.ctx.rank, self.ctx.world, rows)

    return ShardedBPREngine
