"""BPR pairwise model on the HIP hot path — src/models/BPRModel.py:49-74,124-144 and the
stand-alone src/models/bpr.py:141-201.

Graph: shared item embedding (positive + negative lookups), user embedding, Lambda triplet loss
`1 - sigmoid(u.p - u.n)` (NOT -log sigmoid: BPRModel.py:144), identityLoss = mean (BPRModel.py:124-126),
Adam(1e-3) (BPRModel.py:70).  One fused launch does the 3 gathers, 2 dots, the loss partials and the
3 per-triplet row gradients; the shared item table receives its two IndexedSlices concatenated
[pos rows | neg rows] and deduplicated by one sort over the 2B ids.
"""
from __future__ import annotations

import os

import torch

from . import _lib, ops


class BPREngine:
    """optimizer "adam_dense" = Keras' non-lazy sparse Adam (every row of both tables moves every step [TF-sem]); dense_impl
    "deferred" (default) reaches the untouched rows by per-row replay (include/binrec.h "Deferred dense Adam": the lookup replays a
    row's missing g = 0 steps in registers, the optimizer launch applies them, flush() brings every row to the current step before
    anything else reads the tables) instead of sweeping 6 x 4 B per table element per step ("sweep"); bit-equal tables."""

    ALPHA_RING = _lib.parse_enums()["BR_ALPHA_RING"]
    BETA1, BETA2, EPS = 0.9, 0.999, 1e-7

    def __init__(self, num_users: int, num_items: int, num_factor: int, device, max_batch: int, lr: float = 1e-3,
                 optimizer: str = "adam_dense", id_dtype=torch.int32, init_seed: int = 0, dense_impl: str = "deferred", replay: str | None = None):
        assert optimizer in ("adam_dense", "adam_lazy") and dense_impl in ("deferred", "sweep")
        self.replay = os.environ.get("BR_REPLAY", "fast") if replay is None else replay      # NeuMFConfig.replay: form of the deferred replay
        assert self.replay in ("fast", "exact")
        self.device, self.max_batch, self.lr, self.optimizer, self.id_dtype = torch.device(device), int(max_batch), lr, optimizer, id_dtype
        self.dim = int(num_factor)
        self.deferred = optimizer == "adam_dense" and dense_impl == "deferred"
        self._stale, self._flush_t = False, 0
        dev = self.device
        self._init_tables(num_users, num_items, init_seed)
        self.user_m, self.user_v = torch.zeros_like(self._user), torch.zeros_like(self._user)
        self.item_m, self.item_v = torch.zeros_like(self._item), torch.zeros_like(self._item)
        B = self.max_batch
        self.g_user = torch.empty(B, self.dim, device=dev)
        self.g_item = torch.empty(2 * B, self.dim, device=dev)
        self.per_triplet = torch.empty(B, device=dev)
        self.loss_slots = torch.zeros(ops.SUM_SLOTS, dtype=torch.float64, device=dev)
        self.item_ids = torch.empty(2 * B, dtype=id_dtype, device=dev)
        self.user_index = ops.RowIndex(B, id_dtype, dev)
        self.item_index = ops.RowIndex(2 * B, id_dtype, dev)
        self.err = ops.new_err_flag(dev)
        if self.deferred:
            self.user_last = torch.zeros(self._user.shape[0], dtype=torch.int32, device=dev)
            self.item_last = torch.zeros(self._item.shape[0], dtype=torch.int32, device=dev)
            self.step_state = ops.new_step_state(dev, self.BETA1, self.BETA2, self.EPS, self.replay)
            _lib.check(_lib.load().brStepStateSet(self.step_state.data_ptr(), 0, lr, self.BETA1, self.BETA2, ops._stream()), "brStepStateSet")
            self.r_user = torch.empty(B, self.dim, device=dev)            # rows as of the previous step (replayed in registers)
            self.r_item = torch.empty(2 * B, self.dim, device=dev)
            self.pos_b = torch.arange(2 * B, device=dev).to(id_dtype)
        elif optimizer == "adam_dense":
            self.user_mark = torch.zeros(self._user.shape[0], dtype=torch.uint8, device=dev)
            self.item_mark = torch.zeros(self._item.shape[0], dtype=torch.uint8, device=dev)
        self.t = 0
        self.n_seen = 0

    # the tables as a caller sees them: flushed (deferred mode) before they are handed out
    @property
    def user(self):
        self.flush()
        return self._user

    @user.setter
    def user(self, t):
        self._user = t

    @property
    def item(self):
        self.flush()
        return self._item

    @item.setter
    def item(self, t):
        self._item = t

    def flush(self):
        """deferred dense Adam: apply the pending g = 0 steps to every row of both tables (brAdamFlush).  No-op otherwise."""
        if not (self.deferred and self._stale):
            return
        lib = _lib.load()
        for tab, m, v, last in ((self._user, self.user_m, self.user_v, self.user_last), (self._item, self.item_m, self.item_v, self.item_last)):
            _lib.check(lib.brAdamFlush(tab.data_ptr(), m.data_ptr(), v.data_ptr(), last.data_ptr(), tab.shape[0], tab.shape[1], self.step_state.data_ptr(),
                                       self.BETA1, self.BETA2, self.EPS, ops._stream()), "brAdamFlush")
        self._stale, self._flush_t = False, self.t

    def _advance(self):
        """step counter + alpha ring of the device step state (the replay reads each missed step's alpha from the ring)"""
        if self.t + 1 - self._flush_t >= self.ALPHA_RING - 8:
            self.flush()
        self._stale = True
        _lib.check(_lib.load().brStepStateAdvance(self.step_state.data_ptr(), self.lr, self.BETA1, self.BETA2, None, 0, ops._stream()), "brStepStateAdvance")

    def _init_tables(self, num_users, num_items, init_seed):
        """[TF-sem] Keras Embedding init U(-0.05, 0.05).  (A hook: the row-sharded engine allocates only its shard.)"""
        g = torch.Generator(device="cpu").manual_seed(init_seed)
        self._user = (torch.rand(num_users, self.dim, generator=g) * 0.1 - 0.05).to(self.device)
        self._item = (torch.rand(num_items, self.dim, generator=g) * 0.1 - 0.05).to(self.device)

    def train_step(self, users, pos, neg, batch_total: int | None = None):
        """fit step (BPRModel.py:109): users/pos/neg device ids (B,). No host sync."""
        B = users.shape[0]
        if B == 0:
            return
        if B > self.max_batch:
            raise ValueError("batch exceeds max_batch")
        if self.deferred and self.t + 1 - self._flush_t >= self.ALPHA_RING - 8:      # the replay reads each missed step's alpha from a ring
            self.flush()
        ids2 = self.item_ids[:2 * B]
        for t in (users, pos, neg):
            if t.dtype != self.id_dtype or not t.is_cuda or not t.is_contiguous() or t.shape[0] != B:
                raise TypeError(f"ids must be contiguous {self.id_dtype} device tensors of one length")
        # [pos | neg] side by side (the shared item table gets one index over both): one launch instead of two copies
        _lib.check(_lib.load().brStageBatch(ids2.data_ptr(), ids2[B:].data_ptr(), None, pos.data_ptr(), neg.data_ptr(), None,
                                            ops.I64 if self.id_dtype == torch.int64 else ops.I32, B, ops._stream()), "brStageBatch")
        gr = getattr(self, "_graph", None)
        if gr is not None and B == gr["batch"] and (batch_total is None or batch_total == B):
            gr["users"].copy_(users)
            self.t += 1
            self._stale = self.deferred
            gr["graph"].replay()
            self.n_seen += B
            return
        self.t += 1
        self._stale = self.deferred
        self._step_body(users, ids2, B, B if batch_total is None else batch_total)
        self.n_seen += B

    def enable_graph(self, batch: int | None = None):
        """Replay the step for batches of exactly `batch` triplets as ONE hipGraph (for hosts whose per-launch overhead is the bound:
        at batch 65 536 on MI355X the eager step is GPU-bound at 0.195 ms and the replay, with its two forked index branches, takes
        0.214 ms).  Deferred dense Adam only: then every per-step scalar (step counter, alpha) lives in the device step state.  The
        user ids are copied into a static buffer per step."""
        if not self.deferred:
            raise ValueError("graph replay needs optimizer='adam_dense' with dense_impl='deferred' (per-step scalars on the device)")
        B = self.max_batch if batch is None else int(batch)
        if not 0 < B <= self.max_batch:
            raise ValueError("graph batch must be in (0, max_batch]")
        users = torch.zeros(B, dtype=self.id_dtype, device=self.device)
        ids2 = self.item_ids[:2 * B]
        ids2.zero_()
        # every kernel runs once outside a capture first (code objects load on first launch); the model state is put back afterwards.
        # The dry run's ids are all 0: it moves row 0 of each table and nothing else (deferred tables) - that row, with its moments and
        # `last`, is what is kept.  NOT state_dict(): that flushes, and a flush in the middle of a run resets every row's lag (and, with the
        # fast replay, is no longer bit-neutral: one catch-up over (s, t] and two over (s, u], (u, t] round differently).
        z = torch.zeros(1, dtype=torch.long, device=self.device)
        keep = [(t, t[z].clone()) for t in (self._user, self.user_m, self.user_v, self.user_last, self._item, self.item_m, self.item_v, self.item_last)]
        keep_loss = self.loss_slots.clone()

        def restore():
            for t, row in keep:
                t[z] = row
            self.loss_slots.copy_(keep_loss)
            _lib.check(_lib.load().brStepStateSet(self.step_state.data_ptr(), self.t, self.lr, self.BETA1, self.BETA2, ops._stream()), "brStepStateSet")
        self._step_body(users, ids2, B, B)
        torch.cuda.synchronize(self.device)
        restore()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._step_body(users, ids2, B, B)
        restore()                               # (the capture executes nothing; this re-syncs the device step counter after the dry run)
        self._graph = {"batch": B, "users": users, "graph": g}

    def disable_graph(self):
        self._graph = None

    def _step_body(self, users, ids2, B, bt):
        """the step's launches (ids2 = [pos | neg] already in place)"""
        gi = self.g_item[:2 * B]
        U, I = self._user, self._item
        # one-wave-per-row shapes at a batch that fills the chip: the chunk sorts of both id streams ride in the gather's launch, the
        # chunk-rank launch advances the step state - five launches on one stream, no fork (include/binrec.h brGatherRowsDeferredPairWithIndex)
        fused = self.deferred and self.dim in (64, 128, 256) and B >= 1024 and 2 * B <= 524288 and max(U.shape[0], I.shape[0]) < (1 << 31) - 2 \
            and os.environ.get("BR_FUSED_SORT", "1") != "0"
        if fused:
            hp = (self.BETA1, self.BETA2, self.EPS)
            ru, ri = ops.gather_rows_deferred_pair_with_index(U, self.user_m, self.user_v, self.user_last, users, self.r_user[:B], self.user_index,
                                                              I, self.item_m, self.item_v, self.item_last, ids2, self.r_item[:2 * B], self.item_index,
                                                              self.step_state, self.lr, *hp, err_flag=self.err)
            ar = self.pos_b[:B]
            ops.bpr_forward_backward(ru, ri, ar, ar, self.pos_b[B:2 * B], 1.0 / bt, self.loss_slots, self.g_user[:B], gi, self.per_triplet[:B], self.err)
            ops.adam_rows_sorted_deferred_pair_replayed(U, self.user_m, self.user_v, self.user_last, self.user_index, self.g_user[:B], ru,
                                                        I, self.item_m, self.item_v, self.item_last, self.item_index, gi, ri, 0, self.step_state, *hp)
            return
        if self.deferred:
            _lib.check(_lib.load().brStepStateAdvance(self.step_state.data_ptr(), self.lr, self.BETA1, self.BETA2, None, 0, ops._stream()), "brStepStateAdvance")
        # the two dedup indexes depend only on the ids: each on a side stream of its own (their sort kernels fill 8 and 16 CUs), beside
        # the lookups and the triplet kernel; joined before the optimizer launches.  (On the launch stream they were 92 of the step's 248 us.)
        main = torch.cuda.current_stream(self.device)
        if getattr(self, "_side", None) is None:
            self._side = (torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device))
            self._ev = (torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event())
        self._ev[0].record(main)                                   # the ids are complete; the previous step's readers of the indexes are done
        for k, (idx, ids, rows) in enumerate(((self.user_index, users, U.shape[0]), (self.item_index, ids2, I.shape[0]))):
            self._side[k].wait_event(self._ev[0])
            with torch.cuda.stream(self._side[k]):
                idx.build(ids, rows)
                self._ev[1 + k].record(self._side[k])
        if self.deferred:
            hp = (self.BETA1, self.BETA2, self.EPS)
            pair = self.dim in (64, 128, 256)       # one-wave-per-row shapes: both tables served / updated by one launch each
            if pair:
                ru, ri = ops.gather_rows_deferred_pair(U, self.user_m, self.user_v, self.user_last, users, self.r_user[:B],
                                                       I, self.item_m, self.item_v, self.item_last, ids2, self.r_item[:2 * B], self.step_state, *hp, err_flag=self.err)
            else:
                ru = ops.gather_rows_deferred(U, self.user_m, self.user_v, self.user_last, users, self.step_state, *hp, out=self.r_user[:B], err_flag=self.err)
                ri = ops.gather_rows_deferred(I, self.item_m, self.item_v, self.item_last, ids2, self.step_state, *hp, out=self.r_item[:2 * B], err_flag=self.err)
            ar = self.pos_b[:B]
            ops.bpr_forward_backward(ru, ri, ar, ar, self.pos_b[B:2 * B], 1.0 / bt, self.loss_slots, self.g_user[:B], gi, self.per_triplet[:B], self.err)
            main.wait_event(self._ev[1]); main.wait_event(self._ev[2])
            # the gathered rows ARE the tables' rows replayed to step t-1: the optimizer takes theta from them and replays m, v only
            if pair:
                ops.adam_rows_sorted_deferred_pair_replayed(U, self.user_m, self.user_v, self.user_last, self.user_index, self.g_user[:B], ru,
                                                            I, self.item_m, self.item_v, self.item_last, self.item_index, gi, ri, 0, self.step_state, *hp)
            else:
                ops.adam_rows_sorted_deferred(U, self.user_m, self.user_v, self.user_last, self.user_index, self.g_user[:B], self.dim, self.step_state, *hp,
                                              replayed=ru)
                ops.adam_rows_sorted_deferred(I, self.item_m, self.item_v, self.item_last, self.item_index, gi, self.dim, self.step_state, *hp, replayed=ri)
            return
        pos, neg = ids2[:B], ids2[B:]
        ops.bpr_forward_backward(U, I, users, pos, neg, 1.0 / bt, self.loss_slots, self.g_user[:B], gi,
                                 self.per_triplet[:B], self.err)
        main.wait_event(self._ev[1]); main.wait_event(self._ev[2])
        a = ops.adam_alpha(self.lr, self.t)
        dense = self.optimizer == "adam_dense"
        ops.adam_rows_sorted(U, self.user_m, self.user_v, self.user_index, self.g_user[:B], self.dim, a,
                             mark=self.user_mark if dense else None)
        ops.adam_rows_sorted(I, self.item_m, self.item_v, self.item_index, gi, self.dim, a,
                             mark=self.item_mark if dense else None)
        if dense:
            ops.adam_dense_sweep(U, self.user_m, self.user_v, a, mark=self.user_mark)
            ops.adam_dense_sweep(I, self.item_m, self.item_v, a, mark=self.item_mark)

    def pop_loss(self) -> float:
        """Host sync: mean triplet loss since the last call."""
        s = float(self.loss_slots.sum().item())
        n = max(1, self.n_seen)
        self.loss_slots.zero_()
        self.n_seen = 0
        return s / n

    def predict_scores(self, user_ids, item_ids=None):
        """bpr_predict (src/models/bpr.py:122-133): user vectors x item matrix^T."""
        u = ops.gather_rows([self.user], [user_ids])[0]
        it = self.item if item_ids is None else ops.gather_rows([self.item], [item_ids])[0]
        return ops.score_matrix(u, it)

    def check_ids(self):
        ops.raise_if_flag(self.err)

    # tables + Adam slots (model.save / restoreFromLatestCheckPoint, RModel.py:139,172); row-sharded engines hold their shard
    STATE_TABLES = ("user", "item", "user_m", "user_v", "item_m", "item_v")

    def state_dict(self) -> dict:
        self.flush()
        sd = {"t": self.t}
        sd.update({k: getattr(self, k) for k in self.STATE_TABLES})
        return sd

    def load_state_dict(self, sd: dict):
        self.t = int(sd["t"])
        for k in self.STATE_TABLES:
            getattr(self, "_" + k if k in ("user", "item") else k).copy_(sd[k])
        if self.deferred:                   # a checkpoint holds flushed tables: every row includes step t
            self.user_last.fill_(self.t); self.item_last.fill_(self.t)
            self._stale, self._flush_t = False, self.t
            _lib.check(_lib.load().brStepStateSet(self.step_state.data_ptr(), self.t, self.lr, self.BETA1, self.BETA2, ops._stream()), "brStepStateSet")
