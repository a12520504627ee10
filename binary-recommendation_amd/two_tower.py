"""TwoTower retrieval model on the HIP hot path — trainers/twoTower.py:19-111.

  userTower = StringLookup -> Embedding(nbrUser+2, embedDim) -> Dense(semb)      (twoTower.py:33-40)
  itemTower = StringLookup -> Embedding(nbrItem+2, embedDim) -> Dense(semb)      (twoTower.py:35-41)
  computeLossTfrs : tfrs.tasks.Retrieval(loss=None) with candidate_ids = batch item ids
                    = in-batch softmax, accidental hits masked, SUM reduction [TF-sem]  (:47,82-83)
  computeLossRdZero: sigmoid(Dot(q, c)) vs RATING_TYPE, BinaryCrossentropy                (:85-87)
  train_step: GradientTape over both towers, optimizer.apply_gradients                      (:89-102)
  optimizer : getOptimizer("Adagrad", 0.1) (twoTower.py:209,278-279; helper file missing in the
              reference) => Keras Adagrad, initial accumulator 0.1, eps 1e-7 [TF-sem]
StringLookup is host-side vocabulary mapping: index 0 = mask, 1 = OOV, vocabulary from 2 (hence +2 rows).
"""
from __future__ import annotations

import math

import torch

from . import ops


class StringLookup:
    """tf.keras StringLookup(vocabulary=...) [TF-sem TF 2.3]: '' -> 0, OOV -> 1, vocabulary[i] -> i + 2."""

    def __init__(self, vocabulary):
        self.index = {str(v): i + 2 for i, v in enumerate(vocabulary)}

    def __call__(self, keys, device=None, dtype=torch.int32):
        idx = [0 if k == "" else self.index.get(str(k), 1) for k in keys]
        return torch.tensor(idx, dtype=dtype, device=device)


class TwoTowerEngine:
    def __init__(self, embed_dim: int, nbr_item: int, nbr_user: int, semb: int, device, max_batch: int, lr: float = 0.1,
                 optimizer: str = "Adagrad", rd_zero: bool = False, id_dtype=torch.int32, init_seed: int = 0):
        assert optimizer in ("Adagrad", "Adam")
        if embed_dim > 128 or semb > 128:
            raise ValueError("embedDim and semb <= 128 in this build")
        dev = self.device = torch.device(device)
        self.E, self.S, self.max_batch, self.lr, self.optimizer, self.rd_zero, self.id_dtype = embed_dim, semb, int(max_batch), lr, optimizer, rd_zero, id_dtype
        g = torch.Generator(device="cpu").manual_seed(init_seed)
        E, S, B = embed_dim, semb, self.max_batch
        self._init_tables(nbr_user + 2, nbr_item + 2, g)
        # dense params flat: [Wu (E x S) | bu (S) | Wi (E x S) | bi (S)]  ([W|b] adjacent: slab layout)
        lim = math.sqrt(6.0 / (E + S))
        self.theta = torch.zeros(2 * (E * S + S), device=dev)
        self.grad = torch.zeros_like(self.theta)
        for o in (0, E * S + S):
            self.theta[o:o + E * S].copy_(((torch.rand(E * S, generator=g) * 2 - 1) * lim).to(dev))
        acc0 = 0.1 if optimizer == "Adagrad" else 0.0
        mk = lambda t: torch.full_like(t, acc0)
        self.user_acc, self.item_acc, self.theta_acc = mk(self.user_emb), mk(self.item_emb), mk(self.theta)
        if optimizer == "Adam":
            self.user_v, self.item_v, self.theta_v = (torch.zeros_like(t) for t in (self.user_emb, self.item_emb, self.theta))
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        self.eu, self.ei, self.q, self.c = f(B, E), f(B, E), f(B, S), f(B, S)
        self.dq, self.dc, self.deu, self.dei = f(B, S), f(B, S), f(B, E), f(B, E)
        self.lse, self.z, self.dz, self.prob = f(B), f(B), f(B), f(B)
        self.loss_slots = torch.zeros(ops.SUM_SLOTS, dtype=torch.float64, device=dev)
        self.ns = ops.dense_backward_slabs(B, E, S)
        self.slabs = f(self.ns * (E * S + S))
        self.dz_ws = f(ops.dense_backward_ws_floats(B, E, S))
        self.user_index, self.item_index = ops.RowIndex(B, id_dtype, dev), ops.RowIndex(B, id_dtype, dev)
        self.err = ops.new_err_flag(dev)
        self.t, self.n_seen = 0, 0

    def _init_tables(self, user_rows, item_rows, g):
        """[TF-sem] Embedding init U(-0.05, 0.05) (rows = vocabulary + 2: '' and OOV).  A hook: the row-sharded engine allocates
        only its shard."""
        self.user_emb = (torch.rand(user_rows, self.E, generator=g) * 0.1 - 0.05).to(self.device)
        self.item_emb = (torch.rand(item_rows, self.E, generator=g) * 0.1 - 0.05).to(self.device)

    # views
    def W(self, tower):
        o = 0 if tower == "user" else self.E * self.S + self.S
        return self.theta[o:o + self.E * self.S].view(self.E, self.S), self.theta[o + self.E * self.S:o + self.E * self.S + self.S]

    def _gW(self, tower):
        o = 0 if tower == "user" else self.E * self.S + self.S
        return self.grad[o:o + self.E * self.S + self.S]

    # ---- hooks the row-sharded / data-parallel subclass overrides (parallel.py) ----
    dist = None

    def _lookup(self, users, items, B):
        """G1: both embedding lookups in one launch -> eu, ei (B x E)."""
        ops.gather_rows([self.user_emb, self.item_emb], [users, items], [self.eu[:B], self.ei[:B]], err_flag=self.err)

    def _softmax(self, q, c, items, B, dq, dc):
        """L4: in-batch softmax loss (+ gradients when dq is given) over the local batch."""
        if dq is None:
            ops.inbatch_softmax_lse(q, c, items, items, 0, self.lse[:B], self.loss_slots)
            return
        ops.inbatch_softmax_lse_grad_q(q, c, items, items, 0, self.lse[:B], self.loss_slots, dq)      # lse + loss + dQ in one sweep
        ops.inbatch_softmax_grad(q, c, items, items, 0, self.lse[:B], None, dc)

    def _start_indexes(self, users, items):
        """the two dedup indexes depend only on the ids: each on a side stream of its own beside the lookup / towers / softmax, joined in
        _apply_tables.  (The row-sharded subclass builds its indexes over the ids it RECEIVES and overrides this with a no-op.)"""
        main = torch.cuda.current_stream(self.device)
        if getattr(self, "_side", None) is None:
            self._side = (torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device))
            self._ev = (torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event())
        self._ev[0].record(main)                   # the previous step's readers of the indexes are behind this point
        for k, (idx, ids, rows) in enumerate(((self.user_index, users, self.user_emb.shape[0]), (self.item_index, items, self.item_emb.shape[0]))):
            self._side[k].wait_event(self._ev[0])
            with torch.cuda.stream(self._side[k]):
                idx.build(ids, rows)
                self._ev[1 + k].record(self._side[k])

    def _apply_tables(self, users, items, B):
        """S1 + O2/O1 on the two embedding tables from the per-pair row gradients deu / dei."""
        main = torch.cuda.current_stream(self.device)
        main.wait_event(self._ev[1]); main.wait_event(self._ev[2])
        self._opt_rows(self.user_emb, self.user_acc, getattr(self, "user_v", None), self.user_index, self.deu[:B])
        self._opt_rows(self.item_emb, self.item_acc, getattr(self, "item_v", None), self.item_index, self.dei[:B])

    def _opt_rows(self, table, acc, v, index, row_grads):
        if self.optimizer == "Adagrad":
            ops.adagrad_rows_sorted(table, acc, index, row_grads, row_grads.stride(0), self.lr)
        else:
            ops.adam_rows_sorted(table, acc, v, index, row_grads, row_grads.stride(0), ops.adam_alpha(self.lr, self.t))

    def compute_emb(self, users, items, B):
        """computeEmb (twoTower.py:77-80): (q, c) = towers(user ids), towers(item ids)."""
        self._lookup(users, items, B)
        Wu, bu = self.W("user"); Wi, bi = self.W("item")
        ops.dense_forward(self.eu[:B], Wu, bu, self.q[:B], "linear")
        ops.dense_forward(self.ei[:B], Wi, bi, self.c[:B], "linear")
        return self.q[:B], self.c[:B]

    def item_tower(self, items):
        """itemTower over arbitrary ids (setCandidates, twoTower.py:64-69)."""
        n = items.shape[0]
        e = ops.gather_rows([self.item_emb], [items])[0]
        out = torch.empty(n, self.S, device=self.device)
        Wi, bi = self.W("item")
        ops.dense_forward(e, Wi, bi, out, "linear")
        return out

    def user_tower(self, users):
        n = users.shape[0]
        e = ops.gather_rows([self.user_emb], [users])[0]
        out = torch.empty(n, self.S, device=self.device)
        Wu, bu = self.W("user")
        ops.dense_forward(e, Wu, bu, out, "linear")
        return out

    def enable_graph(self, batch: int | None = None):
        """Replay the step for batches of exactly `batch` pairs as ONE hipGraph (the eager step is ~20 launches from the Python host with
        gaps between them: 0.48 ms at batch 8 192, of which the kernels take 0.40).  Adagrad, single GPU: then no per-step scalar is baked
        into a launch.  The first such step runs eagerly on the static input buffers, the body is captured behind it."""
        if self.optimizer != "Adagrad" or self.dist is not None:
            raise ValueError("graph replay covers the single-GPU Adagrad step (Adam's alpha_t is a host scalar per step; the sharded step syncs the host)")
        B = self.max_batch if batch is None else int(batch)
        if not 0 < B <= self.max_batch:
            raise ValueError("graph batch must be in (0, max_batch]")
        z = lambda dt: torch.zeros(B, dtype=dt, device=self.device)
        self._graph = {"batch": B, "graph": None, "users": z(self.id_dtype), "items": z(self.id_dtype), "labels": z(torch.float32), "bt": None}

    def disable_graph(self):
        self._graph = None

    def train_step(self, users, items, labels=None, batch_total=None):
        """train_step (twoTower.py:89-102). labels only for rd_zero (RATING_TYPE)."""
        B = users.shape[0]
        if B == 0:
            if self.dist is not None and self.dist.world > 1:
                # the peers are about to enter the step's collectives: returning here would leave them waiting forever
                raise ValueError("empty local batch in a data-parallel step: give every rank at least one pair (pad or drop the ragged tail)")
            return
        if B > self.max_batch:
            raise ValueError("batch exceeds max_batch")
        g = getattr(self, "_graph", None)
        if g is not None and B == g["batch"] and (g["graph"] is None or g["bt"] == batch_total):
            g["users"].copy_(users); g["items"].copy_(items)
            if labels is not None:
                g["labels"].copy_(labels)
            lab = g["labels"] if self.rd_zero else None
            self.t += 1
            if g["graph"] is None:
                self._step_body(g["users"], g["items"], lab, B, batch_total)       # eager: every kernel once outside a capture
                torch.cuda.synchronize(self.device)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self._step_body(g["users"], g["items"], lab, B, batch_total)
                g["graph"], g["bt"] = graph, batch_total
            else:
                g["graph"].replay()
            self.n_seen += B if self.rd_zero else 1
            return
        self.t += 1
        self._step_body(users, items, labels, B, batch_total)
        self.n_seen += B if self.rd_zero else 1

    def _step_body(self, users, items, labels, B, batch_total):
        """the launches of one step (no host sync, no per-step host scalar with Adagrad)"""
        self._start_indexes(users, items)
        q, c = self.compute_emb(users, items, B)
        dq, dc = self.dq[:B], self.dc[:B]
        if self.rd_zero:
            bt = B if batch_total is None else batch_total
            ops.row_dot(q, c, self.z[:B])
            ops.bce_logits(self.z[:B], labels, 1.0 / bt, prob=self.prob[:B], dz=self.dz[:B], sums=self.loss_slots)
            ops.row_dot_backward(q, c, self.dz[:B], dq, dc)
        else:
            self._softmax(q, c, items, B, dq, dc)
        E, S = self.E, self.S
        ns = ops.dense_backward_slabs(B, E, S)
        for tower, g_out, e_in, d_e in (("user", dq, self.eu[:B], self.deu[:B]), ("item", dc, self.ei[:B], self.dei[:B])):
            Wt, _ = self.W(tower)
            y = self.q[:B] if tower == "user" else self.c[:B]
            ops.dense_backward(g_out, y, e_in, Wt, "linear", self.slabs, ns, gx=d_e, dz_ws=self.dz_ws)
            ops.reduce_slabs(self.slabs, ns, E * S + S, self._gW(tower))
        if self.dist is not None:
            self.dist.all_reduce_sum(self.grad)
        self._apply_tables(users, items, B)
        if self.optimizer == "Adagrad":
            ops.adagrad_flat(self.theta, self.theta_acc, self.grad, self.lr)
        else:
            ops.adam_flat(self.theta, self.theta_acc, self.theta_v, self.grad, ops.adam_alpha(self.lr, self.t))

    def test_step(self, users, items):
        """test_step (twoTower.py:104-111): the retrieval loss without an update."""
        B = users.shape[0]
        q, c = self.compute_emb(users, items, B)
        self._softmax(q, c, items, B, None, None)
        self.n_seen += 1

    def pop_loss(self) -> float:
        """Host sync: mean over steps of the step loss (softmax: SUM over the batch per step, as TFRS
        reports it; rd_zero: mean BCE)."""
        s = float(self.loss_slots.sum().item())
        n = max(1, self.n_seen)
        self.loss_slots.zero_()
        self.n_seen = 0
        return s / n

    def check_ids(self):
        ops.raise_if_flag(self.err)

    # embedding tables, towers and optimizer slots (row-sharded engines hold their shard of the tables)
    def _state_names(self):
        names = ["user_emb", "item_emb", "theta", "user_acc", "item_acc", "theta_acc"]
        if self.optimizer == "Adam":
            names += ["user_v", "item_v", "theta_v"]
        return names

    def state_dict(self) -> dict:
        sd = {"t": self.t}
        sd.update({k: getattr(self, k) for k in self._state_names()})
        return sd

    def load_state_dict(self, sd: dict):
        self.t = int(sd["t"])
        for k in self._state_names():
            getattr(self, k).copy_(sd[k])
