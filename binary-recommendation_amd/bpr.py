"""BPR pairwise model on the HIP hot path — src/models/BPRModel.py:49-74,124-144 and the
stand-alone src/models/bpr.py:141-201.

Graph: shared item embedding (positive + negative lookups), user embedding, Lambda triplet loss
`1 - sigmoid(u.p - u.n)` (NOT -log sigmoid: BPRModel.py:144), identityLoss = mean (BPRModel.py:124-126),
Adam(1e-3) (BPRModel.py:70).  One fused launch does the 3 gathers, 2 dots, the loss partials and the
3 per-triplet row gradients; the shared item table receives its two IndexedSlices concatenated
[pos rows | neg rows] and deduplicated by one sort over the 2B ids.
"""
from __future__ import annotations

import torch

from . import ops


class BPREngine:
    def __init__(self, num_users: int, num_items: int, num_factor: int, device, max_batch: int, lr: float = 1e-3,
                 optimizer: str = "adam_dense", id_dtype=torch.int32, init_seed: int = 0):
        assert optimizer in ("adam_dense", "adam_lazy")
        self.device, self.max_batch, self.lr, self.optimizer, self.id_dtype = torch.device(device), int(max_batch), lr, optimizer, id_dtype
        self.dim = int(num_factor)
        dev = self.device
        self._init_tables(num_users, num_items, init_seed)
        self.user_m, self.user_v = torch.zeros_like(self.user), torch.zeros_like(self.user)
        self.item_m, self.item_v = torch.zeros_like(self.item), torch.zeros_like(self.item)
        B = self.max_batch
        self.g_user = torch.empty(B, self.dim, device=dev)
        self.g_item = torch.empty(2 * B, self.dim, device=dev)
        self.per_triplet = torch.empty(B, device=dev)
        self.loss_slots = torch.zeros(ops.SUM_SLOTS, dtype=torch.float64, device=dev)
        self.item_ids = torch.empty(2 * B, dtype=id_dtype, device=dev)
        self.user_index = ops.RowIndex(B, id_dtype, dev)
        self.item_index = ops.RowIndex(2 * B, id_dtype, dev)
        self.err = ops.new_err_flag(dev)
        if optimizer == "adam_dense":
            self.user_mark = torch.zeros(self.user.shape[0], dtype=torch.uint8, device=dev)
            self.item_mark = torch.zeros(self.item.shape[0], dtype=torch.uint8, device=dev)
        self.t = 0
        self.n_seen = 0

    def _init_tables(self, num_users, num_items, init_seed):
        """[TF-sem] Keras Embedding init U(-0.05, 0.05).  (A hook: the row-sharded engine allocates only its shard.)"""
        g = torch.Generator(device="cpu").manual_seed(init_seed)
        self.user = (torch.rand(num_users, self.dim, generator=g) * 0.1 - 0.05).to(self.device)
        self.item = (torch.rand(num_items, self.dim, generator=g) * 0.1 - 0.05).to(self.device)

    def train_step(self, users, pos, neg, batch_total: int | None = None):
        """fit step (BPRModel.py:109): users/pos/neg device ids (B,). No host sync."""
        B = users.shape[0]
        if B == 0:
            return
        if B > self.max_batch:
            raise ValueError("batch exceeds max_batch")
        self.t += 1
        bt = B if batch_total is None else batch_total
        gi = self.g_item[:2 * B]
        ops.bpr_forward_backward(self.user, self.item, users, pos, neg, 1.0 / bt, self.loss_slots, self.g_user[:B], gi,
                                 self.per_triplet[:B], self.err)
        ids2 = self.item_ids[:2 * B]
        ids2[:B].copy_(pos)
        ids2[B:].copy_(neg)
        self.user_index.build(users, self.user.shape[0])
        self.item_index.build(ids2, self.item.shape[0])
        a = ops.adam_alpha(self.lr, self.t)
        dense = self.optimizer == "adam_dense"
        ops.adam_rows_sorted(self.user, self.user_m, self.user_v, self.user_index, self.g_user[:B], self.dim, a,
                             mark=self.user_mark if dense else None)
        ops.adam_rows_sorted(self.item, self.item_m, self.item_v, self.item_index, gi, self.dim, a,
                             mark=self.item_mark if dense else None)
        if dense:
            ops.adam_dense_sweep(self.user, self.user_m, self.user_v, a, mark=self.user_mark)
            ops.adam_dense_sweep(self.item, self.item_m, self.item_v, a, mark=self.item_mark)
        self.n_seen += B

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
        sd = {"t": self.t}
        sd.update({k: getattr(self, k) for k in self.STATE_TABLES})
        return sd

    def load_state_dict(self, sd: dict):
        self.t = int(sd["t"])
        for k in self.STATE_TABLES:
            getattr(self, k).copy_(sd[k])
