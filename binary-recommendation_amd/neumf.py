"""NeuMF training/inference engine on the HIP hot path (single GPU or one data-parallel rank).

Covers both reference graphs with one parametrisation:
  variant "A": trainers/NFC_plain.py:107-155  (concat [item,user], Dense 100/50/10 sigmoid,
               head concat [mf, mlp], BCE, Adam lr .005)
  variant "B": src/models/NeuMFModel.py:53-100 (concat [user,item], Dense F, F/2, F/4 relu,
               head concat [mlp, mf], MSE, Adam 1e-3)

Every arithmetic op is a libbinrec_hip.so launch (ops.py); torch only owns the buffers, the
stream and (data-parallel) the RCCL collectives.  One training step is the launch sequence in
`NeuMFEngine.train_step` — see DESIGN.md for the kernel-by-kernel data flow.
"""
from __future__ import annotations

import dataclasses
import math

import torch

from . import ops

DENSE_ORDER = ("W1", "b1", "g1", "be1", "W2", "b2", "g2", "be2", "W3", "b3", "W4", "b4")
TABLES = ("user_mlp", "item_mlp", "user_mf", "item_mf")


@dataclasses.dataclass
class NeuMFConfig:
    variant: str = "A"
    dim: int = 10                       # latent_dim (NFC_plain.py:109) / numFactor (RModel.py:35)
    hidden: tuple | None = None
    lr: float | None = None
    dropout: float = 0.2                # NFC_plain.py:138,141,144 ; NeuMFModel.py:67,71,75
    bn_eps: float = 1e-3                # [TF-sem] Keras BatchNormalization defaults
    bn_momentum: float = 0.99
    beta1: float = 0.9                  # [TF-sem] Keras Adam defaults
    beta2: float = 0.999
    adam_eps: float = 1e-7
    optimizer: str = "adam_dense"       # "adam_dense" = Keras semantics (non-lazy sparse apply);
                                        # "adam_lazy"  = touched rows only (throughput deviation)
    seed: int = 0x5EED
    sync_bn: bool = True                # data-parallel: batch statistics over the GLOBAL batch

    def __post_init__(self):
        assert self.variant in ("A", "B")
        assert self.optimizer in ("adam_dense", "adam_lazy")
        if self.variant == "A":
            self.hidden = tuple(self.hidden) if self.hidden else (100, 50, 10)
            self.act, self.item_first, self.mf_first, self.loss = "sigmoid", 1, 1, "bce"
            self.lr = 0.005 if self.lr is None else self.lr
        else:
            self.hidden = tuple(self.hidden) if self.hidden else (self.dim, self.dim // 2, self.dim // 4)
            self.act, self.item_first, self.mf_first, self.loss = "relu", 0, 0, "mse"
            self.lr = 1e-3 if self.lr is None else self.lr

    def dense_shapes(self):
        n1, n2, n3 = self.hidden
        return {"W1": (2 * self.dim, n1), "b1": (n1,), "g1": (n1,), "be1": (n1,),
                "W2": (n1, n2), "b2": (n2,), "g2": (n2,), "be2": (n2,),
                "W3": (n2, n3), "b3": (n3,), "W4": (n3 + 1,), "b4": (1,)}


class _Flat:
    """One flat fp32 buffer with named views (dense parameters / their grads / Adam slots)."""

    def __init__(self, shapes: dict, device):
        self.offsets, off = {}, 0
        for k, shp in shapes.items():
            n = 1
            for s in shp:
                n *= s
            self.offsets[k] = (off, n, shp)
            off += n
        self.numel = off
        self.buf = torch.zeros(off, dtype=torch.float32, device=device)

    def view(self, k):
        off, n, shp = self.offsets[k]
        return self.buf[off:off + n].view(*shp)

    def slice(self, k0, k1):
        """contiguous span covering parameters k0..k1 (adjacent in the layout)."""
        o0 = self.offsets[k0][0]
        o1, n1, _ = self.offsets[k1]
        return self.buf[o0:o1 + n1]


class NeuMFEngine:
    def __init__(self, cfg: NeuMFConfig, num_user_rows: int, num_item_rows: int, device, max_batch: int,
                 id_dtype=torch.int32, init_seed: int = 0, dist=None):
        self.cfg, self.device, self.max_batch, self.id_dtype = cfg, torch.device(device), int(max_batch), id_dtype
        self.num_user_rows, self.num_item_rows = int(num_user_rows), int(num_item_rows)
        self.dist = dist                      # None or a parallel.DataParallelCtx
        D, (n1, n2, n3) = cfg.dim, cfg.hidden
        if 2 * D > 128 or max(n1, n2) > 128 or n3 > 32:
            raise ValueError("tower widths: 2*dim, n1, n2 <= 128 and n3 <= 32 in this build")
        dev = self.device
        g = torch.Generator(device="cpu").manual_seed(init_seed)
        # [TF-sem] Embedding init U(-0.05, 0.05); Dense glorot-uniform; bias 0; BN gamma 1 beta 0
        self.tables = {}
        self._init_tables(g, init_seed)
        self.theta = _Flat(cfg.dense_shapes(), dev)
        for k, (_, _, shp) in self.theta.offsets.items():
            if k.startswith("W"):
                fi, fo = (shp[0], shp[1]) if len(shp) == 2 else (shp[0], 1)
                lim = math.sqrt(6.0 / (fi + fo))
                self.theta.view(k).copy_(((torch.rand(*shp, generator=g) * 2 - 1) * lim).to(dev))
            elif k.startswith("g"):
                self.theta.view(k).fill_(1.0)
        self.grad = _Flat(cfg.dense_shapes(), dev)
        self.adam_m = _Flat(cfg.dense_shapes(), dev)
        self.adam_v = _Flat(cfg.dense_shapes(), dev)
        self.tab_m = {k: torch.zeros_like(v) for k, v in self.tables.items()}
        self.tab_v = {k: torch.zeros_like(v) for k, v in self.tables.items()}
        self.moving = {"mm1": torch.zeros(n1, device=dev), "mv1": torch.ones(n1, device=dev),
                       "mm2": torch.zeros(n2, device=dev), "mv2": torch.ones(n2, device=dev)}
        self.t = 0
        self._alloc(self.max_batch)

    def local_rows(self, name: str) -> int:
        """rows of `name` held by this process (all of them without row-sharding)."""
        return self.num_user_rows if name.startswith("user") else self.num_item_rows

    def _init_tables(self, g, init_seed):
        D, dev = self.cfg.dim, self.device
        for name in TABLES:
            rows = self.local_rows(name)
            if rows * D > (1 << 26):   # big tables: draw on the device (same distribution)
                dg = torch.Generator(device=dev).manual_seed(init_seed + 7919 * (1 + TABLES.index(name)))
                self.tables[name] = torch.rand(rows, D, generator=dg, device=dev, dtype=torch.float32).mul_(0.1).sub_(0.05)
            else:
                self.tables[name] = (torch.rand(rows, D, generator=g) * 0.1 - 0.05).to(dev)

    # ------------------------------------------------------------------ buffers
    def _alloc(self, B):
        cfg, dev = self.cfg, self.device
        D, (n1, n2, n3) = cfg.dim, cfg.hidden
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        self.x0, self.dot = f(B, 2 * D), f(B)
        self.a1, self.a2, self.a3 = f(B, n1), f(B, n2), f(B, n3)
        self.logit, self.prob = f(B), f(B)
        self.da3, self.ddot = f(B, n3), f(B)
        self.gh2, self.gh1, self.dx0 = f(B, n2), f(B, n1), f(B, 2 * D)
        self.g_user_mf, self.g_item_mf = f(B, D), f(B, D)
        # per-step double scratch: [stats1 2n1 | stats2 2n2 | bsum1 2n1 | bsum2 2n2]
        self.dstat = torch.zeros(4 * n1 + 4 * n2, dtype=torch.float64, device=dev)
        o = 0
        self.stats1 = self.dstat[o:o + 2 * n1]; o += 2 * n1
        self.stats2 = self.dstat[o:o + 2 * n2]; o += 2 * n2
        self.bsum1 = self.dstat[o:o + 2 * n1]; o += 2 * n1
        self.bsum2 = self.dstat[o:o + 2 * n2]
        self.msums = torch.zeros(4, dtype=torch.float64, device=dev)   # loss, se, ae, correct (epoch)
        self.bn = {k: f(n) for k, n in (("scale1", n1), ("shift1", n1), ("mean1", n1), ("rstd1", n1),
                                        ("scale2", n2), ("shift2", n2), ("mean2", n2), ("rstd2", n2))}
        self.ns1 = ops.dense_backward_slabs(B, 2 * D, n1)
        self.slabs = f(max(self.ns1, 1) * max(2 * D * n1 + n1, n1 * n2 + n2, n2 * n3 + n3))
        self.nsh = ops.head_slabs(B)
        self.hslabs = f(self.nsh * (n3 + 2))
        self.err = ops.new_err_flag(dev)
        self._alloc_sparse(B)

    def _alloc_sparse(self, B):
        dev = self.device
        self.user_index = ops.RowIndex(B, self.id_dtype, dev)
        self.item_index = ops.RowIndex(B, self.id_dtype, dev)
        if self.cfg.optimizer == "adam_dense":
            self.user_mark = torch.zeros(self.local_rows("user_mf"), dtype=torch.uint8, device=dev)
            self.item_mark = torch.zeros(self.local_rows("item_mf"), dtype=torch.uint8, device=dev)

    # ------------------------------------------------------------------ forward (shared)
    def _forward(self, users, items, B, training, row0, batch_total):
        cfg, th, bn = self.cfg, self.theta, self.bn
        p = cfg.dropout if training else 0.0
        x0, a1, a2, a3 = self.x0[:B], self.a1[:B], self.a2[:B], self.a3[:B]
        self._embed_forward(users, items, B)
        step = self.t
        ops.dense_forward(x0, th.view("W1"), th.view("b1"), a1, cfg.act, drop_p=p, seed=cfg.seed, step=step, site=0,
                          row0=row0, stats=self.stats1 if training else None)
        if training:
            if self.dist is not None and cfg.sync_bn:
                self.dist.all_reduce_sum(self.stats1)
            ops.bn_finalize(self.stats1, batch_total, th.view("g1"), th.view("be1"), cfg.bn_eps, cfg.bn_momentum,
                            self.moving["mm1"], self.moving["mv1"], bn["scale1"], bn["shift1"], bn["mean1"], bn["rstd1"])
        else:
            ops.bn_inference(th.view("g1"), th.view("be1"), self.moving["mm1"], self.moving["mv1"], cfg.bn_eps,
                             bn["scale1"], bn["shift1"])
        ops.dense_forward(a1, th.view("W2"), th.view("b2"), a2, cfg.act, bn["scale1"], bn["shift1"], drop_p=p,
                          seed=cfg.seed, step=step, site=1, row0=row0, stats=self.stats2 if training else None)
        if training:
            if self.dist is not None and cfg.sync_bn:
                self.dist.all_reduce_sum(self.stats2)
            ops.bn_finalize(self.stats2, batch_total, th.view("g2"), th.view("be2"), cfg.bn_eps, cfg.bn_momentum,
                            self.moving["mm2"], self.moving["mv2"], bn["scale2"], bn["shift2"], bn["mean2"], bn["rstd2"])
        else:
            ops.bn_inference(th.view("g2"), th.view("be2"), self.moving["mm2"], self.moving["mv2"], cfg.bn_eps,
                             bn["scale2"], bn["shift2"])
        ops.dense_forward(a2, th.view("W3"), th.view("b3"), a3, cfg.act, bn["scale2"], bn["shift2"], drop_p=p,
                          seed=cfg.seed, step=step, site=2, row0=row0)

    def _embed_forward(self, users, items, B):
        """G1+M1+T1: 4 lookups, GMF dot, MLP concat -> x0, dot (one launch)."""
        t = self.tables
        ops.neumf_embed_forward(t["user_mlp"], t["item_mlp"], t["user_mf"], t["item_mf"], users, items,
                                self.cfg.item_first, self.x0[:B], self.dot[:B], self.err)

    # ------------------------------------------------------------------ one optimizer step
    def train_step(self, users, items, labels, row0: int = 0, batch_total: int | None = None):
        """users/items: device int ids (B,), labels: device float32 (B,).  No host sync."""
        cfg, th, gr, bn = self.cfg, self.theta, self.grad, self.bn
        B = users.shape[0]
        if B > self.max_batch:
            raise ValueError(f"batch {B} > max_batch {self.max_batch}")
        if B == 0:
            return
        batch_total = B if batch_total is None else batch_total
        D, (n1, n2, n3) = cfg.dim, cfg.hidden
        self.t += 1
        step, p, seed = self.t, cfg.dropout, cfg.seed
        self.dstat.zero_()
        self._forward(users, items, B, True, row0, batch_total)
        a1, a2, a3, x0 = self.a1[:B], self.a2[:B], self.a3[:B], self.x0[:B]
        inv_b = 1.0 / batch_total
        nsh = ops.head_slabs(B)
        ops.neumf_head(a3, self.dot[:B], labels, th.view("W4"), th.view("b4"), cfg.mf_first, cfg.loss, inv_b,
                       logit=self.logit[:B], prob=self.prob[:B], sums=self.msums, da3=self.da3[:B], ddot=self.ddot[:B],
                       slabs=self.hslabs, n_slabs=nsh)
        ops.reduce_slabs(self.hslabs, nsh, n3 + 2, gr.slice("W4", "b4"))
        ns = ops.dense_backward_slabs(B, 0, 0)
        # layer 3: no BN after it; its input carries BN2 + dropout site 2
        ops.dense_backward(self.da3[:B], a3, a2, th.view("W3"), cfg.act, self.slabs, ns, gx=self.gh2[:B],
                           in_scale=bn["scale2"], in_shift=bn["shift2"], in_bn=(bn["mean2"], bn["rstd2"]), in_drop_p=p,
                           in_site=2, seed=seed, step=step, row0=row0, in_bn_sums=self.bsum2, batch_total=batch_total)
        ops.reduce_slabs(self.slabs, ns, n2 * n3 + n3, gr.slice("W3", "b3"))
        if self.dist is not None and cfg.sync_bn:
            self.dist.all_reduce_sum(self.bsum2)
        # layer 2
        ops.dense_backward(self.gh2[:B], a2, a1, th.view("W2"), cfg.act, self.slabs, ns, gx=self.gh1[:B],
                           out_bn=(bn["mean2"], bn["rstd2"], th.view("g2")), bn_sums=self.bsum2, batch_total=batch_total,
                           in_scale=bn["scale1"], in_shift=bn["shift1"], in_bn=(bn["mean1"], bn["rstd1"]), in_drop_p=p,
                           in_site=1, seed=seed, step=step, row0=row0, in_bn_sums=self.bsum1)
        ops.reduce_slabs(self.slabs, ns, n1 * n2 + n2, gr.slice("W2", "b2"))
        if self.dist is not None and cfg.sync_bn:
            self.dist.all_reduce_sum(self.bsum1)
        # layer 1: input = raw concat with dropout site 0, no BN below
        ops.dense_backward(self.gh1[:B], a1, x0, th.view("W1"), cfg.act, self.slabs, ns, gx=self.dx0[:B],
                           out_bn=(bn["mean1"], bn["rstd1"], th.view("g1")), bn_sums=self.bsum1, batch_total=batch_total,
                           in_drop_p=p, in_site=0, seed=seed, step=step, row0=row0)
        ops.reduce_slabs(self.slabs, ns, 2 * D * n1 + n1, gr.slice("W1", "b1"))
        # dgamma/dbeta are the BN-backward column sums.  With sync_bn those sums are already global,
        # so they are written AFTER the dense all-reduce; per-replica BN sums are local like the rest.
        synced = self.dist is not None and cfg.sync_bn
        if not synced:
            ops.bn_param_grads(self.bsum2, gr.view("g2"), gr.view("be2"))
            ops.bn_param_grads(self.bsum1, gr.view("g1"), gr.view("be1"))
        if self.dist is not None:
            self.dist.all_reduce_sum(self.grad.buf)
        if synced:
            ops.bn_param_grads(self.bsum2, gr.view("g2"), gr.view("be2"))
            ops.bn_param_grads(self.bsum1, gr.view("g1"), gr.view("be1"))
        self._embed_backward_apply(users, items, B)
        a = ops.adam_alpha(cfg.lr, self.t, cfg.beta1, cfg.beta2)
        ops.adam_flat(self.theta.buf, self.adam_m.buf, self.adam_v.buf, self.grad.buf, a, beta1=cfg.beta1, beta2=cfg.beta2,
                      eps=cfg.adam_eps)

    def row_grad_views(self, B):
        """name -> (tensor, row stride): the MLP tables' row gradients are the halves of dx0."""
        D = self.cfg.dim
        uo, io = (D, 0) if self.cfg.item_first else (0, D)
        dx0 = self.dx0[:B]
        return {"user_mlp": (dx0[:, uo:uo + D], 2 * D), "item_mlp": (dx0[:, io:io + D], 2 * D),
                "user_mf": (self.g_user_mf[:B], D), "item_mf": (self.g_item_mf[:B], D)}

    def _embed_backward_apply(self, users, items, B):
        """B1 row gradients of the 4 tables, S1 dedup index, O1 Adam on the tables."""
        cfg, t = self.cfg, self.tables
        ops.neumf_embed_backward(t["user_mf"], t["item_mf"], users, items, cfg.item_first, None, self.ddot[:B],
                                 self.g_user_mf[:B], self.g_item_mf[:B])
        self.user_index.build(users, self.num_user_rows)
        self.item_index.build(items, self.num_item_rows)
        self._adam_tables(self.row_grad_views(B))

    def _adam_tables(self, rg):
        """rg: name -> (row_grads, row stride) aligned with the positions the indexes were built on."""
        cfg = self.cfg
        a = ops.adam_alpha(cfg.lr, self.t, cfg.beta1, cfg.beta2)
        hp = dict(beta1=cfg.beta1, beta2=cfg.beta2, eps=cfg.adam_eps)
        dense = cfg.optimizer == "adam_dense"
        for name in TABLES:
            idx = self.user_index if name.startswith("user") else self.item_index
            mark = (self.user_mark if name.startswith("user") else self.item_mark) if dense else None
            g, ldg = rg[name]
            ops.adam_rows_sorted(self.tables[name], self.tab_m[name], self.tab_v[name], idx, g, ldg, a, mark=mark, **hp)
            if dense:
                ops.adam_dense_sweep(self.tables[name], self.tab_m[name], self.tab_v[name], a, mark=mark, **hp)

    # ------------------------------------------------------------------ inference
    def predict(self, users, items, out=None):
        """sigmoid output of the graph in inference mode (moving BN stats, no dropout)."""
        B = users.shape[0]
        if out is None:
            out = torch.empty(B, dtype=torch.float32, device=self.device)
        cfg, th = self.cfg, self.theta
        for s in range(0, B, self.max_batch):
            e = min(B, s + self.max_batch)
            n = e - s
            self._forward(users[s:e], items[s:e], n, False, 0, n)
            ops.neumf_head(self.a3[:n], self.dot[:n], None, th.view("W4"), th.view("b4"), cfg.mf_first, cfg.loss, 1.0,
                           prob=out[s:e])
        return out

    def evaluate_batch(self, users, items, labels):
        """inference-mode forward + loss/metric sums accumulated into self.msums (no grads)."""
        cfg, th = self.cfg, self.theta
        B = users.shape[0]
        self._forward(users, items, B, False, 0, B)
        ops.neumf_head(self.a3[:B], self.dot[:B], labels, th.view("W4"), th.view("b4"), cfg.mf_first, cfg.loss, 1.0 / B,
                       logit=self.logit[:B], prob=self.prob[:B], sums=self.msums)

    def pop_metrics(self, n_samples: int) -> dict:
        """Host sync: mean loss / mse / mae / binary_accuracy since the last call (RModel.py:20)."""
        s = self.msums.cpu().tolist()
        self.msums.zero_()
        n = max(1, n_samples)
        return {"loss": s[0] / n, "mse": s[1] / n, "mae": s[2] / n, "binary_accuracy": s[3] / n}

    def check_ids(self):
        ops.raise_if_flag(self.err)

    # ------------------------------------------------------------------ state
    def state_dict(self) -> dict:
        sd = {"t": self.t, "theta": self.theta.buf, "adam_m": self.adam_m.buf, "adam_v": self.adam_v.buf}
        for k in TABLES:
            sd[k], sd[k + ".m"], sd[k + ".v"] = self.tables[k], self.tab_m[k], self.tab_v[k]
        sd.update(self.moving)
        return sd

    def load_state_dict(self, sd: dict):
        self.t = int(sd["t"])
        for k, dst in (("theta", self.theta.buf), ("adam_m", self.adam_m.buf), ("adam_v", self.adam_v.buf)):
            dst.copy_(sd[k])
        for k in TABLES:
            self.tables[k].copy_(sd[k]); self.tab_m[k].copy_(sd[k + ".m"]); self.tab_v[k].copy_(sd[k + ".v"])
        for k in self.moving:
            self.moving[k].copy_(sd[k])

    def load_numpy_params(self, p: dict):
        """Load a parameter dict in the oracle's naming (tests / golden fixtures)."""
        for k in TABLES:
            self.tables[k].copy_(torch.as_tensor(p[k], dtype=torch.float32))
        for k in DENSE_ORDER:
            self.theta.view(k).copy_(torch.as_tensor(p[k], dtype=torch.float32).reshape(self.theta.view(k).shape))
        for k in self.moving:
            self.moving[k].copy_(torch.as_tensor(p[k], dtype=torch.float32))
