"""CPU ORACLE, part 2 (test infrastructure only): torch-CPU fp32 *autograd* restatement.

PARITY UNPINNED (see oracle/binrec_oracle.py header): the reference pins no numbers.

Two jobs:
  1. cross-check oracle/binrec_oracle.py's hand-derived backward against autograd;
  2. bench.py's `cpu_baseline` leg ("kind": "port"): one trainers/NFC_plain.py
     training step (NeuMF-A graph :107-155, BCE from logits, TF-form dense Adam) timed
     on the host cores.  The reference's own Python cannot run here or on the GPU box
     (TensorFlow absent, no network) and never leaves this container.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import math

import torch


def _act(z, act):
    if act == "sigmoid":
        return torch.sigmoid(z)
    if act == "relu":
        return torch.relu(z)
    return z


def _bn_train(a, g, b, eps):
    mu = a.mean(dim=0)
    var = a.var(dim=0, unbiased=False)
    return g * (a - mu) * torch.rsqrt(var + eps) + b, mu, var


def neumf_forward_torch(spec, p, users, items, masks=None, training=True):
    """p: dict name -> torch tensor (requires_grad where wanted). Mirrors
    binrec_oracle.neumf_forward (NFC_plain.py:107-152 / NeuMFModel.py:53-83)."""
    keep = 1.0 - spec.dropout
    e = {k: p[k][users if k.startswith("user") else items] for k in ("user_mlp", "item_mlp", "user_mf", "item_mf")}
    x = torch.cat([e[spec.mlp_concat[0] + "_mlp"], e[spec.mlp_concat[1] + "_mlp"]], dim=1)

    def drop(x, i):
        if training and masks is not None and masks[i] is not None:
            return x * masks[i].to(x.dtype) * (1.0 / keep)
        return x

    stats = {}
    x = drop(x, 0)
    a1 = _act(x @ p["W1"] + p["b1"], spec.act)
    if training:
        h1, stats["mu1"], stats["var1"] = _bn_train(a1, p["g1"], p["be1"], spec.bn_eps)
    else:
        h1 = p["g1"] * (a1 - p["mm1"]) * torch.rsqrt(p["mv1"] + spec.bn_eps) + p["be1"]
    x = drop(h1, 1)
    a2 = _act(x @ p["W2"] + p["b2"], spec.act)
    if training:
        h2, stats["mu2"], stats["var2"] = _bn_train(a2, p["g2"], p["be2"], spec.bn_eps)
    else:
        h2 = p["g2"] * (a2 - p["mm2"]) * torch.rsqrt(p["mv2"] + spec.bn_eps) + p["be2"]
    x = drop(h2, 2)
    a3 = _act(x @ p["W3"] + p["b3"], spec.act)
    dot = (e["user_mf"] * e["item_mf"]).sum(dim=1, keepdim=True)
    comb = torch.cat([dot, a3], dim=1) if spec.head_concat[0] == "mf" else torch.cat([a3, dot], dim=1)
    z = comb @ p["W4"] + p["b4"][0]
    return z, e, stats


def neumf_loss_torch(spec, z, y):
    if spec.loss == "bce":
        return torch.nn.functional.binary_cross_entropy_with_logits(z, y, reduction="mean")
    return ((torch.sigmoid(z) - y) ** 2).mean()


def neumf_autograd(spec, p_np, users, items, labels, masks=None, dtype=torch.float64):
    """Returns loss, logits, dense grads dict, per-batch row grads dict (via retain_grad on
    the gathered rows) — same contract as binrec_oracle.neumf_step_grads."""
    from .binrec_oracle import DENSE_ORDER
    p = {k: torch.tensor(v, dtype=dtype) for k, v in p_np.items()}
    for k in DENSE_ORDER:
        p[k].requires_grad_(True)
    u = torch.as_tensor(users, dtype=torch.long)
    i = torch.as_tensor(items, dtype=torch.long)
    m = None if masks is None else [None if mm is None else torch.as_tensor(mm) for mm in masks]
    # gather outside autograd, then make the gathered rows leaves
    keys = ("user_mlp", "item_mlp", "user_mf", "item_mf")
    rows = {k: p[k][u if k.startswith("user") else i].clone().requires_grad_(True) for k in keys}

    class _Tbl:  # table stand-in whose __getitem__ returns the pre-gathered leaf
        def __init__(self, t):
            self.t = t

        def __getitem__(self, _):
            return self.t

    q = dict(p)
    for k in keys:
        q[k] = _Tbl(rows[k])
    z, _, _ = neumf_forward_torch(spec, q, u, i, masks=m, training=True)
    loss = neumf_loss_torch(spec, z, torch.as_tensor(labels, dtype=dtype))
    loss.backward()
    g = {k: p[k].grad.numpy() for k in DENSE_ORDER}
    rg = {k: rows[k].grad.numpy() for k in keys}
    return float(loss.detach()), z.detach().numpy(), g, rg


# ----------------------------------------------------------------------------
# cpu_baseline: NFC_plain-style training step, torch CPU fp32, all host threads
# ----------------------------------------------------------------------------
class NFCPlainCpuStep:
    """One `model.fit` step of trainers/NFC_plain.py:107-165 restated on torch-CPU fp32:
    4 embeddings, [item,user] concat, Dropout .2 -> Dense 100 sigmoid -> BN -> Dropout ->
    Dense 50 sigmoid -> BN -> Dropout -> Dense 10 sigmoid, Dot, concat [mf, mlp], Dense 1,
    BCE from logits, Keras-Adam (dense m/v decay over the whole table, eps outside)."""

    def __init__(self, spec, num_users, num_items, lr=0.005, seed=1, lazy_adam=False):
        from .binrec_oracle import DENSE_ORDER
        g = torch.Generator().manual_seed(seed)
        D = spec.dim
        self.spec, self.lr, self.t, self.lazy = spec, lr, 0, lazy_adam
        self.p = {}
        for name, rows in (("user_mlp", num_users), ("item_mlp", num_items), ("user_mf", num_users), ("item_mf", num_items)):
            self.p[name] = (torch.rand(rows, D, generator=g) * 0.1 - 0.05).requires_grad_(True)
        for k, shp in spec.dense_shapes.items():
            if k.startswith("W"):
                fi, fo = (shp[0], shp[1]) if len(shp) == 2 else (shp[0], 1)
                lim = math.sqrt(6.0 / (fi + fo))
                self.p[k] = ((torch.rand(*shp, generator=g) * 2 - 1) * lim).requires_grad_(True)
            elif k.startswith("g"):
                self.p[k] = torch.ones(*shp, requires_grad=True)
            else:
                self.p[k] = torch.zeros(*shp, requires_grad=True)
        self.train_keys = ["user_mlp", "item_mlp", "user_mf", "item_mf"] + list(DENSE_ORDER)
        self.m = {k: torch.zeros_like(self.p[k]) for k in self.train_keys}
        self.v = {k: torch.zeros_like(self.p[k]) for k in self.train_keys}
        n1, n2, _ = spec.hidden
        self.mm = [torch.zeros(n1), torch.zeros(n2)]
        self.mv = [torch.ones(n1), torch.ones(n2)]

    def step(self, users, items, labels):
        spec = self.spec
        B = users.shape[0]
        masks = [torch.rand(B, w) >= spec.dropout for w in (2 * spec.dim, spec.hidden[0], spec.hidden[1])]
        z, _, st = neumf_forward_torch(spec, self.p, users, items, masks=masks, training=True)
        loss = neumf_loss_torch(spec, z, labels)
        for k in self.train_keys:
            self.p[k].grad = None
        loss.backward()
        self.t += 1
        a = self.lr * math.sqrt(1 - 0.999 ** self.t) / (1 - 0.9 ** self.t)
        with torch.no_grad():
            for k in self.train_keys:
                g = self.p[k].grad
                if self.lazy and k in ("user_mlp", "item_mlp", "user_mf", "item_mf"):
                    idx = torch.unique(users if k.startswith("user") else items)
                    gi = g[idx]
                    m = self.m[k][idx].mul_(0.9).add_(gi, alpha=0.1)
                    v = self.v[k][idx].mul_(0.999).addcmul_(gi, gi, value=0.001)
                    self.m[k][idx] = m
                    self.v[k][idx] = v
                    self.p[k][idx] -= a * m / (v.sqrt() + 1e-7)
                else:
                    self.m[k].mul_(0.9).add_(g, alpha=0.1)
                    self.v[k].mul_(0.999).addcmul_(g, g, value=0.001)
                    self.p[k].addcdiv_(self.m[k], self.v[k].sqrt().add_(1e-7), value=-a)
            mom = spec.bn_momentum
            self.mm[0].mul_(mom).add_(st["mu1"].detach(), alpha=1 - mom)
            self.mv[0].mul_(mom).add_(st["var1"].detach(), alpha=1 - mom)
            self.mm[1].mul_(mom).add_(st["mu2"].detach(), alpha=1 - mom)
            self.mv[1].mul_(mom).add_(st["var2"].detach(), alpha=1 - mom)
        return float(loss.detach())
