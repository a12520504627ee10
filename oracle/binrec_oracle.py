"""CPU ORACLE (test infrastructure only) for the NeuMF / BPR / TwoTower hot path.

PARITY UNPINNED: the reference (leotimus/binary-recommendation) delegates every
arithmetic op on this path to un-vendored `tensorflow>=2.3.1` (requirements.txt:1)
and an unpinned `tensorflow_recommenders` (trainers/twoTower.py:10); its test/
scripts hold no assertion, seed, fixture or golden vector.
This file is therefore a restatement, written by this build, of the algorithm the
reference *declares* plus the TF-2.3-era Keras semantics it relies on ([TF-sem]).
It is cross-checked against torch-CPU autograd (oracle/torch_ref.py), not against
the reference itself.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module. The product package never does.

Everything is plain numpy; `dt` selects float64 (tolerance anchor) or float32
(bit-level anchor for sequentially-ordered sums such as the duplicate-id
segment sum).

Reference call sites restated here (path:line under /root/reference):
  NeuMF-A graph .............. trainers/NFC_plain.py:107-155, fit :165
  NeuMF-B graph .............. src/models/NeuMFModel.py:53-100
  NeuMF-B negative sampler ... src/models/NeuMFModel.py:102-123
  BPR triplet loss ........... src/models/BPRModel.py:49-74,124-144 ; src/models/bpr.py:141-157
  bpr_predict ................ src/models/bpr.py:122-133
  TwoTower ................... trainers/twoTower.py:19-111
  top-k + HR@k ............... trainers/topKmetrics.py:51-99
"""
from __future__ import annotations

import numpy as np

# ----------------------------------------------------------------------------
# Philox4x32-10 counter RNG: the dropout-mask definition shared (bit-exactly)
# with csrc/philox.h.  Keras' own dropout stream (NFC_plain.py:138,141,144;
# NeuMFModel.py:67,71,75) is a stateful TF generator that cannot be reproduced,
# so the build defines its own: one Philox call yields 4 x u32 = 8 x u16 draws
# for the 8 columns 8q..8q+7 of a row.  Element (row r, col c) of dropout site
# `site` at optimizer step `step` is KEPT iff
#   u16 = (philox(key=(seed_lo, seed_hi), ctr=(r, c>>3, site, step))[(c>>1)&3] >> (16*(c&1))) & 0xFFFF
#   u16 >= floor(p * 65536)
# ----------------------------------------------------------------------------
PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = np.uint64(0x9E3779B9)
PHILOX_W1 = np.uint64(0xBB67AE85)
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10. All inputs uint32-valued arrays (broadcastable)."""
    c0 = np.asarray(c0, dtype=np.uint64) & _MASK32
    c1 = np.asarray(c1, dtype=np.uint64) & _MASK32
    c2 = np.asarray(c2, dtype=np.uint64) & _MASK32
    c3 = np.asarray(c3, dtype=np.uint64) & _MASK32
    k0 = np.uint64(k0) & _MASK32
    k1 = np.uint64(k1) & _MASK32
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    for _ in range(10):
        p0 = PHILOX_M0 * c0
        p1 = PHILOX_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
        n0 = (hi1 ^ c1 ^ k0) & _MASK32
        n1 = lo1
        n2 = (hi0 ^ c3 ^ k1) & _MASK32
        n3 = lo0
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + PHILOX_W0) & _MASK32
        k1 = (k1 + PHILOX_W1) & _MASK32
    return (c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32))


def dropout_threshold(p: float) -> int:
    """u16 draw < threshold  => element dropped."""
    return min(int(np.floor(float(p) * 65536.0)), 0xFFFF)


def dropout_mask(seed: int, step: int, site: int, nrows: int, ncols: int, p: float, row0: int = 0):
    """(nrows, ncols) bool keep-mask. row0 = global row offset (data-parallel shards
    draw the mask of their *global* batch rows, so 1-GPU == N-GPU)."""
    if p <= 0.0:
        return np.ones((nrows, ncols), dtype=bool)
    r = (np.arange(nrows, dtype=np.uint64) + np.uint64(row0))[:, None]
    c = np.arange(ncols, dtype=np.uint64)[None, :]
    out = philox4x32_10(r, c >> np.uint64(3), np.uint64(site), np.uint64(step),
                        seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    word = ((c >> np.uint64(1)) & np.uint64(3)).astype(np.int64)
    word = np.broadcast_to(word, (nrows, ncols))
    stacked = np.stack(out, axis=-1)  # (nrows, ncols, 4)
    w32 = np.take_along_axis(stacked, word[..., None], axis=-1)[..., 0].astype(np.uint32)
    half = np.broadcast_to((c & np.uint64(1)).astype(np.uint32), (nrows, ncols))
    draw = (w32 >> (np.uint32(16) * half)) & np.uint32(0xFFFF)
    return draw >= np.uint32(dropout_threshold(p))


# ----------------------------------------------------------------------------
# G1 gather / M1 dot / S1 scatter
# ----------------------------------------------------------------------------
def gather_rows(table, ids):
    """out[b,:] = table[ids[b],:]  — Embedding lookup (NFC_plain.py:115-126,
    NeuMFModel.py:58-63, BPRModel.py:55-60, twoTower.py:34,36). Bit-exact row copy.
    Out-of-range ids raise (TF-CPU raises InvalidArgument [TF-sem])."""
    ids = np.asarray(ids)
    if ids.size and (ids.min() < 0 or ids.max() >= table.shape[0]):
        raise IndexError("embedding id out of range")
    return table[ids.astype(np.int64)]


def row_dot(a, b):
    """Dot(axes=1) (NFC_plain.py:148, NeuMFModel.py:79) — (B,D),(B,D)->(B,)."""
    return np.sum(a * b, axis=1)


def dedup_rows_sequential(ids, rows, dt=np.float32):
    """[TF-sem] optimizer `_deduplicate_indexed_slices`: unique ids, duplicates summed.
    Sum order = ascending batch position (what a sequential unsorted_segment_sum does);
    returned sorted by id.  In float32 this is the bit-level anchor for the HIP
    sort + ordered segment-sum kernel."""
    ids = np.asarray(ids).astype(np.int64)
    rows = np.asarray(rows, dtype=dt)
    order = np.argsort(ids, kind="stable")
    sid = ids[order]
    uniq, start = np.unique(sid, return_index=True)
    out = np.zeros((len(uniq), rows.shape[1]), dtype=dt)
    ends = list(start[1:]) + [len(sid)]
    for k, (s, e) in enumerate(zip(start, ends)):
        acc = rows[order[s]].astype(dt).copy()
        for j in range(s + 1, e):
            acc = (acc + rows[order[j]]).astype(dt)
        out[k] = acc
    return uniq, out


SEG_BLOCK = 64   # include/binrec.h "The ordered duplicate sum and its scratch"


def ordered_segment_sum(ids, rows, dt=np.float32, block=SEG_BLOCK):
    """The HIP kernels' two-level ordered duplicate sum (seg_ws != NULL), restated: in the stable-sorted order, the head
    of a segment [s, e) adds positions s+1 .. up to the next multiple of `block` one by one; every later block start b
    (multiple of `block`, b < e) contributes P_b = the one-by-one sum of positions b .. min(b + block, e) - 1.
    Segments that do not cross a block boundary equal dedup_rows_sequential bit for bit."""
    ids = np.asarray(ids).astype(np.int64)
    rows = np.asarray(rows, dtype=dt)
    order = np.argsort(ids, kind="stable")
    sid = ids[order]
    uniq, start = np.unique(sid, return_index=True)
    out = np.zeros((len(uniq), rows.shape[1]), dtype=dt)
    ends = list(start[1:]) + [len(sid)]
    for k, (s, e) in enumerate(zip(start, ends)):
        own_end = min(e, (s // block + 1) * block)
        acc = rows[order[s]].astype(dt).copy()
        for j in range(s + 1, own_end):
            acc = (acc + rows[order[j]]).astype(dt)
        for b in range(own_end, e, block):
            part = rows[order[b]].astype(dt).copy()
            for j in range(b + 1, min(b + block, e)):
                part = (part + rows[order[j]]).astype(dt)
            acc = (acc + part).astype(dt)
        out[k] = acc
    return uniq, out


def dedup_rows_unordered(ids, rows, dt=np.float64):
    """Same unique-id sums as dedup_rows_sequential, accumulated by np.add.at (no Python loop): the add order is
    numpy's, so this is a float64 TOLERANCE anchor for full-size batches, never the fp32 bit anchor."""
    ids = np.asarray(ids).astype(np.int64)
    uniq, inv = np.unique(ids, return_inverse=True)
    out = np.zeros((len(uniq), rows.shape[1]), dtype=dt)
    np.add.at(out, inv, np.asarray(rows, dtype=dt))
    return uniq, out


def scatter_add_dense(nrows, ids, rows, dt=np.float64):
    """g_table[ids[b],:] += rows[b,:] (np.add.at) — dense view of the same sum."""
    g = np.zeros((nrows, rows.shape[1]), dtype=dt)
    np.add.at(g, np.asarray(ids).astype(np.int64), rows.astype(dt))
    return g


# ----------------------------------------------------------------------------
# activations / losses
# ----------------------------------------------------------------------------
def sigmoid(x):
    x = np.asarray(x)
    e = np.exp(-np.abs(x))
    return np.where(x >= 0, 1.0 / (1.0 + e), e / (1.0 + e)).astype(x.dtype)


def act_fwd(z, act):
    if act == "sigmoid":
        return sigmoid(z)
    if act == "relu":
        return np.maximum(z, 0).astype(z.dtype)
    if act == "linear":
        return z
    raise ValueError(act)


def act_bwd_from_out(a, act):
    """d act / d z expressed with the activation OUTPUT a (what the kernels keep)."""
    if act == "sigmoid":
        return a * (1 - a)
    if act == "relu":
        return (a > 0).astype(a.dtype)
    if act == "linear":
        return np.ones_like(a)
    raise ValueError(act)


def bce_from_logits(z, y):
    """[TF-sem] Keras BinaryCrossentropy on a terminal Sigmoid in graph mode
    (NFC_plain.py:152-155; twoTower.py:86-87,209): max(z,0) - z*y + log1p(exp(-|z|)),
    mean over batch.  Returns (loss, dloss/dz)."""
    B = z.shape[0]
    per = np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z)))
    return per.mean(dtype=z.dtype), ((sigmoid(z) - y) / B).astype(z.dtype)


def mse_on_sigmoid(z, y):
    """'mean_squared_error' on sigmoid output (NeuMFModel.py:83,90). (loss, dloss/dz)."""
    B = z.shape[0]
    p = sigmoid(z)
    return ((p - y) ** 2).mean(dtype=z.dtype), (2.0 * (p - y) * p * (1 - p) / B).astype(z.dtype)


# ----------------------------------------------------------------------------
# NeuMF (variants A = trainers/NFC_plain.py, B = src/models/NeuMFModel.py)
# ----------------------------------------------------------------------------
class NeuMFSpec:
    """One parametrisation covering both reference graphs."""

    def __init__(self, variant="A", dim=10, hidden=None, dropout=0.2, bn_eps=1e-3, bn_momentum=0.99):
        assert variant in ("A", "B")
        self.variant = variant
        self.dim = dim
        if variant == "A":
            # NFC_plain.py:137 concat [item_mlp, user_mlp]; :139,142,147 Dense 100/50/10 sigmoid;
            # :148 Dot([item_mf,user_mf]); :149 concat [pred_mf, pred_mlp]; :155 BCE.
            self.hidden = tuple(hidden) if hidden else (100, 50, 10)
            self.act = "sigmoid"
            self.mlp_concat = ("item", "user")
            self.head_concat = ("mf", "mlp")
            self.loss = "bce"
        else:
            # NeuMFModel.py:66 concat [user_mlp,item_mlp]; :69,73,78 Dense F, F//2, F//4 relu;
            # :79 Dot([user_mf,item_mf]); :80 concat [mlp, mf]; :90 MSE.
            self.hidden = tuple(hidden) if hidden else (dim, dim // 2, dim // 4)
            self.act = "relu"
            self.mlp_concat = ("user", "item")
            self.head_concat = ("mlp", "mf")
            self.loss = "mse"
        self.dropout = dropout
        self.bn_eps = bn_eps
        self.bn_momentum = bn_momentum

    @property
    def dense_shapes(self):
        n1, n2, n3 = self.hidden
        return {"W1": (2 * self.dim, n1), "b1": (n1,), "g1": (n1,), "be1": (n1,),
                "W2": (n1, n2), "b2": (n2,), "g2": (n2,), "be2": (n2,),
                "W3": (n2, n3), "b3": (n3,), "W4": (n3 + 1,), "b4": (1,)}


DENSE_ORDER = ("W1", "b1", "g1", "be1", "W2", "b2", "g2", "be2", "W3", "b3", "W4", "b4")


def neumf_init(spec, num_user_rows, num_item_rows, seed=0, dt=np.float32):
    """[TF-sem] Keras defaults: Embedding U(-0.05,0.05); Dense glorot-uniform, bias 0;
    BN gamma 1, beta 0, moving mean 0, moving var 1.  (numpy RNG: the reference has no seed
    on this path, so only the *distribution* is restated.)"""
    rng = np.random.default_rng(seed)
    D = spec.dim
    p = {}
    for name, rows in (("user_mlp", num_user_rows), ("item_mlp", num_item_rows),
                       ("user_mf", num_user_rows), ("item_mf", num_item_rows)):
        p[name] = rng.uniform(-0.05, 0.05, size=(rows, D)).astype(dt)
    for k, shp in spec.dense_shapes.items():
        if k.startswith("W"):
            fan_in, fan_out = (shp[0], shp[1]) if len(shp) == 2 else (shp[0], 1)
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            p[k] = rng.uniform(-lim, lim, size=shp).astype(dt)
        elif k.startswith("g"):
            p[k] = np.ones(shp, dtype=dt)
        else:
            p[k] = np.zeros(shp, dtype=dt)
    n1, n2, _ = spec.hidden
    p["mm1"], p["mv1"] = np.zeros(n1, dt), np.ones(n1, dt)
    p["mm2"], p["mv2"] = np.zeros(n2, dt), np.ones(n2, dt)
    return p


def _bn_train(a, gamma, beta, eps):
    mu = a.mean(axis=0)
    var = ((a - mu) ** 2).mean(axis=0)  # biased batch variance [TF-sem]
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (a - mu) * rstd
    return gamma * xhat + beta, (mu, var, rstd, xhat)


def neumf_forward(spec, p, users, items, training=False, masks=None, dt=np.float64):
    """Forward pass. masks = (m0,m1,m2) bool keep-masks for the three Dropout sites
    (training only; None => no dropout).  Returns dict of every intermediate."""
    f = lambda x: np.asarray(x, dtype=dt)
    keep = 1.0 - spec.dropout
    e = {"user_mlp": f(gather_rows(p["user_mlp"], users)), "item_mlp": f(gather_rows(p["item_mlp"], items)),
         "user_mf": f(gather_rows(p["user_mf"], users)), "item_mf": f(gather_rows(p["item_mf"], items))}
    x0 = np.concatenate([e[spec.mlp_concat[0] + "_mlp"], e[spec.mlp_concat[1] + "_mlp"]], axis=1)
    c = {"e": e, "x0": x0}

    def drop(x, i):
        if training and masks is not None and masks[i] is not None:
            return x * masks[i].astype(dt) * dt(1.0 / keep)
        return x

    x0d = drop(x0, 0)
    z1 = x0d @ f(p["W1"]) + f(p["b1"]); a1 = act_fwd(z1, spec.act)
    if training:
        h1, bn1 = _bn_train(a1, f(p["g1"]), f(p["be1"]), dt(spec.bn_eps))
    else:
        rstd = 1.0 / np.sqrt(f(p["mv1"]) + dt(spec.bn_eps))
        h1, bn1 = f(p["g1"]) * (a1 - f(p["mm1"])) * rstd + f(p["be1"]), None
    x1d = drop(h1, 1)
    z2 = x1d @ f(p["W2"]) + f(p["b2"]); a2 = act_fwd(z2, spec.act)
    if training:
        h2, bn2 = _bn_train(a2, f(p["g2"]), f(p["be2"]), dt(spec.bn_eps))
    else:
        rstd = 1.0 / np.sqrt(f(p["mv2"]) + dt(spec.bn_eps))
        h2, bn2 = f(p["g2"]) * (a2 - f(p["mm2"])) * rstd + f(p["be2"]), None
    x2d = drop(h2, 2)
    z3 = x2d @ f(p["W3"]) + f(p["b3"]); a3 = act_fwd(z3, spec.act)
    dot = row_dot(e["user_mf"], e["item_mf"])
    if spec.head_concat[0] == "mf":
        comb = np.concatenate([dot[:, None], a3], axis=1)
    else:
        comb = np.concatenate([a3, dot[:, None]], axis=1)
    z = comb @ f(p["W4"]) + f(p["b4"])[0]
    c.update(x0d=x0d, a1=a1, bn1=bn1, h1=h1, x1d=x1d, a2=a2, bn2=bn2, h2=h2, x2d=x2d, a3=a3,
             dot=dot, comb=comb, logit=z, prob=sigmoid(z))
    return c


def neumf_loss(spec, z, y):
    return bce_from_logits(z, y) if spec.loss == "bce" else mse_on_sigmoid(z, y)


def _bn_bwd(dh, gamma, bn):
    mu, var, rstd, xhat = bn
    dgamma = (dh * xhat).sum(axis=0)
    dbeta = dh.sum(axis=0)
    B = dh.shape[0]
    da = (gamma * rstd) * (dh - dbeta / B - xhat * (dgamma / B))
    return da, dgamma, dbeta


def neumf_step_grads(spec, p, users, items, labels, masks=None, dt=np.float64):
    """One training forward + analytic backward (B1). Returns (loss, cache, dense_grads dict,
    row_grads dict name->(B,D) aligned with the batch, new BN moving stats)."""
    f = lambda x: np.asarray(x, dtype=dt)
    keep = 1.0 - spec.dropout
    c = neumf_forward(spec, p, users, items, training=True, masks=masks, dt=dt)
    y = f(labels)
    loss, dz = neumf_loss(spec, c["logit"], y)
    g = {}
    W4 = f(p["W4"])
    g["W4"] = c["comb"].T @ dz
    g["b4"] = np.array([dz.sum()], dtype=dt)
    dcomb = dz[:, None] * W4[None, :]
    if spec.head_concat[0] == "mf":
        ddot, da3 = dcomb[:, 0], dcomb[:, 1:]
    else:
        da3, ddot = dcomb[:, :-1], dcomb[:, -1]

    def undrop(dx, i):
        if masks is not None and masks[i] is not None:
            return dx * masks[i].astype(dt) * dt(1.0 / keep)
        return dx

    dz3 = da3 * act_bwd_from_out(c["a3"], spec.act)
    g["W3"] = c["x2d"].T @ dz3; g["b3"] = dz3.sum(axis=0)
    dh2 = undrop(dz3 @ f(p["W3"]).T, 2)
    da2, g["g2"], g["be2"] = _bn_bwd(dh2, f(p["g2"]), c["bn2"])
    dz2 = da2 * act_bwd_from_out(c["a2"], spec.act)
    g["W2"] = c["x1d"].T @ dz2; g["b2"] = dz2.sum(axis=0)
    dh1 = undrop(dz2 @ f(p["W2"]).T, 1)
    da1, g["g1"], g["be1"] = _bn_bwd(dh1, f(p["g1"]), c["bn1"])
    dz1 = da1 * act_bwd_from_out(c["a1"], spec.act)
    g["W1"] = c["x0d"].T @ dz1; g["b1"] = dz1.sum(axis=0)
    dx0 = undrop(dz1 @ f(p["W1"]).T, 0)
    D = spec.dim
    rg = {spec.mlp_concat[0] + "_mlp": dx0[:, :D], spec.mlp_concat[1] + "_mlp": dx0[:, D:],
          "user_mf": ddot[:, None] * c["e"]["item_mf"], "item_mf": ddot[:, None] * c["e"]["user_mf"]}
    # Magnitude propagation ("what fp32 rounding is relative to"): the same backward with every
    # subtraction replaced by a sum of absolute values.  Several true gradients cancel almost
    # completely (BN-backward removes mean and slope; pre-BN biases), so tests bound the error by
    # 1e-5 x these magnitudes instead of by the cancelled result.  Test tolerances only.
    A = np.abs

    def bn_abs(dh_abs, gamma, bn):
        _mu, _var, rstd, xhat = bn
        return A(gamma * rstd) * (dh_abs + dh_abs.mean(0) + A(xhat) * (dh_abs * A(xhat)).mean(0))

    dz3_abs = A(dz3)
    dh2_abs = undrop(dz3_abs @ A(f(p["W3"])).T, 2)
    dz2_abs = bn_abs(dh2_abs, f(p["g2"]), c["bn2"]) * act_bwd_from_out(c["a2"], spec.act)
    dh1_abs = undrop(dz2_abs @ A(f(p["W2"])).T, 1)
    dz1_abs = bn_abs(dh1_abs, f(p["g1"]), c["bn1"]) * act_bwd_from_out(c["a1"], spec.act)
    c["gabs"] = {"W4": A(c["comb"]).T @ A(dz), "b4": np.array([A(dz).sum()]),
                 "W3": A(c["x2d"]).T @ dz3_abs, "b3": dz3_abs.sum(0), "g2": (dh2_abs * A(c["bn2"][3])).sum(0), "be2": dh2_abs.sum(0),
                 "W2": A(c["x1d"]).T @ dz2_abs, "b2": dz2_abs.sum(0), "g1": (dh1_abs * A(c["bn1"][3])).sum(0), "be1": dh1_abs.sum(0),
                 "W1": A(c["x0d"]).T @ dz1_abs, "b1": dz1_abs.sum(0)}
    c["rg_abs"] = undrop(dz1_abs @ A(f(p["W1"])).T, 0)
    mom = dt(spec.bn_momentum)
    new_stats = {"mm1": f(p["mm1"]) * mom + c["bn1"][0] * (1 - mom), "mv1": f(p["mv1"]) * mom + c["bn1"][1] * (1 - mom),
                 "mm2": f(p["mm2"]) * mom + c["bn2"][0] * (1 - mom), "mv2": f(p["mv2"]) * mom + c["bn2"][1] * (1 - mom)}
    return loss, c, g, rg, new_stats


def keras_metrics(prob, y):
    """RModel.METRICS = ['mse','mae','binary_accuracy'] (RModel.py:20)."""
    return {"mse": float(np.mean((prob - y) ** 2)), "mae": float(np.mean(np.abs(prob - y))),
            "binary_accuracy": float(np.mean((prob > 0.5) == (y > 0.5)))}


# ----------------------------------------------------------------------------
# optimizers (O1 TF-form Adam, O2 Keras Adagrad)
# ----------------------------------------------------------------------------
def adam_alpha(lr, t, b1=0.9, b2=0.999):
    """[TF-sem] alpha_t = lr*sqrt(1-b2^t)/(1-b1^t), t = 1 on the first step."""
    return lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)


def adam_dense(theta, m, v, g, lr, t, b1=0.9, b2=0.999, eps=1e-7, dt=np.float64):
    """[TF-sem] Keras Adam: eps OUTSIDE the bias correction (!= torch.optim.Adam)."""
    a = dt(adam_alpha(lr, t, b1, b2))
    m = (dt(b1) * m + dt(1 - b1) * g).astype(dt)
    v = (dt(b2) * v + dt(1 - b2) * g * g).astype(dt)
    theta = (theta - a * m / (np.sqrt(v) + dt(eps))).astype(dt)
    return theta, m, v


def adam_sparse_tf(theta, m, v, ids, row_grads, lr, t, lazy=False, b1=0.9, b2=0.999, eps=1e-7, dt=np.float64, dedup=None):
    """[TF-sem] Keras Adam `_resource_apply_sparse` after `_deduplicate_indexed_slices`:
    duplicates summed FIRST (so v gets (sum g)^2), then m,v decayed over the WHOLE table and
    the WHOLE table updated (non-lazy).  lazy=True touches only the gathered rows
    (documented throughput deviation, not the reference's semantics)."""
    theta, m, v = theta.astype(dt).copy(), m.astype(dt).copy(), v.astype(dt).copy()
    uniq, gs = (dedup or dedup_rows_sequential)(ids, row_grads, dt=dt)
    if lazy:
        theta[uniq], m[uniq], v[uniq] = adam_dense(theta[uniq], m[uniq], v[uniq], gs, lr, t, b1, b2, eps, dt)
        return theta, m, v
    g = np.zeros_like(theta)
    g[uniq] = gs
    return adam_dense(theta, m, v, g, lr, t, b1, b2, eps, dt)


def adagrad_dense(theta, acc, g, lr, eps=1e-7, dt=np.float64):
    """[TF-sem] Keras Adagrad (initial_accumulator_value 0.1): acc += g^2 ; theta -= lr*g/(sqrt(acc)+eps)."""
    acc = (acc + g * g).astype(dt)
    theta = (theta - dt(lr) * g / (np.sqrt(acc) + dt(eps))).astype(dt)
    return theta, acc


def adagrad_sparse(theta, acc, ids, row_grads, lr, eps=1e-7, dt=np.float64):
    """Sparse apply touches only the (deduplicated) rows [TF-sem]."""
    theta, acc = theta.astype(dt).copy(), acc.astype(dt).copy()
    uniq, gs = dedup_rows_sequential(ids, row_grads, dt=dt)
    theta[uniq], acc[uniq] = adagrad_dense(theta[uniq], acc[uniq], gs, lr, eps, dt)
    return theta, acc


# ----------------------------------------------------------------------------
# BPR (src/models/BPRModel.py:49-74,124-144; src/models/bpr.py:141-157)
# ----------------------------------------------------------------------------
def bpr_step_grads(user_table, item_table, u, pos, neg, dt=np.float64):
    """x = u.p - u.n ; l = 1 - sigmoid(x) (NOT -log sigmoid: BPRModel.py:144); loss = mean l
    (identityLoss, BPRModel.py:124-126).  Returns loss, per-triplet l, row grads (gu,gp,gn)."""
    f = lambda x: np.asarray(x, dtype=dt)
    eu, ep, en = f(gather_rows(user_table, u)), f(gather_rows(item_table, pos)), f(gather_rows(item_table, neg))
    x = row_dot(eu, ep) - row_dot(eu, en)
    s = sigmoid(x)
    l = 1.0 - s
    B = x.shape[0]
    dx = (-s * (1.0 - s) / B).astype(dt)
    gu = dx[:, None] * (ep - en)
    gp = dx[:, None] * eu
    gn = -dx[:, None] * eu
    return l.mean(dtype=dt), l, (gu, gp, gn)


def bpr_predict(user_table, item_table, user_id, item_ids):
    """scores = user_vector . item_matrix^T (src/models/bpr.py:122-133)."""
    return item_table[np.asarray(item_ids)] @ user_table[user_id]


# ----------------------------------------------------------------------------
# TwoTower (trainers/twoTower.py:19-111)
# ----------------------------------------------------------------------------
MIN_FLOAT = float(np.finfo(np.float32).min) / 100.0  # [TF-sem] tfrs RemoveAccidentalHits


def twotower_embed(p, users, items, dt=np.float64):
    """computeEmb (twoTower.py:77-80): Embedding -> Dense(semb) linear (twoTower.py:40-41)."""
    f = lambda x: np.asarray(x, dtype=dt)
    eu, ei = f(gather_rows(p["user_emb"], users)), f(gather_rows(p["item_emb"], items))
    q = eu @ f(p["Wu"]) + f(p["bu"])
    c = ei @ f(p["Wi"]) + f(p["bi"])
    return eu, ei, q, c


def inbatch_softmax_loss(q, c, cand_ids):
    """[TF-sem, TFRS unpinned] tfrs.tasks.Retrieval(loss=None) with candidate_ids
    (twoTower.py:47,82-83): S = q c^T, labels = I, accidental hits
    (same candidate id, off-diagonal) get + MIN_FLOAT, loss = SUM_i [logsumexp_j S_ij - S_ii].
    Returns loss, dq, dc."""
    dt = q.dtype
    S = q @ c.T
    ids = np.asarray(cand_ids)
    dup = (ids[:, None] == ids[None, :]).astype(dt) - np.eye(len(ids), dtype=dt)
    S = (S + dup * MIN_FLOAT).astype(dt)
    mx = S.max(axis=1, keepdims=True)
    ex = np.exp(S - mx)
    den = ex.sum(axis=1, keepdims=True)
    lse = (mx + np.log(den))[:, 0]
    loss = (lse - np.diag(S)).sum()
    P = ex / den
    dS = P - np.eye(len(ids), dtype=dt)
    return loss, dS @ c, dS.T @ q


def inbatch_softmax_stripe(q, c, q_ids, c_ids, diag_offset, chunk=8192):
    """The data-parallel form of inbatch_softmax_loss: a stripe of Bq queries against Bc (all-gathered)
    candidates; query i's positive is candidate i + diag_offset (columns outside [0, Bc) have no positive in this stripe),
    accidental hits = same id, not the positive.  Column chunks, so Bq x Bc is never held.  Returns (loss over the
    stripe's rows, row lse, dq (Bq, d), dc contribution of these queries to every candidate (Bc, d))."""
    dt = q.dtype
    Bq, Bc = q.shape[0], c.shape[0]
    qi, ci = np.asarray(q_ids), np.asarray(c_ids)
    pos = np.arange(Bq) + diag_offset

    def tile(c0, c1):
        S = q @ c[c0:c1].T
        hit = (qi[:, None] == ci[None, c0:c1]) & (pos[:, None] != np.arange(c0, c1)[None, :])
        return (S + hit.astype(dt) * MIN_FLOAT).astype(dt)

    mx = np.full(Bq, -np.inf, dt)
    for c0 in range(0, Bc, chunk):
        mx = np.maximum(mx, tile(c0, min(Bc, c0 + chunk)).max(axis=1))
    den = np.zeros(Bq, dt)
    for c0 in range(0, Bc, chunk):
        den += np.exp(tile(c0, min(Bc, c0 + chunk)) - mx[:, None]).sum(axis=1)
    lse = mx + np.log(den)
    has = (pos >= 0) & (pos < Bc)
    sdiag = np.einsum("ij,ij->i", q[has], c[pos[has]])
    loss = lse[has].sum() - sdiag.sum()
    dq, dc = np.zeros_like(q), np.zeros_like(c)
    for c0 in range(0, Bc, chunk):
        c1 = min(Bc, c0 + chunk)
        P = np.exp(tile(c0, c1) - lse[:, None])
        inr = has & (pos >= c0) & (pos < c1)
        P[np.nonzero(inr)[0], pos[inr] - c0] -= 1.0
        dq += P @ c[c0:c1]
        dc[c0:c1] = P.T @ q
    return loss, lse, dq, dc


def twotower_step_grads(p, users, items, labels=None, rd_zero=False, dt=np.float64):
    """train_step (twoTower.py:89-102). rd_zero: sigmoid(dot(q,c)) + BCE (twoTower.py:85-87)."""
    f = lambda x: np.asarray(x, dtype=dt)
    eu, ei, q, c = twotower_embed(p, users, items, dt)
    if rd_zero:
        z = row_dot(q, c)
        loss, dz = bce_from_logits(z, f(labels))
        dq, dc = dz[:, None] * c, dz[:, None] * q
    else:
        loss, dq, dc = inbatch_softmax_loss(q, c, items)
    g = {"Wu": eu.T @ dq, "bu": dq.sum(axis=0), "Wi": ei.T @ dc, "bi": dc.sum(axis=0)}
    rg = {"user_emb": dq @ f(p["Wu"]).T, "item_emb": dc @ f(p["Wi"]).T}
    return loss, (q, c), g, rg


# ----------------------------------------------------------------------------
# E1/E2: top-k and HR@k (trainers/topKmetrics.py:51-99)
# ----------------------------------------------------------------------------
def topk_reference_order(scores, item_ids, k):
    """__topk/__insertSorted (topKmetrics.py:51-72) restated literally: seed with the first k,
    stable-sort descending, then for each later item replace the tail iff STRICTLY greater
    (ties keep the earlier item position) and insert before the first strictly-smaller entry
    scanning from the tail (so among equals the newcomer lands AFTER)."""
    l = [(float(s), i) for s, i in zip(scores, item_ids)]
    res = l[:k]
    res.sort(reverse=True, key=lambda x: x[0])
    for i in range(k, len(l)):
        if l[i][0] > res[-1][0]:
            res.pop()
            # __insertSorted: walk from the tail while the element above is < val
            j = len(res)
            cur = res[-1][0] if res else None
            while j > 0 and cur < l[i][0]:
                j -= 1
                cur = res[j - 1][0]  # j==0 reads res[-1] like the reference; loop ends on j>0
            res.insert(j, l[i])
    return res


def topk_metrics(predictions, positives, users_id, items_id):
    """topKMetrics (topKmetrics.py:74-99): hitRate = hits / len(usersId) over ALL users."""
    nbr_user, nbr_item = len(users_id), len(items_id)
    total = nbr_user * nbr_item
    real = set(positives)
    tp = fp = hits = 0
    for u, topk in predictions:
        hit = False
        for _r, i in topk:
            if (u, i) in real:
                tp += 1
                hit = True
            else:
                fp += 1
        if hit:
            hits += 1
    fn = len(real) - tp
    tn = total - tp - fp - fn
    return {"tp": tp, "tn": tn, "fp": fp, "fn": fn, "precision": tp / (tp + fp),
            "recall": tp / (tp + fn), "hitRate": hits / nbr_user}


# ----------------------------------------------------------------------------
# negative sampling / batch construction, restating csrc/sampling.hip bit for bit
#   NeuMFModel.bootstrapDataset          src/models/NeuMFModel.py:102-109
#   BPR triplets                         src/models/BPRModel.py:94-98,111-119 (sampled instead of enumerated)
#   generateNegativeFeedback             Data handling/synthetic.py:208-223,237-256
# [pandas-sem] the reference's samplers are unseeded pandas calls: the distributions are restated (rows sampled with
# replacement, a permuted item column, independently shuffled columns, distinct non-interacted pairs), not the streams.
# ----------------------------------------------------------------------------
def mix32(x):
    """murmur3 finaliser on uint32 arrays."""
    x = np.asarray(x, dtype=np.uint64) & _MASK32
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x85ebca6b)) & _MASK32
    x ^= x >> np.uint64(13); x = (x * np.uint64(0xc2b2ae35)) & _MASK32
    x ^= x >> np.uint64(16)
    return x


def perm_key(seed: int, stream: int):
    lo, hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    return [int(mix32((lo + 0x9E3779B9 * (r + 1)) & 0xFFFFFFFF)) ^ int(mix32(hi ^ ((stream * 0x85ebca6b + r) & 0xFFFFFFFF))) for r in range(4)]


def feistel_perm(x, M: int, key):
    """keyed bijection of [0, M): 4-round Feistel over ceil(log2 M) bits, cycle walking (vectorised over x)."""
    x = np.asarray(x, dtype=np.uint64).copy()
    if M <= 1:
        return np.zeros_like(x)
    b = 1
    while b < 32 and (1 << b) < M:
        b += 1
    hb = np.uint64((b + 1) >> 1)
    mask = np.uint64((1 << int(hb)) - 1)
    todo = np.ones(x.shape, dtype=bool)
    while todo.any():
        v = x[todo]
        L, R = v >> hb, v & mask
        for r in range(4):
            F = mix32(R ^ np.uint64(key[r])) & mask
            L, R = R, L ^ F
        v = (L << hb) | R
        x[todo] = v
        todo[todo] = v >= np.uint64(M)
    return x.astype(np.int64)


def draw_below(seed: int, c0, c1, stream: int, n: int):
    """uniform integer in [0, n): floor(philox(...).x * n / 2^32)."""
    d = philox4x32_10(c0, c1, np.uint64(stream), np.uint64(0), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)[0]
    return ((d.astype(np.uint64) * np.uint64(n)) >> np.uint64(32)).astype(np.int64)


def bootstrap_dataset(users, items, n_neg: int, seed: int):
    """brBootstrapDataset: positives (label 1) + n_neg rows sampled with replacement whose item column is permuted (label 0,
    no collision check: NeuMFModel.py:103-105), all shuffled (:109).  -> (users, items, labels)."""
    users, items = np.asarray(users), np.asarray(items)
    n, K = len(users), int(n_neg)
    t = np.arange(n + K, dtype=np.uint64)
    src = feistel_perm(t, n + K, perm_key(seed, 1))
    ou, oi, oy = np.empty(n + K, users.dtype), np.empty(n + K, items.dtype), np.empty(n + K, np.float32)
    pos = src < n
    ou[pos], oi[pos], oy[pos] = users[src[pos]], items[src[pos]], 1.0
    j = (src[~pos] - n).astype(np.uint64)
    a = draw_below(seed, j, 0, 2, n)
    b = draw_below(seed, feistel_perm(j, K, perm_key(seed, 3)).astype(np.uint64), 0, 2, n)
    ou[~pos], oi[~pos], oy[~pos] = users[a], items[b], 0.0
    return ou, oi, oy


def _positive_sets(users, items):
    sets = {}
    for u, i in zip(np.asarray(users).tolist(), np.asarray(items).tolist()):
        sets.setdefault(u, set()).add(i)
    return sets


def bpr_sample_triplets(users, items, neg_per_pos: int, seed: int, n_cand: int, cand_items=None, max_tries: int = 16):
    """brBprSampleTriplets: per positive row neg_per_pos negatives, uniform over the candidates, re-drawn while they are
    positives of the user (at most max_tries draws; the last one stands)."""
    users, items = np.asarray(users), np.asarray(items)
    pos = _positive_sets(users, items)
    n = len(users)
    T = n * neg_per_pos
    ou, op, on = np.empty(T, users.dtype), np.empty(T, items.dtype), np.empty(T, items.dtype)
    draws = np.stack([draw_below(seed, np.arange(T, dtype=np.uint64), a, 4, n_cand) for a in range(max_tries)], axis=1)
    for t in range(T):
        row = t // neg_per_pos
        u = users[row]
        neg = 0
        for a in range(max_tries):
            c = int(draws[t, a])
            neg = int(cand_items[c]) if cand_items is not None else c
            if neg not in pos[int(u)]:
                break
        ou[t], op[t], on[t] = u, items[row], neg
    return ou, op, on


def ncf_negative_candidates(users, items, n_cand: int, num_items: int, seed: int):
    """brNcfNegativeCandidates: candidate j = (users[perm_u_round(slot)], items[perm_i_round(slot)]) with round = j // n,
    slot = j % n (generateSyntethic: independently shuffled columns); key = user * num_items + item, -1 for a positive pair."""
    users, items = np.asarray(users).astype(np.int64), np.asarray(items).astype(np.int64)
    n = len(users)
    pos = set((users * num_items + items).tolist())
    keys = np.empty(n_cand, np.int64)
    for r in range((n_cand + n - 1) // n):
        m = min(n, n_cand - r * n)
        slot = np.arange(m, dtype=np.uint64)
        u = users[feistel_perm(slot, n, perm_key(seed, 16 + 2 * r))]
        i = items[feistel_perm(slot, n, perm_key(seed, 17 + 2 * r))]
        k = u * num_items + i
        keys[r * n:r * n + m] = np.where(np.isin(k, list(pos)), -1, k)
    return keys


def ncf_negatives(users, items, num_items: int, size: int, seed: int, n_cand: int):
    """the whole generateNegativeFeedback contract on n_cand candidates: distinct valid keys in ascending order, then
    out[t] = pool[feistel_perm(t, len(pool))] for t < size (brSortUniqueKeys64 + brGatherPermutedPairs)."""
    keys = ncf_negative_candidates(users, items, n_cand, num_items, seed)
    pool = np.unique(keys[keys >= 0])
    assert len(pool) >= size
    sel = pool[feistel_perm(np.arange(size, dtype=np.uint64), len(pool), perm_key(seed, 5))]
    return sel // num_items, sel % num_items


# ----------------------------------------------------------------------------
# BPR notebook evaluation (src/models/bpr.py:230-289)
# ----------------------------------------------------------------------------
def roc_auc(truth, scores):
    """sklearn.metrics.roc_auc_score for binary truth: the Mann-Whitney statistic, ties one half."""
    truth = np.asarray(truth).astype(bool)
    s = np.asarray(scores, dtype=np.float64)
    pos, neg = s[truth], s[~truth]
    if len(pos) == 0 or len(neg) == 0:
        return float("nan")
    below = (neg[None, :] < pos[:, None]).sum() + 0.5 * (neg[None, :] == pos[:, None]).sum()
    return float(below) / (len(pos) * len(neg))


def full_auc(score_rows, ground_truth, items):
    """full_auc (bpr.py:230-254): mean over the users with positives of roc_auc_score(ground truth over ALL items, scores).
    score_rows[u] = scores of `items` for the u-th (user_id, true_item_ids) pair of ground_truth."""
    out = []
    for row, (_user, true_items) in zip(score_rows, ground_truth):
        grnd = np.zeros(len(items), dtype=np.int32)
        for p_ in true_items:
            grnd[items.index(p_)] = 1
        if true_items:
            out.append(roc_auc(grnd, row))
    return sum(out) / len(out), out


def mean_average_precision_k(score_rows, ground_truth, items, k=100):
    """mean_average_precision_k (bpr.py:257-289), literally: stable descending sort, AP over the top k / min(len(actual), k)."""
    scores = []
    for row, (_user, actual) in zip(score_rows, ground_truth):
        pred = sorted(dict(zip(items, row)).items(), key=lambda kv: kv[1], reverse=True)[:k]
        score, hits = 0.0, 0.0
        for i, (p_, _s) in enumerate(pred):
            if p_ in actual:
                hits += 1.0
                score += hits / (i + 1.0)
        scores.append(score / min(len(actual), k))
    return float(np.mean(scores)), scores


# ----------------------------------------------------------------------------
# negative sampling (NeuMFModel.bootstrapDataset, NeuMFModel.py:102-123)
# ----------------------------------------------------------------------------
def bootstrap_negatives(users, items, neg_ratio=3.0, seed=0):
    """posDf + negDf: sample rows with replacement (frac=negRatio), permute the item column of
    the sample, label 0; no collision check (NeuMFModel.py:103-109).  numpy RNG stands in for
    pandas' (the reference is unseeded)."""
    rng = np.random.default_rng(seed)
    n = len(users)
    k = int(round(neg_ratio * n))
    pick = rng.integers(0, n, size=k)
    nu, ni = users[pick], items[pick][rng.permutation(k)]
    U = np.concatenate([users, nu]); I = np.concatenate([items, ni])
    Y = np.concatenate([np.ones(n, np.float32), np.zeros(k, np.float32)])
    perm = rng.permutation(n + k)
    return U[perm], I[perm], Y[perm]
