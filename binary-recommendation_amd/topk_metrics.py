"""HR@k / precision / recall and the full-catalogue top-k behind them — trainers/topKmetrics.py:17-99,
with the scoring and the selection on the GPU (the reference runs a python loop per user and a
python insertion sort per item: topKmetrics.py:26-41,51-72).

Same outputs: `topKRatings` -> [(user, [(score, item), ...k])] in descending score with ties keeping the
LOWER item position (strict '>' in __topk, :59,68); `topKMetrics` -> {tp, tn, fp, fn, precision, recall,
hitRate} with hitRate = hits / len(usersId) over ALL users, train positives not excluded (:98).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


def topk_scores_neumf(engine, user_ids, item_ids, k, users_per_chunk=None):
    """Score every (user, item) pair of user_ids x item_ids through `engine.predict` and keep the k best
    per user.  -> (scores (U,k) float32, index into item_ids (U,k) int32), both on the device."""
    dev = engine.device
    users = torch.as_tensor(np.asarray(user_ids), device=dev).to(engine.id_dtype)
    items = torch.as_tensor(np.asarray(item_ids), device=dev).to(engine.id_dtype)
    U, I = users.shape[0], items.shape[0]
    if users_per_chunk is None:
        users_per_chunk = max(1, min(U, (1 << 22) // max(1, I)))
    out_s = torch.empty(U, k, dtype=torch.float32, device=dev)
    out_i = torch.empty(U, k, dtype=torch.int32, device=dev)
    for s in range(0, U, users_per_chunk):
        e = min(U, s + users_per_chunk)
        uu = users[s:e].repeat_interleave(I).contiguous()      # index plumbing only
        ii = items.repeat(e - s).contiguous()
        scores = engine.predict(uu, ii).view(e - s, I)
        ts, ti = ops.topk_rows(scores, k)
        out_s[s:e], out_i[s:e] = ts, ti
    return out_s, out_i


def topKRatings(k, model, usersId, itemsId, mtype=None):
    """trainers/topKmetrics.py:17-43.  `model` is a NeuMF engine / Keras-like wrapper (mtype "NFC"), or
    any object with `topk(users, items, k) -> (scores, index)` (TwoTower BruteForce)."""
    engine = getattr(model, "engine", model)
    if mtype == "NFC" or hasattr(engine, "predict"):
        ts, ti = topk_scores_neumf(engine, usersId, itemsId, k)
    else:
        ts, ti = model.topk(usersId, itemsId, k)
    ts, ti = ts.cpu().numpy(), ti.cpu().numpy()
    items = list(itemsId)
    return [(u, [(float(ts[n, j]), items[int(ti[n, j])]) for j in range(ts.shape[1])]) for n, u in enumerate(usersId)]


def topKMetrics(predictions, positives, usersId, itemsId):
    """trainers/topKmetrics.py:74-99: {tp, tn, fp, fn, precision, recall, hitRate}; hitRate = users with at least one hit /
    len(usersId) over ALL users.  The reference walks every (user, item) of every top-k list against a python set (:85-93); here
    the lists become an index matrix, the positives a CSR, and brMapAtK counts the hits per user on the GPU."""
    nbrUser, nbrItem = len(usersId), len(itemsId)
    total = nbrUser * nbrItem
    real = set(positives)
    icol = {}
    for _u, lst in predictions:
        for _r, i in lst:
            icol.setdefault(i, len(icol))
    k = max((len(lst) for _u, lst in predictions), default=0)
    tp = hits = n_pred = 0
    if predictions and k:
        dev = torch.device("cuda", torch.cuda.current_device())
        topk = np.full((len(predictions), k), -1, dtype=np.int32)          # ragged lists: -1 never matches
        for n, (_u, lst) in enumerate(predictions):
            topk[n, :len(lst)] = [icol[i] for _r, i in lst]
            n_pred += len(lst)
        # the truth CSR has one row per PREDICTION row (a user listed twice in `predictions` is counted twice, as the reference's loop
        # over `predictions` does, topKmetrics.py:85-93)
        by_user = {}
        for (u, i) in real:
            if i in icol:
                by_user.setdefault(u, []).append(icol[i])
        rows = [n for n, (u, _l) in enumerate(predictions) for _ in by_user.get(u, ())]
        cols = [c for (u, _l) in predictions for c in by_user.get(u, ())]
        off, idx = ops.truth_csr(len(predictions), rows, cols, dev)
        _, h = ops.map_at_k(torch.from_numpy(topk).to(dev), off, idx, want_ap=False)
        h = h.cpu().numpy()
        tp, hits = int(h.sum()), int((h > 0).sum())
    fp = n_pred - tp
    fn = len(real) - tp
    tn = total - tp - fp - fn
    return {"tp": tp, "tn": tn, "fp": fp, "fn": fn, "precision": tp / (tp + fp) if tp + fp else 0.0,
            "recall": tp / (tp + fn) if tp + fn else 0.0, "hitRate": hits / nbrUser}
