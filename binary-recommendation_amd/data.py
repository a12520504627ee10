"""Data contracts either side of the hot path (host side, numpy).

  * binarised MovieLens loader — trainers/loadBinaryMovieLens.py:8-39 (every rating -> ratedVal,
    ids kept as opaque keys; the reference reads ml-100k `u.data`, ML-1M's `ratings.dat` uses '::')
  * NCF negative generation — Data handling/synthetic.py:152-164,237-256: `2 x len(data)` (user,item)
    pairs that are NOT positives, no duplicate pairs, RATING_TYPE 0; positives get RATING_TYPE 1
  * k time-ordered chunks, one held out — synthetic.py:258-261, trainers/twoTower.py:179-188
  * ML-1M-shaped synthetic set (MovieLens-1M is not available offline): 6 040 users x 3 706 items x
    1 000 209 positives, power-law item popularity / user activity
"""
from __future__ import annotations

import os

import numpy as np

ML1M_USERS, ML1M_ITEMS, ML1M_RATINGS = 6040, 3706, 1_000_209


def load_movielens(path: str, rated_val: float = 1.0):
    """-> dict(users, items, ratings, usersId, moviesId, nbrUser, nbrMovie, realRat) with dense int ids.
    Accepts ml-100k `u.data` (tab separated) or ML-1M `ratings.dat` ('::')."""
    sep = "::" if path.endswith(".dat") else "\t"
    u, m = [], []
    with open(path, encoding="latin-1") as f:
        for line in f:
            parts = line.rstrip("\n").split(sep)
            if len(parts) >= 3:
                u.append(parts[0]); m.append(parts[1])
    users_id, uidx = np.unique(np.array(u), return_inverse=True)
    movies_id, midx = np.unique(np.array(m), return_inverse=True)
    return {"users": uidx.astype(np.int32), "items": midx.astype(np.int32), "ratings": np.full(len(u), float(rated_val), np.float32),
            "usersId": users_id, "moviesId": movies_id, "nbrUser": len(users_id), "nbrMovie": len(movies_id),
            "realRat": set(zip(uidx.tolist(), midx.tolist()))}


def find_ml1m():
    for p in ("data/ml-1m/ratings.dat", "/data/ml-1m/ratings.dat", os.path.expanduser("~/ml-1m/ratings.dat")):
        if os.path.exists(p):
            return p
    return None


def ml1m_shaped(seed: int = 0, n_users=ML1M_USERS, n_items=ML1M_ITEMS, n_pos=ML1M_RATINGS):
    """Seeded positives with ML-1M's shape: unique (user,item) pairs, Zipf-like item popularity and
    log-normal user activity.  Returns (users int32, items int32) in a time-like order."""
    rng = np.random.default_rng(seed)
    item_w = 1.0 / np.arange(1, n_items + 1) ** 0.9
    item_w /= item_w.sum()
    user_w = rng.lognormal(0.0, 1.0, n_users)
    user_w /= user_w.sum()
    pairs = np.empty(0, dtype=np.int64)
    while pairs.size < n_pos:
        need = int((n_pos - pairs.size) * 1.3) + 1000
        u = rng.choice(n_users, size=need, p=user_w)
        i = rng.choice(n_items, size=need, p=item_w)
        pairs = np.unique(np.concatenate([pairs, u.astype(np.int64) * n_items + i]))
    pairs = rng.permutation(pairs)[:n_pos]
    return (pairs // n_items).astype(np.int32), (pairs % n_items).astype(np.int32)


def _dev_ids(a, device):
    import torch
    t = a if torch.is_tensor(a) else torch.as_tensor(np.ascontiguousarray(np.asarray(a)))
    if t.dtype not in (torch.int32, torch.int64):
        t = t.to(torch.int32)
    return t.to(device).contiguous()


def generate_negative_feedback(users, items, n_users, n_items, size, seed=0, device="cuda:0"):
    """generateNegativeFeedback (synthetic.py:237-256) on the device (csrc/sampling.hip): `size` DISTINCT (user,item) pairs
    outside the positives; customers and products are drawn by shuffling the two columns of the data independently, round
    after round (generateSyntethic, synthetic.py:208-223).  -> (users, items) numpy int32.
    oracle/binrec_oracle.py::ncf_negatives restates it bit for bit.
    Deviations from the reference's pandas code (same marginals, not the same sample): the reference reshuffles the GROWING frame of
    negatives each round and `drop_duplicates(keep=False)` removes every copy of a pair that came up twice; here the candidates of all
    rounds are independent shuffles of the two columns, duplicates keep ONE copy (sort + unique) and positives are rejected against the
    customer's CSR.  The reference's exact frame depends on pandas' / numpy's global RNG state and is not reproducible anyway."""
    from . import ops
    u, i = _dev_ids(users, device), _dev_ids(items, device)
    off, pit = ops.positives_csr(u, i, n_users, u.device)
    nu, ni = ops.ncf_negatives(u, i, off, pit, n_items, int(size), int(seed))
    return nu.cpu().numpy().astype(np.int32), ni.cpu().numpy().astype(np.int32)


def make_ncf_chunks(users, items, n_users, n_items, k=5, neg_per_pos=2, seed=0, device="cuda:0"):
    """makeNCFDatasets (synthetic.py:152-164): k chunks, each = its slice of the positives (label 1)
    + its slice of the `neg_per_pos x` negatives (label 0)."""
    nu, ni = generate_negative_feedback(users, items, n_users, n_items, neg_per_pos * len(users), seed, device)
    chunks = []
    for pu, pi, qu, qi in zip(np.array_split(users, k), np.array_split(items, k), np.array_split(nu, k), np.array_split(ni, k)):
        chunks.append({"users": np.concatenate([pu, qu]), "items": np.concatenate([pi, qi]),
                       "labels": np.concatenate([np.ones(len(pu), np.float32), np.zeros(len(qu), np.float32)])})
    return chunks


def bootstrap_dataset(users, items, neg_ratio=3.0, seed=0, device="cuda:0"):
    """NeuMFModel.bootstrapDataset (NeuMFModel.py:102-109) on the device: positives + round(neg_ratio x n) rows sampled with
    replacement whose item column is permuted (no collision check), shuffled.  -> device tensors (users, items, labels);
    oracle/binrec_oracle.py::bootstrap_dataset restates it bit for bit."""
    from . import ops
    u, i = _dev_ids(users, device), _dev_ids(items, device)
    return ops.bootstrap_dataset(u, i, int(round(neg_ratio * u.shape[0])), int(seed))


def sample_bpr_triplets(users, items, n_users, n_items, neg_per_pos=1, seed=0, cand_items=None, device="cuda:0"):
    """BPR training triplets (customer, positive, negative) - the sampled replacement of BPRModel.extractPositivesNegatives'
    O(U*I) enumeration (src/models/BPRModel.py:111-119): neg_per_pos negatives per positive, uniform over the candidate products
    and never a positive of the customer.  -> device tensors (users, positives, negatives)."""
    from . import ops
    u, i = _dev_ids(users, device), _dev_ids(items, device)
    off, pit = ops.positives_csr(u, i, n_users, u.device)
    cand = None if cand_items is None else _dev_ids(cand_items, device).to(u.dtype)
    return ops.bpr_sample_triplets(u, i, off, pit, neg_per_pos, seed, int(n_items if cand is None else cand.shape[0]), cand)
