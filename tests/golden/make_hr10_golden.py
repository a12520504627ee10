"""CPU-oracle side of the HR@10 protocol of trainers/NFC_plain.py:128-134,165 (20 epochs, batch 50 000, Adam 0.005, dropout 0.2) on
the seeded ML-1M-shaped set (MovieLens-1M itself is not available offline), generated ONCE in the build container:

    python tests/golden/make_hr10_golden.py            # ~40 min of numpy float64 on 8 cores -> hr10_ml1m_shaped_e20.npz

Everything the GPU leg needs to repeat the run bit-for-bit on its side is derived from seeds: the positives (data.ml1m_shaped),
the negatives (oracle restatement of the device sampler, bit-exact with csrc/sampling.hip: tests/test_gpu_rows_f.py), the initial
parameters (oracle neumf_init), the epoch orders (numpy Generator) and the dropout masks (Philox, keyed by cfg seed / step).
The fixture holds the oracle's results: per-epoch loss, HR@10 / precision / recall on the held-out chunk and every user's top-10.
tests/test_gpu_hr10.py trains the HIP path on the same data and compares (|dHR@10| <= 0.002 is BASELINE.json's bar)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import binrec_oracle as O  # noqa: E402

PROTOCOL = dict(n_users=6040, n_items=3706, n_pos=1_000_209, epochs=20, batch=50000, dim=10, seed=0, k=10, cfg_seed=424242, lr=0.005, dropout=0.2,
                neg_per_pos=2, folds=5)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hr10_ml1m_shaped_e20.npz")


def positives(p=PROTOCOL):
    from importlib import import_module
    data = import_module("binary-recommendation_amd.data")          # numpy only
    return data.ml1m_shaped(p["seed"], p["n_users"], p["n_items"], p["n_pos"])


def split(users, items, nu, ni, p=PROTOCOL):
    """makeNCFDatasets (synthetic.py:152-164): 5 chunks of positives + negatives, chunk 0 held out (twoTower.py:182)."""
    k = p["folds"]
    chunks = [{"users": np.concatenate([a, c]), "items": np.concatenate([b, d]),
               "labels": np.concatenate([np.ones(len(a), np.float32), np.zeros(len(c), np.float32)])}
              for a, b, c, d in zip(np.array_split(users, k), np.array_split(items, k), np.array_split(nu, k), np.array_split(ni, k))]
    test = chunks[0]
    tr = {q: np.concatenate([c[q] for c in chunks[1:]]) for q in ("users", "items", "labels")}
    return tr, test


def negatives_cpu(users, items, p=PROTOCOL):
    size = p["neg_per_pos"] * len(users)
    return O.ncf_negatives(users, items, p["n_items"], size, p["seed"], int(size * 1.3) + len(users))


def epoch_orders(n, p=PROTOCOL):
    rng = np.random.default_rng(p["seed"] + 1)
    return [rng.permutation(n) for _ in range(p["epochs"])]


def initial_params(p=PROTOCOL):
    return O.neumf_init(O.NeuMFSpec("A", dim=p["dim"]), p["n_users"] + 1, p["n_items"] + 1, seed=p["seed"], dt=np.float32)


def main():
    p = PROTOCOL
    users, items = positives()
    nu, ni = negatives_cpu(users, items)
    tr, test = split(users, items, nu, ni)
    n = len(tr["users"])
    orders = epoch_orders(n)
    spec = O.NeuMFSpec("A", dim=p["dim"])
    P = {k: v.astype(np.float64) for k, v in initial_params().items()}
    tables = ("user_mlp", "item_mlp", "user_mf", "item_mf")
    keys = list(O.DENSE_ORDER) + list(tables)
    M = {k: np.zeros_like(P[k]) for k in keys}
    V = {k: np.zeros_like(P[k]) for k in keys}
    t, t0, losses = 0, time.time(), []
    for ep in range(p["epochs"]):
        o, ep_loss = orders[ep], 0.0
        for s in range(0, n, p["batch"]):
            idx = o[s:s + p["batch"]]
            B = len(idx)
            t += 1
            u, i, y = tr["users"][idx], tr["items"][idx], tr["labels"][idx]
            masks = [O.dropout_mask(p["cfg_seed"], t, st, B, w, p["dropout"]) for st, w in enumerate((2 * p["dim"], spec.hidden[0], spec.hidden[1]))]
            loss, c, g, rg, ns = O.neumf_step_grads(spec, P, u, i, y, masks, dt=np.float64)
            for k in O.DENSE_ORDER:
                P[k], M[k], V[k] = O.adam_dense(P[k], M[k], V[k], g[k], p["lr"], t)
            for k in tables:
                ids = u if k.startswith("user") else i
                P[k], M[k], V[k] = O.adam_dense(P[k], M[k], V[k], O.scatter_add_dense(P[k].shape[0], ids, rg[k]), p["lr"], t)   # Keras non-lazy sparse apply
            P.update(ns)
            ep_loss += loss * B
        losses.append(ep_loss / n)
        print(f"epoch {ep + 1}/{p['epochs']}: loss {losses[-1]:.6f} ({time.time() - t0:.0f} s)", flush=True)
    nU, nI, k = p["n_users"], p["n_items"], p["k"]
    all_items = np.arange(nI)
    top_i, top_s = np.empty((nU, k), np.int16), np.empty((nU, k), np.float64)
    for u0 in range(0, nU, 256):
        uu = np.repeat(np.arange(u0, min(u0 + 256, nU)), nI)
        sc = O.neumf_forward(spec, P, uu, np.tile(all_items, len(uu) // nI), training=False, dt=np.float64)["prob"].reshape(-1, nI)
        order = np.argsort(-sc, axis=1, kind="stable")[:, :k]
        top_i[u0:u0 + len(sc)] = order
        top_s[u0:u0 + len(sc)] = np.take_along_axis(sc, order, axis=1)
    pos = list(zip(test["users"][test["labels"] > 0].tolist(), test["items"][test["labels"] > 0].tolist()))
    preds = [(int(u), [(float(top_s[u, j]), int(top_i[u, j])) for j in range(k)]) for u in range(nU)]
    m = O.topk_metrics(preds, pos, list(range(nU)), list(range(nI)))
    print(m, flush=True)
    np.savez_compressed(OUT, top_items=top_i, top_scores=top_s, losses=np.array(losses), hit_rate=m["hitRate"], precision=m["precision"], recall=m["recall"],
                        tp=m["tp"], fp=m["fp"], fn=m["fn"], n_train=n, n_test_pos=len(pos), data_checksum=int(np.sum(tr["users"].astype(np.int64) * 7919 + tr["items"]) % (1 << 62)),
                        fit_seconds=time.time() - t0)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
