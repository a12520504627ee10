"""HR@10 parity (BASELINE.json metric, second half): train NeuMF-A (trainers/NFC_plain.py graph) on the GPU
path and on the CPU oracle with identical data, sample order and dropout masks; rank the full catalogue per
user (trainers/topKmetrics.py) and compare HR@10.  |dHR| <= 0.002 is the bar.

    python tools/hr10_parity.py [--users 6040 --items 3706 --pos 1000209 --epochs 20 --batch 50000 --dim 10]
Uses data/ml-1m/ratings.dat if present, else the seeded ML-1M-shaped synthetic set (data.py)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from importlib import import_module
from oracle import binrec_oracle as O


def run(n_users, n_items, n_pos, epochs, batch, dim, seed=0, k=10, cpu=True, log=print):
    data = import_module("binary-recommendation_amd.data")
    models = import_module("binary-recommendation_amd.models")
    neumf = import_module("binary-recommendation_amd.neumf")
    tkm = import_module("binary-recommendation_amd.topk_metrics")
    path = data.find_ml1m()
    if path and n_users == data.ML1M_USERS:
        d = data.load_movielens(path); users, items, n_users, n_items = d["users"], d["items"], d["nbrUser"], d["nbrMovie"]; src = path
    else:
        users, items = data.ml1m_shaped(seed, n_users, n_items, n_pos); src = "ML-1M-shaped synthetic (data.ml1m_shaped)"
    chunks = data.make_ncf_chunks(users, items, n_users, n_items, k=5, neg_per_pos=2, seed=seed)
    test = chunks[0]                                       # one chunk held out (twoTower.py:182)
    tr = {k_: np.concatenate([c[k_] for c in chunks[1:]]) for k_ in ("users", "items", "labels")}
    n = len(tr["users"])
    rng = np.random.default_rng(seed + 1)
    orders = [rng.permutation(n) for _ in range(epochs)]
    dev = torch.device("cuda:0")
    cfg = neumf.NeuMFConfig(variant="A", dim=dim, optimizer="adam_dense", seed=424242)
    eng = neumf.NeuMFEngine(cfg, n_users + 1, n_items + 1, dev, max_batch=max(batch, 1 << 16), init_seed=seed)
    spec = O.NeuMFSpec("A", dim=dim)
    P = {k_: eng.tables[k_].cpu().numpy().astype(np.float64) for k_ in neumf.TABLES}
    P.update({k_: eng.theta.view(k_).cpu().numpy().astype(np.float64).reshape(spec.dense_shapes[k_]) for k_ in neumf.DENSE_ORDER})
    P.update({k_: v.cpu().numpy().astype(np.float64) for k_, v in eng.moving.items()})
    model = models.KerasLikeNeuMF(eng)
    t0 = time.time()
    hist = model.fit([tr["users"], tr["items"]], tr["labels"], epochs=epochs, batch_size=batch, orders=orders)
    torch.cuda.synchronize(); t_gpu = time.time() - t0
    log(f"GPU fit: {t_gpu:.1f} s, loss {hist.history['loss'][0]:.5f} -> {hist.history['loss'][-1]:.5f}")
    all_users, all_items = np.arange(n_users), np.arange(n_items)
    positives = list(zip(test["users"][test["labels"] > 0].tolist(), test["items"][test["labels"] > 0].tolist()))
    top_gpu = tkm.topKRatings(k, model, all_users.tolist(), all_items.tolist(), "NFC")
    m_gpu = tkm.topKMetrics(top_gpu, positives, all_users.tolist(), all_items.tolist())
    out = {"data": src, "users": int(n_users), "items": int(n_items), "train_samples": int(n), "epochs": epochs, "batch": batch, "dim": dim,
           "gpu": {"hitRate@10": m_gpu["hitRate"], "precision": m_gpu["precision"], "recall": m_gpu["recall"], "final_loss": hist.history["loss"][-1], "fit_s": t_gpu}}
    if cpu:
        keys = list(O.DENSE_ORDER) + list(neumf.TABLES)
        M = {k_: np.zeros_like(P[k_]) for k_ in keys}; V = {k_: np.zeros_like(P[k_]) for k_ in keys}
        t, t0, losses = 0, time.time(), []
        for ep in range(epochs):
            o = orders[ep]; ep_loss = 0.0
            for s in range(0, n, batch):
                idx = o[s:s + batch]; B = len(idx); t += 1
                u, i, y = tr["users"][idx], tr["items"][idx], tr["labels"][idx]
                masks = [O.dropout_mask(cfg.seed, t, st, B, w, cfg.dropout) for st, w in enumerate((2 * dim, spec.hidden[0], spec.hidden[1]))]
                loss, c, g, rg, ns = O.neumf_step_grads(spec, P, u, i, y, masks, dt=np.float64)
                for k_ in O.DENSE_ORDER:
                    P[k_], M[k_], V[k_] = O.adam_dense(P[k_], M[k_], V[k_], g[k_], cfg.lr, t)
                for k_ in neumf.TABLES:
                    ids = u if k_.startswith("user") else i
                    gd = O.scatter_add_dense(P[k_].shape[0], ids, rg[k_])
                    P[k_], M[k_], V[k_] = O.adam_dense(P[k_], M[k_], V[k_], gd, cfg.lr, t)     # == Keras non-lazy sparse apply
                P.update(ns); ep_loss += loss * B
            losses.append(ep_loss / n)
            log(f"CPU oracle epoch {ep + 1}/{epochs}: loss {losses[-1]:.5f} ({time.time() - t0:.0f} s)")
        uu = np.repeat(all_users, n_items); ii = np.tile(all_items, n_users)
        sc = np.concatenate([O.neumf_forward(spec, P, uu[s:s + (1 << 20)], ii[s:s + (1 << 20)], training=False, dt=np.float64)["prob"]
                             for s in range(0, len(uu), 1 << 20)]).reshape(n_users, n_items)
        order = np.argsort(-sc, axis=1, kind="stable")[:, :k]
        top_cpu = [(int(u_), [(float(sc[u_, j]), int(j)) for j in order[u_]]) for u_ in all_users]
        m_cpu = O.topk_metrics(top_cpu, positives, all_users.tolist(), all_items.tolist())
        same = np.mean([set(i for _, i in a[1]) == set(i for _, i in b[1]) for a, b in zip(top_gpu, top_cpu)])
        out["cpu_oracle"] = {"hitRate@10": m_cpu["hitRate"], "precision": m_cpu["precision"], "recall": m_cpu["recall"], "final_loss": losses[-1], "fit_s": time.time() - t0}
        out["abs_delta_hitRate@10"] = abs(m_cpu["hitRate"] - m_gpu["hitRate"])
        out["users_with_identical_top10_sets"] = float(same)
        out["loss_rel_diff_last_epoch"] = abs(losses[-1] - hist.history["loss"][-1]) / abs(losses[-1])
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=6040); ap.add_argument("--items", type=int, default=3706)
    ap.add_argument("--pos", type=int, default=1000209); ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--batch", type=int, default=50000); ap.add_argument("--dim", type=int, default=10)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    print(json.dumps(run(a.users, a.items, a.pos, a.epochs, a.batch, a.dim, cpu=not a.no_cpu, log=lambda m: print(m, file=sys.stderr, flush=True))))
