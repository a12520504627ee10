"""-m "not gpu": the oracle itself.  The reference pins nothing (PARITY UNPINNED), so
the oracle is pinned by (a) torch autograd for every hand-derived backward, (b) the published
Random123 known-answer vectors for Philox4x32-10, (c) literal restatements of the reference's
top-k / HR@k python, (d) the committed golden fixtures."""
import os

import numpy as np
import pytest

from oracle import binrec_oracle as O
from oracle import torch_ref as T

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_philox_known_answers():
    # Random123 kat_vectors: philox4x32 10 rounds
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = O.philox4x32_10(*[np.array([c]) for c in ctr], *key)
        assert tuple(int(g[0]) for g in got) == want


def test_dropout_mask_rate_and_row_offset():
    m = O.dropout_mask(123456789012345, 3, 1, 4096, 100, 0.2)
    assert abs(m.mean() - 0.8) < 0.01
    # global-row definition: a shard's mask is a slice of the global mask
    part = O.dropout_mask(123456789012345, 3, 1, 1000, 100, 0.2, row0=2000)
    assert np.array_equal(part, m[2000:3000])
    assert O.dropout_mask(1, 1, 0, 5, 7, 0.0).all()


@pytest.mark.parametrize("variant", ["A", "B"])
def test_neumf_backward_matches_autograd(variant):
    spec = O.NeuMFSpec(variant, dim=8)
    p = O.neumf_init(spec, 97, 53, seed=3, dt=np.float64)
    rng = np.random.default_rng(0)
    for k in ("b1", "b2", "b3", "b4", "be1", "be2"):
        p[k] = rng.normal(0, 0.1, p[k].shape)
    B = 64
    u, i = rng.integers(0, 97, B), rng.integers(0, 53, B)
    u[:9] = u[0]
    y = (rng.random(B) < 0.25).astype(np.float64)
    masks = [O.dropout_mask(7, 1, s, B, w, 0.2) for s, w in enumerate((16, spec.hidden[0], spec.hidden[1]))]
    loss, c, g, rg, _ = O.neumf_step_grads(spec, p, u, i, y, masks)
    l2, z2, g2, rg2 = T.neumf_autograd(spec, p, u, i, y, masks)
    assert abs(loss - l2) < 1e-12
    np.testing.assert_allclose(c["logit"], z2, atol=1e-12)
    for k in g:
        np.testing.assert_allclose(g[k], g2[k], atol=1e-12, err_msg=k)
    for k in rg:
        np.testing.assert_allclose(rg[k], rg2[k], atol=1e-12, err_msg=k)


def test_bpr_and_twotower_backward_match_autograd():
    import torch
    rng = np.random.default_rng(1)
    U, I, D, B = 31, 17, 6, 40
    ut, it = rng.normal(size=(U, D)), rng.normal(size=(I, D))
    u, p, n = rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B)
    loss, l, (gu, gp, gn) = O.bpr_step_grads(ut, it, u, p, n)
    eu = torch.tensor(ut[u], requires_grad=True); ep = torch.tensor(it[p], requires_grad=True); en = torch.tensor(it[n], requires_grad=True)
    lt = (1 - torch.sigmoid((eu * ep).sum(1) - (eu * en).sum(1))).mean()
    lt.backward()
    assert abs(float(lt) - loss) < 1e-14
    for a, b in ((gu, eu.grad), (gp, ep.grad), (gn, en.grad)):
        np.testing.assert_allclose(a, b.numpy(), atol=1e-14)
    # TwoTower: in-batch softmax with accidental hits (duplicate candidate ids) + rdZero BCE
    E, S = 5, 4
    prm = {"user_emb": rng.normal(size=(U, E)), "item_emb": rng.normal(size=(I, E)), "Wu": rng.normal(size=(E, S)),
           "bu": rng.normal(size=S), "Wi": rng.normal(size=(E, S)), "bi": rng.normal(size=S)}
    items = rng.integers(0, I, B)  # B > I => duplicates guaranteed
    for rd in (False, True):
        y = (rng.random(B) < 0.5).astype(np.float64)
        loss, (q, c), g, rg = O.twotower_step_grads(prm, u, items, y, rd_zero=rd)
        tp = {k: torch.tensor(v, requires_grad=True) for k, v in prm.items()}
        eu = tp["user_emb"][torch.as_tensor(u)]; ei = tp["item_emb"][torch.as_tensor(items)]
        eu.retain_grad(); ei.retain_grad()
        qt, ct = eu @ tp["Wu"] + tp["bu"], ei @ tp["Wi"] + tp["bi"]
        if rd:
            lt = torch.nn.functional.binary_cross_entropy_with_logits((qt * ct).sum(1), torch.tensor(y))
        else:
            S_ = qt @ ct.T
            ids = torch.as_tensor(items)
            dup = (ids[:, None] == ids[None, :]).double() - torch.eye(B, dtype=torch.float64)
            S_ = S_ + dup * O.MIN_FLOAT
            lt = (torch.logsumexp(S_, dim=1) - torch.diagonal(S_)).sum()
        lt.backward()
        assert abs(float(lt) - loss) < 1e-9 * max(1, abs(loss))
        for k in ("Wu", "bu", "Wi", "bi"):
            np.testing.assert_allclose(g[k], tp[k].grad.numpy(), atol=1e-10, err_msg=k)
        np.testing.assert_allclose(rg["user_emb"], eu.grad.numpy(), atol=1e-10)
        np.testing.assert_allclose(rg["item_emb"], ei.grad.numpy(), atol=1e-10)


def test_adam_tf_form_vs_closed_form():
    # first step from zero slots: theta -= lr*sqrt(1-b2)/(1-b1) * (1-b1) g / (sqrt((1-b2) g^2) + eps)
    g = np.array([0.5, -2.0, 1e-4]); th = np.zeros(3)
    t1, m, v = O.adam_dense(th, np.zeros(3), np.zeros(3), g, 0.01, 1)
    want = -0.01 * np.sqrt(1 - 0.999) / (1 - 0.9) * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-7)
    np.testing.assert_allclose(t1, want, rtol=1e-12)
    # non-lazy sparse apply: an untouched row with non-zero m keeps moving; lazy leaves it alone
    th = np.ones((4, 2)); m0 = np.full((4, 2), 0.1); v0 = np.full((4, 2), 0.01)
    ids = np.array([1, 1, 3]); rg = np.ones((3, 2))
    d, _, _ = O.adam_sparse_tf(th, m0, v0, ids, rg, 0.01, 5, lazy=False)
    z, _, _ = O.adam_sparse_tf(th, m0, v0, ids, rg, 0.01, 5, lazy=True)
    assert not np.allclose(d[0], th[0]) and np.allclose(z[0], th[0])
    np.testing.assert_allclose(d[[1, 3]], z[[1, 3]])          # touched rows agree
    # duplicates are summed BEFORE the square: row 1 saw g = 2
    _, _, vv = O.adam_sparse_tf(th, np.zeros((4, 2)), np.zeros((4, 2)), ids, rg, 0.01, 1, lazy=True)
    np.testing.assert_allclose(vv[1], 0.001 * 4.0)


def test_topk_reference_tie_break_and_metrics():
    # trainers/topKmetrics.py:51-72: strict '>' keeps the lower position among equal scores
    scores = [0.5, 0.9, 0.9, 0.1, 0.9, 0.5]
    items = ["a", "b", "c", "d", "e", "f"]
    top = O.topk_reference_order(scores, items, 3)
    assert [i for _, i in top] == ["b", "c", "e"]
    top2 = O.topk_reference_order(scores, items, 4)
    assert [i for _, i in top2] == ["b", "c", "e", "a"]
    order = sorted(range(6), key=lambda j: (-scores[j], j))      # == stable descending sort
    assert [items[j] for j in order[:4]] == [i for _, i in top2]
    preds = [("u1", top), ("u2", [(0.3, "d"), (0.2, "a"), (0.1, "f")])]
    m = O.topk_metrics(preds, [("u1", "c"), ("u2", "b"), ("u3", "a")], ["u1", "u2", "u3"], items)
    assert m["tp"] == 1 and m["fp"] == 5 and m["fn"] == 2 and m["hitRate"] == pytest.approx(1 / 3)
    assert m["tn"] == 18 - 1 - 5 - 2


def test_dedup_sequential_is_ordered_fp32():
    rng = np.random.default_rng(3)
    ids = np.array([5, 2, 5, 5, 2, 9])
    rows = rng.normal(size=(6, 4)).astype(np.float32) * np.array([1e8, 1, 1e-8, 1], np.float32)
    uniq, out = O.dedup_rows_sequential(ids, rows, np.float32)
    assert list(uniq) == [2, 5, 9]
    want5 = ((rows[0] + rows[2]).astype(np.float32) + rows[3]).astype(np.float32)
    assert np.array_equal(out[1], want5)


def test_golden_fixtures_reproduce():
    """tests/golden/*.npz were written by tests/golden/make_golden.py from this oracle."""
    files = sorted(f for f in os.listdir(GOLD) if f.endswith(".npz") and not f.startswith("hr10_"))   # hr10_*: 40 min, make_hr10_golden.py
    assert files, "no golden fixtures committed"
    from tests.golden import make_golden as mg
    for f in files:
        z = np.load(os.path.join(GOLD, f), allow_pickle=False)
        fresh = mg.CASES[f[:-4]]()
        for k in z.files:
            np.testing.assert_allclose(fresh[k], z[k], rtol=1e-12, atol=1e-15, err_msg=f"{f}:{k}")


def test_hr10_fixture_matches_its_protocol():
    """the 20-epoch HR@10 fixture (make_hr10_golden.py): shapes, monotone loss, and the data it was trained on is what the
    protocol's seeds still produce (the negatives come from the oracle restatement of the device sampler)."""
    import sys
    sys.path.insert(0, GOLD)
    import make_hr10_golden as G
    p = G.PROTOCOL
    z = np.load(os.path.join(GOLD, "hr10_ml1m_shaped_e20.npz"), allow_pickle=False)
    assert z["top_items"].shape == (p["n_users"], p["k"]) and z["losses"].shape == (p["epochs"],)
    assert np.all(np.diff(z["losses"]) < 0) and 0.7 < float(z["hit_rate"]) < 0.85
    assert np.all(np.diff(z["top_scores"], axis=1) <= 0)                      # every user's list is sorted by score
    users, items = G.positives()
    nu, ni = G.negatives_cpu(users, items)
    tr, test = G.split(users, items, nu, ni)
    assert len(tr["users"]) == int(z["n_train"]) and int((test["labels"] > 0).sum()) == int(z["n_test_pos"])
    assert int(np.sum(tr["users"].astype(np.int64) * 7919 + tr["items"]) % (1 << 62)) == int(z["data_checksum"])


def test_two_level_segment_sum_restatement():
    """ordered_segment_sum (the HIP kernels' order for hot ids) vs the sequential sum: identical on segments that stay
    inside one 64-position block of the sorted order, equal to fp32 rounding on the long ones, exact in float64 sense."""
    rng = np.random.default_rng(3)
    n = 1000
    ids = rng.integers(0, 400, n)
    ids[:300] = 7                                     # one hot id: 300+ positions, crosses 4-5 block boundaries
    g = rng.normal(size=(n, 6)).astype(np.float32)
    u1, seq = O.dedup_rows_sequential(ids, g, dt=np.float32)
    u2, two = O.ordered_segment_sum(ids, g, dt=np.float32)
    assert np.array_equal(u1, u2)
    np.testing.assert_allclose(two, seq, rtol=1e-5, atol=1e-5)
    _, ref64 = O.dedup_rows_sequential(ids, g.astype(np.float64), dt=np.float64)
    np.testing.assert_allclose(two, ref64, rtol=1e-5, atol=1e-5)
    order = np.argsort(ids, kind="stable"); sid = ids[order]
    uniq, start = np.unique(sid, return_index=True)
    ends = np.append(start[1:], n)
    inside = (start // 64) == ((ends - 1) // 64)       # segments within one block: bit-equal
    assert inside.sum() > 100
    assert np.array_equal(two[inside].view(np.uint32), seq[inside].view(np.uint32))
