"""-m gpu parity: L4 in-batch softmax (accidental-hit mask, streaming LSE, gradients), TwoTower
train_step (softmax + rdZero, Adagrad), BPR engine steps (Adam), E1 scoring + stable top-k."""
import os
from importlib import import_module

import numpy as np
import pytest
import torch

from oracle import binrec_oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _m(name):
    return import_module("binary-recommendation_amd." + name)


@pytest.mark.parametrize("Bq,dim", [(64, 50), (200, 64), (1000, 50), (37, 16), (130, 128)])
def test_inbatch_softmax_kernels(dev, Bq, dim):
    ops = _m("ops")
    rng = np.random.default_rng(Bq + dim)
    q = rng.normal(0, 0.4, (Bq, dim)).astype(np.float32); c = rng.normal(0, 0.4, (Bq, dim)).astype(np.float32)
    ids = rng.integers(0, max(3, Bq // 3), Bq)            # many accidental hits
    td = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    qd, cd, idd = td(q), td(c), td(ids).int()
    lse = torch.empty(Bq, device=dev); ls = torch.zeros(64, dtype=torch.float64, device=dev)
    ops.inbatch_softmax_lse(qd, cd, idd, idd, 0, lse, ls)
    dq, dc = torch.empty(Bq, dim, device=dev), torch.empty(Bq, dim, device=dev)
    ops.inbatch_softmax_grad(qd, cd, idd, idd, 0, lse, dq, dc)
    loss, rdq, rdc = O.inbatch_softmax_loss(q.astype(np.float64), c.astype(np.float64), ids)
    assert abs(ls.sum().item() - loss) <= 1e-5 * abs(loss)
    s = np.abs(rdq).max()
    np.testing.assert_allclose(dq.cpu().numpy(), rdq, rtol=1e-4, atol=1e-5 * s)
    lse2 = torch.empty(Bq, device=dev); ls2 = torch.zeros(64, dtype=torch.float64, device=dev); dq2 = torch.empty(Bq, dim, device=dev)
    ops.inbatch_softmax_lse_grad_q(qd, cd, idd, idd, 0, lse2, ls2, dq2)                # the one-sweep form
    assert abs(ls2.sum().item() - loss) <= 1e-5 * abs(loss)
    np.testing.assert_allclose(lse2.cpu().numpy(), lse.cpu().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(dq2.cpu().numpy(), rdq, rtol=1e-4, atol=1e-5 * s)
    np.testing.assert_allclose(dc.cpu().numpy(), rdc, rtol=1e-4, atol=1e-5 * np.abs(rdc).max())


@pytest.mark.parametrize("Bq,Bc,dim,idt", [(1, 1, 1, torch.int32), (3, 130, 7, torch.int64), (129, 63, 33, torch.int32), (260, 2, 100, torch.int32),
                                           (1, 700, 128, torch.int64), (513, 257, 104, torch.int32), (64, 64, 57, torch.int32)])
@pytest.mark.parametrize("with_ws", [True, False])
def test_inbatch_softmax_ragged_shapes_and_workspace(dev, Bq, Bc, dim, idt, with_ws):
    """rectangular / ragged (rows not multiples of 128 / 64, features not multiples of 4 / 8 / 32, one row, one column), int64 ids,
    a diagonal that leaves the matrix for some queries, with the split workspace and through the C-ABI without one (ws = NULL:
    one workgroup per 128 rows writes its results directly); no ids = no accidental-hit mask."""
    ops, lib = _m("ops"), _m("_lib").load()
    rng = np.random.default_rng(Bq * 1000 + Bc + dim)
    q = rng.normal(0, 0.5, (Bq, dim)).astype(np.float32); c = rng.normal(0, 0.5, (Bc, dim)).astype(np.float32)
    off = 0 if Bq <= Bc else 0                               # query i's positive is candidate i (+ off); queries past Bc have none
    cid = rng.integers(0, 9, Bc); qid = np.where(np.arange(Bq) < Bc, cid[np.minimum(np.arange(Bq), Bc - 1)], rng.integers(0, 9, Bq))
    td = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    qd, cd, qi, ci = td(q), td(c), td(qid).to(idt), td(cid).to(idt)
    # float64 reference with the same rule: S_ij += float32.min / 100 where cand id == query's id and j is not the diagonal
    S = q.astype(np.float64) @ c.astype(np.float64).T
    diag = np.zeros((Bq, Bc), bool); ii = np.arange(Bq); ok = ii + off < Bc; diag[ii[ok], ii[ok] + off] = True
    S = S + np.where((qid[:, None] == cid[None, :]) & ~diag, float(np.finfo(np.float32).min) / 100, 0.0)
    m = S.max(1, keepdims=True); lse_ref = (m + np.log(np.exp(S - m).sum(1, keepdims=True)))[:, 0]
    P = np.exp(S - lse_ref[:, None]) - diag
    rdq, rdc = P @ c.astype(np.float64), P.T @ q.astype(np.float64)
    loss_ref = float((lse_ref - np.where(ok, S[ii, np.minimum(ii + off, Bc - 1)], 0.0)).sum())
    lse = torch.empty(Bq, device=dev); ls = torch.zeros(64, dtype=torch.float64, device=dev)
    dq, dc = torch.full((Bq, dim), 7.0, device=dev), torch.full((Bc, dim), 7.0, device=dev)
    if with_ws:
        ops.inbatch_softmax_lse(qd, cd, qi, ci, off, lse, ls)
        ops.inbatch_softmax_grad(qd, cd, qi, ci, off, lse, dq, dc)
    else:
        ty = ops.I64 if idt == torch.int64 else ops.I32
        assert lib.brInBatchSoftmaxLse(qd.data_ptr(), cd.data_ptr(), qi.data_ptr(), ci.data_ptr(), ty, Bq, Bc, dim, off, lse.data_ptr(), ls.data_ptr(), None, 0, ops._stream()) == 0
        assert lib.brInBatchSoftmaxGrad(qd.data_ptr(), cd.data_ptr(), qi.data_ptr(), ci.data_ptr(), ty, Bq, Bc, dim, off, lse.data_ptr(), dq.data_ptr(), dc.data_ptr(), None, 0,
                                        ops._stream()) == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(lse.cpu().numpy(), lse_ref, rtol=1e-5, atol=1e-5)
    assert abs(ls.sum().item() - loss_ref) <= 1e-5 * max(1.0, abs(loss_ref))
    np.testing.assert_allclose(dq.cpu().numpy(), rdq, rtol=1e-4, atol=1e-5 * max(1e-3, np.abs(rdq).max()))
    np.testing.assert_allclose(dc.cpu().numpy(), rdc, rtol=1e-4, atol=1e-5 * max(1e-3, np.abs(rdc).max()))
    sm = ops.score_matrix(qd, cd).cpu().numpy()
    np.testing.assert_allclose(sm, q.astype(np.float64) @ c.astype(np.float64).T, rtol=1e-5, atol=1e-5)
    # lse + loss + dQ in one sweep (online softmax; above 64 features it runs the two passes itself)
    lse2 = torch.empty(Bq, device=dev); ls2 = torch.zeros(64, dtype=torch.float64, device=dev); dq2 = torch.full((Bq, dim), 7.0, device=dev)
    if with_ws:
        ops.inbatch_softmax_lse_grad_q(qd, cd, qi, ci, off, lse2, ls2, dq2)
    else:
        ty = ops.I64 if idt == torch.int64 else ops.I32
        assert lib.brInBatchSoftmaxLseGradQ(qd.data_ptr(), cd.data_ptr(), qi.data_ptr(), ci.data_ptr(), ty, Bq, Bc, dim, off, lse2.data_ptr(), ls2.data_ptr(), dq2.data_ptr(),
                                            None, 0, ops._stream()) == 0
    np.testing.assert_allclose(lse2.cpu().numpy(), lse_ref, rtol=1e-5, atol=1e-5)
    assert abs(ls2.sum().item() - loss_ref) <= 1e-5 * max(1.0, abs(loss_ref))
    np.testing.assert_allclose(dq2.cpu().numpy(), rdq, rtol=1e-4, atol=1e-5 * max(1e-3, np.abs(rdq).max()))


def test_inbatch_softmax_sharded_columns(dev):
    """data-parallel form: a rank's Bq queries against the all-gathered Bc candidates, diag_offset = rank*Bq."""
    ops = _m("ops")
    rng = np.random.default_rng(5)
    W, Bq, dim = 3, 70, 50
    Bc = W * Bq
    Q = rng.normal(0, 0.4, (Bc, dim)); C = rng.normal(0, 0.4, (Bc, dim)); ids = rng.integers(0, 40, Bc)
    loss, rdq, rdc = O.inbatch_softmax_loss(Q, C, ids)
    td = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
    Cd, idsd = td(C), td(ids, torch.int32)
    tot, dQ, lse_all = 0.0, [], []
    for r in range(W):
        sl = slice(r * Bq, (r + 1) * Bq)
        lse = torch.empty(Bq, device=dev); ls = torch.zeros(64, dtype=torch.float64, device=dev)
        ops.inbatch_softmax_lse(td(Q[sl]), Cd, idsd[sl].contiguous(), idsd, r * Bq, lse, ls)
        dq = torch.empty(Bq, dim, device=dev)
        ops.inbatch_softmax_grad(td(Q[sl]), Cd, idsd[sl].contiguous(), idsd, r * Bq, lse, dq, None)
        tot += ls.sum().item(); dQ.append(dq.cpu().numpy()); lse_all.append(lse)
    assert abs(tot - loss) <= 1e-5 * abs(loss)
    np.testing.assert_allclose(np.concatenate(dQ), rdq, rtol=1e-4, atol=1e-5 * np.abs(rdq).max())
    # dC for rank r's candidates against ALL queries: diag_offset = -r*Bq
    lse_g = torch.cat(lse_all)
    for r in range(W):
        sl = slice(r * Bq, (r + 1) * Bq)
        dc = torch.empty(Bq, dim, device=dev)
        ops.inbatch_softmax_grad(td(Q), td(C[sl]), idsd, idsd[sl].contiguous(), -r * Bq, lse_g, None, dc)
        np.testing.assert_allclose(dc.cpu().numpy(), rdc[sl], rtol=1e-4, atol=1e-5 * np.abs(rdc).max())


@pytest.mark.parametrize("rd_zero", [False, True])
def test_twotower_step_vs_golden(dev, rd_zero):
    tt = _m("two_tower")
    z = np.load(os.path.join(GOLD, "twotower_e75_s50_b64.npz"), allow_pickle=False)
    tag = "rdzero" if rd_zero else "softmax"
    E, S = z["p_Wu"].shape
    B = z["users"].shape[0]
    eng = tt.TwoTowerEngine(E, z["p_item_emb"].shape[0] - 2, z["p_user_emb"].shape[0] - 2, S, dev, B, lr=0.1, rd_zero=rd_zero)
    td = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    eng.user_emb.copy_(td(z["p_user_emb"])); eng.item_emb.copy_(td(z["p_item_emb"]))
    Wu, bu = eng.W("user"); Wi, bi = eng.W("item")
    Wu.copy_(td(z["p_Wu"])); bu.copy_(td(z["p_bu"])); Wi.copy_(td(z["p_Wi"])); bi.copy_(td(z["p_bi"]))
    eng.train_step(td(z["users"]), td(z["items"]), td(z["labels"]))
    torch.cuda.synchronize(); eng.check_ids()
    np.testing.assert_allclose(eng.q[:B].cpu().numpy(), z[tag + "_q"], rtol=1e-5, atol=5e-6 * np.abs(z[tag + "_q"]).max())
    loss = eng.pop_loss()
    assert abs(loss - float(z[tag + "_loss"])) <= 1e-5 * abs(float(z[tag + "_loss"])), (loss, float(z[tag + "_loss"]))
    gWu = eng._gW("user").cpu().numpy()
    ref = np.concatenate([z[tag + "_g_Wu"].reshape(-1), z[tag + "_g_bu"]])
    # dW = eu^T (P - I) c: the P and I parts cancel (sum_j P_ij = 1), so the fp32 error scale is the
    # magnitude-propagated |eu|^T (P + I) |c| ~ 50x the result here, not the result itself
    np.testing.assert_allclose(gWu, ref, rtol=1e-4, atol=2e-4 * np.abs(ref).max())
    np.testing.assert_allclose(eng.dei[:B].cpu().numpy(), z[tag + "_rg_item_emb"], rtol=1e-4, atol=1e-5 * np.abs(z[tag + "_rg_item_emb"]).max())
    # Keras Adagrad on the (deduplicated) item rows
    acc0 = np.full(z["p_item_emb"].shape, 0.1)
    ref_t, _ = O.adagrad_sparse(z["p_item_emb"], acc0, z["items"], z[tag + "_rg_item_emb"], 0.1)
    np.testing.assert_allclose(eng.item_emb.cpu().numpy(), ref_t, rtol=1e-5, atol=2e-3 * 0.1)
    assert np.median(np.abs(eng.item_emb.cpu().numpy() - ref_t)) <= 1e-7


def test_twotower_compute_loss_vs_golden(dev):
    """TwoTowerModel.computeLoss / computeLossTfrs / computeLossRdZero (trainers/twoTower.py:42-47,82-87) on given (q, c, info): the committed
    oracle fixture's q, c and losses (tests/golden/twotower_e75_s50_b64.npz: TFRS in-batch softmax with accidental-hit mask, SUM; sigmoid-dot
    BCE, mean) within 1e-5 relative; also through computeEmb on the fixture's parameters, and the step-loss bookkeeping after a pop_loss()."""
    models = _m("models")
    z = np.load(os.path.join(GOLD, "twotower_e75_s50_b64.npz"), allow_pickle=False)
    E, S = z["p_Wu"].shape
    nU, nI = z["p_user_emb"].shape[0] - 2, z["p_item_emb"].shape[0] - 2
    users, items = [f"u{k}" for k in range(nU)], [f"m{k}" for k in range(nI)]
    td = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    info = {"CUSTOMER_ID": [users[k - 2] for k in z["users"]], "MATERIAL": [items[k - 2] for k in z["items"]], "RATING_TYPE": z["labels"]}
    for rd in (False, True):
        tag = "rdzero" if rd else "softmax"
        m = models.TwoTowerModel(E, nI, nU, "CUSTOMER_ID", "MATERIAL", users, items, semb=S, max_batch=64, rdZero=rd, resKey="RATING_TYPE")
        got = float(m.computeLoss(td(z[tag + "_q"]), td(z[tag + "_c"]), info))
        want = float(z[tag + "_loss"])
        assert abs(got - want) <= 1e-5 * abs(want), (tag, got, want)
        direct = float((m.computeLossRdZero if rd else m.computeLossTfrs)(td(z[tag + "_q"]), td(z[tag + "_c"]), info))
        assert direct == got
        e = m.engine
        e.user_emb.copy_(td(z["p_user_emb"])); e.item_emb.copy_(td(z["p_item_emb"]))
        Wu, bu = e.W("user"); Wi, bi = e.W("item")
        Wu.copy_(td(z["p_Wu"])); bu.copy_(td(z["p_bu"])); Wi.copy_(td(z["p_Wi"])); bi.copy_(td(z["p_bi"]))
        q, c = m.computeEmb(info)                                           # twoTower.py:77-80 -> 82-87
        assert abs(float(m.computeLoss(q, c, info)) - want) <= 1e-5 * abs(want)
        l1 = float(m.train_step(info)["loss"])                              # the same loss through the step (first step: the fixture's parameters)
        assert abs(l1 - want) <= 1e-5 * abs(want), (tag, l1, want)
        e.pop_loss()                                                        # somebody reads and clears the engine's running sum ...
        l2 = float(m.train_step(info)["loss"])                              # ... the next step still reports its own loss (not a difference against the old total)
        assert 0.0 < l2 < l1


@pytest.mark.parametrize("optimizer", ["adam_dense", "adam_lazy"])
def test_bpr_engine_three_steps(dev, optimizer):
    bpr = _m("bpr")
    rng = np.random.default_rng(21)
    U, I, F, B = 211, 89, 32, 256
    eng = bpr.BPREngine(U, I, F, dev, B, optimizer=optimizer)
    ut, it = eng.user.cpu().numpy().astype(np.float64), eng.item.cpu().numpy().astype(np.float64)
    mu, vu, mi, vi = (np.zeros_like(x) for x in (ut, ut, it, it))
    td = lambda a: torch.from_numpy(a).to(dev).int()
    losses = []
    for t in range(1, 4):
        u, p, n = rng.integers(0, U, B), rng.integers(0, I, B), rng.integers(0, I, B)
        p[:40] = 7; n[:10] = 7
        eng.train_step(td(u), td(p), td(n))
        loss, _, (gu, gp, gn) = O.bpr_step_grads(ut, it, u, p, n)
        losses.append(loss)
        lazy = optimizer == "adam_lazy"
        ut, mu, vu = O.adam_sparse_tf(ut, mu, vu, u, gu, 1e-3, t, lazy=lazy)
        it, mi, vi = O.adam_sparse_tf(it, mi, vi, np.concatenate([p, n]), np.concatenate([gp, gn]), 1e-3, t, lazy=lazy)
    torch.cuda.synchronize(); eng.check_ids()
    assert abs(eng.pop_loss() - np.mean(losses)) <= 1e-5 * np.mean(losses)
    np.testing.assert_allclose(eng.user.cpu().numpy(), ut, rtol=1e-5, atol=5e-3 * 3e-3)
    np.testing.assert_allclose(eng.item.cpu().numpy(), it, rtol=1e-5, atol=5e-3 * 3e-3)
    assert np.median(np.abs(eng.item.cpu().numpy() - it)) <= 1e-7
    sc = eng.predict_scores(td(np.array([3, 5]))).cpu().numpy()
    np.testing.assert_allclose(sc[0], O.bpr_predict(ut, it, 3, np.arange(I)), rtol=1e-4, atol=1e-6)


def test_bpr_deferred_adam_is_bit_equal_to_the_sweep(dev):
    """dense_impl="deferred" (per-row replay of the missed g = 0 steps) and "sweep" (every row every step) give the same tables bit
    for bit - rows that sit out several steps, rows hit as positive and negative, a reload in between, the alpha ring wrapping."""
    bpr = _m("bpr")
    g = torch.Generator().manual_seed(4)
    U, I, F, B = 400, 150, 16, 128
    a = bpr.BPREngine(U, I, F, dev, B, optimizer="adam_dense", dense_impl="deferred", init_seed=3, replay="exact")      # (the fast form: test_gpu_sparse_optim.py)
    b = bpr.BPREngine(U, I, F, dev, B, optimizer="adam_dense", dense_impl="sweep", init_seed=3)
    assert a.deferred and not b.deferred and torch.equal(a.user, b.user)
    draw = lambda N: torch.randint(0, N, (B,), generator=g).int().to(dev)
    for t in range(1, 1101):                                   # > BR_ALPHA_RING - 8 steps: one forced flush on the way
        u, p, n = draw(U), draw(I), draw(I)
        if t % 7 == 0:
            p[:20] = 5; n[:9] = 5
        a.train_step(u, p, n); b.train_step(u, p, n)
        if t in (3, 40, 1100):
            assert torch.equal(a.user, b.user) and torch.equal(a.item, b.item), t
            assert torch.equal(a.user_m, b.user_m) and torch.equal(a.item_v, b.item_v), t
        if t == 40:                                            # reload into a fresh deferred engine and carry on
            c = bpr.BPREngine(U, I, F, dev, B, optimizer="adam_dense", dense_impl="deferred", init_seed=99, replay="exact")
            c.load_state_dict({k: (v.clone() if torch.is_tensor(v) else v) for k, v in a.state_dict().items()})
            a = c
    a.check_ids(); b.check_ids()


def test_bpr_graph_replay_equals_eager(dev):
    """BPREngine.enable_graph: the step as one hipGraph (dedup indexes on forked side streams inside the capture) == the eager launches,
    bit for bit, incl. a ragged batch in between that falls back to the eager path."""
    bpr = _m("bpr")
    g = torch.Generator().manual_seed(6)
    U, I, F, B = 3000, 700, 32, 512
    a = bpr.BPREngine(U, I, F, dev, B, init_seed=4)
    b = bpr.BPREngine(U, I, F, dev, B, init_seed=4)
    draw = lambda N, n=B: torch.randint(0, N, (n,), generator=g).int().to(dev)
    for _ in range(2):                                           # enable after some training: the capture must not disturb the state
        u, p, n = draw(U), draw(I), draw(I)
        a.train_step(u, p, n); b.train_step(u, p, n)
    a.enable_graph(B)
    for t in range(6):
        nb = 300 if t == 3 else B
        u, p, n = draw(U, nb), draw(I, nb), draw(I, nb)
        a.train_step(u, p, n); b.train_step(u, p, n)
    a.check_ids(); b.check_ids()
    assert a.t == b.t == 8
    assert torch.equal(a.user, b.user) and torch.equal(a.item, b.item) and torch.equal(a.item_v, b.item_v)
    assert abs(a.pop_loss() - b.pop_loss()) < 1e-9


@pytest.mark.parametrize("graph", [False, True])
def test_bpr_fused_gather_sort_step_equals_the_side_stream_step(dev, graph):
    """Batches >= 1024 at 64 / 128 / 256 factors take the five-launch step (brGatherRowsDeferredPairWithIndex: the chunk sorts of the user
    stream (B ids) and the [pos | neg] stream (2 B ids) ride in the gather's launch, the chunk-rank launch advances the step state behind
    it; no side stream).  Against the same engine with BR_FUSED_SORT=0 (step-state launch, indexes on two side streams, plain pair gather):
    tables, moments and last[] bit for bit over steps with lags, hot ids, a ragged batch and, optionally, the hipGraph replay."""
    bpr = _m("bpr")
    g = torch.Generator().manual_seed(12)
    U, I, F, B = 20000, 3000, 64, 4096
    a = bpr.BPREngine(U, I, F, dev, B, init_seed=7)
    b = bpr.BPREngine(U, I, F, dev, B, init_seed=7)
    draw = lambda N, n=B: torch.randint(0, N, (n,), generator=g).int().to(dev)
    if graph:
        a.enable_graph(B)
    for t in range(6):
        nb = 3000 if t == 2 else B
        u, p, n = draw(U, nb), draw(I, nb), draw(I, nb)
        u[:500] = 11; p[:300] = 5; n[:100] = 5                   # hot rows: runs that cross strips and 64-blocks; an id that is positive and negative
        a.train_step(u, p, n)
        os.environ["BR_FUSED_SORT"] = "0"
        try:
            b.train_step(u, p, n)
        finally:
            del os.environ["BR_FUSED_SORT"]
    torch.cuda.synchronize()
    a.check_ids(); b.check_ids()
    assert a.t == b.t == 6 and int(a.step_state[0].item()) == int(b.step_state[0].item()) == 6
    for k in ("_user", "_item", "user_m", "user_v", "item_m", "item_v", "user_last", "item_last"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert abs(a.pop_loss() - b.pop_loss()) < 1e-9


@pytest.mark.parametrize("rd_zero", [False, True])
def test_twotower_graph_replay_equals_eager(dev, rd_zero):
    """TwoTowerEngine.enable_graph: the step as one hipGraph (dedup indexes on forked side streams inside the capture) == the eager launches,
    incl. a ragged batch in between that falls back to the eager path."""
    tt = _m("two_tower")
    g = torch.Generator().manual_seed(8)
    U, I, E, S, B = 900, 300, 24, 16, 256
    a = tt.TwoTowerEngine(E, I, U, S, dev, B, rd_zero=rd_zero, init_seed=2)
    b = tt.TwoTowerEngine(E, I, U, S, dev, B, rd_zero=rd_zero, init_seed=2)
    draw = lambda N, n=B: (torch.randint(0, N, (n,), generator=g) + 2).int().to(dev)
    lab = lambda n=B: (torch.rand(n, generator=g) < 0.4).float().to(dev)
    a.enable_graph(B)
    for t in range(7):
        nb = 100 if t == 4 else B
        u, i, y = draw(U, nb), draw(I, nb), lab(nb)
        u[:9] = u[0]                                                 # duplicate ids in both streams
        a.train_step(u, i, y); b.train_step(u, i, y)
    torch.cuda.synchronize()
    a.check_ids(); b.check_ids()
    assert a._graph["graph"] is not None and a.t == b.t == 7
    for k in ("user_emb", "item_emb", "theta", "user_acc", "theta_acc"):
        np.testing.assert_allclose(getattr(a, k).cpu().numpy(), getattr(b, k).cpu().numpy(), rtol=1e-6, atol=1e-7, err_msg=k)
    la, lb = a.pop_loss(), b.pop_loss()
    assert abs(la - lb) <= 1e-6 * abs(lb)


def test_topk_ties_and_scores(dev):
    ops = _m("ops")
    z = np.load(os.path.join(GOLD, "topk_ties.npz"), allow_pickle=False)
    s = torch.from_numpy(z["scores"]).to(dev)
    ts, ti = ops.topk_rows(s, z["topk_index"].shape[1])
    assert np.array_equal(ti.cpu().numpy(), z["topk_index"])           # ties keep the lower item position
    assert np.array_equal(ts.cpu().numpy(), z["topk_scores"])
    rng = np.random.default_rng(2)
    q = rng.normal(size=(150, 50)).astype(np.float32); c = rng.normal(size=(333, 50)).astype(np.float32)
    sm = ops.score_matrix(torch.from_numpy(q).to(dev), torch.from_numpy(c).to(dev)).cpu().numpy()
    np.testing.assert_allclose(sm, q.astype(np.float64) @ c.astype(np.float64).T, rtol=1e-5, atol=1e-5)
    # k == n_items edge, single row
    one = torch.tensor([[0.5, 0.5, 0.1, 0.9]], device=dev)
    ts, ti = ops.topk_rows(one, 4)
    assert ti.cpu().tolist() == [[3, 0, 1, 2]]
