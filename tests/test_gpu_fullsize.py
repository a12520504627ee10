"""-m gpu parity AT THE BENCHMARKED SIZES (BASELINE.json configs 2-5), against oracle/binrec_oracle.py in float64.

  config 2: NeuMF-A dim 64, 1 M users x 100 K items, batch 65 536 (what bench.py times): uniform and Zipf(1.05) ids,
            int32 and int64 ids, a ragged 40 000-row batch, and the same through the hipGraph replay; three steps, so the
            deferred Keras-Adam replay sees rows that sat out a step.  At batch > 32 768 every wave of the dense kernels
            walks a second 16-row tile (register-prefetched hand-over), brHeadSlabs = 256 and the 8 BatchNorm stat
            replicas run under real contention: this is the only place those paths are value-checked.
  config 3: BPR, 65 536 triplets, same tables.
  config 4: in-batch softmax stripe: 1 024 queries x 65 536 all-gathered candidates (per-rank shape of the 8-GPU job).
  config 5: dim 128 (first layer K = 256 as two K-halves) with int64 ids.
The oracle runs on the COMPACT tables (rows any step touches + a sample of rows none does): Keras' non-lazy Adam is
row-independent, so this equals the full-table oracle on those rows and keeps the CPU side to seconds.
Observed maximum errors (relative to each bound) are written to gpurun_out/parity_errors_fullsize.json."""
import json
import os
from importlib import import_module

import numpy as np
import pytest
import torch

from oracle import binrec_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ERRS = {}


def _m(name):
    return import_module("binary-recommendation_amd." + name)


def _record(case, name, err, bound):
    """keep the worst observed error/bound ratio per (case, tensor); asserted by the caller"""
    r = float(np.max(err / bound)) if np.size(err) else 0.0
    d = _ERRS.setdefault(case, {})
    d[name] = max(d.get(name, 0.0), r)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity_errors_fullsize.json"), "w") as f:
            json.dump({"note": "max |got - oracle| / bound per tensor (<= 1 passes); bounds are written in tests/test_gpu_fullsize.py", "cases": _ERRS}, f, indent=1)
    except OSError:
        pass
    return r


def _check(case, name, got, ref, rtol, atol):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    bound = atol + rtol * np.abs(ref)
    r = _record(case, name, np.abs(got - ref), bound)
    assert r <= 1.0, f"{case}: {name}: max error / bound = {r:.3g}"


def _zipf(rng, n, N, a=1.05):
    u = rng.random(n)
    r = ((N ** (1 - a) - 1) * u + 1) ** (1 / (1 - a))
    return np.clip(r.astype(np.int64), 1, N) - 1


def _ids(rng, kind, n, N):
    return _zipf(rng, n, N) if kind == "zipf" else rng.integers(0, N, n)


def _compact(id_lists, N, rng, n_extra=2000):
    """sorted unique rows: every id of every step + n_extra random others; ids -> positions"""
    keep = np.unique(np.concatenate(list(id_lists) + [rng.integers(0, N, n_extra)]))
    return keep, [np.searchsorted(keep, x) for x in id_lists]


K1, M1 = 100_000, 1_000_000
NEUMF_CASES = [("uniform", torch.int32, 65536, False, 64, M1, K1), ("zipf", torch.int32, 65536, False, 64, M1, K1), ("uniform", torch.int64, 65536, False, 64, M1, K1),
               ("uniform", torch.int32, 40000, False, 64, M1, K1), ("uniform", torch.int32, 65536, True, 64, M1, K1), ("uniform", torch.int64, 40000, False, 128, M1, K1),
               # a user table of 17 M x 128 floats = 2.18 G elements (8.7 GB, 26 GB with the Adam slots): rows past element 2^31 are
               # looked up, replayed and updated
               ("uniform", torch.int32, 16384, False, 64, 17_000_000, K1),
               # BASELINE config 5's per-GPU shard as it stands on one of the 8 GPUs: 12.5 M x 1.25 M rows of 256 floats (3.2 G + 0.32 G
               # elements, 42 GB with the Adam slots), int64 ids, batch 65 536: the VEC = 4 wave kernels (lookup, Adam rows, flush) at
               # element offsets >= 2^31 and byte offsets >= 2^33
               ("uniform", torch.int64, 65536, False, 128, 12_500_000, 1_250_000)]
NEUMF_CASES = [c + ("A",) for c in NEUMF_CASES] + [
    # the other reference graph (src/models/NeuMFModel.py:53-100: user-first concat, relu tower dim -> dim/2 -> dim/4, MSE) at the benchmarked size
    ("uniform", torch.int32, 65536, False, 64, M1, K1, "B")]


@pytest.mark.parametrize("kind,idt,B,graph,dim,U,I,variant", NEUMF_CASES)
def test_neumf_steps_at_bench_size(dev, kind, idt, B, graph, dim, U, I, variant):
    neumf = _m("neumf")
    case = f"neumf_{kind}_{'i64' if idt == torch.int64 else 'i32'}_b{B}_d{dim}{'_graph' if graph else ''}" + (f"_u{U}" if U != M1 else "") + (f"_i{I}" if I != K1 else "") + ("" if variant == "A" else "_" + variant)
    D, steps = dim, 3
    cfg = neumf.NeuMFConfig(variant=variant, dim=D, optimizer="adam_dense", dense_impl="deferred", seed=0x1234ABCD5)
    eng = neumf.NeuMFEngine(cfg, U, I, dev, B, id_dtype=idt, init_seed=1)
    rng = np.random.default_rng(17)
    # non-trivial biases / BatchNorm parameters / moving statistics
    n1, n2, n3 = cfg.hidden
    for k, (mu, sd) in {"b1": (0, .1), "b2": (0, .1), "b3": (0, .1), "b4": (0, .1), "be1": (0, .1), "be2": (0, .1), "g1": (1, .1), "g2": (1, .1)}.items():
        eng.theta.view(k).copy_(torch.from_numpy(rng.normal(mu, sd, eng.theta.view(k).shape).astype(np.float32)))
    us = [_ids(rng, kind, B, U) for _ in range(steps)]
    its = [_ids(rng, kind, B, I) for _ in range(steps)]
    ys = [(rng.random(B) < 0.25).astype(np.float32) for _ in range(steps)]
    us[2][:B // 2] = us[0][:B // 2]            # rows touched at step 1, idle at step 2, touched at step 3 (replay lag 1)
    if U * 2 * D > (1 << 31):                  # the table's last rows (element offsets >= 2^31) in every step
        for t in range(steps):
            us[t][-256:] = U - 1 - rng.integers(0, 100_000, 256)
            us[t][-512:-256] = (1 << 31) // (2 * D) + rng.integers(-64, 64, 256)      # rows on either side of element 2^31
        assert (min(int(x[-256:].min()) for x in us) * 2 * D) >= (1 << 31)
    ku, cu = _compact(us, U, rng)
    ki, ci = _compact(its, I, rng)
    spec = O.NeuMFSpec(variant, dim=D)
    tku, tki = torch.from_numpy(ku).to(dev), torch.from_numpy(ki).to(dev)
    P = {k: eng.theta.view(k).cpu().numpy().astype(np.float64) for k in O.DENSE_ORDER}
    P["W4"] = P["W4"].reshape(-1)
    for k in neumf.TABLES:
        P[k] = eng.tables[k][tku if k.startswith("user") else tki].cpu().numpy().astype(np.float64)
    P.update({k: eng.moving[k].cpu().numpy().astype(np.float64) for k in ("mm1", "mv1", "mm2", "mv2")})
    names = list(O.DENSE_ORDER) + list(neumf.TABLES)
    M = {k: np.zeros_like(P[k]) for k in names}
    V = {k: np.zeros_like(P[k]) for k in names}
    if graph:
        eng.enable_graph(B)
    td = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
    for t in range(1, steps + 1):
        u, i, y = us[t - 1], its[t - 1], ys[t - 1]
        eng.train_step(td(u, idt), td(i, idt), td(y, torch.float32))
        torch.cuda.synchronize()
        eng.check_ids()
        masks = [O.dropout_mask(cfg.seed, t, s, B, w, cfg.dropout) for s, w in enumerate((2 * D, n1, n2))]
        loss, c, g, rg, ns = O.neumf_step_grads(spec, P, cu[t - 1], ci[t - 1], y, masks, dt=np.float64)
        if t == 1:     # G1/T1: the concat of the four gathered rows, bit for bit (no replay yet: every row is at step 0)
            x0 = np.concatenate([P["item_mlp"][ci[0]], P["user_mlp"][cu[0]]] if cfg.item_first else [P["user_mlp"][cu[0]], P["item_mlp"][ci[0]]], axis=1).astype(np.float32)
            assert np.array_equal(eng.x0[:B].cpu().numpy().view(np.uint32), x0.view(np.uint32)), "x0 not bit-exact"
        # t = 1: both sides hold the same fp32 parameters -> the north_star bounds (1e-5 relative).  t >= 2: the fp32 parameters
        # have taken Adam steps whose m / (sqrt(v) + eps) amplifies the rounding of nearly cancelled gradients (bounded at the
        # end of this test), so the activations of the two sides are those of slightly different models: 8x the bounds.
        f = 1.0 if t == 1 else 8.0
        zmax = np.abs(c["logit"]).max()
        _check(case, f"logit[t={t}]", eng.logit[:B].cpu().numpy(), c["logit"], f * 1e-5, f * 5e-6 * zmax)      # north_star: 1e-5 relative
        _check(case, f"prob[t={t}]", eng.prob[:B].cpu().numpy(), c["prob"], f * 1e-5, f * 5e-6)
        m = eng.pop_metrics(B)
        _check(case, f"loss[t={t}]", m["loss"], loss, f * 1e-5, 0.0)
        _check(case, f"a1[t={t}]", eng.a1[:B].cpu().numpy(), c["a1"], f * 1e-5, f * 5e-6)
        _check(case, f"a2[t={t}]", eng.a2[:B].cpu().numpy(), c["a2"], f * 1e-5, f * 5e-6)
        _check(case, f"a3[t={t}]", eng.a3[:B].cpu().numpy(), c["a3"], f * 1e-5, f * 5e-6)
        for k in O.DENSE_ORDER:        # batch sums, many of them nearly cancelled: bound = 1e-5 x sum of |summands|
            got = eng.grad.view(k).cpu().numpy().reshape(g[k].shape)
            _check(case, f"grad {k}[t={t}]", got, g[k], 0.0, f * 1e-5 * c["gabs"][k] + 1e-12)
        for k, (gt, _ld) in eng.row_grad_views(B).items():
            _check(case, f"row grad {k}[t={t}]", gt.cpu().numpy(), rg[k], f * 1e-4, f * 1e-5 * np.abs(rg[k]).max())
        for k in ("mm1", "mv1", "mm2", "mv2"):
            _check(case, f"{k}[t={t}]", eng.moving[k].cpu().numpy(), ns[k], f * 1e-5, f * 5e-6 * np.abs(ns[k]).max())
        for k in O.DENSE_ORDER:
            P[k], M[k], V[k] = O.adam_dense(P[k], M[k], V[k], g[k], cfg.lr, t)
        for k in neumf.TABLES:
            ids = cu[t - 1] if k.startswith("user") else ci[t - 1]
            P[k], M[k], V[k] = O.adam_sparse_tf(P[k], M[k], V[k], ids, rg[k], cfg.lr, t, lazy=False, dedup=O.dedup_rows_unordered)
        P.update(ns)
    # Keras-Adam state after 3 steps (deferred -> flush): touched rows + the sample of never-touched rows.
    # Adam's m / (sqrt(v) + eps) turns the fp32 rounding of a nearly cancelled gradient into an O(lr) move, so the
    # bound is 1e-5 relative + 0.5 % (tables) / 2 % (dense) of the distance 3 steps can travel, with the median
    # required to agree to 1e-7 (test_gpu_neumf.py::test_three_optimizer_steps, same bounds at toy size).
    travel = steps * cfg.lr
    for k in neumf.TABLES:
        sel = tku if k.startswith("user") else tki
        got = eng.tables[k][sel].cpu().numpy()
        _check(case, "table " + k, got, P[k], 1e-5, 5e-3 * travel)
        assert np.median(np.abs(got - P[k])) <= 1e-7, k
        _check(case, "adam m " + k, eng.tab_m[k][sel].cpu().numpy(), M[k], 1e-4, 1e-5 * np.abs(M[k]).max())
        _check(case, "adam v " + k, eng.tab_v[k][sel].cpu().numpy(), V[k], 1e-4, 1e-5 * np.abs(V[k]).max())
    for k in O.DENSE_ORDER:
        _check(case, "theta " + k, eng.theta.view(k).cpu().numpy().reshape(P[k].shape), P[k], 1e-5, 2e-2 * travel)


@pytest.mark.parametrize("kind", ["uniform", "zipf"])
def test_bpr_steps_at_bench_size(dev, kind):
    """config 3: BPR 1-sigmoid triplet step, 65 536 triplets on 1 M x 100 K tables, Keras-Adam (non-lazy), two steps."""
    bpr = _m("bpr")
    case = "bpr_" + kind
    U, I, F, B = 1_000_000, 100_000, 64, 65536
    eng = bpr.BPREngine(U, I, F, dev, B, optimizer="adam_dense")
    rng = np.random.default_rng(3)
    us = [_ids(rng, kind, B, U) for _ in range(2)]
    ps = [_ids(rng, kind, B, I) for _ in range(2)]
    ns_ = [_ids(rng, kind, B, I) for _ in range(2)]
    ku, cu = _compact(us, U, rng)
    ki, cpn = _compact(ps + ns_, I, rng)
    cp, cn = cpn[:2], cpn[2:]
    tku, tki = torch.from_numpy(ku).to(dev), torch.from_numpy(ki).to(dev)
    ut, it = eng.user[tku].cpu().numpy().astype(np.float64), eng.item[tki].cpu().numpy().astype(np.float64)
    mu, vu, mi, vi = (np.zeros_like(x) for x in (ut, ut, it, it))
    td = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).int()
    losses = []
    for t in range(1, 3):
        eng.train_step(td(us[t - 1]), td(ps[t - 1]), td(ns_[t - 1]))
        loss, per, (gu, gp, gn) = O.bpr_step_grads(ut, it, cu[t - 1], cp[t - 1], cn[t - 1])
        losses.append(loss)
        torch.cuda.synchronize(); eng.check_ids()
        _check(case, f"per-triplet loss[t={t}]", eng.per_triplet[:B].cpu().numpy(), per, 1e-5, 1e-6)
        _check(case, f"g_user[t={t}]", eng.g_user[:B].cpu().numpy(), gu, 1e-4, 1e-5 * np.abs(gu).max())
        _check(case, f"g_item[t={t}]", eng.g_item[:2 * B].cpu().numpy(), np.concatenate([gp, gn]), 1e-4, 1e-5 * np.abs(gp).max())
        ut, mu, vu = O.adam_sparse_tf(ut, mu, vu, cu[t - 1], gu, 1e-3, t, dedup=O.dedup_rows_unordered)
        it, mi, vi = O.adam_sparse_tf(it, mi, vi, np.concatenate([cp[t - 1], cn[t - 1]]), np.concatenate([gp, gn]), 1e-3, t, dedup=O.dedup_rows_unordered)
    _check(case, "mean loss", eng.pop_loss(), np.mean(losses), 1e-5, 0.0)
    travel = 2 * 1e-3
    for name, got, ref in (("user", eng.user[tku], ut), ("item", eng.item[tki], it)):
        g = got.cpu().numpy()
        _check(case, "table " + name, g, ref, 1e-5, 5e-3 * travel)
        assert np.median(np.abs(g - ref)) <= 1e-7


def test_inbatch_softmax_stripe_1024_x_65536(dev):
    """config 4, one rank's share of the 8-GPU step: its queries against the all-gathered 65 536 candidates (streaming LSE +
    accidental-hit mask + dQ), diag_offset = rank * Bq; oracle in column chunks.  Item ids from 20 000 items: ~3 accidental
    hits per row."""
    ops = _m("ops")
    case = "softmax_1024x65536"
    rng = np.random.default_rng(8)
    Bq, Bc, dim, rank = 1024, 65536, 64, 5
    off = rank * Bq
    C = rng.normal(0, 0.35, (Bc, dim)); Q = rng.normal(0, 0.35, (Bq, dim))
    cid = rng.integers(0, 20000, Bc); qid = cid[off:off + Bq].copy()
    td = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
    Qd, Cd, qd, cd = td(Q), td(C), td(qid, torch.int32), td(cid, torch.int32)
    lse = torch.empty(Bq, device=dev); ls = torch.zeros(64, dtype=torch.float64, device=dev)
    ops.inbatch_softmax_lse(Qd, Cd, qd, cd, off, lse, ls)
    dq = torch.empty(Bq, dim, device=dev)
    ops.inbatch_softmax_grad(Qd, Cd, qd, cd, off, lse, dq, None)
    torch.cuda.synchronize()
    Q32, C32 = Q.astype(np.float32).astype(np.float64), C.astype(np.float32).astype(np.float64)
    loss, rlse, rdq, _ = O.inbatch_softmax_stripe(Q32, C32, qid, cid, off)
    _check(case, "loss", ls.sum().item(), loss, 1e-5, 0.0)
    _check(case, "row lse", lse.cpu().numpy(), rlse, 1e-5, 1e-6)
    _check(case, "dQ", dq.cpu().numpy(), rdq, 1e-4, 1e-5 * np.abs(rdq).max())
    # the one-sweep form (lse + loss + dQ, online softmax, 64 column splits combined afterwards)
    lse2 = torch.empty(Bq, device=dev); ls2 = torch.zeros(64, dtype=torch.float64, device=dev); dq2 = torch.empty(Bq, dim, device=dev)
    ops.inbatch_softmax_lse_grad_q(Qd, Cd, qd, cd, off, lse2, ls2, dq2)
    torch.cuda.synchronize()
    _check(case, "loss (one sweep)", ls2.sum().item(), loss, 1e-5, 0.0)
    _check(case, "row lse (one sweep)", lse2.cpu().numpy(), rlse, 1e-5, 1e-6)
    _check(case, "dQ (one sweep)", dq2.cpu().numpy(), rdq, 1e-4, 1e-5 * np.abs(rdq).max())


def test_inbatch_softmax_dc_stripe_8192(dev):
    """dC of one rank's 1 024 candidates against ALL 8 192 queries (diag_offset = -rank * Bc_local), row lse from the kernel's
    own full pass checked against the oracle first."""
    ops = _m("ops")
    case = "softmax_dC_8192x1024"
    rng = np.random.default_rng(9)
    Bt, Bl, dim, rank = 8192, 1024, 64, 3
    Q = rng.normal(0, 0.35, (Bt, dim)).astype(np.float32); C = rng.normal(0, 0.35, (Bt, dim)).astype(np.float32)
    ids = rng.integers(0, 3000, Bt)
    td = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
    Qd, Cd, idd = td(Q), td(C), td(ids, torch.int32)
    lse = torch.empty(Bt, device=dev); ls = torch.zeros(64, dtype=torch.float64, device=dev)
    ops.inbatch_softmax_lse(Qd, Cd, idd, idd, 0, lse, ls)
    loss, rdq, rdc = O.inbatch_softmax_loss(Q.astype(np.float64), C.astype(np.float64), ids)
    _check(case, "loss", ls.sum().item(), loss, 1e-5, 0.0)
    sl = slice(rank * Bl, (rank + 1) * Bl)
    dc = torch.empty(Bl, dim, device=dev)
    ops.inbatch_softmax_grad(Qd, Cd[sl].contiguous(), idd, idd[sl].contiguous(), -rank * Bl, lse, None, dc)
    _check(case, "dC", dc.cpu().numpy(), rdc[sl], 1e-4, 1e-5 * np.abs(rdc).max())
    dq = torch.empty(Bt, dim, device=dev); dc_all = torch.empty(Bt, dim, device=dev)
    ops.inbatch_softmax_grad(Qd, Cd, idd, idd, 0, lse, dq, dc_all)
    _check(case, "dQ (square)", dq.cpu().numpy(), rdq, 1e-4, 1e-5 * np.abs(rdq).max())
    _check(case, "dC (square)", dc_all.cpu().numpy(), rdc, 1e-4, 1e-5 * np.abs(rdc).max())


def test_inbatch_softmax_square_4096_dim50(dev):
    """4 096 x 4 096 at 50 features (padded to 56 / 64 inside the kernel: the k-blocks and feature tiles past the data must stay zero) -
    the smallest streamed axis and an odd feature count on the bf16 matrix pipe (TP / TT piece images): lse, loss, dQ and dC
    by the two-pass kernels and lse + loss + dQ by the one-sweep kernel against the float64 oracle."""
    ops = _m("ops")
    case = "softmax_4096x4096x50"
    rng = np.random.default_rng(21)
    Bt, dim = 4096, 50
    Q = rng.normal(0, 0.4, (Bt, dim)).astype(np.float32); C = rng.normal(0, 0.4, (Bt, dim)).astype(np.float32)
    ids = rng.integers(0, 1500, Bt)
    td = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
    Qd, Cd, idd = td(Q), td(C), td(ids, torch.int32)
    lse = torch.empty(Bt, device=dev); ls = torch.zeros(64, dtype=torch.float64, device=dev)
    ops.inbatch_softmax_lse(Qd, Cd, idd, idd, 0, lse, ls)
    loss, rdq, rdc = O.inbatch_softmax_loss(Q.astype(np.float64), C.astype(np.float64), ids)
    _check(case, "loss", ls.sum().item(), loss, 1e-5, 0.0)
    dq = torch.empty(Bt, dim, device=dev); dc = torch.empty(Bt, dim, device=dev)
    ops.inbatch_softmax_grad(Qd, Cd, idd, idd, 0, lse, dq, dc)
    _check(case, "dQ", dq.cpu().numpy(), rdq, 1e-4, 1e-5 * np.abs(rdq).max())
    _check(case, "dC", dc.cpu().numpy(), rdc, 1e-4, 1e-5 * np.abs(rdc).max())
    lse2 = torch.empty(Bt, device=dev); ls2 = torch.zeros(64, dtype=torch.float64, device=dev); dq2 = torch.empty(Bt, dim, device=dev)
    ops.inbatch_softmax_lse_grad_q(Qd, Cd, idd, idd, 0, lse2, ls2, dq2)
    _check(case, "loss (one sweep)", ls2.sum().item(), loss, 1e-5, 0.0)
    _check(case, "row lse (one sweep) vs two-pass", lse2.cpu().numpy(), lse.cpu().numpy(), 1e-6, 1e-6)
    _check(case, "dQ (one sweep)", dq2.cpu().numpy(), rdq, 1e-4, 1e-5 * np.abs(rdq).max())
