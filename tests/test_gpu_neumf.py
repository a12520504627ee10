"""-m gpu parity of the NeuMF step (T1-T4, L1/L2, B1, S1, O1) against oracle/binrec_oracle.py.
Tolerance (north_star): 1e-5 relative on logits / loss; gather bit-exact (test_gpu_embedding)."""
from importlib import import_module

import numpy as np
import pytest
import torch

from oracle import binrec_oracle as O

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def _mods():
    return import_module("binary-recommendation_amd.ops"), import_module("binary-recommendation_amd.neumf")


def _close(got, ref, name, rtol=RTOL, atol_frac=5e-6):
    ref = np.asarray(ref, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    scale = np.abs(ref).max() + 1e-30
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol_frac * scale, err_msg=name)


def _setup(variant, dim, B, dev, U=97, I=53, seed=3, optimizer="adam_lazy", hidden=None, dense_impl="deferred", replay=None):
    ops, neumf = _mods()
    spec = O.NeuMFSpec(variant, dim=dim, hidden=hidden)
    p = O.neumf_init(spec, U, I, seed=seed, dt=np.float32)
    rng = np.random.default_rng(seed + 1)
    for k in ("b1", "b2", "b3", "b4", "be1", "be2"):
        p[k] = rng.normal(0, 0.1, p[k].shape).astype(np.float32)
    for k in ("g1", "g2"):
        p[k] = (1 + rng.normal(0, 0.1, p[k].shape)).astype(np.float32)
    for k in ("mm1", "mm2"):
        p[k] = rng.normal(0.4, 0.1, p[k].shape).astype(np.float32)
    for k in ("mv1", "mv2"):
        p[k] = rng.uniform(0.05, 0.3, p[k].shape).astype(np.float32)
    cfg = neumf.NeuMFConfig(variant=variant, dim=dim, hidden=hidden, optimizer=optimizer, seed=0xABCDEF12345, dense_impl=dense_impl, replay=replay)
    eng = neumf.NeuMFEngine(cfg, U, I, dev, max_batch=B)
    eng.load_numpy_params(p)
    u = rng.integers(0, U, B); i = rng.integers(0, I, B)
    u[: B // 8] = u[0]; i[: B // 6] = i[1]  # duplicate ids
    y = (rng.random(B) < 0.25).astype(np.float32)
    return ops, eng, spec, cfg, p, u, i, y


def _masks(cfg, spec, step, B, row0=0):
    widths = (2 * spec.dim, spec.hidden[0], spec.hidden[1])
    return [O.dropout_mask(cfg.seed, step, s, B, w, cfg.dropout, row0) for s, w in enumerate(widths)]


@pytest.mark.parametrize("variant,dim,B", [("A", 64, 300), ("B", 64, 257), ("A", 10, 64), ("B", 32, 7), ("A", 8, 1000), ("A", 10, 2048),
                                           ("A", 128, 300), ("B", 100, 129)])      # 2*dim > 128: the first layer runs as two K-halves (config 5)
def test_forward_backward_parity(dev, variant, dim, B):
    ops, eng, spec, cfg, p, u, i, y = _setup(variant, dim, B, dev)
    td = lambda a, dt: torch.from_numpy(a).to(dev).to(dt)
    eng.train_step(td(u, torch.int32), td(i, torch.int32), td(y, torch.float32))
    torch.cuda.synchronize()
    eng.check_ids()
    masks = _masks(cfg, spec, 1, B)
    loss, c, g, rg, ns = O.neumf_step_grads(spec, p, u, i, y, masks, dt=np.float64)
    # BatchNorm over a handful of rows divides by a tiny batch variance: fp32 rounding of the
    # activations is amplified by rstd, so small-batch cases get 4x the bound (inherent to fp32 BN,
    # not to the kernels); the headline 1e-5 holds from a few hundred rows up.
    f = 1.0 if B >= 256 else 4.0
    _close(eng.logit[:B].cpu().numpy(), c["logit"], "logit", rtol=f * RTOL, atol_frac=f * 5e-6)
    _close(eng.prob[:B].cpu().numpy(), c["prob"], "prob", rtol=f * RTOL, atol_frac=f * 5e-6)
    m = eng.pop_metrics(B)
    assert abs(m["loss"] - loss) <= f * RTOL * abs(loss), (m["loss"], loss)
    km = O.keras_metrics(c["prob"], y)
    assert abs(m["mse"] - km["mse"]) < 1e-6 and abs(m["mae"] - km["mae"]) < 1e-6
    assert abs(m["binary_accuracy"] - km["binary_accuracy"]) < 1e-9
    _close(eng.a1[:B].cpu().numpy(), c["a1"], "a1")
    _close(eng.a3[:B].cpu().numpy(), c["a3"], "a3", rtol=f * RTOL, atol_frac=f * 5e-6)
    # dense gradients: fp32 sums over the batch; the error scale is the sum of |summands|
    # (c["gabs"]), not the (often almost fully cancelled) result
    for k in O.DENSE_ORDER:
        got = eng.grad.view(k).cpu().numpy().reshape(g[k].shape).astype(np.float64)
        assert np.all(np.abs(got - g[k]) <= f * 1e-5 * c["gabs"][k] + 1e-12), "grad " + k
    # per-pair row gradients (IndexedSlices values)
    for k, (gt, _ld) in eng.row_grad_views(B).items():
        _close(gt.cpu().numpy(), rg[k], "row grad " + k, rtol=1e-4, atol_frac=1e-5)
    # moving statistics
    for k in ("mm1", "mv1", "mm2", "mv2"):
        _close(eng.moving[k].cpu().numpy(), ns[k], k)


@pytest.mark.parametrize("variant,dim,optimizer,impl", [("A", 64, "adam_dense", "deferred"), ("A", 64, "adam_dense", "sweep"), ("A", 64, "adam_lazy", "sweep"),
                                                        ("B", 32, "adam_dense", "deferred"), ("A", 10, "adam_dense", "deferred"), ("A", 10, "adam_dense", "sweep"),
                                                        ("A", 128, "adam_dense", "deferred")])
def test_three_optimizer_steps(dev, variant, dim, optimizer, impl):
    """Parameters after 3 steps of Keras-Adam (dense = non-lazy sparse apply [TF-sem]).
    Adam divides by sqrt(v): a gradient that is itself a nearly cancelled fp32 sum (pre-BN biases,
    a few W1 entries) turns its rounding noise into an O(lr) difference, so the end-to-end bound
    is 1e-5 relative + 2 % (dense) / 0.5 % (tables) of the distance Adam can travel in 3 steps
    (a ReLU unit sitting at 0 can also flip between fp32 and fp64); the tight checks are the
    single-step gradients above and the optimizer kernels on identical inputs
    (test_gpu_sparse_optim.py)."""
    B = 200
    ops, eng, spec, cfg, p, u, i, y = _setup(variant, dim, B, dev, optimizer=optimizer, dense_impl=impl)
    td = lambda a, dt: torch.from_numpy(a).to(dev).to(dt)
    rng = np.random.default_rng(11)
    P = {k: v.astype(np.float64) for k, v in p.items()}
    M = {k: np.zeros_like(P[k]) for k in list(O.DENSE_ORDER) + ["user_mlp", "item_mlp", "user_mf", "item_mf"]}
    V = {k: np.zeros_like(P[k]) for k in M}
    for t in range(1, 4):
        u = rng.integers(0, 97, B); i = rng.integers(0, 53, B); u[:20] = u[0]
        y = (rng.random(B) < 0.25).astype(np.float32)
        eng.train_step(td(u, torch.int32), td(i, torch.int32), td(y, torch.float32))
        masks = _masks(cfg, spec, t, B)
        loss, c, g, rg, ns = O.neumf_step_grads(spec, P, u, i, y, masks, dt=np.float64)
        for k in O.DENSE_ORDER:
            P[k], M[k], V[k] = O.adam_dense(P[k], M[k], V[k], g[k], cfg.lr, t)
        for k in ("user_mlp", "item_mlp", "user_mf", "item_mf"):
            ids = u if k.startswith("user") else i
            P[k], M[k], V[k] = O.adam_sparse_tf(P[k], M[k], V[k], ids, rg[k], cfg.lr, t, lazy=(optimizer == "adam_lazy"))
        P.update(ns)
    torch.cuda.synchronize()
    travel = 3 * cfg.lr
    for k in O.DENSE_ORDER:
        np.testing.assert_allclose(eng.theta.view(k).cpu().numpy().reshape(P[k].shape), P[k], rtol=1e-5, atol=2e-2 * travel, err_msg=k)
    for k in ("user_mlp", "item_mlp", "user_mf", "item_mf"):
        np.testing.assert_allclose(eng.tables[k].cpu().numpy(), P[k], rtol=1e-5, atol=5e-3 * travel, err_msg=k)
        # and most of the table must agree far more tightly than that bound
        d = np.abs(eng.tables[k].cpu().numpy() - P[k])
        assert np.median(d) <= 1e-7


@pytest.mark.parametrize("variant,dim", [("A", 64), ("B", 64), ("A", 10)])
def test_predict_inference_mode(dev, variant, dim):
    B = 333
    ops, eng, spec, cfg, p, u, i, y = _setup(variant, dim, B, dev)
    td = lambda a, dt: torch.from_numpy(a).to(dev).to(dt)
    out = eng.predict(td(u, torch.int32), td(i, torch.int32)).cpu().numpy()
    c = O.neumf_forward(spec, p, u, i, training=False, dt=np.float64)
    _close(out, c["prob"], "predict")


@pytest.mark.parametrize("K,N,act", [(20, 100, "sigmoid"), (75, 50, "linear"), (128, 128, "relu"), (100, 50, "sigmoid"), (50, 10, "relu"), (3, 1, "linear"),
                                     (256, 100, "sigmoid"), (200, 64, "relu"), (130, 17, "linear")])
def test_dense_layer_shapes(dev, K, N, act):
    """brDenseForward / brDenseBackward alone on ragged K, N, B (padding paths), with input
    BN-affine + dropout and the BN-backward column sums."""
    ops, _ = _mods()
    rng = np.random.default_rng(K * 131 + N)
    B = 150
    x = rng.normal(size=(B, K)).astype(np.float32); W = rng.normal(scale=0.2, size=(K, N)).astype(np.float32)
    b = rng.normal(size=N).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, K).astype(np.float32); sh = rng.normal(size=K).astype(np.float32)
    mean_in = rng.normal(size=K).astype(np.float32); rstd_in = rng.uniform(0.5, 2, K).astype(np.float32)
    seed, step, site, p, row0 = 12345678901234, 7, 2, 0.2, 1000
    td = lambda a: torch.from_numpy(a).to(dev)
    y = torch.empty(B, N, device=dev); stats = torch.zeros(8, 2 * N, dtype=torch.float64, device=dev)   # BR_STAT_REPLICAS
    ops.dense_forward(td(x), td(W), td(b), y, act, td(sc), td(sh), p, seed, step, site, row0, stats)
    mask = O.dropout_mask(seed, step, site, B, K, p, row0)
    tx = (x.astype(np.float64) * sc + sh) * mask / (1 - p)
    yr = O.act_fwd(tx @ W.astype(np.float64) + b, act)
    _close(y.cpu().numpy(), yr, "y")
    st = stats.sum(0).cpu().numpy()
    _close(st[:N], yr.sum(0), "colsum", rtol=1e-6)
    _close(st[N:], (yr ** 2).sum(0), "colsumsq", rtol=1e-6)
    # backward without out-BN: gy is d/dy
    gy = rng.normal(size=(B, N)).astype(np.float32)
    ns = ops.dense_backward_slabs(B, K, N)
    slabs = torch.empty(ns * (K * N + N), device=dev); gx = torch.empty(B, K, device=dev)
    insum = torch.zeros(8, 2 * K, dtype=torch.float64, device=dev)
    wide = K > 128      # split-K layers (first layer of a tower): no BatchNorm on the input side
    ops.dense_backward(td(gy), y, td(x), td(W), act, slabs, ns, gx=gx, in_scale=td(sc), in_shift=td(sh),
                       in_bn=None if wide else (td(mean_in), td(rstd_in)), in_drop_p=p, in_site=site, seed=seed, step=step, row0=row0,
                       in_bn_sums=None if wide else insum)
    out = torch.empty(K * N + N, device=dev)
    ops.reduce_slabs(slabs, ns, K * N + N, out)
    dz = gy.astype(np.float64) * O.act_bwd_from_out(yr, act)
    _close(out.cpu().numpy()[: K * N].reshape(K, N), tx.T @ dz, "dW", rtol=1e-4, atol_frac=1e-5)
    _close(out.cpu().numpy()[K * N:], dz.sum(0), "db", rtol=1e-4, atol_frac=1e-5)
    gxr = (dz @ W.astype(np.float64).T) * mask / (1 - p)
    _close(gx.cpu().numpy(), gxr, "gx", rtol=1e-4, atol_frac=1e-5)
    if wide:
        return
    xhat = (x.astype(np.float64) - mean_in) * rstd_in
    ins = insum.sum(0).cpu().numpy()
    _close(ins[:K], gxr.sum(0), "sum dh", rtol=1e-4, atol_frac=1e-5)
    _close(ins[K:], (gxr * xhat).sum(0), "sum dh*xhat", rtol=1e-4, atol_frac=1e-5)


@pytest.mark.parametrize("K,N,B,act", [(128, 100, 1000, "sigmoid"), (100, 50, 777, "relu"), (64, 112, 640, "linear"), (37, 23, 333, "sigmoid"), (128, 64, 65, "relu"),
                                       (16, 16, 64, "linear"), (112, 128, 300, "sigmoid"), (128, 120, 200, "relu"), (50, 10, 4097, "sigmoid"), (96, 33, 129, "relu"),
                                       (128, 100, 13, "sigmoid"), (256, 100, 500, "sigmoid")])
def test_dense_layer_shapes_padded_rows(dev, K, N, B, act):
    """The same checks with rows padded to a multiple of 4 floats (what the engines allocate): these are the shapes that take the FUSED
    backward kernel and the 16-B forward - on the bf16 pipe where the piece images fit the LDS (odd n-tile counts = a half
    block, N > 112 with K > 96 = the 16-slot dz image, 128 x 120 = no fit, fp32 kernels), with ragged last tiles and batches smaller than a
    tile, input BatchNorm-affine + dropout + the producer's BatchNorm-backward sums."""
    ops, _ = _mods()
    rng = np.random.default_rng(K * 977 + N * 31 + B)
    ldk, ldn = (K + 3) & ~3, (N + 3) & ~3
    x = rng.normal(size=(B, K)).astype(np.float32); W = rng.normal(scale=0.2, size=(K, N)).astype(np.float32)
    b = rng.normal(size=N).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, K).astype(np.float32); sh = rng.normal(size=K).astype(np.float32)
    mean_in = rng.normal(size=K).astype(np.float32); rstd_in = rng.uniform(0.5, 2, K).astype(np.float32)
    seed, step, site, p, row0 = 987654321, 3, 1, 0.2, 64
    td = lambda a: torch.from_numpy(a).to(dev)
    pad = lambda a, ld: (lambda buf: (buf[:, :a.shape[1]].copy_(td(a)), buf[:, :a.shape[1]])[1])(torch.full((a.shape[0], ld), float("nan"), device=dev))
    xv = pad(x, ldk)
    yv = torch.full((B, ldn), float("nan"), device=dev)[:, :N]
    stats = torch.zeros(8, 2 * N, dtype=torch.float64, device=dev)
    ops.dense_forward(xv, td(W), td(b), yv, act, td(sc), td(sh), p, seed, step, site, row0, stats)
    mask = O.dropout_mask(seed, step, site, B, K, p, row0)
    tx = (x.astype(np.float64) * sc + sh) * mask / (1 - p)
    yr = O.act_fwd(tx @ W.astype(np.float64) + b, act)
    _close(yv.cpu().numpy(), yr, "y")
    st = stats.sum(0).cpu().numpy()
    _close(st[:N], yr.sum(0), "colsum", rtol=1e-6)
    _close(st[N:], (yr ** 2).sum(0), "colsumsq", rtol=1e-6)
    gy = rng.normal(size=(B, N)).astype(np.float32)
    gyv = pad(gy, ldn)
    ns = ops.dense_backward_slabs(B, K, N)
    slabs = torch.full((ns * (K * N + N),), float("nan"), device=dev)
    gxv = torch.full((B, ldk), float("nan"), device=dev)[:, :K]
    insum = torch.zeros(8, 2 * K, dtype=torch.float64, device=dev)
    wide = K > 128
    ops.dense_backward(gyv, yv, xv, td(W), act, slabs, ns, gx=gxv, in_scale=td(sc), in_shift=td(sh),
                       in_bn=None if wide else (td(mean_in), td(rstd_in)), in_drop_p=p, in_site=site, seed=seed, step=step, row0=row0,
                       in_bn_sums=None if wide else insum)
    out = torch.empty(K * N + N, device=dev)
    ops.reduce_slabs(slabs, ns, K * N + N, out)
    dz = gy.astype(np.float64) * O.act_bwd_from_out(yr, act)
    _close(out.cpu().numpy()[: K * N].reshape(K, N), tx.T @ dz, "dW", rtol=1e-4, atol_frac=1e-5)
    _close(out.cpu().numpy()[K * N:], dz.sum(0), "db", rtol=1e-4, atol_frac=1e-5)
    gxr = (dz @ W.astype(np.float64).T) * mask / (1 - p)
    _close(gxv.cpu().numpy(), gxr, "gx", rtol=1e-4, atol_frac=1e-5)
    if wide:
        return
    xhat = (x.astype(np.float64) - mean_in) * rstd_in
    ins = insum.sum(0).cpu().numpy()
    _close(ins[:K], gxr.sum(0), "sum dh", rtol=1e-4, atol_frac=1e-5)
    _close(ins[K:], (gxr * xhat).sum(0), "sum dh*xhat", rtol=1e-4, atol_frac=1e-5)


def test_empty_batch_is_noop(dev):
    ops, eng, spec, cfg, p, u, i, y = _setup("A", 8, 16, dev)
    e = torch.empty(0, dtype=torch.int32, device=dev)
    before = eng.theta.buf.clone()
    eng.train_step(e, e, torch.empty(0, device=dev))
    assert torch.equal(before, eng.theta.buf) and eng.t == 0


@pytest.mark.parametrize("optimizer,impl,eager", [("adam_dense", "sweep", ()), ("adam_dense", "sweep", ("SWEEP_USER",)), ("adam_lazy", "sweep", ("BWD1",)),
                                                  ("adam_dense", "deferred", ()), ("adam_dense", "deferred", ("BWD2", "BWD1"))])
def test_graph_replay_matches_eager_steps(dev, optimizer, impl, eager):
    """hipGraph replay of the step (NeuMFEngine.enable_graph) against the eager launch sequence over 4
    steps with fresh batches, a ragged (eager) batch in between, and a state reload.  The only arithmetic
    difference is alpha_t: computed on the device in double and rounded to fp32 (eager: on the host),
    so parameters agree to an fp32 ulp of alpha (1e-6 relative), everything else is the same launches."""
    B = 192
    ops, eg, spec, cfg, p, u, i, y = _setup("A", 64, B, dev, optimizer=optimizer, dense_impl=impl)
    _, gr, *_ = _setup("A", 64, B, dev, optimizer=optimizer, dense_impl=impl)
    gr.enable_graph(B, eager_phases=eager)
    assert gr.t == 0
    rng = np.random.default_rng(11)
    td = lambda a, dt: torch.from_numpy(a).to(dev).to(dt)
    for step in range(5):
        n = 77 if step == 2 else B      # step 2: ragged batch -> eager path on both engines
        uu, ii = rng.integers(0, 97, n), rng.integers(0, 53, n)
        yy = (rng.random(n) < 0.3).astype(np.float32)
        for e in (eg, gr):
            e.train_step(td(uu, torch.int32), td(ii, torch.int32), td(yy, torch.float32))
        if step == 3:   # reload: the device step counter must follow t
            gr.load_state_dict({k: (v.clone() if torch.is_tensor(v) else v) for k, v in gr.state_dict().items()})
    torch.cuda.synchronize()
    gr.check_ids()
    assert gr.t == eg.t == 5
    assert int(gr.step_state[0].item()) == 5
    gr.flush(); eg.flush()
    for k in ("user", "item"):
        _close(gr.fused[k].cpu().numpy(), eg.fused[k].cpu().numpy(), "table " + k, rtol=2e-6, atol_frac=1e-6)
        _close(gr.fused_v[k].cpu().numpy(), eg.fused_v[k].cpu().numpy(), "v " + k, rtol=2e-6, atol_frac=1e-6)
    _close(gr.theta.buf.cpu().numpy(), eg.theta.buf.cpu().numpy(), "theta", rtol=2e-6, atol_frac=1e-6)
    _close(gr.moving_buf.cpu().numpy(), eg.moving_buf.cpu().numpy(), "moving", rtol=2e-6, atol_frac=1e-6)
    _close(gr.msums.sum(0).cpu().numpy(), eg.msums.sum(0).cpu().numpy(), "metric sums", rtol=1e-6)


def test_multi_step_graph_equals_single_step_graph(dev):
    """NeuMFEngine.enable_graph_multi / train_steps: four steps captured as ONE hipGraph against the same four steps replayed one graph
    launch each - the same kernels on the same device-side step state, so every table, moment, dense parameter, moving statistic and
    metric sum must be BIT-equal; a single step and a reload sit between the groups (the device step counter and the prefetched dropout
    planes must carry over in both directions)."""
    B, S = 192, 4
    ops, one, spec, cfg, p, u, i, y = _setup("A", 64, B, dev, optimizer="adam_dense")
    _, grp, *_ = _setup("A", 64, B, dev, optimizer="adam_dense")
    one.enable_graph(B)
    grp.enable_graph_multi(B, steps=S)
    rng = np.random.default_rng(23)
    td = lambda a, dt: torch.from_numpy(a).to(dev).to(dt)

    def group():
        uu, ii = rng.integers(0, 97, (S, B)), rng.integers(0, 53, (S, B))
        yy = (rng.random((S, B)) < 0.3).astype(np.float32)
        for k in range(S):
            one.train_step(td(uu[k], torch.int32), td(ii[k], torch.int32), td(yy[k], torch.float32))
        grp.train_steps(td(uu, torch.int32), td(ii, torch.int32), td(yy, torch.float32))

    group()
    uu, ii, yy = rng.integers(0, 97, B), rng.integers(0, 53, B), (rng.random(B) < 0.3).astype(np.float32)
    for e in (one, grp):
        e.train_step(td(uu, torch.int32), td(ii, torch.int32), td(yy, torch.float32))
    group()
    grp.load_state_dict({k: (v.clone() if torch.is_tensor(v) else v) for k, v in grp.state_dict().items()})
    one.load_state_dict({k: (v.clone() if torch.is_tensor(v) else v) for k, v in one.state_dict().items()})
    group()
    torch.cuda.synchronize()
    grp.check_ids()
    assert grp.t == one.t == 3 * S + 1 and int(grp.step_state[0].item()) == grp.t
    grp.flush(); one.flush()
    for k in ("user", "item"):
        assert torch.equal(grp.fused[k], one.fused[k]), "table " + k
        assert torch.equal(grp.fused_m[k], one.fused_m[k]) and torch.equal(grp.fused_v[k], one.fused_v[k]), "moments " + k
    assert torch.equal(grp.theta.buf, one.theta.buf) and torch.equal(grp.adam_v.buf, one.adam_v.buf)
    assert torch.equal(grp.moving_buf, one.moving_buf)
    assert torch.equal(grp.msums.sum(0), one.msums.sum(0))


def test_graph_replay_against_oracle(dev):
    """Three replayed steps against the oracle (dropout masks keyed by the DEVICE step counter, alpha_t
    computed on the device); same bounds as test_three_optimizer_steps."""
    B = 200
    ops, eng, spec, cfg, p, u, i, y = _setup("A", 64, B, dev, optimizer="adam_dense")
    eng.enable_graph(B)
    td = lambda a, dt: torch.from_numpy(a).to(dev).to(dt)
    rng = np.random.default_rng(11)
    P = {k: v.astype(np.float64) for k, v in p.items()}
    names = ("user_mlp", "item_mlp", "user_mf", "item_mf")
    M = {k: np.zeros_like(P[k]) for k in list(O.DENSE_ORDER) + list(names)}
    V = {k: np.zeros_like(P[k]) for k in M}
    for t in range(1, 4):
        u = rng.integers(0, 97, B); i = rng.integers(0, 53, B); u[:20] = u[0]
        y = (rng.random(B) < 0.25).astype(np.float32)
        eng.train_step(td(u, torch.int32), td(i, torch.int32), td(y, torch.float32))
        loss, c, g, rg, ns = O.neumf_step_grads(spec, P, u, i, y, _masks(cfg, spec, t, B), dt=np.float64)
        for k in O.DENSE_ORDER:
            P[k], M[k], V[k] = O.adam_dense(P[k], M[k], V[k], g[k], cfg.lr, t)
        for k in names:
            P[k], M[k], V[k] = O.adam_sparse_tf(P[k], M[k], V[k], u if k.startswith("user") else i, rg[k], cfg.lr, t, lazy=False)
        P.update(ns)
    torch.cuda.synchronize()
    eng.check_ids()
    travel = 3 * cfg.lr
    for k in O.DENSE_ORDER:
        np.testing.assert_allclose(eng.theta.view(k).cpu().numpy().reshape(P[k].shape), P[k], rtol=1e-5, atol=2e-2 * travel, err_msg=k)
    for k in names:
        np.testing.assert_allclose(eng.tables[k].cpu().numpy(), P[k], rtol=1e-5, atol=5e-3 * travel, err_msg=k)
        assert np.median(np.abs(eng.tables[k].cpu().numpy() - P[k])) <= 1e-7


def _two_engines(dev, B, U, I, dim=16):
    """the same model twice: dense Adam by per-step sweep and by deferred replay, both reading alpha_t from
    the device step state (so the two use the same fp32 alpha)."""
    _, sw, spec, cfg, *_ = _setup("A", dim, B, dev, U=U, I=I, optimizer="adam_dense", dense_impl="sweep")
    # replay="exact": the form of the deferred replay that issues the sweep's own fp32 operations.  The default ("fast", include/binrec.h
    # BR_REPLAY_FAST) agrees with it to ~1e-6 of a row's movement PER CATCH-UP (tests/test_gpu_sparse_optim.py::test_fast_replay_*), which
    # two whole training runs cannot show: Adam's m / (sqrt(v) + eps) with eps = 1e-7 amplifies a one-ulp difference of a gradient by
    # alpha (1 - b1) / eps = 5000 per step, so two runs either agree bit for bit or drift apart within a dozen steps (tools/diag/replay_debug.py)
    _, de, *_ = _setup("A", dim, B, dev, U=U, I=I, optimizer="adam_dense", dense_impl="deferred", replay="exact")
    sw._alloc_step_state(sw.step_struct)
    return sw, de


def _assert_same_state(sw, de):
    de.flush()
    for k in ("user", "item"):
        assert torch.equal(sw.fused[k], de.fused[k]), "table " + k
        assert torch.equal(sw.fused_m[k], de.fused_m[k]), "m " + k
        assert torch.equal(sw.fused_v[k], de.fused_v[k]), "v " + k
    assert torch.equal(sw.theta.buf, de.theta.buf)


def test_deferred_adam_is_bit_equal_to_the_sweep(dev):
    """O1, Keras non-lazy sparse apply [TF-sem]: the deferred replay must give the SAME tables as sweeping
    every row every step — bit for bit (same fp32 operations, adam_math.h).  1500 users, 48 pairs per step:
    most rows sit out many steps between two touches; an inference call in the middle flushes."""
    B, U, I = 48, 1500, 400
    sw, de = _two_engines(dev, B, U, I)
    rng = np.random.default_rng(5)
    td = lambda a, dt: torch.from_numpy(a).to(dev).to(dt)
    for step in range(40):
        n = B if step % 7 else B - 11
        uu, ii = rng.integers(0, U, n), rng.integers(0, I, n)
        uu[:5] = uu[0]
        yy = (rng.random(n) < 0.3).astype(np.float32)
        for e in (sw, de):
            e.train_step(td(uu, torch.int32), td(ii, torch.int32), td(yy, torch.float32))
        if step == 17:
            q = td(rng.integers(0, U, 64), torch.int32), td(rng.integers(0, I, 64), torch.int32)
            assert torch.equal(sw.predict(*q), de.predict(*q))
    torch.cuda.synchronize()
    de.check_ids()
    assert de._stale
    _assert_same_state(sw, de)
    # a reload (all rows at step t) and more steps
    de.load_state_dict({k: (v.clone() if torch.is_tensor(v) else v) for k, v in de.state_dict().items()})
    for step in range(5):
        uu, ii = rng.integers(0, U, B), rng.integers(0, I, B)
        yy = (rng.random(B) < 0.3).astype(np.float32)
        for e in (sw, de):
            e.train_step(td(uu, torch.int32), td(ii, torch.int32), td(yy, torch.float32))
    _assert_same_state(sw, de)


def test_deferred_adam_survives_the_alpha_ring(dev):
    """more steps than BR_ALPHA_RING: the engine flushes before a replay could read an overwritten alpha."""
    B, U, I = 16, 300, 200
    sw, de = _two_engines(dev, B, U, I, dim=8)
    rng = np.random.default_rng(9)
    steps = de.ALPHA_RING + 40
    us = torch.from_numpy(rng.integers(0, U, (steps, B))).to(dev).to(torch.int32)
    its = torch.from_numpy(rng.integers(0, I, (steps, B))).to(dev).to(torch.int32)
    ys = torch.from_numpy((rng.random((steps, B)) < 0.3).astype(np.float32)).to(dev)
    us[:, 0] = 7      # row 7 every step; rows >= 290 of the user table never
    us.clamp_(max=289)
    for s in range(steps):
        for e in (sw, de):
            e.train_step(us[s], its[s], ys[s])
    torch.cuda.synchronize()
    assert de._flush_t > 0
    _assert_same_state(sw, de)


def test_graph_resident_probe_times_a_kernel_inside_the_replay(dev):
    """brProbeGraph*: event-record nodes placed around one tagged launch while the step is captured, pointed at a fresh event pair
    before every replay.  The durations read back are positive, of the order of the eager probe's figure for the same launch, and
    the replays that carry the nodes produce the same tables as an engine without them (to the order of the BatchNorm atomics)."""
    import ctypes
    from importlib import import_module
    _lib = import_module("binary-recommendation_amd._lib")
    lib = _lib.load()
    TAG = {k[7:]: v for k, v in _lib.parse_enums().items() if k.startswith("BR_TAG_")}
    B = 4096
    ops, plain, spec, cfg, p, u, i, y = _setup("A", 64, B, dev, optimizer="adam_dense")
    _, probed, *_ = _setup("A", 64, B, dev, optimizer="adam_dense")
    plain.enable_graph(B)
    assert lib.brProbeGraphSelect(TAG["ADAM_ROWS_USER"]) == 0
    probed.enable_graph(B, keep_graph=True)
    assert lib.brProbeGraphSelect(-1) == 0
    assert lib.brProbeGraphNodes() == 2                       # one in front of the Adam-rows kernel (behind the partials), one behind
    ex = ctypes.c_void_p(int(probed._graph["graphs"][0].raw_cuda_graph_exec()))
    steps = 6
    assert lib.brProbeGraphEnable(steps) == 0
    rng = np.random.default_rng(5)
    td = lambda a, dt: torch.from_numpy(a).to(dev).to(dt)
    for k in range(steps):
        uu, ii = rng.integers(0, 97, B), rng.integers(0, 53, B)
        yy = (rng.random(B) < 0.3).astype(np.float32)
        assert lib.brProbeGraphArm(ex, k) == 0, lib.brGetLastError()
        for e in (plain, probed):
            e.train_step(td(uu, torch.int32), td(ii, torch.int32), td(yy, torch.float32))
    torch.cuda.synchronize()
    ms = ctypes.c_float()
    got = []
    for k in range(steps):
        assert lib.brProbeGraphRead(k, ctypes.byref(ms)) == 0, lib.brGetLastError()
        got.append(ms.value * 1e3)
    assert all(1.0 < g < 2000.0 for g in got), got            # microseconds of one small launch, every replay its own pair
    assert lib.brProbeGraphArm(ex, steps) == -1               # slot out of range
    lib.brProbeGraphEnable(0)
    plain.flush(); probed.flush()
    for k in ("user", "item"):      # (same launches; the BatchNorm column sums are double atomics, so not bit for bit)
        _close(probed.fused[k].cpu().numpy(), plain.fused[k].cpu().numpy(), "table " + k, rtol=2e-6, atol_frac=1e-6)
    _close(probed.theta.buf.cpu().numpy(), plain.theta.buf.cpu().numpy(), "theta", rtol=2e-6, atol_frac=1e-6)


@pytest.mark.parametrize("graph", [False, True])
def test_fused_lookup_sort_step_is_bit_equal_to_the_sweep(dev, graph):
    """Batches above 16 384 take the fork-free step: the chunk sorts ride in the lookup's launch (lookup_sort_kernel) and, once the
    dropout planes are prefetched, the chunk-rank launch advances the step state behind the lookup (which computes its step as
    ss->step + 1).  Against the sweep engine (plain lookup, dedup sorts on the aux stream, a step-state launch of its own): tables and
    moments bit for bit over steps with lags, a ragged batch (eager path in graph mode) and duplicate-heavy ids."""
    B, U, I = 20000, 30000, 7000
    sw, de = _two_engines(dev, B, U, I, dim=64)
    if graph:
        de.enable_graph(B)
    rng = np.random.default_rng(21)
    td = lambda a, dt: torch.from_numpy(a).to(dev).to(dt)
    for step in range(7):
        n = B if step != 3 else 17000          # 17 000 > 16 384: the fused launch on the eager path as well
        uu, ii = rng.integers(0, U, n), rng.integers(0, I, n)
        uu[:300] = uu[0]
        if step % 2:
            uu[1000:9000] = rng.integers(0, 500, 8000)      # rows touched again after a short lag
        yy = (rng.random(n) < 0.3).astype(np.float32)
        for e in (sw, de):
            e.train_step(td(uu, torch.int32), td(ii, torch.int32), td(yy, torch.float32))
    torch.cuda.synchronize()
    de.check_ids()
    assert de.t == sw.t == 7 and int(de.step_state[0].item()) == 7
    if graph:      # alpha_t is computed on the device in double by the replayed step, on the host by the eager one: an fp32 ulp of alpha
        de.flush(); sw.flush()
        for k in ("user", "item"):
            _close(de.fused[k].cpu().numpy(), sw.fused[k].cpu().numpy(), "table " + k, rtol=2e-6, atol_frac=1e-6)
            _close(de.fused_v[k].cpu().numpy(), sw.fused_v[k].cpu().numpy(), "v " + k, rtol=2e-6, atol_frac=1e-6)
    else:
        _assert_same_state(sw, de)
