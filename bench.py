#!/usr/bin/env python3
"""Headline benchmark: (user,item) pairs/s of one NeuMF training step at batch 65 536, dim 64
(BASELINE.json configs[1]: NeuMF embed_dim=64, synthetic 1M users x 100K items, 1 x MI355X).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = embedding gather -> GMF dot + MLP tower (fp32 MFMA) -> sigmoid-BCE -> backward ->
dedup of the per-pair row gradients -> Keras-Adam on the four tables and the dense parameters,
on one synthetic batch already resident in HBM.  Default optimizer = "adam_dense": Keras'
NON-lazy sparse Adam (every row of every table decays m, v and moves each step [TF-sem]) — the
reference's semantics; the lazy-row variant is timed too and reported under "adam_lazy".

Rank 0 prints ONE JSON line (metric/value/... + "roofline" for the kernel that takes the most time
per step - chosen from the per-kernel HIP-event table of an eager pass, then bracketed by events on the
launch stream INSIDE the timed region - + "kernels": one entry per kernel, + "legs": the plain 4-table
gather (north_star's HBM-read target), Zipf(1.05) ids, the BPR step (config 3) and the TwoTower
in-batch-softmax step (config 4 on one GPU), + "cpu_baseline": the torch-CPU fp32 port of
trainers/NFC_plain.py's step from oracle/torch_ref.py on a bounded sample).
"""
from __future__ import annotations

import argparse
import ctypes
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32 dense peak (same guide)
MLP_MATH = "f32" if os.environ.get("BR_MLP_MATH", "").startswith("f") else "bf16x6"      # csrc/dense.h mlp_bf16x6()


class PyProbe:
    """Row-sharded runs launch their table kernels from Python (parallel.py), outside the C step
    driver's probe: bracket those launches with torch.cuda.Event on the current (= launch) stream."""

    def __init__(self, tags):
        self.tags, self.pairs, self._open = tags, {}, None

    def before(self, name, args):
        if name in self.tags:
            s = torch.cuda.Event(enable_timing=True)
            s.record()
            self._open = (self.tags[name](args), s)

    def after(self, name, args):
        if self._open is not None:
            tag, s = self._open
            self._open = None
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.pairs.setdefault(tag, []).append((s, e))

    def merge_into(self, per_tag):
        for tag, v in self.pairs.items():
            per_tag[tag] = (sum(s.elapsed_time(e) for s, e in v) / len(v) * 1e3, len(v))


def read_probe(lib):
    """{tag: (mean_us, launches)} from the C-level HIP-event probe (include/binrec.h, brProbe*)."""
    import ctypes
    acc = {}
    tag, ms = ctypes.c_int(0), ctypes.c_float(0.0)
    for i in range(lib.brProbeCount()):
        if lib.brProbeRead(i, ctypes.byref(tag), ctypes.byref(ms)) != 0:
            raise RuntimeError(lib.brGetLastError().decode())
        t = acc.setdefault(tag.value, [0.0, 0])
        t[0] += ms.value * 1e3
        t[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def seed_steady_state(eng, uniq_u, uniq_i, since_flush, t0):
    """--steady-state-lags: put the deferred tables into the state of a LONG run without running it.  In a long run every row carries
    moments and sits some steps behind the step counter: a row comes up in a step with probability p = unique rows per step / rows,
    so its lag is geometric with mean 1 / p - 1 (191 steps for the 12.5 M-row user shard of config 5 at 65 536 pairs per step) - capped
    by the engine's periodic flush, `since_flush` steps ago here.  A short cycle of batches cannot produce that state (a 25-batch cycle
    caps every lag at 25 and leaves most rows at m = v = 0: profiles/r02/bench_config5_shard_r02.json).  Seeds: m, v of every row from a
    gradient scale of 1e-6 (|g| of a batch-mean loss at this batch size), last[row] = t0 - 1 - lag, the step state (alpha ring
    included) at step t0.  -> what was seeded, for the bench line."""
    _lib = importlib.import_module("binary-recommendation_amd._lib")
    ops = importlib.import_module("binary-recommendation_amd.ops")
    lib, cfg, info = _lib.load(), eng.cfg, {}
    assert eng.deferred and t0 > since_flush >= 0 and t0 >= eng.ALPHA_RING
    gen = torch.Generator(device=eng.device).manual_seed(777)
    for k, uniq in (("user", uniq_u), ("item", uniq_i)):
        rows = eng.fused[k].shape[0]
        gsc = torch.empty_like(eng.fused[k]).normal_(0.0, 1e-6, generator=gen)
        eng.fused_m[k].copy_(gsc).mul_(1.0 - cfg.beta1)
        eng.fused_v[k].copy_(gsc).square_().mul_(1.0 - cfg.beta2)
        del gsc
        p = min(1.0, uniq / rows)
        lag = torch.empty(rows, device=eng.device, dtype=torch.float32).geometric_(p, generator=gen).sub_(1.0).clamp_(max=float(since_flush))
        eng.last[k].copy_((t0 - 1 - lag).to(torch.int32))
        info[k] = {"rows": rows, "p_touch_per_step": p, "mean_lag": float(lag.mean().item()), "frac_capped_by_flush": float((lag >= since_flush).float().mean().item())}
        del lag
    # the alpha ring of the last BR_ALPHA_RING steps: set the state RING steps back and advance it to t0 (1024 one-block launches)
    _lib.check(lib.brStepStateSet(eng.step_state.data_ptr(), t0 - eng.ALPHA_RING, cfg.lr, cfg.beta1, cfg.beta2, ops._stream()), "brStepStateSet")
    for _ in range(eng.ALPHA_RING):
        _lib.check(lib.brStepStateAdvance(eng.step_state.data_ptr(), cfg.lr, cfg.beta1, cfg.beta2, None, 0, ops._stream()), "brStepStateAdvance")
    torch.cuda.synchronize()
    eng.t, eng._flush_t, eng._stale = t0, t0 - since_flush, True
    info.update({"t0": t0, "steps_since_flush": since_flush, "replay": cfg.replay,
                 "note": "lags geometric with mean rows / unique rows per step - 1, capped by the last flush; every row carries moments"})
    return info


def make_batches(n, B, U, I, dev, seed, zipf, idt=torch.int32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = []
    for _ in range(n):
        if zipf:
            # Zipf(alpha=1.05)-like ranks by inverse-CDF on a truncated power law
            users, items = zipf_ids(B, U, g), zipf_ids(B, I, g)
        else:
            users = torch.randint(0, U, (B,), generator=g)
            items = torch.randint(0, I, (B,), generator=g)
        labels = (torch.rand(B, generator=g) < 0.25).float()   # 1 pos : 3 neg (NeuMFModel.py:102)
        out.append((users.to(idt).to(dev), items.to(idt).to(dev), labels.to(dev)))
    return out


def zipf_ids(n, N, g, a=1.05):
    u = torch.rand(n, generator=g, dtype=torch.float64)
    r = ((N ** (1 - a) - 1) * u + 1) ** (1 / (1 - a))
    return r.long().clamp_(1, N) - 1


def time_us(fn, reps=20, rounds=5):
    """mean microseconds of one fn() (a short launch sequence on torch's current stream): `reps` calls captured into a hipGraph
    (no host launch gaps), replayed `rounds` times between two events; eager loop when the capture is refused."""
    fn()
    torch.cuda.synchronize()
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        replay, mode = g.replay, "hipGraph replay"
    except Exception:  # noqa: BLE001
        replay, mode = (lambda: [fn() for _ in range(reps)]), "eager loop"
    replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(rounds):
        replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (reps * rounds), mode


def loop_ms(step, n=20, warm=5, windows=3):
    """ms per call of an eager step: the best of `windows` back-to-back windows of n calls (a shared box shows one-off hiccups of tens of
    milliseconds - allocator / clock / neighbours - that a single mean would fold into every step)"""
    for _ in range(warm):
        step()
    best = None
    for _ in range(windows):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e3
        best = dt if best is None else min(best, dt)
    return best


def gather_leg(ops, dev, U, I, D, B, seed):
    """north_star's target: the embedding gather alone at batch 65 536 / dim 64 - (a) brGatherRows over the four separate
    (rows x D) tables of the reference graph (NFC_plain.py:118-126), (b) the engine's form: two fused [mlp | mf] tables, lookup +
    GMF dot + MLP concat in one launch (brNeumfEmbedForward).  frac_read = gathered bytes / time / 8 TB/s (reads only, the
    north_star figure); the kernels also write what they read (frac_all counts both directions + ids)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    tabs = [torch.rand(U, D, device=dev), torch.rand(I, D, device=dev), torch.rand(U, D, device=dev), torch.rand(I, D, device=dev)]
    fu, fi = torch.rand(U, 2 * D, device=dev), torch.rand(I, 2 * D, device=dev)
    outs = [torch.empty(B, D, device=dev) for _ in range(4)]
    x0, dot = torch.empty(B, 2 * D, device=dev), torch.empty(B, device=dev)
    rd = B * 4 * D * 4
    for name, draw in (("uniform", lambda N: torch.randint(0, N, (B,), generator=g)), ("zipf", lambda N: zipf_ids(B, N, g))):
        users, items = draw(U).int().to(dev), draw(I).int().to(dev)
        t4, mode = time_us(lambda: ops.gather_rows(tabs, [users, items, users, items], outs))
        tf, _ = time_us(lambda: ops.neumf_embed_forward(fu[:, :D], fi[:, :D], fu[:, D:], fi[:, D:], users, items, 1, x0, dot))
        out[name] = {"gather_rows_4_tables": {"kernel": "gather_rows_kernel", "us": t4, "read_GBps": rd / t4 * 1e-3, "frac_read": rd / t4 * 1e-3 / HBM_PEAK_GBPS,
                                              "frac_all": (2 * rd + 16 * B) / t4 * 1e-3 / HBM_PEAK_GBPS},
                     "neumf_embed_fwd_fused_tables": {"kernel": "neumf_embed_fwd_kernel", "us": tf, "read_GBps": rd / tf * 1e-3, "frac_read": rd / tf * 1e-3 / HBM_PEAK_GBPS,
                                                      "frac_all": (rd + B * 2 * D * 4 + 12 * B) / tf * 1e-3 / HBM_PEAK_GBPS}}
    out.update({"batch": B, "dim": D, "tables": f"{U} x {D} (x2), {I} x {D} (x2)", "gathered_bytes": rd, "peak_GBps": HBM_PEAK_GBPS, "timing": mode,
                "target": "north_star: >= 0.40 of the HBM-read roofline on the gather at batch 65 536 / dim 64"})
    return out


def bpr_leg(dev, U, I, F, B, seed):
    """BASELINE configs[2]: the BPR triplet step (src/models/BPRModel.py:38-74) at the same synthetic scale, Keras-Adam."""
    bpr = importlib.import_module("binary-recommendation_amd.bpr")
    g = torch.Generator().manual_seed(seed)
    out = {"workload": f"BPR step, {U} users x {I} items, {F} factors, {B} triplets, uniform ids, 16 batches cycled (tables in steady state)"}
    for name, opt, impl in (("adam_dense", "adam_dense", "deferred"), ("adam_dense_sweep", "adam_dense", "sweep"), ("adam_lazy", "adam_lazy", "sweep")):
        e = bpr.BPREngine(U, I, F, dev, B, optimizer=opt, dense_impl=impl)
        # a cycle of 16 distinct batches, applied once before the clock starts: rows then carry moments and lags as in a running job
        # (one batch repeated leaves every lag at 0 and the deferred replay with nothing to do)
        nb = 16
        trip = [tuple(torch.randint(0, N, (B,), generator=g).int().to(dev) for N in (U, I, I)) for _ in range(nb)]
        k = [0]

        def one():
            e.train_step(*trip[k[0] % nb]); k[0] += 1
        for _ in range(nb):
            one()
        ms = loop_ms(one)
        e.check_ids()
        ms_graph = None
        if impl == "deferred" and opt == "adam_dense":
            # the same step replayed as one hipGraph (BPREngine.enable_graph): the eager step is a dozen launches from the Python host
            try:
                e.enable_graph(B)
                ms_graph = loop_ms(one)
                e.check_ids()
            except Exception as exc:  # noqa: BLE001
                log(f"BPR graph capture failed: {type(exc).__name__}: {exc}")
        u, p, n = trip[0]
        uu = int(torch.unique(u).numel()) + int(torch.unique(torch.cat([p, n])).numel())
        # 3 rows in, 3 row gradients out and in again, touched rows of table + m + v read and written; the sweep: every row of both tables
        alg = B * 3 * F * 4 * 3 + (6 * 4 * F * (U + I) if impl == "sweep" and opt == "adam_dense" else uu * 6 * 4 * F)
        out[name] = {"ms_per_step": ms, "triplets_per_s": B / ms * 1e3, "algorithmic_bytes_per_step": alg, "frac_of_hbm_peak": alg / ms * 1e-6 / HBM_PEAK_GBPS}
        if ms_graph is not None:
            out[name]["hipgraph_replay"] = {"ms_per_step": ms_graph, "triplets_per_s": B / ms_graph * 1e3}
        del e
        torch.cuda.empty_cache()
    out["adam_dense"]["note"] = "Keras' non-lazy Adam by per-row deferred replay (bit-equal to adam_dense_sweep, tests/test_gpu_twotower_bpr.py)"
    out["adam_lazy"]["note"] = "touched rows only: NOT the reference's (Keras non-lazy) semantics"
    return out


def twotower_leg(ops, dev, U, I, E, S, B, seed):
    """BASELINE configs[3] on one GPU: trainers/twoTower.py's step (2 lookups, two E x S towers, in-batch softmax over the batch's B
    candidates, Adagrad) + the in-batch softmax kernels alone against the fp32 MFMA peak (3 GEMMs of B x B x S: S = Q C^T, dQ = P C,
    dC = P^T Q; the recomputation of S in the gradient sweep is not counted)."""
    tt = importlib.import_module("binary-recommendation_amd.two_tower")
    g = torch.Generator().manual_seed(seed)
    e = tt.TwoTowerEngine(E, I, U, S, dev, B)
    u, it = (torch.randint(0, U, (B,), generator=g) + 2).int().to(dev), (torch.randint(0, I, (B,), generator=g) + 2).int().to(dev)
    ms = loop_ms(lambda: e.train_step(u, it))
    e.check_ids()
    ms_graph = None
    try:      # the same step replayed as one hipGraph (TwoTowerEngine.enable_graph)
        e.enable_graph(B)
        ms_graph = loop_ms(lambda: e.train_step(u, it))
        e.check_ids()
    except Exception as exc:  # noqa: BLE001
        log(f"TwoTower graph capture failed: {type(exc).__name__}: {exc}")
    del e
    q, c = torch.randn(B, S, device=dev) * 0.3, torch.randn(B, S, device=dev) * 0.3
    lse, slots = torch.empty(B, device=dev), torch.zeros(64, dtype=torch.float64, device=dev)
    dq, dc = torch.empty_like(q), torch.empty_like(c)
    t_lse, mode = time_us(lambda: ops.inbatch_softmax_lse(q, c, it, it, 0, lse, slots), reps=10)
    t_grad, _ = time_us(lambda: ops.inbatch_softmax_grad(q, c, it, it, 0, lse, dq, dc), reps=10)
    t_fused, _ = time_us(lambda: ops.inbatch_softmax_lse_grad_q(q, c, it, it, 0, lse, slots, dq), reps=10)       # what the step runs: lse + dQ in one sweep,
    t_dc, _ = time_us(lambda: ops.inbatch_softmax_grad(q, c, it, it, 0, lse, None, dc), reps=10)                 # then dC
    flop = 2.0 * B * B * S
    # the per-rank shape of BASELINE configs[3] on 8 GPUs: 1 024 local queries against the 65 536 all-gathered candidates (lse + loss + dQ in
    # one sweep), then the rank's 1 024 candidates against the 65 536 all-gathered queries (dC): 3 GEMMs of 1 024 x 65 536 x S
    Bq, Bc = 1024, 65536
    gq = torch.Generator().manual_seed(seed + 1)
    qs, ca = torch.randn(Bq, S, device=dev) * 0.3, torch.randn(Bc, S, device=dev) * 0.3
    qa, cs = torch.randn(Bc, S, device=dev) * 0.3, torch.randn(Bq, S, device=dev) * 0.3
    ida = (torch.randint(0, I, (Bc,), generator=gq) + 2).int().to(dev)
    ids_ = ida[:Bq].contiguous()
    lse_s, lse_a, dqs, dcs = torch.empty(Bq, device=dev), torch.zeros(Bc, device=dev), torch.empty(Bq, S, device=dev), torch.empty(Bq, S, device=dev)
    t_s1, _ = time_us(lambda: ops.inbatch_softmax_lse_grad_q(qs, ca, ids_, ida, 0, lse_s, slots, dqs), reps=10)
    lse_a.fill_(10.0)                                   # any finite lse: the dC sweep's time does not depend on its values
    t_s2, _ = time_us(lambda: ops.inbatch_softmax_grad(qa, cs, ida, ids_, 0, lse_a, None, dcs), reps=10)
    sflop = 2.0 * Bq * Bc * S
    stripe = {"shape": f"{Bq} queries x {Bc} gathered candidates, semb {S}", "lse_dq_one_sweep_us": t_s1, "dc_us": t_s2, "algorithmic_flop": 3 * sflop,
              "achieved_TFLOPs": 3 * sflop / (t_s1 + t_s2) * 1e-6, "frac": 3 * sflop / (t_s1 + t_s2) * 1e-6 / MFMA_F32_PEAK_TFLOPS, "bound": "mfma"}
    return {"workload": f"TwoTower step, {U} users x {I} items, embed {E}, semb {S}, batch {B} (in-batch negatives: {B} candidates), Adagrad, uniform ids",
            "ms_per_step": ms, "pairs_per_s": B / ms * 1e3, "launch_mode": "eager launches from the Python host",
            "hipgraph_replay": None if ms_graph is None else {"ms_per_step": ms_graph, "pairs_per_s": B / ms_graph * 1e3},
            "config4_per_rank_stripe": stripe,
            "inbatch_softmax": {"lse_dq_one_sweep_us": t_fused, "dc_us": t_dc, "lse_us": t_lse, "grad_us": t_grad, "bound": "mfma", "algorithmic_flop": 3 * flop,
                                "achieved_TFLOPs": 3 * flop / (t_fused + t_dc) * 1e-6, "frac": 3 * flop / (t_fused + t_dc) * 1e-6 / MFMA_F32_PEAK_TFLOPS,
                                "frac_separate_passes": 3 * flop / (t_lse + t_grad) * 1e-6 / MFMA_F32_PEAK_TFLOPS,
                                "lse_frac": flop / t_lse * 1e-6 / MFMA_F32_PEAK_TFLOPS, "grad_frac": 2 * flop / t_grad * 1e-6 / MFMA_F32_PEAK_TFLOPS,
                                "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "timing": mode}}


_CURSOR = {}


_GROUPS = {}


def make_groups(batches, S):
    """the cycle's batches back to back in groups of S (what NeuMFEngine.train_steps stages with one launch); a cycle that S does not
    divide keeps its last batches for single steps"""
    g = [tuple(torch.cat([batches[c + k][j] for k in range(S)]) for j in range(3)) for c in range(0, len(batches) - len(batches) % S, S)]
    _GROUPS[id(batches)] = (S, g)
    return g


def run_steps(eng, batches, n, row0, batch_total, arm=None, arm_group=None):
    """n steps over the batch cycle, CONTINUING where the last call on this cycle stopped: every leg of a run then sees batches in cycle
    order (restarting at batch 0 per call replayed the same ~20 batches in every leg - their rows came back with lags of a few steps
    while the seeded long-run state of --steady-state-lags sat untouched in the rest of the table).
    With a multi-step graph on the engine (enable_graph_multi) and the cycle grouped (make_groups), every run of S steps that starts on a
    group boundary is ONE train_steps call; the steps in front of the first boundary and behind the last are single train_step calls."""
    nb = len(batches)
    c0 = _CURSOR.get(id(batches), 0)
    _CURSOR[id(batches)] = c0 + n
    gm = getattr(eng, "_graph_multi", None)
    S, groups = _GROUPS.get(id(batches), (0, None))
    use_groups = gm is not None and groups and gm["S"] == S and row0 == 0 and batch_total == batches[0][0].shape[0]
    s = 0
    while s < n:
        c = (c0 + s) % nb
        if use_groups and c % S == 0 and c // S < len(groups) and n - s >= S:
            if arm_group is not None:
                arm_group(s)    # chooses the probed or the plain capture for this group and arms the probed one's record nodes
            eng.train_steps(*groups[c // S])
            s += S
            continue
        u, i, y = batches[c]
        if arm is not None:
            arm(s)              # points the graph's event-record nodes at this replay's event pair (brProbeGraphArm)
        eng.train_step(u, i, y, row0=row0, batch_total=batch_total)
        s += 1


def timed(eng, batches, steps, warmup, ctx, row0, batch_total, arm=None, arm_group=None):
    run_steps(eng, batches, warmup, row0, batch_total)
    S, groups = _GROUPS.get(id(batches), (0, None))
    if groups and getattr(eng, "_graph_multi", None) is not None:
        # start the timed region on a group boundary of the batch cycle (a few more untimed steps): K timed steps are then K // S graph launches
        run_steps(eng, batches, (-_CURSOR.get(id(batches), 0)) % S, row0, batch_total)
    if ctx is not None:
        ctx.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(eng, batches, steps, row0, batch_total, arm, arm_group)
    torch.cuda.synchronize()
    if ctx is not None:
        ctx.barrier()
    dt = time.perf_counter() - t0
    if ctx is not None and not ctx.local:
        t = torch.tensor([dt], dtype=torch.float64, device=eng.device if ctx.backend == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def cpu_baseline(args, spec_dim):
    """torch-CPU fp32 port of one trainers/NFC_plain.py step (oracle/torch_ref.py), all host threads."""
    from oracle import binrec_oracle as O
    from oracle import torch_ref as T
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 16))   # the 1-GPU box's CPU share is 16 cores; more threads only oversubscribe
    torch.set_num_threads(threads)
    spec = O.NeuMFSpec("A", dim=spec_dim)
    U, I, B = args.users, args.items, args.batch
    step = T.NFCPlainCpuStep(spec, U, I, lr=0.005, seed=1, lazy_adam=False)
    g = torch.Generator().manual_seed(1234)
    n = args.cpu_steps
    data = [(torch.randint(0, U, (B,), generator=g), torch.randint(0, I, (B,), generator=g),
             (torch.rand(B, generator=g) < 0.25).float()) for _ in range(n + 1)]
    step.step(*data[0])  # warm-up
    t0 = time.perf_counter()
    for k in range(1, n + 1):
        step.step(*data[k])
    dt = time.perf_counter() - t0
    return {"value": B * n / dt, "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} steps of batch {B} (NeuMF-A dim {spec_dim}, {U}x{I} tables, dense Keras-Adam) after 1 warm-up, "
                      f"{dt:.1f} s; oracle/torch_ref.py::NFCPlainCpuStep"}


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--users", type=int, default=1_000_000)
    ap.add_argument("--items", type=int, default=100_000)
    ap.add_argument("--variant", default="A")
    ap.add_argument("--optimizer", default="adam_dense", choices=["adam_dense", "adam_lazy"])
    ap.add_argument("--dense-impl", default=None, choices=["deferred", "sweep"],
                    help="adam_dense: per-row deferred replay (default) or one table sweep per step")
    ap.add_argument("--sync-bn", action="store_true", help="N > 1: BatchNorm statistics over the global batch (4 extra all-reduces per step)")
    ap.add_argument("--zipf", action="store_true", help="Zipf(1.05) ids instead of uniform")
    ap.add_argument("--cpu-steps", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lazy", action="store_true", help="skip the extra adam_lazy timing")
    ap.add_argument("--no-graph", action="store_true", help="single GPU: time the eager launch sequence instead of hipGraph replay")
    ap.add_argument("--steps-per-launch", type=int, default=4,
                    help="single GPU, graph mode: training steps captured per hipGraph launch (NeuMFEngine.enable_graph_multi; 1 = one graph launch per step, "
                         "what rounds 1-2 timed)")
    ap.add_argument("--no-legs", action="store_true", help="skip the gather / Zipf / BPR / TwoTower legs")
    ap.add_argument("--cycle", type=int, default=256, help="distinct synthetic batches cycled (the tables are primed with one pass over them)")
    ap.add_argument("--profile-steps", type=int, default=10, help="graph mode: eager steps probed per kernel before the timed region")
    ap.add_argument("--int64", action="store_true", help="int64 ids (config 5: 100 M users)")
    ap.add_argument("--steady-state-lags", action="store_true",
                    help="single GPU, deferred tables: seed every row with moments and a lag from the long-run distribution instead of priming with the "
                         "batch cycle (seed_steady_state) - the state a table far larger than steps x batch is in after a long run")
    ap.add_argument("--since-flush", type=int, default=700, help="--steady-state-lags: steps since the engine's last periodic flush (caps the lags)")
    ap.add_argument("--replay", default=None, choices=["fast", "exact"], help="form of the deferred replay (NeuMFConfig.replay; default fast)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    backend = os.environ.get("BR_BENCH_BACKEND", "nccl")     # "gloo": rehearsal of the N > 1 path on a 1-GPU box (ranks share the card)
    dev = torch.device("cuda", local_rank if backend == "nccl" else local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)

    neumf = importlib.import_module("binary-recommendation_amd.neumf")
    par = importlib.import_module("binary-recommendation_amd.parallel")
    _lib = importlib.import_module("binary-recommendation_amd._lib")
    _lib.load()

    ctx = None
    # BR_BENCH_FORCE_SHARDED=1 (one-GPU box): the whole N > 1 path - RCCL process group, row-sharded engine, every collective really
    # issued - on a 1-rank group; a rehearsal of the multi-GPU code, its figure is not a bench line
    force_sharded = world == 1 and os.environ.get("BR_BENCH_FORCE_SHARDED") == "1"
    json_out = sys.stdout
    if world > 1 or force_sharded:
        # RCCL prints a version banner on file descriptor 1 when the communicator comes up: everything native goes to stderr, the ONE
        # JSON line of the contract to the descriptor stdout had
        sys.stdout.flush()
        json_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if force_sharded:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)
        ctx = par.DistCtx(force_collectives=force_sharded)

    B, D = args.batch, args.dim
    # weak scaling: per-GPU batch AND per-GPU table shard are fixed as N grows
    U, I = args.users * world, args.items * world
    batch_total = B * world
    row0 = rank * B

    impl = args.dense_impl or "deferred"
    idt = torch.int64 if args.int64 else torch.int32

    def build(optimizer, dense_impl=impl):
        # N > 1: per-replica BatchNorm = what the reference's MirroredStrategy does with a plain BatchNormalization
        # [TF-sem] (and no BatchNorm collective in the step); --sync-bn makes the statistics global
        cfg = neumf.NeuMFConfig(variant=args.variant, dim=D, optimizer=optimizer, seed=20261004, dense_impl=dense_impl, sync_bn=args.sync_bn, replay=args.replay)
        if ctx is None:
            return neumf.NeuMFEngine(cfg, U, I, dev, B, init_seed=1, id_dtype=idt)
        # fixed-capacity exchange (no host sync in the step; duplicate ids are merged before they travel, so Zipf heads fit too)
        return par.make_sharded_engine(neumf.NeuMFEngine)(cfg, U, I, dev, B, ctx, init_seed=1, id_dtype=idt, exchange="padded")

    log(f"building engine: {U} users x {I} items, dim {D}, batch {B}/GPU, world {world}, {args.optimizer}")
    eng = build(args.optimizer)
    log("engine built")
    # distinct batches cycled.  Long enough that (almost) every row of the tables comes up within a pass: the deferred kernels' replay work per
    # step is the number of rows that carry moments - a 32-batch cycle reaches 88 % of 1 M users and flatters the step by ~4 %
    n_batches = max(1, args.cycle)
    batches = make_batches(n_batches, B, U, I, dev, 1234 + rank, args.zipf, idt)

    # ---- per-kernel table, then the timed region.
    #      single GPU (default): (1) an eager pass with a HIP event pair around every launch of the step driver, the dedup sorts kept
    #      on the launch stream so each kernel runs alone -> the per-kernel table; (2) the kernel with the largest time per step is
    #      left OUTSIDE the hipGraphs (graph A -> eager launch with events -> graph B) so it is bracketed by events INSIDE the timed
    #      region (timed events cannot be recorded into a capture on ROCm 7.2); (3) the whole step in one graph is timed too.
    lib = _lib.load()
    TAG = {k[7:]: v for k, v in _lib.parse_enums().items() if k.startswith("BR_TAG_")}
    n1, n2, n3 = eng.cfg.hidden
    l1, l2 = (n1 + 3) & ~3, (n2 + 3) & ~3
    loc_users, loc_items = eng.local_rows("user_mf"), eng.local_rows("item_mf")
    deferred_mode = bool(getattr(eng, "deferred", False))
    fused_rows = deferred_mode and ctx is None      # the pair launch forms the MF-row gradients itself (neumf_step.cpp fuse_mf)
    uniq_u = sum(int(torch.unique(b[0]).numel()) for b in batches) / len(batches)
    uniq_i = sum(int(torch.unique(b[1]).numel()) for b in batches) / len(batches)
    # deferred lookup: a pair whose row was NOT updated in the previous step also reads the row's m and v (the replay needs them): measured
    # on the primed tables for the batch the cursor stands at (steady state: ~94 % of the user rows, ~52 % of the item rows at config 2)
    lag_u = lag_i = 0.0      # (measured behind the timed region, where the tables are in steady state)
    # algorithmic work per launch (DESIGN.md "Algorithmic bytes"): name, kernel symbol, bound, work, phases to keep eager
    rows_pair = 8 + 2 * D * 4 + (4 if fused_rows else 0)            # sorted id + position, one 2D-float gradient row (+ ddot)
    per_uniq = 6 * 2 * D * 4 + (8 if deferred_mode else 0)           # table, m, v rows read and written (+ last[])
    ROWS_ALL = ("OPT_TABLES", "ROWS_USER", "SWEEP_USER", "ROWS_ITEM", "SWEEP_ITEM")
    # single GPU, deferred, batch > 16 384, embed_dim 64 / 128: the chunk sorts ride in the lookup's launch (neumf_step.cpp fused_index)
    fused_sort = deferred_mode and ctx is None and B > 16384 and D in (64, 128) and os.environ.get("BR_FUSED_SORT", "1") != "0" and os.environ.get("BR_WAVE_ROWS", "1") != "0"
    # riders of the pair launch (single GPU, dropout on): the next step's keep-bit planes written + the dense finalize's slabs read and
    # theta / m / v / grad of the dense vector updated - their algorithmic bytes count towards that launch
    riders = deferred_mode and ctx is None and eng.cfg.dropout > 0 and os.environ.get("BR_KEEP_PREFETCH", "2") == "2"
    rider_bytes = (B * 4 * sum((w + 31) // 32 for w in (2 * D, n1, n2)) + 4 * int(eng.slabs.numel()) + 7 * 4 * int(eng.theta.buf.numel())) if riders else 0
    kb = lambda width: (B * ((width + 31) // 32) * 4) if eng.cfg.dropout > 0 else 0      # keep-bit plane of a layer's input
    SPEC = {
        "EMBED_FWD": (((("lookup on deferred tables (4 lookups + replay of lagging rows + GMF dot + concat) + the chunk sorts of both id streams riding in its grid",
                         "lookup_sort_kernel") if fused_sort else
                        ("neumf_embed_fwd_deferred (4 lookups + replay of lagging rows + GMF dot + concat)", "neumf_embed_fwd_deferred_wave_kernel")) + ("hbm",
                       B * (4 * D * 4 + 2 * D * 4 + 2 * D * 4 + 12))) if deferred_mode and ctx is None else
                      ("neumf_embed_fwd (4 lookups + GMF dot + concat)", "neumf_embed_fwd_kernel", "hbm", B * (4 * D * 4 + 2 * D * 4 + 12)), ("FWD1",)),
        "EMBED_BWD": (("mf_grad_inplace" if deferred_mode and ctx is None else "neumf_embed_bwd", "neumf_embed_bwd_kernel", "hbm", B * (4 * D * 4 + 8)), ("OPT_TABLES",)),
        "KEEP_BITS": ((f"dropout keep-bit planes (Philox4x32-10, {2 * D} + {n1} + {n2} bits per row)", "keep_bits_kernel", None, None), ("FWD1",)),
        "STEP_STATE": (("step counter / alpha_t advance + BatchNorm sum reset", "step_state_advance_kernel", None, None), ("FWD1",)),
        # dense layers: (fp32-equivalent flop, algorithmic bytes = the activation rows in and out + the keep-bit plane of the input)
        "FWD_L1": ((f"dense_fwd[{2 * D}x{n1}]", "dense_fwd_kernel", "mfma", (2.0 * B * 2 * D * n1, B * (2 * D + n1) * 4 + kb(2 * D))), ("FWD1",)),
        "FWD_L2": ((f"dense_fwd[{n1}x{n2}] (+ BatchNorm 1 finalize)", "dense_fwd_kernel", "mfma", (2.0 * B * n1 * n2, B * (n1 + n2) * 4 + kb(n1))), ("FWD2",)),
        "FWD_L3": ((f"dense_fwd[{n2}x{n3}]", "dense_fwd_kernel", "mfma", (2.0 * B * n2 * n3, B * (n2 + n3) * 4 + kb(n2))), ("FWD3",)),
        # (an HBM kernel: 0.2 GFLOP over 2 x a2 / gh2 rows + a3, logits, probabilities - SURVEY.md 8d's per-pair bytes of the tail)
        "HEAD": ((f"tail: BatchNorm 2 finalize + dense {n2}x{n3} fwd + head + loss + their backward (one launch)", "neumf_tail_mfma_kernel", "hbm",
                  B * (2 * l2 * 4 + n3 * 4 + 5 * 4) + (B * ((n2 + 31) // 32) * 4 if eng.cfg.dropout > 0 else 0)), ("FWD3",)),
        "BWD_L3": ((f"dense_bwd[{n2}x{n3}]", "dense_bwd_kernel", "mfma", (4.0 * B * n2 * n3, B * 2 * (n2 + n3) * 4 + kb(n2))), ("FWD3",)),
        "BWD_L2": ((f"dense_bwd[{n1}x{n2}] (dx + dW + db + BatchNorm sums, one launch)", "dense_bwd_kernel", "mfma", (4.0 * B * n1 * n2, B * 2 * (n1 + n2) * 4 + kb(n1))), ("BWD2",)),
        "BWD_L1": ((f"dense_bwd[{2 * D}x{n1}] (dx + dW + db, one launch)", "dense_bwd_kernel", "mfma", (4.0 * B * 2 * D * n1, B * 2 * (2 * D + n1) * 4 + kb(2 * D))), ("BWD1",)),
        "INDEX_SORT": (("row index, both tables: chunk sort", "chunk_sort_kernel", None, None), ("FWD1",)),
        "INDEX_USER": (("row index, both tables: chunk rank / merge", "chunk_rank_kernel", None, None), ("FWD1",)),
        "INDEX_ITEM": (("row index [item]", "chunk_rank_kernel", None, None), ("FWD1",)),
        "SEG_PARTIALS": (("segment partial sums of long duplicate runs (both tables)", "segment_partials_kernel", None, None), ROWS_ALL),
        "ADAM_ROWS_USER": ((("adam_rows[user + item, one launch" + (" + riders: next step's keep-bit planes, dense finalize]" if riders else "]"),
                             "adam_rows_wave_kernel", "hbm", 2 * B * rows_pair + (uniq_u + uniq_i) * per_uniq + rider_bytes) if ctx is None else
                            ("adam_rows_sorted[user]", "adam_rows_sorted_kernel", "hbm", B * rows_pair + uniq_u * per_uniq)), ROWS_ALL),
        "ADAM_ROWS_ITEM": (("adam_rows_sorted[item]", "adam_rows_sorted_kernel", "hbm", B * rows_pair + uniq_i * per_uniq), ROWS_ALL),
        "SWEEP_USER": ((f"adam_dense_sweep[user {loc_users}x{2 * D}]", "adam_dense_sweep_kernel", "hbm", 6 * 4 * loc_users * 2 * D), ("SWEEP_USER",)),
        "SWEEP_ITEM": ((f"adam_dense_sweep[item {loc_items}x{2 * D}]", "adam_dense_sweep_kernel", "hbm", 6 * 4 * loc_items * 2 * D), ("SWEEP_ITEM",)),
        "ADAM_FLAT": (("dense finalize: slab reduce x3 + BatchNorm grads + Adam (one launch)" if ctx is None else "adam_flat", "dense_finalize_kernel", None, None), ("OPT_DENSE",)),
        "REDUCE": (("reduce_slabs (each)", "reduce_slabs_kernel", None, None), ("BWD1",)),
        "SMALL": (("bn / small (each)", "-", None, None), ("BNG",)),
    }
    use_graph = ctx is None and not args.no_graph
    graph_exec = None
    sharded_graph = None
    if ctx is not None and not args.no_graph and deferred_mode:
        # N > 1: the sharded step with its four RCCL collectives as ONE hipGraph per step (captured behind the first eager step; the engine
        # keeps the eager sequence if the runtime refuses the capture)
        try:
            eng.enable_graph(B)
            sharded_graph = eng
        except Exception as exc:  # noqa: BLE001
            log(f"sharded graph not enabled: {exc}")
    # warm-up outside the probe, at least one pass over the batch cycle: the tables are then in the state of a RUNNING job (every row of
    # the cycle carries moments and a lag).  On fresh tables (m = v = 0) the deferred kernels skip their replay arithmetic and look
    # ~30 % faster than they are in steady state.
    seeded = None
    if args.steady_state_lags:
        if not (deferred_mode and ctx is None):
            raise SystemExit("--steady-state-lags: single GPU, adam_dense by deferred replay")
        seeded = seed_steady_state(eng, uniq_u, uniq_i, args.since_flush, 4 * eng.ALPHA_RING)
        log(f"steady-state lags seeded: user mean lag {seeded['user']['mean_lag']:.1f}, item {seeded['item']['mean_lag']:.1f} steps")
        need = 3 * args.warmup + 2 * args.steps + args.profile_steps + 8
        if need > n_batches:
            log(f"WARNING: the legs of this run take ~{need} steps but the cycle has {n_batches} batches: repeated batches come back with lags of a few steps")
    prime = max(args.warmup, n_batches if (deferred_mode and seeded is None) else 0)
    run_steps(eng, batches, prime, row0, batch_total)
    pyprobe = None
    eager_profile = None
    graph_error = None
    dom_tag = None

    def per_step_us(per_tag, nsteps):
        return {t: us * n / nsteps for t, (us, n) in per_tag.items()}

    def pick_dominant(per_tag, nsteps):
        share = per_step_us(per_tag, nsteps)
        inv = {v: k for k, v in TAG.items()}
        cands = [(share[t], inv[t]) for t in share if inv.get(t) in SPEC and SPEC[inv[t]][0][2] is not None]
        return max(cands)[1] if cands else None

    if use_graph:
        np_ = max(1, min(args.profile_steps, args.steps))
        aux = eng.step_struct.aux_stream
        eng.step_struct.aux_stream = None        # dedup sorts on the launch stream: every kernel alone between its two events
        # (two un-probed steps in this arrangement first: without the aux stream the planes / finalize are launches of their own, and a
        #  kernel's first launch loads its code object - milliseconds that would sit in a 10-step mean)
        run_steps(eng, batches, 2, row0, batch_total)
        torch.cuda.synchronize()
        if lib.brProbeEnable(64 * np_) != 0:
            raise RuntimeError(lib.brGetLastError().decode())
        dte = timed(eng, batches, np_, 0, ctx, row0, batch_total)
        eng.step_struct.aux_stream = aux
        per_tag = read_probe(lib)
        lib.brProbeEnable(0)
        eager_profile = {"steps": np_, "ms_per_step": dte / np_ * 1e3, "value": B * np_ / dte, "unit": "pairs/s",
                         "note": "eager, serial launch sequence with a HIP event pair around every launch (source of the per-kernel table)"}
        log(f"eager profiling pass: {dte / np_ * 1e3:.3f} ms/step")
        dom_tag = pick_dominant(per_tag, np_)
        # the timed region replays the WHOLE step as one hipGraph; the dominant kernel is bracketed inside it by event-record nodes
        # (brProbeGraph*: a node per replay pointed at that replay's own event pair), so its duration is measured live in every step
        # Every record node is a barrier in the replayed graph (~7 us each, measured), so the timed region alternates two captures of
        # the same step: the plain graph, and every PROBE_EVERY-th step the one that carries the two nodes.
        # Round 3: S = --steps-per-launch consecutive steps are ONE graph launch (NeuMFEngine.enable_graph_multi: the gap between two graph
        # launches and the staging launch are paid once per S steps); the record nodes then sit around the LAST step's launch of every
        # second group.
        PROBE_EVERY = 4
        S_MULTI = args.steps_per_launch if (ctx is None and args.steps_per_launch > 1 and row0 == 0 and batch_total == B) else 1
        graph_exec = g_probe = g_plain = gm_probe = gm_plain = None
        try:
            if S_MULTI > 1:
                make_groups(batches, S_MULTI)
                gm_probe = eng.enable_graph_multi(B, S_MULTI, keep_graph=True, probe_tag=TAG[dom_tag])
                gobj = gm_probe["graph"]
            else:
                lib.brProbeGraphSelect(TAG[dom_tag])
                eng.enable_graph(B, keep_graph=True)
                g_probe = eng._graph
                gobj = g_probe["graphs"][0]
            if lib.brProbeGraphNodes() > 0 and hasattr(gobj, "raw_cuda_graph_exec"):
                graph_exec = ctypes.c_void_p(int(gobj.raw_cuda_graph_exec()))
                if lib.brProbeGraphEnable(args.steps) != 0 or lib.brProbeGraphArm(graph_exec, 0) != 0:
                    log(f"graph probe refused: {lib.brGetLastError().decode()}")
                    graph_exec = None
            lib.brProbeGraphSelect(-1)
            if graph_exec is None:
                # no record nodes in this runtime: keep the dominant kernel between two graphs instead (graph A -> eager launch with
                # events -> graph B), the round-1 arrangement
                eng.disable_graph()
                S_MULTI = 1
                eng.enable_graph(B, eager_phases=SPEC[dom_tag][1])
            elif S_MULTI > 1:
                gm_plain = eng.enable_graph_multi(B, S_MULTI)
                g_plain = eng._graph              # the single-step graph: steps off the group boundaries
            else:
                eng.enable_graph(B)
                g_plain = eng._graph
            run_steps(eng, batches, max(2, args.warmup), row0, batch_total)
        except Exception as exc:  # noqa: BLE001  - a driver / runtime that cannot capture this step: time the eager sequence
            log(f"hipGraph capture failed ({type(exc).__name__}: {exc}); falling back to eager launches")
            eng.disable_graph()
            use_graph = False
            graph_error = f"{type(exc).__name__}: {exc}"
    if use_graph and graph_exec is not None and S_MULTI > 1:
        probed, seen = [], [0]

        def arm_group(k):
            if seen[0] % 2 == 0:
                if lib.brProbeGraphArm(graph_exec, seen[0]) != 0:
                    raise RuntimeError(lib.brGetLastError().decode())
                eng._graph_multi = gm_probe
                probed.append(seen[0])
            else:
                eng._graph_multi = gm_plain
            seen[0] += 1
        eng._graph = g_plain
        dt = timed(eng, batches, args.steps, 0, ctx, row0, batch_total, None, arm_group)
        eng._graph_multi = gm_plain
        ms, tot = ctypes.c_float(), 0.0
        for k in probed:
            if lib.brProbeGraphRead(k, ctypes.byref(ms)) != 0:
                raise RuntimeError(lib.brGetLastError().decode())
            tot += ms.value
        if probed:
            per_tag[TAG[dom_tag]] = (tot / len(probed) * 1e3, len(probed))
        probe_src = {t: ((f"timed region: event-record nodes inside the replayed graph around the last step of every 2nd group of {S_MULTI} steps "
                          f"({len(probed)} samples in {args.steps} steps)", len(probed))
                         if (t == TAG[dom_tag] and probed) else ("eager profiling pass", np_)) for t in per_tag}
        lib.brProbeGraphEnable(0)
    elif use_graph and graph_exec is not None:
        probed = [k for k in range(args.steps) if k % PROBE_EVERY == 0]

        def arm(k):
            if k % PROBE_EVERY == 0:
                if lib.brProbeGraphArm(graph_exec, k) != 0:
                    raise RuntimeError(lib.brGetLastError().decode())
                eng._graph = g_probe
            else:
                eng._graph = g_plain
        dt = timed(eng, batches, args.steps, 0, ctx, row0, batch_total, arm)
        eng._graph = g_plain
        ms, tot = ctypes.c_float(), 0.0
        for k in probed:
            if lib.brProbeGraphRead(k, ctypes.byref(ms)) != 0:
                raise RuntimeError(lib.brGetLastError().decode())
            tot += ms.value
        per_tag[TAG[dom_tag]] = (tot / len(probed) * 1e3, len(probed))
        probe_src = {t: ((f"timed region: event-record nodes inside the replayed graph, every {PROBE_EVERY}th step ({len(probed)} of {args.steps})", len(probed))
                         if t == TAG[dom_tag] else ("eager profiling pass", np_)) for t in per_tag}
        lib.brProbeGraphEnable(0)
    elif use_graph:
        if lib.brProbeEnable(16 * args.steps) != 0:
            raise RuntimeError(lib.brGetLastError().decode())
        dt = timed(eng, batches, args.steps, 0, ctx, row0, batch_total)
        live = read_probe(lib)          # only the launches outside the graphs
        lib.brProbeEnable(0)
        live = {t: v for t, v in live.items() if t == TAG[dom_tag]}     # its phase neighbours keep their serial-pass figures
        per_tag.update(live)
        probe_src = {t: (("timed region", args.steps) if t in live else ("eager profiling pass", np_)) for t in per_tag}
    else:
        if lib.brProbeEnable(64 * args.steps) != 0:
            raise RuntimeError(lib.brGetLastError().decode())
        if ctx is not None:
            users_rows = eng.local_rows("user_mf")
            pyprobe = PyProbe({"brAdamDenseSweep": lambda a: TAG["SWEEP_USER"] if int(a[3]) == users_rows else TAG["SWEEP_ITEM"],
                               "brAdamRowsSorted": lambda a: TAG["ADAM_ROWS_USER"] if int(a[3]) == users_rows else TAG["ADAM_ROWS_ITEM"],
                               "brAdamRowsSortedDeferred": lambda a: TAG["ADAM_ROWS_USER"] if int(a[4]) == users_rows else TAG["ADAM_ROWS_ITEM"],
                               "brNeumfEmbedForward": lambda a: TAG["EMBED_FWD"], "brNeumfEmbedBackward": lambda a: TAG["EMBED_BWD"]})
            _lib.set_probe(pyprobe)
        dt = timed(eng, batches, args.steps, 0, ctx, row0, batch_total)
        _lib.set_probe(None)
        per_tag = read_probe(lib)
        if pyprobe is not None:
            pyprobe.merge_into(per_tag)
        lib.brProbeEnable(0)
        probe_src = {t: ("timed region", args.steps) for t in per_tag}
        dom_tag = pick_dominant(per_tag, args.steps)
    eng.check_ids()
    log(f"timed region done: {dt / args.steps * 1e3:.3f} ms/step")
    pairs_per_s = B * world * args.steps / dt

    if deferred_mode and ctx is None and hasattr(eng, "last"):
        ub, ib, _ = batches[_CURSOR.get(id(batches), 0) % len(batches)]
        lag_u = float((eng.last["user"][ub.long()].long() + 1 < eng.t + 1).float().mean().item())      # the next step is t + 1: rows must hold steps <= t
        lag_i = float((eng.last["item"][ib.long()].long() + 1 < eng.t + 1).float().mean().item())
    kernels = {}
    gpu_us_per_step = 0.0
    for tname, ((name, sym, bound, work), _ph) in SPEC.items():
        if TAG.get(tname) not in per_tag:
            continue
        if tname == "EMBED_FWD" and bound == "hbm" and deferred_mode and ctx is None:
            work = work + int(B * (lag_u + lag_i) * 2 * 2 * D * 4)      # m and v rows of the lagging rows (fused [mlp | mf] rows: 2 D floats)
        us, n = per_tag[TAG[tname]]
        src, nsteps = probe_src[TAG[tname]]
        k = {"kernel": sym, "us": us, "launches_per_step": n / nsteps, "us_per_step": us * n / nsteps, "measured_in": src}
        if bound == "hbm":
            k.update({"bound": "hbm", "bytes": work, "achieved_GBps": work / us * 1e-3, "frac": work / us * 1e-3 / HBM_PEAK_GBPS})
        elif bound == "mfma":
            # the tower's GEMMs: fp32-equivalent flop against the fp32-MFMA peak (`frac_mfma_fp32_equiv`, comparable across rounds), the share of
            # the launch the matrix pipe is busy (bf16x6, DESIGN.md 4c: six 16-cycle bf16 MFMAs do the work of eight 32-cycle fp32 ones ->
            # 0.375 x), and the activation traffic against HBM; `bound` names the nearer roofline and `frac` is the fraction of THAT one
            flop, nbytes = work
            frac = flop / us * 1e-6 / MFMA_F32_PEAK_TFLOPS
            busy = frac * (0.375 if MLP_MATH == "bf16x6" else 1.0)
            fh = nbytes / us * 1e-3 / HBM_PEAK_GBPS
            k.update({"bound": "hbm" if fh > busy else "mfma", "frac": max(fh, busy), "math": MLP_MATH, "flop": flop, "achieved_TFLOPs": flop / us * 1e-6,
                      "frac_mfma_fp32_equiv": frac, "matrix_pipe_busy": busy, "bytes": nbytes, "achieved_GBps": nbytes / us * 1e-3, "frac_hbm": fh})
        kernels[name] = k
        gpu_us_per_step += k["us_per_step"]
    for k in kernels.values():
        k["share_of_kernel_time"] = k["us_per_step"] / gpu_us_per_step

    # the roofline entry = the kernel with the largest time per step (measured, not assumed)
    roofline = traffic = tmeta = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if dom_tag is not None:
        (dom_key, dom_sym, _b, _w), _ = SPEC[dom_tag]
        dom = kernels[dom_key]
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("hbm_bytes_per_launch", {}).get(dom_sym)
                tmeta = {"source": tj.get("source"), "commit": tj.get("commit")}
            except Exception:  # noqa: BLE001
                traffic = None
        roofline = {"bound": dom["bound"], "kernel": f"{dom_sym} ({dom_key})", "achieved": dom["achieved_GBps" if dom["bound"] == "hbm" else "achieved_TFLOPs"],
                    "peak": HBM_PEAK_GBPS if dom["bound"] == "hbm" else MFMA_F32_PEAK_TFLOPS, "unit": "GB/s" if dom["bound"] == "hbm" else "TFLOP/s",
                    "frac": dom["frac"], "traffic": traffic, "traffic_from": tmeta, "avg_launch_us": dom["us"], "measured_in": dom["measured_in"],
                    "share_of_kernel_time": dom["share_of_kernel_time"], "chosen_by": "largest measured time per step among the kernels of the eager pass"}
        roofline["algorithmic_bytes_per_launch" if dom["bound"] == "hbm" else "algorithmic_flop_per_launch"] = dom["bytes" if dom["bound"] == "hbm" else "flop"]
        if dom["bound"] == "hbm" and dom_tag.startswith("ADAM_ROWS"):
            roofline["unique_rows_per_batch"] = {"user": uniq_u, "item": uniq_i}
    else:
        dom_key = None

    lazy = sweep_leg = flush_info = full_graph = None
    if use_graph:
        # the same graph without the two event-record nodes (what a training loop runs)
        try:
            if S_MULTI > 1:
                eng._graph_multi = gm_plain
                eng._graph = g_plain
            else:
                eng.enable_graph(B)
            # (these two sub-legs are the median of three runs of K steps: one run is ~15 ms, short enough for a clock ramp to swap their order)
            med3 = lambda: sorted(timed(eng, batches, args.steps, args.warmup if r == 0 else 0, ctx, row0, batch_total) for r in range(3))[1]
            dtg = med3()
            full_graph = {"value": B * args.steps / dtg, "unit": "pairs/s", "ms_per_step": dtg / args.steps * 1e3, "steps_per_graph_launch": S_MULTI,
                          "timing": "median of 3 runs of --steps steps"}
            log(f"whole-step graph: {dtg / args.steps * 1e3:.3f} ms/step")
            if S_MULTI > 1:
                # one graph launch per step, as rounds 1-2 timed it
                eng._graph_multi = None
                dt1 = med3()
                full_graph["one_step_per_launch"] = {"value": B * args.steps / dt1, "unit": "pairs/s", "ms_per_step": dt1 / args.steps * 1e3}
                log(f"whole-step graph, one step per launch: {dt1 / args.steps * 1e3:.3f} ms/step")
                eng._graph_multi = gm_plain
        except Exception as exc:  # noqa: BLE001
            log(f"whole-step graph leg failed: {exc}")
            eng.disable_graph()
    if deferred_mode and ctx is None and seeded is None:
        # the deferred tables must be flushed at least every BR_ALPHA_RING-8 steps: run up to that point and
        # time the flush a long job pays there (worst case: every row replays a full ring of steps)
        period = eng.ALPHA_RING - 8
        todo = max(0, period - 1 - (eng.t - eng._flush_t))
        while todo > 0:        # in slices: ~1000 graph launches queued at once stalled rocprofv3's --pmc pass on the fork-free graph
            run_steps(eng, batches, min(64, todo), row0, batch_total)
            torch.cuda.synchronize()
            todo -= 64
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.flush()
        torch.cuda.synchronize()
        fl = time.perf_counter() - t0
        ms = dt / args.steps * 1e3
        flush_info = {"flush_ms": fl * 1e3, "every_steps": period, "amortized_ms_per_step": fl * 1e3 / period,
                      "value_including_flush": B * period / (period * ms * 1e-3 + fl), "unit": "pairs/s"}
        log(f"flush after {period} steps: {fl * 1e3:.2f} ms")

    def extra_leg(optimizer, dense_impl):
        torch.cuda.empty_cache()
        e2 = build(optimizer, dense_impl)
        if use_graph:
            run_steps(e2, batches, 2, row0, batch_total)
            try:
                e2.enable_graph(B)
            except Exception as exc:  # noqa: BLE001
                log(f"graph capture failed in an extra leg: {exc}")
                e2.disable_graph()
        d2 = timed(e2, batches, args.steps, args.warmup, ctx, row0, batch_total)
        del e2
        return {"value": B * world * args.steps / d2, "unit": "pairs/s", "ms_per_step": d2 / args.steps * 1e3}

    if not args.no_lazy and args.optimizer == "adam_dense" and ctx is None:     # the extra legs are single-GPU information
        del eng
        if deferred_mode:
            sweep_leg = extra_leg("adam_dense", "sweep")
            sweep_leg["note"] = "same Keras semantics by sweeping every table row every step (bit-equal tables, tests/test_gpu_neumf.py)"
            log(f"adam_dense sweep done: {sweep_leg['ms_per_step']:.3f} ms/step")
        torch.cuda.empty_cache()
        eng2 = build("adam_lazy")
        if use_graph:
            run_steps(eng2, batches, 2, row0, batch_total)
            eng2.enable_graph(B)
        dt2 = timed(eng2, batches, args.steps, args.warmup, ctx, row0, batch_total)
        log(f"adam_lazy done: {dt2 / args.steps * 1e3:.3f} ms/step")
        lazy = {"value": B * world * args.steps / dt2, "unit": "pairs/s", "ms_per_step": dt2 / args.steps * 1e3,
                "note": "touched-rows-only Adam: NOT the reference's (Keras non-lazy) semantics"}
        del eng2

    legs = None
    if rank == 0 and ctx is None and not args.no_legs:
        ops = importlib.import_module("binary-recommendation_amd.ops")
        try:
            del eng
        except NameError:
            pass
        torch.cuda.empty_cache()
        legs = {}
        log("legs: gather")
        legs["gather"] = gather_leg(ops, dev, args.users, args.items, D, B, 99)
        torch.cuda.empty_cache()
        log("legs: zipf")
        zb = make_batches(n_batches, B, U, I, dev, 4321, True, idt)
        ez = build(args.optimizer)
        run_steps(ez, zb, len(zb), row0, batch_total)       # one pass over the cycle: steady state, as for the headline
        ez.enable_graph(B)
        dz = timed(ez, zb, args.steps, args.warmup, None, row0, batch_total)
        ez.check_ids()
        legs["zipf"] = {"workload": "the headline step with Zipf(1.05) user and item ids (whole step in one hipGraph)", "value": B * args.steps / dz, "unit": "pairs/s",
                        "ms_per_step": dz / args.steps * 1e3,
                        "unique_rows_per_batch": {"user": sum(int(torch.unique(b[0]).numel()) for b in zb) / len(zb),
                                                  "item": sum(int(torch.unique(b[1]).numel()) for b in zb) / len(zb)}}
        del ez, zb
        torch.cuda.empty_cache()
        log("legs: bpr")
        legs["bpr"] = bpr_leg(dev, args.users, args.items, D, B, 7)
        log("legs: twotower")
        legs["twotower"] = twotower_leg(ops, dev, args.users, args.items, 64, 64, 8192, 11)
        torch.cuda.empty_cache()

    cpu = None
    if rank == 0 and ctx is None and not args.no_cpu_baseline:
        log("cpu baseline start")
        cpu = cpu_baseline(args, D)
        log("cpu baseline done")

    if rank == 0:
        line = {
            "metric": "(user,item) pairs/sec at batch 65536 dim 64", "value": pairs_per_s, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"NeuMF-{args.variant} full training step (trainers/NFC_plain.py graph: 4 embeddings, "
                                   f"{2 * D}->{'->'.join(map(str, eng_hidden(args, neumf)))}->1, BCE, Keras-Adam {args.optimizer}), "
                                   f"embed_dim={D}, {args.users} users x {args.items} items per GPU, batch {B} per GPU, "
                                   f"{'Zipf(1.05)' if args.zipf else 'uniform'} ids",
                       "global_batch": batch_total, "parallelism": "single GPU" if ctx is None else f"row-sharded tables x{world} + dp{world}, {'global' if args.sync_bn else 'per-replica'} BatchNorm",
                       "steps_per_graph_launch": (S_MULTI if use_graph else None),
                       "optimizer": args.optimizer + (f" ({'deferred replay' if deferred_mode else 'per-step sweep'})" if args.optimizer == "adam_dense" else "")},
            "roofline": roofline, "cpu_baseline": cpu, "legs": legs, "adam_lazy": lazy, "adam_dense_sweep": sweep_leg, "whole_step_graph": full_graph, "deferred_flush": flush_info, "gpu_kernel_us_per_step": gpu_us_per_step,
            "unique_rows_per_batch": {"user": uniq_u, "item": uniq_i}, "lagging_rows_per_batch": {"user": lag_u, "item": lag_i}, "steady_state_lags": seeded,
            "launch_mode": ((((f"hipGraph replay, {S_MULTI} consecutive steps per graph launch (NeuMFEngine.enable_graph_multi; the group's ids / labels staged by one launch); "
                              f"{dom_key} bracketed by event-record nodes around the last step of every 2nd group (a HIP event pair each)") if S_MULTI > 1 else
                             f"hipGraph replay of the whole step, one graph per step; {dom_key} bracketed by event-record nodes inside every 4th replay (a HIP event pair each)")
                             if graph_exec is not None else f"hipGraph replay (graph A -> eager {dom_key} with HIP events -> graph B)") if use_graph
                            else (("sharded step incl. its RCCL collectives as one hipGraph per step" if sharded_graph is not None and sharded_graph.graph_active
                                   else "eager launches (brNeumfStepRun phases + torch.distributed collectives)" + (f" [graph refused: {sharded_graph._sgraph['refused']}]" if sharded_graph is not None and sharded_graph._sgraph else ""))
                                  if ctx is not None else "eager launches (brNeumfStepRun)")),
            "eager": eager_profile, "graph_error": graph_error,
            "kernels": kernels,
        }
        print(json.dumps(line), file=json_out, flush=True)
    if ctx is not None:
        sys.stdout.flush(); sys.stderr.flush()
        if sharded_graph is not None and sharded_graph.graph_active:
            # no process-group teardown behind a capture that holds RCCL nodes: destroy_process_group() did not return on ROCm 7.2 / RCCL 2.26
            ctx.barrier()
            os._exit(0)
        torch.distributed.destroy_process_group()


def eng_hidden(args, neumf):
    return neumf.NeuMFConfig(variant=args.variant, dim=args.dim).hidden


if __name__ == "__main__":
    main()
