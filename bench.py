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

Rank 0 prints ONE JSON line (metric/value/... + "roofline" for the dominant kernel, measured
with HIP events on the launch stream inside the timed region, + "cpu_baseline": the torch-CPU
fp32 port of trainers/NFC_plain.py's step from oracle/torch_ref.py on a bounded sample).
"""
from __future__ import annotations

import argparse
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


def make_batches(n, B, U, I, dev, seed, zipf):
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = []
    for _ in range(n):
        if zipf:
            # Zipf(alpha=1.05)-like ranks by inverse-CDF on a truncated power law
            def draw(N):
                u = torch.rand(B, generator=g, dtype=torch.float64)
                a = 1.05
                r = ((N ** (1 - a) - 1) * u + 1) ** (1 / (1 - a))
                return (r.long().clamp_(1, N) - 1)
            users, items = draw(U), draw(I)
        else:
            users = torch.randint(0, U, (B,), generator=g)
            items = torch.randint(0, I, (B,), generator=g)
        labels = (torch.rand(B, generator=g) < 0.25).float()   # 1 pos : 3 neg (NeuMFModel.py:102)
        out.append((users.int().to(dev), items.int().to(dev), labels.to(dev)))
    return out


def run_steps(eng, batches, n, row0, batch_total):
    nb = len(batches)
    for s in range(n):
        u, i, y = batches[s % nb]
        eng.train_step(u, i, y, row0=row0, batch_total=batch_total)


def timed(eng, batches, steps, warmup, ctx, row0, batch_total):
    run_steps(eng, batches, warmup, row0, batch_total)
    if ctx is not None:
        ctx.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(eng, batches, steps, row0, batch_total)
    torch.cuda.synchronize()
    if ctx is not None:
        ctx.barrier()
    dt = time.perf_counter() - t0
    if ctx is not None and ctx.world > 1:
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
    ap.add_argument("--profile-steps", type=int, default=10, help="graph mode: eager steps probed per kernel before the timed region")
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
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)
        ctx = par.DistCtx()

    B, D = args.batch, args.dim
    # weak scaling: per-GPU batch AND per-GPU table shard are fixed as N grows
    U, I = args.users * world, args.items * world
    batch_total = B * world
    row0 = rank * B

    impl = args.dense_impl or "deferred"

    def build(optimizer, dense_impl=impl):
        # N > 1: per-replica BatchNorm = what the reference's MirroredStrategy does with a plain BatchNormalization
        # [TF-sem] (and no BatchNorm collective in the step); --sync-bn makes the statistics global
        cfg = neumf.NeuMFConfig(variant=args.variant, dim=D, optimizer=optimizer, seed=20261004, dense_impl=dense_impl, sync_bn=args.sync_bn)
        if world == 1:
            return neumf.NeuMFEngine(cfg, U, I, dev, B, init_seed=1)
        return par.make_sharded_engine(neumf.NeuMFEngine)(cfg, U, I, dev, B, ctx, init_seed=1)

    log(f"building engine: {U} users x {I} items, dim {D}, batch {B}/GPU, world {world}, {args.optimizer}")
    eng = build(args.optimizer)
    log("engine built")
    n_batches = min(args.steps + args.warmup, 32)
    batches = make_batches(n_batches, B, U, I, dev, 1234 + rank, args.zipf)

    # ---- timed region.  HIP events recorded by the step driver on the launch stream bracket the kernel
    #      launches; the roofline figures come from these.
    #      single GPU (default): the step is replayed as hipGraph A -> eager adam_dense_sweep[user] -> hipGraph B,
    #      so the dominant kernel is still bracketed by events INSIDE the timed region (timed events cannot be
    #      recorded into a capture on ROCm 7.2); the other kernels' table comes from an eager pass before it.
    lib = _lib.load()
    TAG = {k[7:]: v for k, v in _lib.parse_enums().items() if k.startswith("BR_TAG_")}
    use_graph = world == 1 and not args.no_graph
    run_steps(eng, batches, args.warmup, row0, batch_total)   # warm-up outside the probe
    pyprobe = None
    eager_profile = None
    graph_error = None
    if use_graph:
        np_ = max(1, min(args.profile_steps, args.steps))
        if lib.brProbeEnable(64 * np_) != 0:
            raise RuntimeError(lib.brGetLastError().decode())
        dte = timed(eng, batches, np_, 0, ctx, row0, batch_total)
        per_tag = read_probe(lib)
        lib.brProbeEnable(0)
        eager_profile = {"steps": np_, "ms_per_step": dte / np_ * 1e3, "value": B * np_ / dte, "unit": "pairs/s",
                         "note": "eager launch sequence with a HIP event pair around every launch (source of the per-kernel table)"}
        log(f"eager profiling pass: {dte / np_ * 1e3:.3f} ms/step")
        sweeping = args.optimizer == "adam_dense" and not eng.deferred
        try:
            eng.enable_graph(B, eager_phases=("SWEEP_USER",) if sweeping else ("BWD1",))
            run_steps(eng, batches, max(2, args.warmup), row0, batch_total)
        except Exception as exc:  # noqa: BLE001  - a driver / runtime that cannot capture this step: time the eager sequence
            log(f"hipGraph capture failed ({type(exc).__name__}: {exc}); falling back to eager launches")
            eng.disable_graph()
            use_graph = False
            graph_error = f"{type(exc).__name__}: {exc}"
    if use_graph:
        if lib.brProbeEnable(4 * args.steps) != 0:
            raise RuntimeError(lib.brGetLastError().decode())
        dt = timed(eng, batches, args.steps, 0, ctx, row0, batch_total)
        live = read_probe(lib)          # only the launches outside the graphs
        lib.brProbeEnable(0)
        per_tag.update(live)
        probe_src = {t: (("timed region", args.steps) if t in live else ("eager profiling pass", np_)) for t in per_tag}
    else:
        if lib.brProbeEnable(64 * args.steps) != 0:
            raise RuntimeError(lib.brGetLastError().decode())
        if world > 1:
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
    eng.check_ids()
    log(f"timed region done: {dt / args.steps * 1e3:.3f} ms/step")
    pairs_per_s = B * world * args.steps / dt

    n1, n2, n3 = eng.cfg.hidden
    loc_users, loc_items = eng.local_rows("user_mf"), eng.local_rows("item_mf")
    kernels = {}

    def hbm(name, tag, bytes_):
        if TAG[tag] in per_tag:
            us, n = per_tag[TAG[tag]]
            kernels[name] = {"_tag": TAG[tag], "us": us, "launches": n, "bound": "hbm", "achieved_GBps": bytes_ / us * 1e-3,
                             "frac": bytes_ / us * 1e-3 / HBM_PEAK_GBPS, "bytes": bytes_}

    def mfma(name, tag, flop):
        if TAG[tag] in per_tag:
            us, n = per_tag[TAG[tag]]
            kernels[name] = {"_tag": TAG[tag], "us": us, "launches": n, "bound": "mfma", "achieved_TFLOPs": flop / us * 1e-6,
                             "frac": flop / us * 1e-6 / MFMA_F32_PEAK_TFLOPS, "flop": flop}

    # algorithmic work per launch: gather 4*D*4 B + 8 B ids per pair; Adam sweep 6*4 B per
    # table element; Adam rows: g row + read/write theta,m,v = 7*4 B per element of a touched row (upper bound: no dups)
    if getattr(eng, "deferred", False) and not eng.sharded:
        # deferred lookup: a lagging row also brings its m and v (upper bound: every row lags), plus x0 and the MF stash out
        hbm("neumf_embed_fwd_deferred(gather4 + replay + dot + concat)", "EMBED_FWD", B * (3 * 4 * D * 4 + 4 * D * 4 + 16))
        hbm("mf_grad_inplace", "EMBED_BWD", B * (2 * 2 * D * 4 + 4))
    else:
        hbm("neumf_embed_fwd(gather4+dot+concat)", "EMBED_FWD", B * (4 * D * 4 + 8))
        hbm("neumf_embed_bwd", "EMBED_BWD", B * (4 * D * 4 + 8))
    hbm(f"adam_dense_sweep[user {loc_users}x{2 * D}]", "SWEEP_USER", 6 * 4 * loc_users * 2 * D)
    hbm(f"adam_dense_sweep[item {loc_items}x{2 * D}]", "SWEEP_ITEM", 6 * 4 * loc_items * 2 * D)
    if world == 1:   # the step driver updates both fused tables in one launch
        hbm("adam_rows_sorted[user + item, one launch]", "ADAM_ROWS_USER", 2 * B * 7 * 2 * D * 4)
    else:
        hbm("adam_rows_sorted[user]", "ADAM_ROWS_USER", B * 7 * 2 * D * 4)
        hbm("adam_rows_sorted[item]", "ADAM_ROWS_ITEM", B * 7 * 2 * D * 4)
    mfma(f"dense_fwd[{2 * D}x{n1}]", "FWD_L1", 2.0 * B * 2 * D * n1)
    mfma(f"dense_fwd[{n1}x{n2}]", "FWD_L2", 2.0 * B * n1 * n2)
    mfma(f"dense_fwd[{n2}x{n3}]", "FWD_L3", 2.0 * B * n2 * n3)
    mfma(f"dense_bwd[{2 * D}x{n1}]", "BWD_L1", 4.0 * B * 2 * D * n1)
    mfma(f"dense_bwd[{n1}x{n2}]", "BWD_L2", 4.0 * B * n1 * n2)
    mfma(f"dense_bwd[{n2}x{n3}]", "BWD_L3", 4.0 * B * n2 * n3)
    for name, tag in (("row_index_build[user]", "INDEX_USER"), ("row_index_build[item]", "INDEX_ITEM"), ("tail: dense 3 fwd + head + loss + their backward (one launch)", "HEAD"),
                      ("reduce_slabs (each)", "REDUCE"), ("bn/small (each)", "SMALL"),
                      ("dense finalize: slab reduce x3 + BN grads + Adam (one launch)" if world == 1 else "adam_flat", "ADAM_FLAT")):
        if TAG[tag] in per_tag:
            kernels[name] = {"_tag": TAG[tag], "us": per_tag[TAG[tag]][0], "launches": per_tag[TAG[tag]][1]}
    gpu_us_per_step = 0.0
    for v in kernels.values():
        v["measured_in"], nsteps = probe_src[v.pop("_tag")]
        gpu_us_per_step += v["us"] * v["launches"] / nsteps

    # dominant kernel of the step
    if args.optimizer == "adam_dense" and not eng.deferred:
        dom_key, dom_name = f"adam_dense_sweep[user {loc_users}x{2 * D}]", "adam_dense_sweep_kernel"
    else:
        dom_key, dom_name = f"dense_bwd[{2 * D}x{n1}]", "dense_dx_kernel+dense_dw_kernel"
    dom = kernels.get(dom_key)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(dom_name)
        except Exception:  # noqa: BLE001
            traffic = None
    roofline = None
    if dom is not None:
        if dom["bound"] == "hbm":
            roofline = {"bound": "hbm", "kernel": f"{dom_name} ({dom_key})", "achieved": dom["achieved_GBps"], "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": dom["frac"], "traffic": traffic, "avg_launch_us": dom["us"],
                        "algorithmic_bytes_per_launch": dom["bytes"], "measured_in": dom["measured_in"]}
        else:
            roofline = {"bound": "mfma", "kernel": f"{dom_name} ({dom_key})", "achieved": dom["achieved_TFLOPs"],
                        "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": dom["frac"], "traffic": traffic,
                        "avg_launch_us": dom["us"], "algorithmic_flop_per_launch": dom["flop"], "measured_in": dom["measured_in"]}

    deferred_mode = bool(getattr(eng, "deferred", False))
    lazy = sweep_leg = flush_info = full_graph = None
    if use_graph:
        # the same engine with the WHOLE step in one graph (what a training loop runs; no launch is left
        # outside, so nothing can be bracketed by events)
        try:
            eng.enable_graph(B)
            dtg = timed(eng, batches, args.steps, args.warmup, ctx, row0, batch_total)
            full_graph = {"value": B * args.steps / dtg, "unit": "pairs/s", "ms_per_step": dtg / args.steps * 1e3}
            log(f"whole-step graph: {dtg / args.steps * 1e3:.3f} ms/step")
        except Exception as exc:  # noqa: BLE001
            log(f"whole-step graph leg failed: {exc}")
            eng.disable_graph()
    if deferred_mode and world == 1:
        # the deferred tables must be flushed at least every BR_ALPHA_RING-8 steps: run up to that point and
        # time the flush a long job pays there (worst case: every row replays a full ring of steps)
        period = eng.ALPHA_RING - 8
        run_steps(eng, batches, max(0, period - 1 - (eng.t - eng._flush_t)), row0, batch_total)
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

    if not args.no_lazy and args.optimizer == "adam_dense" and world == 1:     # the extra legs are single-GPU information
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

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
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
                       "global_batch": batch_total, "parallelism": "single GPU" if world == 1 else f"row-sharded tables x{world} + dp{world}, {'global' if args.sync_bn else 'per-replica'} BatchNorm",
                       "optimizer": args.optimizer + (f" ({'deferred replay' if deferred_mode else 'per-step sweep'})" if args.optimizer == "adam_dense" else "")},
            "roofline": roofline, "cpu_baseline": cpu, "adam_lazy": lazy, "adam_dense_sweep": sweep_leg, "whole_step_graph": full_graph, "deferred_flush": flush_info, "gpu_kernel_us_per_step": gpu_us_per_step,
            "launch_mode": (f"hipGraph replay (graph A -> eager {dom_key} with HIP events -> graph B)" if use_graph
                            else "eager launches (brNeumfStepRun)"),
            "eager": eager_profile, "graph_error": graph_error,
            "kernels": kernels,
        }
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


def eng_hidden(args, neumf):
    return neumf.NeuMFConfig(variant=args.variant, dim=args.dim).hidden


if __name__ == "__main__":
    main()
