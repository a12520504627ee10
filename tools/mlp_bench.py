"""Ablation timing of the MLP kernels (HIP events, back-to-back launches)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from importlib import import_module
ops = import_module("binary-recommendation_amd.ops")

def timeit(fn, iters=30, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

dev = torch.device("cuda:0")
B = 65536
res = {}
for (K, N) in ((128, 100), (100, 50), (50, 10), (128, 64), (128, 128)):
    x = torch.randn(B, K, device=dev); W = torch.randn(K, N, device=dev) * 0.1; b = torch.zeros(N, device=dev)
    y = torch.empty(B, N, device=dev); sc = torch.ones(K, device=dev); sh = torch.zeros(K, device=dev)
    stats = torch.zeros(8 * 2 * N, dtype=torch.float64, device=dev)
    keep = ops.dropout_keep_bits(0.2, 1, 1, 0, B, [0], [K])[0]
    res[f"keep_bits[{K}]"] = round(timeit(lambda: ops.dropout_keep_bits(0.2, 1, 1, 0, B, [0], [K], [keep])), 1)
    for tag, kw in (("linear,nodrop", dict(act="linear")), ("sigmoid,nodrop", dict(act="sigmoid")),
                    ("linear,drop", dict(act="linear", drop_p=0.2, keep=keep)), ("sigmoid,drop,bn,stats", dict(act="sigmoid", drop_p=0.2, keep=keep, in_scale=sc, in_shift=sh, stats=stats))):
        act = kw.pop("act")
        t = timeit(lambda: ops.dense_forward(x, W, b, y, act, **kw))
        res[f"fwd[{K}x{N}] {act},{tag}"] = round(t, 1)
    gy = torch.randn(B, N, device=dev); gx = torch.empty(B, K, device=dev)
    ns = ops.dense_backward_slabs(B, K, N); slabs = torch.empty(ns * (K * N + N), device=dev)
    mean = torch.zeros(N, device=dev); rstd = torch.ones(N, device=dev); gam = torch.ones(N, device=dev)
    bns = torch.zeros(8 * 2 * N, dtype=torch.float64, device=dev); ins = torch.zeros(8 * 2 * K, dtype=torch.float64, device=dev)
    mi = torch.zeros(K, device=dev); ri = torch.ones(K, device=dev)
    for tag, kw in (("plain", dict(act="linear")), ("sigmoid,drop", dict(act="sigmoid", in_drop_p=0.2, keep=keep)),
                    ("full", dict(act="sigmoid", in_drop_p=0.2, keep=keep, out_bn=(mean, rstd, gam), bn_sums=bns, in_scale=sc, in_shift=sh, in_bn=(mi, ri), in_bn_sums=ins))):
        act = kw.pop("act")
        dzw = torch.empty(ops.dense_backward_ws_floats(B, K, N), device=dev)
        t = timeit(lambda: ops.dense_backward(gy, y, x, W, act, slabs, ns, gx=gx, dz_ws=dzw, **kw))
        res[f"bwd[{K}x{N}] {tag}"] = round(t, 1)
    out = torch.empty(K * N + N, device=dev)
    res[f"reduce[{K}x{N}] ns={ns}"] = round(timeit(lambda: ops.reduce_slabs(slabs, ns, K * N + N, out)), 1)
print(json.dumps(res, indent=1))
