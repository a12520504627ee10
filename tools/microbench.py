"""Per-kernel micro-benchmark at BASELINE config 2 sizes (1M users x 100K items, dim 64,
batch 65 536).  HIP-event timing on the launch stream; prints GB/s of ALGORITHMIC bytes."""
import json
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from importlib import import_module

ops = import_module("binary-recommendation_amd.ops")


def timeit(fn, iters=50, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    dev = torch.device("cuda:0")
    U, I, D, B = 1_000_000, 100_000, int(os.environ.get("DIM", 64)), 65536
    g = torch.Generator(device="cpu").manual_seed(1234)
    tabs = {k: (torch.rand(r, D, generator=g) * 0.1 - 0.05).to(dev) for k, r in
            (("user_mlp", U), ("item_mlp", I), ("user_mf", U), ("item_mf", I))}
    users = torch.randint(0, U, (B,), generator=g).int().to(dev)
    items = torch.randint(0, I, (B,), generator=g).int().to(dev)
    res = {}
    outs = [torch.empty(B, D, device=dev) for _ in range(4)]
    t = timeit(lambda: ops.gather_rows([tabs["user_mlp"], tabs["item_mlp"], tabs["user_mf"], tabs["item_mf"]],
                                       [users, items, users, items], outs))
    res["gather4"] = {"us": t * 1e6, "GBps_read": B * 4 * D * 4 / t / 1e9}
    x0 = torch.empty(B, 2 * D, device=dev); dot = torch.empty(B, device=dev)
    t = timeit(lambda: ops.neumf_embed_forward(tabs["user_mlp"], tabs["item_mlp"], tabs["user_mf"], tabs["item_mf"],
                                               users, items, 1, x0, dot))
    res["embed_fwd"] = {"us": t * 1e6, "GBps_read": B * 4 * D * 4 / t / 1e9}
    # BPR
    neg = torch.randint(0, I, (B,), generator=g).int().to(dev)
    ls = torch.zeros(64, dtype=torch.float64, device=dev)
    gu = torch.empty(B, D, device=dev); gi = torch.empty(2 * B, D, device=dev)
    t = timeit(lambda: ops.bpr_forward_backward(tabs["user_mf"], tabs["item_mf"], users, items, neg, 1.0 / B, ls, gu, gi))
    res["bpr_fwd_bwd"] = {"us": t * 1e6, "GBps_rw": B * 6 * D * 4 / t / 1e9}
    # index + adam
    idx = ops.RowIndex(B, torch.int32, dev)
    t = timeit(lambda: idx.build(users, U))
    res["row_index_build"] = {"us": t * 1e6}
    m = torch.zeros_like(tabs["user_mf"]); v = torch.zeros_like(tabs["user_mf"])
    t = timeit(lambda: ops.adam_rows_sorted(tabs["user_mf"], m, v, idx, gu, D, 1e-3))
    res["adam_rows_lazy"] = {"us": t * 1e6, "GBps_rw": B * 7 * D * 4 / t / 1e9}
    mark = torch.zeros(U, dtype=torch.uint8, device=dev)
    t = timeit(lambda: ops.adam_dense_sweep(tabs["user_mf"], m, v, 1e-3, mark=mark), iters=20)
    res["adam_dense_sweep_1M"] = {"us": t * 1e6, "GBps_rw": U * D * 4 * 6 / t / 1e9}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
