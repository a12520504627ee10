import os, sys, json, time
sys.path.insert(0, "/root/repo")
import torch
from importlib import import_module
ops = import_module("binary-recommendation_amd.ops"); tt = import_module("binary-recommendation_amd.two_tower")
from bench import time_us
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
U, I, B, S = 1_000_000, 100_000, 8192, 64
w = 1.0 / torch.arange(1, I + 1, device=dev, dtype=torch.float64) ** 1.05
it_z = (torch.multinomial((w / w.sum()).float(), B, replacement=True, generator=g) + 2).to(torch.int32)
it_u = (torch.randint(0, I, (B,), generator=g, device=dev) + 2).to(torch.int32)
q, c = torch.randn(B, S, device=dev) * 0.3, torch.randn(B, S, device=dev) * 0.3
lse, slots = torch.empty(B, device=dev), torch.zeros(64, dtype=torch.float64, device=dev)
dq, dc = torch.empty_like(q), torch.empty_like(c)
for name, it in (("uniform", it_u), ("zipf", it_z)):
    t1, _ = time_us(lambda: ops.inbatch_softmax_lse(q, c, it, it, 0, lse, slots), reps=5)
    t2, _ = time_us(lambda: ops.inbatch_softmax_grad(q, c, it, it, 0, lse, dq, dc), reps=5)
    print(name, "unique", int(torch.unique(it).numel()), "lse", round(t1, 1), "grad", round(t2, 1))
    e = tt.TwoTowerEngine(64, I, U, 64, dev, B)
    u = (torch.randint(0, U, (B,), generator=g, device=dev) + 2).to(torch.int32)
    for _ in range(5): e.train_step(u, it)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): e.train_step(u, it)
    torch.cuda.synchronize(); print("  step ms", (time.perf_counter() - t0) / 20 * 1e3)
