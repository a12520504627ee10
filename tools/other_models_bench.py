"""BPR (config 3) and TwoTower (config 4, one GPU) steps at synthetic scale, uniform and Zipf ids: pairs/s and ms/step.
A pathology check (hot ids, accidental hits) more than a tuned benchmark."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from importlib import import_module
bpr = import_module("binary-recommendation_amd.bpr"); tt = import_module("binary-recommendation_amd.two_tower")
dev = torch.device("cuda:0")

def ids(n, N, zipf, g):
    if not zipf:
        return torch.randint(0, N, (n,), generator=g, device=dev, dtype=torch.int32)
    w = 1.0 / torch.arange(1, N + 1, device=dev, dtype=torch.float64) ** 1.05
    return torch.multinomial((w / w.sum()).float(), n, replacement=True, generator=g).to(torch.int32)

def timeit(step, n=20, warm=5):
    for _ in range(warm): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

res = {}
g = torch.Generator(device=dev).manual_seed(5)
for zipf in (False, True):
    U, I, F, B = 1_000_000, 100_000, 64, 65536
    for opt in ("adam_dense", "adam_lazy"):
        e = bpr.BPREngine(U, I, F, dev, B, optimizer=opt)
        u, p, n = ids(B, U, zipf, g), ids(B, I, zipf, g), ids(B, I, zipf, g)
        dt = timeit(lambda: e.train_step(u, p, n))
        e.check_ids()
        res[f"bpr {opt} {'zipf' if zipf else 'uniform'}"] = {"ms_per_step": round(dt * 1e3, 3), "triplets_per_s": round(B / dt)}
        del e
    Bt = 8192
    e = tt.TwoTowerEngine(64, I, U, 64, dev, Bt)
    u, it = ids(Bt, U, zipf, g) + 2, ids(Bt, I, zipf, g) + 2
    dt = timeit(lambda: e.train_step(u, it))
    e.check_ids()
    res[f"twotower b{Bt} {'zipf' if zipf else 'uniform'}"] = {"ms_per_step": round(dt * 1e3, 3), "pairs_per_s": round(Bt / dt)}
    del e
print(json.dumps(res, indent=1))
