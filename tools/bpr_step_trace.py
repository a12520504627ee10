"""BPR step (config 3) alone, for a rocprofv3 kernel trace: 1 M x 100 K tables, 64 factors, 65 536 triplets, Keras-Adam by deferred replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from importlib import import_module
bpr = import_module("binary-recommendation_amd.bpr")
dev = torch.device("cuda:0"); U, I, F, B = 1_000_000, 100_000, 64, 65536
g = torch.Generator().manual_seed(1)
e = bpr.BPREngine(U, I, F, dev, B)
bs = [tuple(torch.randint(0, N, (B,), generator=g).int().to(dev) for N in (U, I, I)) for _ in range(8)]
if os.environ.get("BR_BPR_GRAPH", "1") == "1":
    e.enable_graph(B)
for s in range(10): e.train_step(*bs[s % 8])
torch.cuda.synchronize(); t0 = time.perf_counter()
for s in range(40): e.train_step(*bs[s % 8])
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / 40 * 1e3)
