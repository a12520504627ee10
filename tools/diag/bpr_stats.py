"""kernel mix of the steady-state BPR step (config 3) for rocprofv3 --kernel-trace --stats"""
import os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
bpr = importlib.import_module("binary-recommendation_amd.bpr")
dev = torch.device("cuda:0"); U, I, F, B = 1_000_000, 100_000, 64, 65536
g = torch.Generator().manual_seed(5)
e = bpr.BPREngine(U, I, F, dev, B, optimizer="adam_dense", dense_impl="deferred")
trip = [tuple(torch.randint(0, N, (B,), generator=g).int().to(dev) for N in (U, I, I)) for _ in range(16)]
for k in range(16 + 200):
    e.train_step(*trip[k % 16])
torch.cuda.synchronize(); e.check_ids()
