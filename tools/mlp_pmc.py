"""Few launches of the L1 forward/backward MLP kernels for rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from importlib import import_module
ops = import_module("binary-recommendation_amd.ops")
dev = torch.device("cuda:0")
B, K, N = 65536, 128, 100
x = torch.randn(B, K, device=dev); W = torch.randn(K, N, device=dev) * 0.1; b = torch.zeros(N, device=dev)
y = torch.empty(B, N, device=dev)
gy = torch.randn(B, N, device=dev); gx = torch.empty(B, K, device=dev)
ns = ops.dense_backward_slabs(B, K, N); slabs = torch.empty(ns * (K * N + N), device=dev)
for _ in range(5):
    ops.dense_forward(x, W, b, y, "linear", seed=1, step=1, site=0)
    ops.dense_backward(gy, y, x, W, "linear", slabs, ns, gx=gx, seed=1, step=1)
torch.cuda.synchronize()
print("done")
