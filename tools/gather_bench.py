"""G1 alone: the fused-table lookup kernel (4 lookups + GMF dot + MLP concat, brNeumfEmbedForward) back to back at
BASELINE config 2 (1 M x 128 and 100 K x 128 fused tables, batch 65 536) - GB/s of the gathered bytes and of all bytes."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from importlib import import_module
ops = import_module("binary-recommendation_amd.ops")
dev = torch.device("cuda:0"); U, I, D, B = 1_000_000, 100_000, 64, 65536
fu = torch.rand(U, 2 * D, device=dev); fi = torch.rand(I, 2 * D, device=dev)
users = torch.randint(0, U, (B,), device=dev, dtype=torch.int32); items = torch.randint(0, I, (B,), device=dev, dtype=torch.int32)
x0 = torch.empty(B, 2 * D, device=dev); dot = torch.empty(B, device=dev)
def run(): ops.neumf_embed_forward(fu[:, :D], fi[:, :D], fu[:, D:], fi[:, D:], users, items, 1, x0, dot)
for _ in range(5): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(100): run()
e.record(); torch.cuda.synchronize()
t = s.elapsed_time(e) / 100 * 1e3
print(json.dumps({"kernel": "neumf_embed_fwd_kernel", "us": round(t, 2), "gathered_GBps": round(B * 4 * D * 4 / t / 1e3, 1),
                  "frac_of_8TBps_read_only": round(B * 4 * D * 4 / t / 1e3 / 8000, 3), "all_bytes_GBps": round(B * 6 * D * 4 / t / 1e3, 1)}))
