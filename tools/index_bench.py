"""Time the dedup index build (brRowIndexBuildPair) alone, back to back on one stream."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from importlib import import_module
ops = import_module("binary-recommendation_amd.ops")
lib = import_module("binary-recommendation_amd._lib").load()
dev = torch.device("cuda:0")
res = {}
for n, ru, ri in ((65536, 1_000_000, 100_000), (8192, 1_000_000, 100_000), (131072, 1_000_000, 100_000), (262144, 1_000_000, 100_000)):
    u = torch.randint(0, ru, (n,), device=dev, dtype=torch.int32); i = torch.randint(0, ri, (n,), device=dev, dtype=torch.int32)
    a, b = ops.RowIndex(n, torch.int32, dev), ops.RowIndex(n, torch.int32, dev)
    def pair():
        lib.brRowIndexBuildPair(u.data_ptr(), ru, a.sorted_ids.data_ptr(), a.sorted_pos.data_ptr(), a.ws.data_ptr(), a.ws_bytes,
                                i.data_ptr(), ri, b.sorted_ids.data_ptr(), b.sorted_pos.data_ptr(), b.ws.data_ptr(), b.ws_bytes, ops.I32, n, ops._stream())
    for _ in range(3): pair()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): pair()
    e.record(); torch.cuda.synchronize()
    res[f"pair n={n}"] = round(s.elapsed_time(e) / 20 * 1e3, 1)
print(json.dumps(res))
