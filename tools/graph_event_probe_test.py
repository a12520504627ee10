import torch
dev = torch.device("cuda:0")
x = torch.randn(1 << 26, device=dev); y = torch.empty_like(x)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    y.copy_(x); y.mul_(2.0)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
try:
    with torch.cuda.graph(g):
        e0.record()
        y.copy_(x)
        e1.record()
        y.mul_(2.0)
        e2.record()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("in-graph events ok:", e0.elapsed_time(e1), e1.elapsed_time(e2))
except Exception as ex:
    print("in-graph events FAILED:", repr(ex)[:300])
