"""bit-identity of the fused lookup's forms (BR_LOOKUP_PAIRS, read once per process): prints a digest of x0 / dot / tables after a few
steady-state steps.  Run once per form and compare the lines."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from importlib import import_module
neumf = import_module("binary-recommendation_amd.neumf")
dev = torch.device("cuda:0")
U, I, B = 200_000, 30_000, 65536
eng = neumf.NeuMFEngine(neumf.NeuMFConfig("A", dim=64, seed=7), U, I, dev, max_batch=B)
g = torch.Generator().manual_seed(11)
bat = [(torch.randint(0, U, (B,), generator=g).int().to(dev), torch.randint(0, I, (B,), generator=g).int().to(dev), (torch.rand(B, generator=g) < 0.25).float().to(dev)) for _ in range(12)]
for k in range(40):
    eng.train_step(*bat[k % 12])
torch.cuda.synchronize()
h = hashlib.sha256()
for t in (eng.x0[:B], eng.dot[:B], eng.g_user[:B], eng.g_item[:B], eng.logit[:B]):
    h.update(t.detach().cpu().numpy().tobytes())
eng.flush()
for k in ("user", "item"):
    h.update(eng.fused[k].cpu().numpy().tobytes())
print("digest", os.environ.get("BR_LOOKUP_PAIRS", "default"), h.hexdigest()[:32])
