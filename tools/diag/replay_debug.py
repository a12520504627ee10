"""debug: fast vs exact deferred replay, step by step (first divergence)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from importlib import import_module
neumf = import_module("binary-recommendation_amd.neumf")
dev = torch.device("cuda:0")
B, U, I, dim = 48, 1500, 400, 16
def mk(replay, impl="deferred"):
    cfg = neumf.NeuMFConfig(variant="A", dim=dim, optimizer="adam_dense", seed=5, dense_impl=impl, replay=replay)
    return neumf.NeuMFEngine(cfg, U, I, dev, max_batch=B, init_seed=3)
ex, fa, sw = mk("exact"), mk("fast"), mk("exact", "sweep")
sw._alloc_step_state(sw.step_struct)
print("state tail fast:", fa.step_state[4152 // 4: 4152 // 4 + 4].tolist(), fa.step_state.view(torch.float32)[4152 // 4 + 2: 4152 // 4 + 8].tolist())
print("pow2[0..3]:", fa.step_state.view(torch.float32)[4152 // 4 + 4 + 1024: 4152 // 4 + 4 + 1028].tolist())
rng = np.random.default_rng(5)
td = lambda a, dt: torch.from_numpy(a).to(dev).to(dt)
init = {k: ex.fused[k].clone() for k in ("user", "item")}
for step in range(40):
    n = B
    uu, ii = rng.integers(0, U, n), rng.integers(0, I, n)
    yy = (rng.random(n) < 0.3).astype(np.float32)
    for e in (ex, fa, sw):
        e.train_step(td(uu, torch.int32), td(ii, torch.int32), td(yy, torch.float32))
    torch.cuda.synchronize()
    # raw (unflushed) state comparison of the two deferred engines
    for k in ("user", "item"):
        for nm, a, b in (("th", ex.fused[k], fa.fused[k]), ("m", ex.fused_m[k], fa.fused_m[k]), ("v", ex.fused_v[k], fa.fused_v[k])):
            d = (a.double() - b.double()).abs()
            rel = d / (a.double().abs() + 1e-30)
            w = int(torch.argmax(d).item())
            r, c = divmod(w, a.shape[1])
            if nm == "th":
                mv = (a.double() - init[k].double()).abs().flatten()[w].item()
                print(f"step {step+1} {k} {nm}: max abs diff {d.max().item():.3e} at row {r} col {c} (move {mv:.3e}, last ex {int(ex.last[k][r])} fa {int(fa.last[k][r])})")
            else:
                print(f"step {step+1} {k} {nm}: max rel diff {rel.max().item():.3e}")
    assert torch.equal(ex.last["user"], fa.last["user"])
    if step in (4, 20):
        ex.flush(); fa.flush()
