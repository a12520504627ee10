"""Per-GPU cost of the row-sharded NeuMF step on a 1-rank group (no bytes cross a link): (a) collectives short-cut to local copies,
(b) every collective issued through RCCL, eager, (c) the same step with its RCCL collectives captured into ONE hipGraph per step
(ShardedNeuMFEngine.enable_graph), next to the fused single-GPU engine's figure from bench.py.  Shows what the exchange bookkeeping and
the per-call latency of the collectives cost before any communication."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
from importlib import import_module
neumf = import_module("binary-recommendation_amd.neumf"); par = import_module("binary-recommendation_amd.parallel")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
B, U, I = 65536, 1_000_000, 100_000
zipf = "--zipf" in sys.argv
g = torch.Generator(device=dev).manual_seed(3)
def ids(N):
    if not zipf:
        return torch.randint(0, N, (B,), device=dev, dtype=torch.int32, generator=g)
    u = torch.rand(B, device=dev, generator=g, dtype=torch.float64)
    return (((N ** (-0.05) - 1) * u + 1) ** (1 / -0.05)).long().clamp_(1, N).sub_(1).int()
batches = [(ids(U), ids(I), (torch.rand(B, device=dev, generator=g) < 0.25).float()) for _ in range(16)]
out = {"workload": f"row-sharded NeuMF-A dim 64, {U} x {I}, batch {B}, {'Zipf(1.05)' if zipf else 'uniform'} ids, 1-rank nccl group, per-replica BatchNorm, 16 batches cycled (steady state)"}
def measure(eng, tag):
    def run(n):
        for s in range(n):
            u, i, y = batches[s % 16]; eng.train_step(u, i, y)
    run(21); torch.cuda.synchronize()      # one pass over the batch cycle first: tables in steady state
    best = None
    for _ in range(3):
        t0 = time.perf_counter(); run(30); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
        best = dt if best is None else min(best, dt)
    t0 = time.perf_counter(); run(30); host = (time.perf_counter() - t0) / 30; torch.cuda.synchronize()
    eng.check_ids()
    out[tag] = {"ms_per_step": best * 1e3, "host_enqueue_ms_per_step": host * 1e3}
cfg = neumf.NeuMFConfig(variant="A", dim=64, seed=1, sync_bn=False)
mk = lambda force: par.make_sharded_engine(neumf.NeuMFEngine)(cfg, U, I, dev, B, par.DistCtx(force_collectives=force), init_seed=1)
e = mk(False); measure(e, "local_copies_eager"); del e; torch.cuda.empty_cache()
if "--local-only" in sys.argv:      # (for rocprofv3: a normal interpreter exit, no capture)
    print(json.dumps(out), flush=True)
    dist.destroy_process_group()
    sys.exit(0)
e = mk(True); measure(e, "rccl_eager"); del e; torch.cuda.empty_cache()
e = mk(True); e.enable_graph(B); measure(e, "rccl_hipgraph")
out["rccl_hipgraph"].update({"graph_active": e.graph_active, "refused": e._sgraph["refused"]})
print(json.dumps(out), flush=True)
if os.environ.get("BR_PROFILE"):
    import cProfile, pstats
    e.disable_graph()
    pr = cProfile.Profile(); pr.enable()
    for s in range(20):
        e.train_step(*batches[s % 16])
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
os._exit(0)      # (no process-group teardown behind a capture that holds RCCL nodes: it did not return on ROCm 7.2 / RCCL 2.26)
