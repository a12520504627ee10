"""Experiment: what does the per-step staging launch cost the replayed step?  Two sets of static input buffers and a capture of the step
per set; the NEXT batch is copied into the other set on a side stream while the current graph runs (same batches, same lags)."""
import os, sys, time, json, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
neumf = importlib.import_module("binary-recommendation_amd.neumf")
dev = torch.device("cuda:0"); B, U, I = 65536, 1_000_000, 100_000
cfg = neumf.NeuMFConfig(variant="A", dim=64, optimizer="adam_dense", seed=1, dense_impl="deferred")
eng = neumf.NeuMFEngine(cfg, U, I, dev, B, init_seed=1)
g = torch.Generator(device=dev).manual_seed(3)
NB = 25
batches = [(torch.randint(0, U, (B,), device=dev, dtype=torch.int32, generator=g), torch.randint(0, I, (B,), device=dev, dtype=torch.int32, generator=g),
            (torch.rand(B, device=dev, generator=g) < 0.25).float()) for _ in range(NB)]
for s in range(NB): eng.train_step(*batches[s])
graphs, bufs = [], []
for k in range(2):
    eng.in_users = None
    eng.enable_graph(B)
    graphs.append(eng._graph); bufs.append((eng.in_users, eng.in_items, eng.in_labels))
main = torch.cuda.current_stream(dev); side = torch.cuda.Stream(device=dev)
ev_done = [torch.cuda.Event(), torch.cuda.Event()]; ev_staged = [torch.cuda.Event(), torch.cuda.Event()]
for e in ev_done: e.record(main)
state = {"s": 0}
def stage(k, batch):
    side.wait_event(ev_done[k])
    with torch.cuda.stream(side):
        for d, srct in zip(bufs[k], batch): d.copy_(srct, non_blocking=True)
        ev_staged[k].record(side)
def run_staged(n):
    eng._graph = graphs[0]; eng.in_users, eng.in_items, eng.in_labels = bufs[0]
    for _ in range(n):
        eng.train_step(*batches[state["s"] % NB]); state["s"] += 1
def run_ahead(n):
    k = 0
    stage(k, batches[state["s"] % NB])
    for _ in range(n):
        stage(1 - k, batches[(state["s"] + 1) % NB])          # next batch into the other set, beside this step's graph
        main.wait_event(ev_staged[k])
        eng._graph = graphs[k]; eng.in_users, eng.in_items, eng.in_labels = bufs[k]
        eng.train_step(*bufs[k])                                # pointers equal the static buffers: no staging launch
        ev_done[k].record(main)
        state["s"] += 1; k = 1 - k
out = {}
for name, fn in (("stage_launch_per_step", run_staged), ("staged_one_step_ahead_on_a_side_stream", run_ahead), ("stage_launch_per_step_2", run_staged), ("ahead_2", run_ahead)):
    fn(10); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); fn(40); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 40)
    out[name] = best * 1e3
eng.check_ids()
print(json.dumps(out))
