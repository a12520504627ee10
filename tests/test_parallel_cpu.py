"""-m "not gpu": the row-sharded exchange on CPU with gloo, world_size 2.
The HIP kernels are absent here, so the owner-side compute is the ORACLE (test injection only);
what is under test is the product's index plumbing + collectives in parallel.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, U, D, B, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from importlib import import_module
    from oracle import binrec_oracle as O
    par = import_module("binary-recommendation_amd.parallel")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = par.DistCtx()
        rng = np.random.default_rng(42)
        table = rng.normal(size=(U, D)).astype(np.float32)          # the GLOBAL table (same on all ranks)
        shard = table[rank::world]                                   # what this rank owns
        assert shard.shape[0] == par.shard_rows(U, rank, world)
        ids_all = rng.integers(0, U, size=(world, B))
        ids_all[:, :5] = 3                                           # duplicates across ranks
        ids = torch.from_numpy(ids_all[rank])
        x = par.ShardExchange(ctx).plan(ids)
        x2 = par.ShardExchange(ctx).plan(torch.from_numpy(ids_all[rank][::-1].copy()))
        x.exchange_counts(x2)
        assert sum(x.send_counts) == B and x.n_recv == sum(x.recv_counts)
        served = x.send_ids()
        rows = torch.from_numpy(O.gather_rows(shard, served.numpy()))        # owner-side lookup (oracle stands in)
        got_bucket = x.return_rows(rows)
        got = got_bucket[x.inv]                                              # bucket order -> batch order
        assert np.array_equal(got.numpy(), table[ids_all[rank]]), "sharded lookup != global lookup"
        # second stream planned in the same count exchange
        served2 = x2.send_ids()
        got2 = x2.return_rows(torch.from_numpy(O.gather_rows(shard, served2.numpy())))[x2.inv]
        assert np.array_equal(got2.numpy(), table[ids_all[rank][::-1]])
        # all-to-all #3: row grads to owners, then the owner's dedup == the global dedup restricted to its rows
        g_all = rng.normal(size=(world, B, D)).astype(np.float32)
        g_bucket = torch.from_numpy(g_all[rank])[x.order]
        g_recv = x.send_row_grads(g_bucket)
        dense_local = O.scatter_add_dense(shard.shape[0], served.numpy(), g_recv.numpy(), dt=np.float64)
        dense_global = O.scatter_add_dense(U, ids_all.reshape(-1), g_all.reshape(-1, D), dt=np.float64)
        np.testing.assert_allclose(dense_local, dense_global[rank::world], rtol=1e-12, atol=1e-12)
        # dense all-reduce
        t = torch.full((7,), float(rank + 1), dtype=torch.float64)
        ctx.all_reduce_sum(t)
        assert torch.all(t == sum(range(1, world + 1)))
        ag = ctx.all_gather_rows(torch.full((2, 3), float(rank)))
        assert ag.shape == (2 * world, 3) and float(ag[-1, 0]) == world - 1
        out_q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        out_q.put((rank, "FAIL: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_shard_exchange_gloo_world2(world):
    """world 8 = the node the row-sharding is designed for: owner = id mod 8, eight-way count exchange and splits"""
    port = _free_port()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_worker, args=(r, world, port, 101, 16, 64, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for p in procs:      # never leave a child behind: the interpreter would wait for it at exit
        if p.is_alive():
            p.kill()
    assert all(r[1] == "ok" for r in res), res


def test_shard_rows_partition():
    from importlib import import_module
    sys.path.insert(0, ROOT)
    par = import_module("binary-recommendation_amd.parallel")
    for total in (0, 1, 7, 8, 9, 1000003):
        for w in (1, 2, 8):
            assert sum(par.shard_rows(total, r, w) for r in range(w)) == total
