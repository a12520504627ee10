"""bench.py's batch-cycle walk (no GPU): with a multi-step graph on the engine, every run of S steps that starts on a group boundary of the
cycle is ONE train_steps call, everything else a single step; the cursor carries over between calls, so the legs of a run see the cycle in
order and the number of steps is exact (the contract times EXACTLY --steps steps)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


class _FakeEngine:
    def __init__(self, S, B):
        self.calls = []
        self._graph_multi = {"S": S, "batch": B}

    def train_step(self, u, i, y, row0=0, batch_total=None):
        self.calls.append((1, int(u[0])))

    def train_steps(self, u, i, y):
        self.calls.append((self._graph_multi["S"], int(u[0])))


def test_run_steps_groups_on_cycle_boundaries_and_counts_every_step():
    B, S = 8, 4
    batches = [(torch.full((B,), k), torch.zeros(B), torch.zeros(B)) for k in range(10)]       # a cycle S does not divide
    groups = bench.make_groups(batches, S)
    assert len(groups) == 2 and groups[1][0].shape[0] == S * B and int(groups[1][0][B]) == 5
    e = _FakeEngine(S, B)
    bench.run_steps(e, batches, 3, 0, B)
    bench.run_steps(e, batches, 13, 0, B)
    assert e.calls == [(1, 0), (1, 1), (1, 2), (1, 3), (4, 4), (1, 8), (1, 9), (4, 0), (1, 4), (1, 5)]
    assert sum(n for n, _ in e.calls) == 16
    e2 = _FakeEngine(S, B)
    e2._graph_multi = None                                   # no multi-step graph: single steps only
    bench.run_steps(e2, batches, 5, 0, B)
    assert [n for n, _ in e2.calls] == [1] * 5
    e3 = _FakeEngine(S, B)
    bench.run_steps(e3, batches, 8, 3, 2 * B)                # a data-parallel slice (row0 != 0): never grouped
    assert [n for n, _ in e3.calls] == [1] * 8
