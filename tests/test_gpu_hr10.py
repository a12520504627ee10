"""-m gpu: HR@10 parity (BASELINE metric) on a scaled-down ML-1M-shaped set: the GPU path and the CPU
oracle train NeuMF-A with the same data, order and dropout masks; |dHR@10| <= 0.002.  The full-size run
(6 040 x 3 706, 20 epochs, batch 50 000: trainers/NFC_plain.py:128-134,165) is tools/hr10_parity.py."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_hr10_parity_small(dev):
    import hr10_parity
    out = hr10_parity.run(n_users=800, n_items=500, n_pos=40000, epochs=4, batch=8192, dim=10, log=lambda m: None)
    assert out["abs_delta_hitRate@10"] <= 0.002, out
    assert out["loss_rel_diff_last_epoch"] <= 1e-4, out
    assert out["users_with_identical_top10_sets"] >= 0.97, out
    assert 0.0 < out["gpu"]["hitRate@10"] <= 1.0
