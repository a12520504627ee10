"""The tower's GEMMs as fp32 products on the bf16 matrix pipe ("bf16x6"; csrc/dense.h) against float64.

brDenseForward / brDenseBackward (trainers/NFC_plain.py:137-152: the Dense layers of the tower) run every fp32 product as six bf16 MFMAs on
three-piece operands by default; BR_MLP_MATH=f32 keeps the exact-fp32 MFMA kernels.  The switch is read once per process, so the fp32 mode
runs in a child process.  Asserted per shape, on errors relative to the sum of |terms| of each output element:
  * bf16x6 stays within 1.5e-6 (a 128-term fp32 accumulation is allowed ~2.5e-7 by the parity bars; six rounded products per 32-deep block);
  * bf16x6 is not worse than 2 x the fp32-MFMA kernels' error on the same inputs (+ a 1e-8 floor): "at least as accurate" is a measured claim.
Shapes: the two benchmarked layers, an odd number of n-tiles (half block in the backward), K and N that are not multiples of 4 or 16,
a batch with a ragged last tile, and 128 x 128 (no piece image fits the backward's LDS: both modes run the fp32 kernel there).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(128, 100, 4096), (100, 50, 4096), (50, 10, 1000), (128, 64, 2048), (37, 23, 333), (128, 128, 1024), (64, 112, 640)]

CHILD = r"""
import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from importlib import import_module
ops = import_module("binary-recommendation_amd.ops")
dev = torch.device("cuda:0")
out = {}
for (K, N, B) in json.loads(sys.argv[1]):
    rng = np.random.default_rng(1000 * K + N)
    ldk, ldn = (K + 3) & ~3, (N + 3) & ~3
    x = rng.standard_normal((B, K)).astype(np.float32); W = (0.1 * rng.standard_normal((K, N))).astype(np.float32)
    b = (0.1 * rng.standard_normal(N)).astype(np.float32); gy = (1e-3 * rng.standard_normal((B, N))).astype(np.float32)
    xp = torch.zeros(B, ldk, device=dev); xp[:, :K] = torch.from_numpy(x).to(dev)
    yp = torch.zeros(B, ldn, device=dev); gyp = torch.zeros(B, ldn, device=dev); gyp[:, :N] = torch.from_numpy(gy).to(dev)
    Wd, bd = torch.from_numpy(W).to(dev), torch.from_numpy(b).to(dev)
    xv, yv, gyv = xp[:, :K], yp[:, :N], gyp[:, :N]
    ops.dense_forward(xv, Wd, bd, yv, "linear")
    x64, W64 = x.astype(np.float64), W.astype(np.float64)
    z = x64 @ W64 + b
    mag = np.abs(x64) @ np.abs(W64) + np.abs(b)
    e_fwd = float(np.max(np.abs(yv.cpu().numpy().astype(np.float64) - z) / mag))
    # backward with a linear activation: dz = gy
    ns = ops.dense_backward_slabs(B, K, N)
    slabs = torch.zeros(ns * (K * N + N), device=dev)
    gxp = torch.zeros(B, ldk, device=dev)
    ops.dense_backward(gyv, yv, xv, Wd, "linear", slabs, ns, gx=gxp[:, :K])
    red = torch.empty(K * N + N, device=dev)
    ops.reduce_slabs(slabs, ns, K * N + N, red)
    g64 = gy.astype(np.float64)
    gx_ref, dW_ref, db_ref = g64 @ W64.T, x64.T @ g64, g64.sum(0)
    e_gx = float(np.max(np.abs(gxp[:, :K].cpu().numpy().astype(np.float64) - gx_ref) / (np.abs(g64) @ np.abs(W64).T)))
    r = red.cpu().numpy().astype(np.float64)
    e_dw = float(np.max(np.abs(r[:K * N].reshape(K, N) - dW_ref) / (np.abs(x64).T @ np.abs(g64))))
    e_db = float(np.max(np.abs(r[K * N:] - db_ref) / np.abs(g64).sum(0)))
    out[f"{K}x{N}x{B}"] = {"fwd": e_fwd, "gx": e_gx, "dW": e_dw, "db": e_db}
print("RESULT " + json.dumps(out))
"""


def _run(mode):
    env = dict(os.environ)
    env.pop("BR_MLP_MATH", None)
    if mode == "f32":
        env["BR_MLP_MATH"] = "f32"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", CHILD, json.dumps(SHAPES)], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_bf16x6_products_against_float64_and_the_fp32_mfma_kernels():
    assert torch.cuda.is_available()
    emu, f32 = _run("bf16x6"), _run("f32")
    report = {}
    for key in emu:
        for what in ("fwd", "gx", "dW", "db"):
            e, f = emu[key][what], f32[key][what]
            report[f"{key} {what}"] = (e, f)
            assert e <= 1.5e-6, (key, what, e)
            assert e <= 2.0 * f + 1e-8, (key, what, e, f)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "mlp_math_errors.json"), "w") as fh:
            json.dump({"note": "max |got - float64| / sum |terms| per output element: [bf16x6, fp32 MFMA]", "cases": report}, fh, indent=1)
