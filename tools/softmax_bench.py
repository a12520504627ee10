"""In-batch softmax kernels alone (brInBatchSoftmaxLse / Grad) against the fp32 MFMA peak: B x B x S, 3 algorithmic GEMMs."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from importlib import import_module
ops = import_module("binary-recommendation_amd.ops")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_us, MFMA_F32_PEAK_TFLOPS
dev = torch.device("cuda:0")
res = {}
for (Bq, Bc, S) in ((8192, 8192, 64), (8192, 8192, 50), (8192, 8192, 100), (8192, 8192, 128), (1024, 65536, 64), (65536, 1024, 64), (2048, 2048, 64)):
    q, c = torch.randn(Bq, S, device=dev) * 0.3, torch.randn(Bc, S, device=dev) * 0.3
    ids_c = torch.randint(0, 100000, (Bc,), device=dev, dtype=torch.int32)
    ids_q = ids_c[:Bq].contiguous() if Bq <= Bc else torch.randint(0, 100000, (Bq,), device=dev, dtype=torch.int32)
    lse, slots = torch.empty(Bq, device=dev), torch.zeros(64, dtype=torch.float64, device=dev)
    dq, dc = torch.empty_like(q), torch.empty_like(c)
    t_lse, _ = time_us(lambda: ops.inbatch_softmax_lse(q, c, ids_q, ids_c, 0, lse, slots), reps=10)
    t_q, _ = time_us(lambda: ops.inbatch_softmax_grad(q, c, ids_q, ids_c, 0, lse, dq, None), reps=10)
    t_c, _ = time_us(lambda: ops.inbatch_softmax_grad(q, c, ids_q, ids_c, 0, lse, None, dc), reps=10)
    t_f, _ = time_us(lambda: ops.inbatch_softmax_lse_grad_q(q, c, ids_q, ids_c, 0, lse, slots, dq), reps=10)
    fl = 2.0 * Bq * Bc * S
    res[f"{Bq}x{Bc}x{S}"] = {"lse_us": round(t_lse, 1), "dq_us": round(t_q, 1), "dc_us": round(t_c, 1), "lse+dq_one_sweep_us": round(t_f, 1),
                             "algorithmic_frac_3gemm_one_sweep": round(3 * 2.0 * Bq * Bc * S / (t_f + t_c) * 1e-6 / MFMA_F32_PEAK_TFLOPS, 3),
                             "lse_frac": round(fl / t_lse * 1e-6 / MFMA_F32_PEAK_TFLOPS, 3), "dq_frac_2gemm": round(2 * fl / t_q * 1e-6 / MFMA_F32_PEAK_TFLOPS, 3),
                             "dc_frac_2gemm": round(2 * fl / t_c * 1e-6 / MFMA_F32_PEAK_TFLOPS, 3),
                             "algorithmic_frac_3gemm": round(3 * fl / (t_lse + t_q + t_c) * 1e-6 / MFMA_F32_PEAK_TFLOPS, 3)}
print(json.dumps(res, indent=1))
