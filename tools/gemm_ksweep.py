#!/usr/bin/env python3
"""K-sweep of the 16-bit GEMM: separates the main-loop rate from per-workgroup fixed cost (prologue + epilogue)."""
import os, sys, itertools
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip
lib = hip.load()
d = torch.device("cuda:0")
dt, td = hip.DT_F16, torch.float16
M = 42880 // 256 * 256
for variant, N, K in itertools.product([0, 3], [768, 3072], [768, 3072]):
    lib.ruart_gemm_set_variant(variant)
    A = torch.randn(M, K, device=d).to(td)
    W = (torch.randn(N, K, device=d) * 0.05).to(td)
    C = torch.empty(M, N, dtype=td, device=d)
    def run():
        assert lib.ruart_gemm_16_nt(hip.ptr(A), K, hip.ptr(W), K, None, None, N, dt, hip.ptr(C), N, dt, M, N, K, 0, dt, hip.stream_ptr()) == 0
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print("variant %d N %4d K %5d: %7.1f us  %6.0f TF/s" % (variant, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)
