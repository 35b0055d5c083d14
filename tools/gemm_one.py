#!/usr/bin/env python3
"""One BERT GEMM shape, a few launches - the target of rocprofv3 --pmc passes.
   python tools/gemm_one.py --shape ff2 --variant 5 [--iters 5]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip
ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="ff2")
ap.add_argument("--variant", type=int, default=5)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--rows", type=int, default=43008)
ap.add_argument("--dtype", default="fp16")
a = ap.parse_args()
lib = hip.load(); d = torch.device("cuda:0")
dt = hip.PRECISION[a.dtype]; td = hip.TORCH_DTYPE[dt]
N, K, act, res = {"qkv": (2304, 768, hip.ACT_NONE, False), "ao": (768, 768, hip.ACT_NONE, True), "ff1": (3072, 768, hip.ACT_GELU, False),
                  "ff2": (768, 3072, hip.ACT_NONE, True), "sq4k": (4096, 4096, hip.ACT_NONE, False),
                  "ff1n": (3072, 768, hip.ACT_NONE, False), "aon": (768, 768, hip.ACT_NONE, False), "ff2n": (768, 3072, hip.ACT_NONE, False)}[a.shape]
M = 4096 if a.shape == "sq4k" else a.rows
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).to(td).to(d); W = (torch.randn(N, K, generator=g) * 0.05).to(td).to(d)
bias = torch.randn(N, generator=g).to(d); R = torch.randn(M, N, generator=g).to(td).to(d) if res else None
C = torch.empty(M, N, dtype=torch.float32 if res else td, device=d)
lib.ruart_gemm_set_tile_order(8); assert lib.ruart_gemm_set_variant(a.variant) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(a.iters + 2):
    if i == 2: e0.record()
    rc = lib.ruart_gemm_16_nt(hip.ptr(A), K, hip.ptr(W), K, hip.ptr(bias), hip.ptr(R), N, dt, hip.ptr(C), N, hip.DT_F32 if res else dt, M, N, K,
                              act, dt, hip.stream_ptr())
    assert rc == 0
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / a.iters
print("%s variant %d: %.1f us  %.0f TF/s" % (a.shape, a.variant, us, 2.0 * M * N * K / us / 1e6))
