#!/usr/bin/env python3
"""Micro-benchmark of the BERT projection GEMMs (the roofline kernel) at the bench's token count.
   python tools/gemm_bench.py [--rows 42880] [--dtype fp16] [--orders 0,4,8,16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=42880)
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--orders", default="0,4,8,16")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--variants", default="3,5")
a = ap.parse_args()
lib = hip.load()
d = torch.device("cuda:0")
dt = hip.PRECISION[a.dtype]
td = hip.TORCH_DTYPE[dt]
M = (a.rows + 255) // 256 * 256
shapes = [("qkv", 2304, 768, hip.ACT_NONE, False), ("ao", 768, 768, hip.ACT_NONE, True), ("ff1", 3072, 768, hip.ACT_GELU, False),
          ("ff2", 768, 3072, hip.ACT_NONE, True)]
g = torch.Generator(device="cpu").manual_seed(0)
import itertools
for variant, order in itertools.product([int(x) for x in a.variants.split(",")], [int(x) for x in a.orders.split(",")]):
    assert lib.ruart_gemm_set_tile_order(order) == 0
    assert lib.ruart_gemm_set_variant(variant) == 0
    tot_t, tot_f = 0.0, 0.0
    line = []
    for name, N, K, act, res in shapes:
        A = torch.randn(M, K, generator=g).to(td).to(d)
        W = (torch.randn(N, K, generator=g) * 0.05).to(td).to(d)
        bias = torch.randn(N, generator=g).to(d)
        R = torch.randn(M, N, generator=g).to(td).to(d) if res else None
        C = torch.empty(M, N, dtype=torch.float32 if res else td, device=d)
        out_dt = hip.DT_F32 if res else dt

        def run():
            rc = lib.ruart_gemm_16_nt(hip.ptr(A), K, hip.ptr(W), K, hip.ptr(bias), hip.ptr(R), N, dt, hip.ptr(C), N, out_dt, M, N, K, act,
                                      dt, hip.stream_ptr())
            assert rc == 0
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        fl = 2.0 * M * N * K
        tot_t += us
        tot_f += fl
        line.append("%s %6.1f us %6.0f TF" % (name, us, fl / us / 1e6))
    print("variant %d order %2d | %s | layer total %7.1f us  %6.0f TF/s" % (variant, order, " | ".join(line), tot_t, tot_f / tot_t / 1e6), flush=True)
