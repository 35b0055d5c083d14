#!/usr/bin/env python3
"""The encoder's four projections at the bench's token count: plain f16 kernel vs the f16 + fp8-correction kernel
(ruart_gemm_16c_nt), per-shape time and the ratio.   python tools/gemm_corr_bench.py [--rows 42880] [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip  # noqa: E402
from ruart_amd.bert import split_f16c  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=42880)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
lib = hip.load()
d = torch.device("cuda:0")
M = (a.rows + 255) // 256 * 256
shapes = [("qkv", 2304, 768, hip.ACT_NONE, False), ("ao", 768, 768, hip.ACT_NONE, True), ("ff1", 3072, 768, hip.ACT_GELU, False),
          ("ff2", 768, 3072, hip.ACT_NONE, True)]
g = torch.Generator(device="cpu").manual_seed(0)


def timed(run):
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / a.iters


tot = [0.0, 0.0]
for name, N, K, act, res in shapes:
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.03
    A16, A8 = [t.to(d) for t in split_f16c(A)]
    hi = W.half().float()
    W16 = W.half().to(d)
    W8 = torch.cat([hi * 2.0 ** hip.f16c_shifts()[2], (W - hi) * 2.0 ** hip.f16c_shifts()[3]], 1).clamp_(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).to(d)
    bias = torch.randn(N, generator=g).to(d)
    R16 = torch.randn(M, N, generator=g).half().to(d) if res else None
    R32 = R16.float() if res else None
    Cp = torch.empty(M, N, dtype=torch.float32 if res else torch.float16, device=d)
    gelu = act == hip.ACT_GELU
    Cc = torch.empty(M, N, dtype=torch.float16 if gelu else torch.float32, device=d)
    C8 = torch.empty(M, 2 * N, dtype=torch.uint8, device=d) if gelu else None

    def plain():
        assert lib.ruart_gemm_16_nt(hip.ptr(A16), K, hip.ptr(W16), K, hip.ptr(bias), hip.ptr(R16), N, hip.DT_F16, hip.ptr(Cp), N,
                                    hip.DT_F32 if res else hip.DT_F16, M, N, K, act, hip.DT_F16, hip.stream_ptr()) == 0

    def corr():
        assert lib.ruart_gemm_16c_nt(hip.ptr(A16), hip.ptr(A8), K, hip.ptr(W16), hip.ptr(W8), K, hip.ptr(bias), hip.ptr(R32), N, hip.ptr(Cc), N,
                                     hip.ptr(C8), M, N, K, act, hip.stream_ptr()) == 0

    tp, tc = timed(plain), timed(corr)
    fl = 2.0 * M * N * K
    tot[0] += tp
    tot[1] += tc
    print("%-4s plain f16 %7.1f us %5.0f TF/s | f16+fp8 %7.1f us %5.0f TF/s-equivalent (%.0f TF/s of MFMA work) | ratio %.2f"
          % (name, tp, fl / tp / 1e6, tc, fl / tc / 1e6, 2 * fl / tc / 1e6, tc / tp), flush=True)
print("layer: plain %.1f us, f16+fp8 %.1f us, ratio %.2f" % (tot[0], tot[1], tot[1] / tot[0]))
