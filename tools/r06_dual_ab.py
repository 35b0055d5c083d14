#!/usr/bin/env python3
"""Round 6: the 256 x 128 two-workgroups-per-CU form of the fp16c projections (ruart_gemm_16c_set_dual) against the 256 x 256 form, in ONE
process on the same buffers, interleaved rounds: bit equality of the outputs, then launch times (us) - QKV and the intermediate dense,
plain and LayerNorm-folded.   python tools/r06_dual_ab.py [--rows 42752] [--rounds 10]"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip
from ruart_amd.bert import split_f16c

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=42752)
ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
L = hip.load()
d = torch.device("cuda:0")
M = (a.rows + 255) // 256 * 256
sa = hip.f16c_shifts()
g = torch.Generator().manual_seed(0)
for name, N, K, act in [("qkv", 2304, 768, hip.ACT_NONE), ("ff1", 3072, 768, hip.ACT_GELU)]:
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.03
    A16, A8 = [t.to(d) for t in split_f16c(A)]
    hi = W.half().float()
    W16 = W.half().to(d)
    W8 = torch.cat([hi * 2.0 ** sa[2], (W - hi) * 2.0 ** sa[3]], 1).clamp_(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).to(d)
    bias = torch.randn(N, generator=g).to(d)
    colc = torch.randn(N, generator=g).to(d)
    part = torch.zeros(M, 4, 2)
    part[:, :3, 0] = torch.randn(M, 3, generator=g) * 3
    part[:, :3, 1] = torch.rand(M, 3, generator=g) * 300 + 100
    part = part.to(d)
    gelu = act == hip.ACT_GELU
    kind = 2 if gelu else 0
    for fold in (False, True):
        outs = []
        for dual in (0, 1):
            C = torch.full((M, N), 7.0, dtype=torch.float16 if gelu else torch.float32, device=d)
            C8 = torch.full((M, 2 * N), 9, dtype=torch.uint8, device=d) if gelu else None
            outs.append((C, C8))

        def run(dual):
            C, C8 = outs[dual]
            L.ruart_gemm_16c_set_dual(dual)
            if fold:
                rc = L.ruart_gemm_16c_nt_fold(hip.ptr(A16), hip.ptr(A8), K, hip.ptr(W16), hip.ptr(W8), K, hip.ptr(bias), kind, hip.ptr(part), 3, hip.ptr(colc), 0.5,
                                              None, 0, None, 0, None, None, hip.ptr(C), N, None, hip.ptr(C8), None, M, N, K, K, 1e-12, hip.stream_ptr())
            else:
                rc = L.ruart_gemm_16c_nt(hip.ptr(A16), hip.ptr(A8), K, hip.ptr(W16), hip.ptr(W8), K, hip.ptr(bias), None, 0, hip.ptr(C), N, hip.ptr(C8), M, N, K, act,
                                         hip.stream_ptr())
            assert rc == 0, rc

        for dual in (0, 1):
            for _ in range(3):
                run(dual)
        torch.cuda.synchronize()
        same = torch.equal(outs[0][0], outs[1][0]) and (not gelu or torch.equal(outs[0][1], outs[1][1]))
        finite = bool(torch.isfinite(outs[1][0].float()).all())
        times = {0: [], 1: []}
        for r in range(a.rounds):
            for dual in ((0, 1) if r % 2 == 0 else (1, 0)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    run(dual)
                e1.record()
                torch.cuda.synchronize()
                times[dual].append(e0.elapsed_time(e1) * 1e3 / a.iters)
        print("%-4s %-7s bit-equal %s finite %s   256x256: median %.1f min %.1f   256x128 dual: median %.1f min %.1f" % (
            name, "folded" if fold else "plain", same, finite, float(np.median(times[0])), min(times[0]), float(np.median(times[1])), min(times[1])), flush=True)
L.ruart_gemm_16c_set_dual(0)
