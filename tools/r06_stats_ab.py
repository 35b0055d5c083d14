#!/usr/bin/env python3
"""In-process A/B of library builds on the four FOLDED projections of the fp16c pass (kinds 0 / 2 with row partials, kind 3 with a
LayerNorm-ed residual): outputs compared bit for bit against the first library, launch times interleaved.
    python tools/r06_stats_ab.py build/libruart_hip_late.so ruart_amd/libruart_hip.so [--rows 42752] [--rounds 10]"""
import argparse, ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip
from ruart_amd.bert import split_f16c

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--rows", type=int, default=42752)
ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--dual", type=int, default=0)
a = ap.parse_args()
hip.load()
libs = []
for p in a.libs:
    L = ctypes.CDLL(os.path.abspath(p))
    f = L.ruart_gemm_16c_nt_fold
    f.restype, f.argtypes = hip._SIGNATURES["ruart_gemm_16c_nt_fold"]
    if a.dual:
        L.ruart_gemm_16c_set_dual(1)
    libs.append((os.path.basename(p).replace("libruart_hip_", "").replace(".so", ""), L))
d = torch.device("cuda:0")
M = (a.rows + 255) // 256 * 256
sa = hip.f16c_shifts()
g = torch.Generator().manual_seed(0)
for name, N, K, kind in [("qkv", 2304, 768, 0), ("ao", 768, 768, 3), ("ff1", 3072, 768, 2), ("ff2", 768, 3072, 3)]:
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.03
    A16, A8 = [t.to(d) for t in split_f16c(A)]
    hi = W.half().float()
    W16 = W.half().to(d)
    W8 = torch.cat([hi * 2.0 ** sa[2], (W - hi) * 2.0 ** sa[3]], 1).clamp_(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).to(d)
    bias, colc, gam, bet = [torch.randn(N, generator=g).to(d) for _ in range(4)]
    part = torch.zeros(M, 4, 2)
    part[:, :3, 0] = torch.randn(M, 3, generator=g) * 3
    part[:, :3, 1] = torch.rand(M, 3, generator=g) * 300 + 100
    part = part.to(d)
    R32 = torch.randn(M, N, generator=g).to(d) if kind == 3 else None
    outs = []
    for _ in libs:
        C = torch.full((M, N), 7.0, dtype=torch.float16 if kind == 2 else torch.float32, device=d)
        C8 = torch.full((M, 2 * N), 9, dtype=torch.uint8, device=d) if kind != 0 else None
        C16 = torch.full((M, N), 5.0, dtype=torch.float16, device=d) if kind == 3 else None
        op = torch.zeros(M, 4, 2, device=d) if kind == 3 else None
        outs.append((C, C8, C16, op))

    def run(i):
        L = libs[i][1]
        C, C8, C16, op = outs[i]
        if kind == 3:
            rc = L.ruart_gemm_16c_nt_fold(hip.ptr(A16), hip.ptr(A8), K, hip.ptr(W16), hip.ptr(W8), K, hip.ptr(bias), 3, None, 0, None, 1.0, hip.ptr(R32), N,
                                          hip.ptr(part), 3, hip.ptr(gam), hip.ptr(bet), hip.ptr(C), N, hip.ptr(C16), hip.ptr(C8), hip.ptr(op), M, N, K, 768, 1e-12,
                                          hip.stream_ptr())
        else:
            rc = L.ruart_gemm_16c_nt_fold(hip.ptr(A16), hip.ptr(A8), K, hip.ptr(W16), hip.ptr(W8), K, hip.ptr(bias), kind, hip.ptr(part), 3, hip.ptr(colc), 0.5,
                                          None, 0, None, 0, None, None, hip.ptr(C), N, None, hip.ptr(C8), None, M, N, K, K, 1e-12, hip.stream_ptr())
        assert rc == 0, rc

    for i in range(len(libs)):
        for _ in range(3):
            run(i)
    torch.cuda.synchronize()
    same = all(all(x is None or torch.equal(x, y) for x, y in zip(outs[0], o)) for o in outs[1:])
    times = [[] for _ in libs]
    for r in range(a.rounds):
        for i in (range(len(libs)) if r % 2 == 0 else reversed(range(len(libs)))):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run(i)
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) * 1e3 / a.iters)
    print("%-4s kind %d  bit-equal %s   %s" % (name, kind, same, "   ".join("%s: median %.1f min %.1f" % (libs[i][0], float(np.median(t)), min(t)) for i, t in enumerate(times))), flush=True)
