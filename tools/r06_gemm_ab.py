#!/usr/bin/env python3
"""In-process A/B of builds of the library on the fp16c encoder projections (cdna_hip_programming.md rule 24: one process, the same
buffers, interleaved rounds; separate processes of this bench differ by +-10 % with the buffers' addresses):
    python tools/r06_gemm_ab.py build/libruart_hip_base.so build/libruart_hip_new.so [--rows 42752] [--rounds 12]
prints per shape the median and minimum of every library's launch time (us) over the rounds."""
import argparse, ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip
from ruart_amd.bert import split_f16c

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--rows", type=int, default=42752)
ap.add_argument("--rounds", type=int, default=12)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
hip.load()                      # (the package's own copy: types, shifts)
libs = []
for p in a.libs:
    L = ctypes.CDLL(os.path.abspath(p))
    for name in ("ruart_gemm_16c_nt", "ruart_gemm_16c_nt_fold"):
        f = getattr(L, name)
        f.restype, f.argtypes = hip._SIGNATURES[name]
    libs.append((os.path.basename(p).replace("libruart_hip_", "").replace(".so", ""), L))
d = torch.device("cuda:0")
M = (a.rows + 255) // 256 * 256
sa = hip.f16c_shifts()
g = torch.Generator().manual_seed(0)
for name, N, K, act, res in [("qkv", 2304, 768, hip.ACT_NONE, False), ("ao", 768, 768, hip.ACT_NONE, True), ("ff1", 3072, 768, hip.ACT_GELU, False),
                             ("ff2", 768, 3072, hip.ACT_NONE, True)]:
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.03
    A16, A8 = [t.to(d) for t in split_f16c(A)]
    hi = W.half().float()
    W16 = W.half().to(d)
    W8 = torch.cat([hi * 2.0 ** sa[2], (W - hi) * 2.0 ** sa[3]], 1).clamp_(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).to(d)
    bias = torch.randn(N, generator=g).to(d)
    R32 = torch.randn(M, N, generator=g).to(d) if res else None
    gelu = act == hip.ACT_GELU
    C = torch.empty(M, N, dtype=torch.float16 if gelu else torch.float32, device=d)
    C8 = torch.empty(M, 2 * N, dtype=torch.uint8, device=d) if gelu else None
    # the folded pass's kind-3 form of the two N = 768 products (what the product runs): split copy + row partials
    C16 = torch.empty(M, N, dtype=torch.float16, device=d) if res else None
    C8r = torch.empty(M, 2 * N, dtype=torch.uint8, device=d) if res else None
    part = torch.zeros(M, 4, 2, device=d) if res else None

    def run(L):
        if res:
            rc = L.ruart_gemm_16c_nt_fold(hip.ptr(A16), hip.ptr(A8), K, hip.ptr(W16), hip.ptr(W8), K, hip.ptr(bias), 3, None, 0, None, 1.0, hip.ptr(R32), N,
                                          None, 0, None, None, hip.ptr(C), N, hip.ptr(C16), hip.ptr(C8r), hip.ptr(part), M, N, K, N, 1e-12, hip.stream_ptr())
        else:
            rc = L.ruart_gemm_16c_nt(hip.ptr(A16), hip.ptr(A8), K, hip.ptr(W16), hip.ptr(W8), K, hip.ptr(bias), None, 0, hip.ptr(C), N, hip.ptr(C8), M, N, K, act,
                                     hip.stream_ptr())
        assert rc == 0, rc

    times = {n: [] for n, _ in libs}
    for n, L in libs:
        for _ in range(3):
            run(L)
    for r in range(a.rounds):
        for n, L in (libs if r % 2 == 0 else libs[::-1]):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run(L)
            e1.record()
            torch.cuda.synchronize()
            times[n].append(e0.elapsed_time(e1) * 1e3 / a.iters)
    print("%-4s %s" % (name + (" (kind 3)" if res else ""), "   ".join("%s: median %.1f min %.1f" % (n, float(np.median(t)), min(t)) for n, t in times.items())), flush=True)
