#!/usr/bin/env python3
"""One fp16c projection with and without the tail split, alone on a CU-masked stream: where does the split's time go?
    python tools/tail_split_one.py [--rows 42752 --N 768 --K 3072 --cus 240]      (under rocprofv3 --kernel-trace --stats for the two kernels)"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip
from ruart_amd.bert import split_f16c

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=42752)
ap.add_argument("--N", type=int, default=768)
ap.add_argument("--K", type=int, default=3072)
ap.add_argument("--cus", type=int, default=240)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
lib = hip.load()
d = torch.device("cuda:0")
M, N, K = a.rows, a.N, a.K
g = torch.Generator().manual_seed(0)
A16, A8 = [t.to(d) for t in split_f16c(torch.randn(M, K, generator=g))]
W = torch.randn(N, K, generator=g) * 0.03
hi = W.half().float()
sh = hip.f16c_shifts()
W16 = W.half().to(d)
W8 = torch.cat([hi * 2.0 ** sh[2], (W - hi) * 2.0 ** sh[3]], 1).clamp_(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).to(d)
bias = torch.randn(N, generator=g).to(d)
R = torch.randn(M, N, generator=g).to(d)
C = torch.empty(M, N, device=d)
st = hip.cu_masked_stream(a.cus, d) if a.cus < 256 else torch.cuda.Stream(device=d)
for plan in (0, a.cus):
    nbytes = int(lib.ruart_gemm_16c_tail_ws_bytes(M, N, K, plan)) if plan else 0
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=d)
    with torch.cuda.stream(st):
        def run():
            assert lib.ruart_gemm_16c_nt_ws(hip.ptr(A16), hip.ptr(A8), K, hip.ptr(W16), hip.ptr(W8), K, hip.ptr(bias), hip.ptr(R), N, hip.ptr(C), N, None,
                                            M, N, K, hip.ACT_NONE, 3, hip.ptr(ws) if nbytes else None, nbytes, plan, hip.stream_ptr()) == 0
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
    print("%d x %d x %d on %d CUs, plan %3d (slab bytes %d): %.1f us per product" % (M, N, K, a.cus, plan, nbytes, e0.elapsed_time(e1) * 1e3 / a.iters), flush=True)
hip.destroy_stream(st) if a.cus < 256 else None
