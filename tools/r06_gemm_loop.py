#!/usr/bin/env python3
"""Back-to-back launches of the fp16c QKV product for SECONDS seconds (power / clock sampling from outside: tools/r06_power.sh)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip
from ruart_amd.bert import split_f16c
lib = hip.load(); d = torch.device("cuda:0"); sa = hip.f16c_shifts()
M, N, K = 42752, 2304, 768
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) * 0.03
A16, A8 = [t.to(d) for t in split_f16c(A)]
hi = W.half().float(); W16 = W.half().to(d)
W8 = torch.cat([hi * 2.0 ** sa[2], (W - hi) * 2.0 ** sa[3]], 1).clamp_(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).to(d)
bias = torch.randn(N, generator=g).to(d); C = torch.empty(M, N, device=d)
secs = float(os.environ.get("SECONDS_", 8))
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        lib.ruart_gemm_16c_nt(hip.ptr(A16), hip.ptr(A8), K, hip.ptr(W16), hip.ptr(W8), K, hip.ptr(bias), None, 0, hip.ptr(C), N, None, M, N, K, hip.ACT_NONE, hip.stream_ptr())
    e1.record(); torch.cuda.synchronize(); n += 200
    last = e0.elapsed_time(e1) * 1e3 / 200
print("%s: %d launches, last 200 at %.1f us each" % (os.environ.get("RUART_HIP_LIB", "product"), n, last), flush=True)
