#!/usr/bin/env python3
"""Back-to-back launches of the plain f16 QKV product (gemm_16_nt_256p8, 42 752 x 2304 x 768) for SECONDS_ seconds; prints us per launch."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip
lib = hip.load(); d = torch.device("cuda:0")
M, N, K = 42752, 2304, int(os.environ.get("K_", 768))
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).half().to(d); W = (torch.randn(N, K, generator=g) * 0.05).half().to(d)
bias = torch.randn(N, generator=g).to(d); C = torch.empty(M, N, dtype=torch.float16, device=d)
secs = float(os.environ.get("SECONDS_", 8)); t0 = time.time(); n = 0
while time.time() - t0 < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        lib.ruart_gemm_16_nt(hip.ptr(A), K, hip.ptr(W), K, hip.ptr(bias), None, 0, hip.DT_F16, hip.ptr(C), N, hip.DT_F16, M, N, K, 0, hip.DT_F16, hip.stream_ptr())
    e1.record(); torch.cuda.synchronize(); n += 200
    last = e0.elapsed_time(e1) * 1e3 / 200
print("%s K=%d: %d launches, last 200 at %.1f us each = %.0f TFLOP/s" % (os.environ.get("RUART_HIP_LIB", "product"), K, n, last, 2.0 * M * N * K / last / 1e6), flush=True)
