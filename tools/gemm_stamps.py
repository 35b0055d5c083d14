#!/usr/bin/env python3
"""In-kernel phase times of the encoder GEMM (diagnostic build: tools/build_variant.sh stamps -DRUART_P8_STAMPS, run with
RUART_HIP_LIB=build/libruart_hip_stamps.so): per workgroup s_memrealtime (100 MHz) at kernel start, after the pipeline fill,
after the K loop and after the epilogue's stores have drained."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip
lib = hip.load(); d = torch.device("cuda:0")
lib.ruart_gemm_set_stamps.argtypes = [ctypes.c_void_p]; lib.ruart_gemm_set_stamps.restype = ctypes.c_int
dt = hip.DT_F16; td = torch.float16
M = int(os.environ.get("ROWS", 43008))
for name, N, K, act, res in [("qkv", 2304, 768, 0, False), ("ao", 768, 768, 0, True), ("ff1", 3072, 768, 1, False), ("ff2", 768, 3072, 0, True)]:
    g = torch.Generator().manual_seed(0)
    A = torch.randn(M, K, generator=g).to(td).to(d); W = (torch.randn(N, K, generator=g) * 0.05).to(td).to(d)
    bias = torch.randn(N, generator=g).to(d); R = torch.randn(M, N, generator=g).to(td).to(d) if res else None
    C = torch.empty(M, N, dtype=torch.float32 if res else td, device=d)
    ntiles = (M // 256) * (N // 256)
    st = torch.zeros(ntiles * 4, dtype=torch.int64, device=d)
    def run():
        assert lib.ruart_gemm_16_nt(hip.ptr(A), K, hip.ptr(W), K, hip.ptr(bias), hip.ptr(R), N, dt, hip.ptr(C), N, hip.DT_F32 if res else dt, M, N, K, act, dt, hip.stream_ptr()) == 0
    lib.ruart_gemm_set_stamps(None)
    for _ in range(3): run()
    lib.ruart_gemm_set_stamps(st.data_ptr()); run(); torch.cuda.synchronize(); lib.ruart_gemm_set_stamps(None)
    t = st.cpu().numpy().reshape(ntiles, 4).astype(np.float64) * 0.01           # us
    fill, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    t0 = t[:, 0].min()
    first = (t[:, 0] - t0) < 5.0                                                  # workgroups of the first round
    print("%-4s tiles %4d | fill %5.2f us (first round %5.2f, later %5.2f) | K loop %6.2f us = %.2f us per K-tile | epilogue+drain %5.2f us | tile %6.2f us | kernel %7.1f us" %
          (name, ntiles, np.median(fill), np.median(fill[first]), np.median(fill[~first]) if (~first).any() else float("nan"), np.median(loop), np.median(loop) / (K // 64),
           np.median(epi), np.median(t[:, 3] - t[:, 0]), t[:, 3].max() - t0))
    # launch ramp and round hand-over: start times of the first 256 workgroups, then start of later ones vs ends of earlier ones
    order = np.argsort(t[:, 0])
    starts, ends = t[order, 0] - t0, np.sort(t[:, 3]) - t0
    n1 = min(256, ntiles)
    print("     first %d workgroups start within %.1f us (p50 %.1f, p90 %.1f); workgroup #%d starts at %.1f us, the earliest end is at %.1f us; "
          "start(k) - end(k - 256), k >= 256: p50 %.1f us, p90 %.1f us; last end %.1f us" %
          (n1, starts[n1 - 1], starts[n1 // 2], starts[int(n1 * 0.9)], n1, starts[n1] if ntiles > n1 else -1, ends[0],
           np.median(starts[256:] - ends[:ntiles - 256]) if ntiles > 256 else -1,
           np.percentile(starts[256:] - ends[:ntiles - 256], 90) if ntiles > 256 else -1, ends[-1]))
