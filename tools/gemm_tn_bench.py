"""Time the weight-gradient product  dW (M x N) = dY (T x M)^T . X (T x N)  on the trainable encoder's four shapes:
ruart_gemm_16_tn_splitk (operands as they lie) against ruart_gemm_16_nt_splitk on pre-transposed copies (transposes not timed).
  python tools/gemm_tn_bench.py [lib.so]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ruart_amd import hip  # noqa: E402

if len(sys.argv) > 1:
    os.environ["RUART_HIP_LIB"] = sys.argv[1]
lib = hip.load()
dev = torch.device("cuda:0")
T = 45056
st = hip.stream_ptr


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, N in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    dY = (torch.randn(T, M, device=dev) * 1e-3).bfloat16()
    X = torch.randn(T, N, device=dev).bfloat16()
    dYt, Xt = dY.t().contiguous(), X.t().contiguous()
    tiles = (M // 256) * (N // 256)
    for cap in (256, 512):
        nz = max(1, min(cap // tiles, T // 128))
        tch = ((T + nz - 1) // nz + 127) // 128 * 128
        nz = (T + tch - 1) // tch
        part = torch.empty(nz, M, N, device=dev)
        t_tn = timed(lambda: lib.ruart_gemm_16_tn_splitk(hip.ptr(dY), M, hip.ptr(X), N, hip.ptr(part), N, M, N, T, tch, hip.DT_BF16, st()))
        t_nt = timed(lambda: lib.ruart_gemm_16_nt_splitk(hip.ptr(dYt), T, hip.ptr(Xt), T, hip.ptr(part), N, M, N, T, tch, hip.DT_BF16, st()))
        fl = 2.0 * M * N * T
        print("M %4d N %4d  slices %3d (x %2d tiles)  TN %7.1f us %6.0f TF   NT %7.1f us %6.0f TF" % (M, N, nz, tiles, t_tn, fl / t_tn / 1e6,
                                                                                                    t_nt, fl / t_nt / 1e6), flush=True)
