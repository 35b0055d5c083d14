#!/usr/bin/env python3
"""The kind-3 product of the LayerNorm-folded encoder pass (ruart_gemm_16c_nt_fold: y = A W^T + b + LN(residual), written fp32 + split +
row partials) alone on the device at the bench shapes, against the plain residual form (ruart_gemm_16c_nt) on the same operands.
With RUART_HIP_LIB=build/libruart_hip_<variant>.so (tools/build_variant.sh nostats -DRUART_ABL_FOLD_NOSTATS / nosplit
-DRUART_ABL_FOLD_NOSPLIT): what the partial sums / the split copy cost the epilogue (valid inputs every launch - a whole-pass ablation
feeds the next kernels garbage and their clock changes with it)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip
from ruart_amd.bert import split_f16c

lib = hip.load()
d = torch.device("cuda:0")
M, N = 42752, 768
g = torch.Generator().manual_seed(1)


def w8(W):
    hi = W.to(torch.float16).to(torch.float32)
    _, _, sw_hi, sw_lo = hip.f16c_shifts()
    pair = torch.cat([hi * float(2.0 ** sw_hi), (W - hi) * float(2.0 ** sw_lo)], 1).clamp_(-448.0, 448.0)
    return W.to(torch.float16).contiguous(), pair.to(torch.float8_e4m3fn).view(torch.uint8).contiguous()


def timeit(f, n=30):
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for K in (768, 3072):
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.03
    A16, A8 = [t.to(d) for t in split_f16c(A)]
    W16, W8 = [t.to(d) for t in w8(W)]
    bias = (torch.randn(N, generator=g) * 0.1).to(d)
    R = torch.randn(M, N, generator=g).to(d)
    gam, bet = torch.ones(N, device=d), torch.zeros(N, device=d)
    part_in = torch.zeros(M, 4, 2, device=d)
    part_in[:, :3, 1] = 256.0                      # mean 0, variance 1
    C = torch.zeros(M, N, device=d)
    C16 = torch.zeros(M, N, dtype=torch.float16, device=d)
    C8 = torch.zeros(M, 2 * N, dtype=torch.uint8, device=d)
    part = torch.zeros(M, 4, 2, device=d)
    s = hip.stream_ptr()
    plain = lambda: hip.check(lib.ruart_gemm_16c_nt(hip.ptr(A16), hip.ptr(A8), K, hip.ptr(W16), hip.ptr(W8), K, hip.ptr(bias), hip.ptr(R), N,
                                                    hip.ptr(C), N, None, M, N, K, hip.ACT_NONE, s), "plain")
    k3 = lambda ln: hip.check(lib.ruart_gemm_16c_nt_fold(hip.ptr(A16), hip.ptr(A8), K, hip.ptr(W16), hip.ptr(W8), K, hip.ptr(bias), 3, None, 0, None,
                                                         1.0, hip.ptr(R), N, hip.ptr(part_in) if ln else None, 3, hip.ptr(gam), hip.ptr(bet),
                                                         hip.ptr(C), N, hip.ptr(C16), hip.ptr(C8), hip.ptr(part), M, N, K, N, 1e-12, s), "kind3")
    print("K = %4d: plain residual epilogue %.1f us | kind 3, plain residual %.1f us | kind 3, residual normalised %.1f us"
          % (K, timeit(plain), timeit(lambda: k3(False)), timeit(lambda: k3(True))), flush=True)
