#!/usr/bin/env python3
"""round 6: where do the NaNs of ruart_gemm_16c_nt at (256, 2304, 768) sit?  Runs the plain fp16c product of
test_gemm_16c_fold_consumer's materialised rows several times and prints the non-finite elements' tile coordinates."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip
from ruart_amd.bert import split_f16c
lib = hip.load(); d = torch.device("cuda:0")
sa = hip.f16c_shifts()
def w8(W):
    hi = W.half().float()
    return W.half(), torch.cat([hi * 2.0 ** sa[2], (W - hi) * 2.0 ** sa[3]], 1).clamp_(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
for (M, N, K) in [(256, 2304, 768), (256, 768, 768), (512, 2304, 768), (256, 2304, 256), (256, 256, 128)]:
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g) * 3
    W = torch.randn(N, K, generator=g) * 0.03
    bias = torch.randn(N, generator=g) * 0.1
    X16, X8 = split_f16c(x)
    V16, V8 = w8(W)
    ref = (x.double() @ W.double().t() + bias.double())
    X16, X8, V16, V8, bd = [t.to(d) for t in (X16, X8, V16, V8, bias)]
    for it in range(6):
        U = torch.full((M, N), 7.0, dtype=torch.float32, device=d)
        rc = lib.ruart_gemm_16c_nt(hip.ptr(X16), hip.ptr(X8), K, hip.ptr(V16), hip.ptr(V8), K, hip.ptr(bd), None, 0, hip.ptr(U), N, None, M, N, K, hip.ACT_NONE, hip.stream_ptr())
        torch.cuda.synchronize()
        Uc = U.cpu()
        bad = ~torch.isfinite(Uc)
        err = float((Uc.double() - ref)[~bad].abs().max())
        msg = ""
        if bad.any():
            idx = bad.nonzero()
            msg = " nonfinite %d rows %d..%d cols %d..%d first %s" % (int(bad.sum()), int(idx[:, 0].min()), int(idx[:, 0].max()), int(idx[:, 1].min()), int(idx[:, 1].max()), idx[:4].tolist())
        print("M %d N %d K %d it %d rc %d max err (finite) %.3e%s" % (M, N, K, it, rc, err, msg), flush=True)
