#!/usr/bin/env python3
"""round 6: the unfolded product of test_gemm_16c_fold_consumer[False-256-2304-768] - which elements are not finite, with and
without the folded launch in front of it?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip
from ruart_amd.bert import split_f16c
lib = hip.load(); d = torch.device("cuda:0")
sa = hip.f16c_shifts()
def w8(W):
    hi = W.half().float()
    return W.half().contiguous(), torch.cat([hi * 2.0 ** sa[2], (W - hi) * 2.0 ** sa[3]], 1).clamp_(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8).contiguous()
M, N, K = 256, 2304, 768
g = torch.Generator().manual_seed(M + N + K)
y = torch.randn(M, K, generator=g) * (0.5 + 2.0 * torch.rand(M, 1, generator=g)) + 0.3 * torch.randn(M, 1, generator=g)
y[:, 5] += 20.0
W = torch.randn(N, K, generator=g) * 0.03
bias = torch.randn(N, generator=g) * 0.1
gam, bet = 1.0 + 0.2 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
gam[7] = 40.0
mu = y.double().mean(1, keepdim=True)
var = ((y.double() - mu) ** 2).mean(1, keepdim=True)
x = (y.double() - mu) / torch.sqrt(var + 1e-12) * gam.double() + bet.double()
ref = x @ W.double().t() + bias.double()
X16, X8 = split_f16c(x.float())
V16, V8 = w8(W)
X16, X8, V16, V8, bd = [t.to(d) for t in (X16, X8, V16, V8, bias)]
for it in range(4):
    U = torch.zeros(M, N, dtype=torch.float32, device=d)
    rc = lib.ruart_gemm_16c_nt(hip.ptr(X16), hip.ptr(X8), K, hip.ptr(V16), hip.ptr(V8), K, hip.ptr(bd), None, 0, hip.ptr(U), N, None, M, N, K, hip.ACT_NONE, hip.stream_ptr())
    torch.cuda.synchronize()
    Uc = U.cpu()
    bad = ~torch.isfinite(Uc)
    msg = ""
    if bad.any():
        idx = bad.nonzero()
        cols = sorted(set(idx[:, 1].tolist())); rows = sorted(set(idx[:, 0].tolist()))
        msg = " nonfinite %d; %d distinct rows (first %s), %d distinct cols (first %s)" % (int(bad.sum()), len(rows), rows[:8], len(cols), cols[:8])
    print("it %d rc %d max err (finite) %.3e%s" % (it, rc, float((Uc.double() - ref)[~bad].abs().max()), msg), flush=True)
# the f16 product alone and each correction half alone (ruart_gemm_16c_nt_sel corr = 0, 1, 2)
for corr in (0, 1, 2, 3):
    U = torch.zeros(M, N, dtype=torch.float32, device=d)
    rc = lib.ruart_gemm_16c_nt_sel(hip.ptr(X16), hip.ptr(X8), K, hip.ptr(V16), hip.ptr(V8), K, hip.ptr(bd), None, 0, hip.ptr(U), N, None, M, N, K, hip.ACT_NONE, corr, hip.stream_ptr())
    torch.cuda.synchronize()
    bad = ~torch.isfinite(U.cpu())
    print("corr %d rc %d nonfinite %d" % (corr, rc, int(bad.sum())))
b = X8.cpu()
print("A8 bytes with all-ones exponent+mantissa:", int(((b & 0x7f) == 0x7f).sum()), "| V8:", int(((V8.cpu() & 0x7f) == 0x7f).sum()))
print("max |x| %.1f  max |a_lo 2^%d| %.1f" % (float(x.abs().max()), sa[0], float(((x.float() - x.float().half().float()) * 2.0 ** sa[0]).abs().max())))
