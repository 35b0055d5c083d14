#!/usr/bin/env python3
"""Host cost of one trunk projection, layer by layer: the bare C call, ops.mm, _Linear forward, forward + backward."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip, ops

dev = torch.device("cuda:0")
lib = hip.load()
M, K, N = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (6400, 250, 500)))
x = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev, requires_grad=True)
b = torch.randn(N, device=dev, requires_grad=True)
out = torch.empty(M, N, device=dev)
nbytes = ctypes.c_size_t(0)
lib.ruart_gemm_x3_plan(M, N, K, 1, 0, None, ctypes.byref(nbytes))
ws = torch.empty(max(nbytes.value // 4, 1), device=dev)
wt = w.detach().t()


def t(f, n=2000):
    for _ in range(50):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    host = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return host


def bare():
    lib.ruart_gemm_x3(hip.ptr(x), K, 1, hip.ptr(wt), 1, K, hip.ptr(b), None, 0, hip.ACT_NONE, hip.ptr(out), N, M, N, K, hip.ptr(ws), nbytes.value,
                      None, None, 1.0, None, 1, hip.stream_ptr())


st = hip.stream_ptr()
px, pw, pb, po, pws = hip.ptr(x), hip.ptr(wt), hip.ptr(b), hip.ptr(out), hip.ptr(ws)


def barest():
    lib.ruart_gemm_x3(px, K, 1, pw, 1, K, pb, None, 0, hip.ACT_NONE, po, N, M, N, K, pws, nbytes.value, None, None, 1.0, None, 1, st)


print("torch.empty(M, N)                      %6.1f us" % t(lambda: torch.empty(M, N, device=dev)))
print("C call, arguments prepared             %6.1f us" % t(barest))
print("C call + hip.ptr / stream_ptr per call %6.1f us" % t(bare))
print("ops.mm(x, w.t(), b)                    %6.1f us" % t(lambda: ops.mm(x, wt, b)))
with torch.no_grad():
    print("ops.linear, no grad                    %6.1f us" % t(lambda: ops.linear(x, w, b)))
print("ops.linear, grad recorded              %6.1f us" % t(lambda: ops.linear(x, w, b)))
xg = x.clone().requires_grad_(True)
g = torch.randn(M, N, device=dev)


def fb():
    y = ops.linear(xg, w, b)
    y.backward(g)
    xg.grad = None; w.grad = None; b.grad = None


print("ops.linear forward + backward          %6.1f us" % t(fb, 500))
print("torch F.linear forward + backward      %6.1f us" % t(lambda: (torch.nn.functional.linear(xg, w, b).backward(g), setattr(xg, 'grad', None), setattr(w, 'grad', None), setattr(b, 'grad', None)), 500))
