#!/usr/bin/env python3
"""Host cost per call of the hot-path wrappers (no device sync inside the loop; the shapes are small so the queue never fills):
how many microseconds of Python / ctypes each launch costs the step's enqueue thread."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import ops, hip
d = torch.device("cuda:0")
def cost(f, n=2000):
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize(); return dt
a, b = torch.randn(256, 512, device=d), torch.randn(512, 256, device=d)
w = torch.randn(256, 512, device=d, requires_grad=True); x = torch.randn(64, 100, 512, device=d)
print("torch.mm                 %6.1f us" % cost(lambda: torch.mm(a, b)))
ops.trunk_gemm = "x3"
print("ops.mm (x3)              %6.1f us" % cost(lambda: ops.mm(a, b)))
print("ops.linear fwd (no grad) %6.1f us" % cost(lambda: ops.linear(x.detach(), w.detach())))
q, k, v = torch.randn(64, 100, 250, device=d), torch.randn(64, 40, 250, device=d), torch.randn(64, 40, 250, device=d)
m = torch.ones(64, 40, dtype=torch.uint8, device=d)
print("ops.fused_attention fwd  %6.1f us" % cost(lambda: ops.fused_attention(q, k, v, m)))
print("hip.stream_ptr()         %6.1f us" % cost(lambda: hip.stream_ptr(), 20000))
print("hip.ptr(tensor)          %6.1f us" % cost(lambda: hip.ptr(a), 20000))
print("torch add                %6.1f us" % cost(lambda: a + a))
