#!/usr/bin/env python3
"""ruart_gemm_x3 (split-bf16 MFMA) vs torch.mm (rocBLAS fp32) on the trunk's projection shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import ops
d = torch.device("cuda:0")
shapes = [("multi2one fwd  x.W^T", 12800, 1200, 1388, "nt"), ("multi2one dX   dY.W", 12800, 1388, 1200, "nn"),
          ("multi2one dW   dY^T.X", 1200, 1388, 12800, "tn"), ("ctx lstm fwd", 6400, 1000, 1250, "nt"), ("ctx lstm dW", 1000, 1250, 6400, "tn"),
          ("att proj fwd", 6400, 250, 1800, "nt"), ("att proj dX", 6400, 1800, 250, "nn"), ("att proj dW", 250, 1800, 6400, "tn"),
          ("lstm W_hh grad", 500, 125, 6400, "tn"), ("q lstm dW", 500, 125, 2304, "tn"), ("small proj", 2560, 250, 800, "nt"),
          ("small dW", 250, 800, 2560, "tn")]
def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
tot_a = tot_b = 0
for name, M, N, K, lay in shapes:
    if lay == "nt":
        a = torch.randn(M, K, device=d); b = torch.randn(N, K, device=d).t()
    elif lay == "nn":
        a = torch.randn(M, K, device=d); b = torch.randn(K, N, device=d)
    else:
        a = torch.randn(K, M, device=d).t(); b = torch.randn(K, N, device=d)
    t_x3 = timeit(lambda: ops.mm(a, b, mode="x3"))
    t_rb = timeit(lambda: torch.mm(a, b))
    tot_a += t_x3; tot_b += t_rb
    fl = 2.0 * M * N * K
    print("%-24s M %5d N %5d K %5d %s | x3 %7.1f us (%5.0f TF/s-equiv) | rocBLAS fp32 %7.1f us (%5.0f TF/s) | %.2fx" %
          (name, M, N, K, lay, t_x3, fl / t_x3 / 1e6, t_rb, fl / t_rb / 1e6, t_rb / t_x3))
print("sum: x3 %.0f us, rocBLAS %.0f us" % (tot_a, tot_b))
