#!/usr/bin/env python3
"""Persistent LSTM recurrence kernels: time per call and per step at the trunk's shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import ops
d = torch.device("cuda:0")
def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
from ruart_amd import hip
for variant in (0, 1):
  assert hip.load().ruart_lstm_set_variant(variant) == 0
  print("variant %d (%s)" % (variant, "16 rows per workgroup, MFMA" if variant else "one row per workgroup, fp32 FMA"))
  for (B, T, h) in [(64, 100, 125), (64, 36, 125), (64, 40, 125)]:
      xp = torch.randn(B, T, 8 * h, device=d, requires_grad=True); whh = (torch.randn(2, 4 * h, h, device=d) * 0.05).requires_grad_(True)
      y = ops._LstmRecurrence.apply(xp, whh, 2); gy = torch.randn_like(y)
      tf = timeit(lambda: ops._LstmRecurrence.apply(xp.detach(), whh.detach(), 2))
      def fb():
          y = ops._LstmRecurrence.apply(xp, whh, 2); y.backward(gy)
      tfb = timeit(fb)
      print("B %d T %3d h %d: fwd %.1f us (%.2f us/step), fwd+bwd(+W_hh grads) %.1f us" % (B, T, h, tf, tf / T, tfb))
