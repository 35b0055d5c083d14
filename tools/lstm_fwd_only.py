import os, sys, torch
sys.path.insert(0, os.getcwd())
from ruart_amd import ops, hip
d = torch.device("cuda:0")
hip.load().ruart_lstm_set_variant(1)
B, T, h = 64, 100, 125
xp = torch.randn(B, T, 8 * h, device=d); whh = torch.randn(2, 4 * h, h, device=d) * 0.05
f = lambda: ops._LstmRecurrence.apply(xp, whh, 2)
for _ in range(3): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("RUART_HIP_LIB", "base"), "fwd %.1f us (%.2f us/step)" % (e0.elapsed_time(e1) * 50, e0.elapsed_time(e1) * 50 / T))
