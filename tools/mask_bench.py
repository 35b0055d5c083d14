import os, sys, torch
sys.path.insert(0, "/root/repo")
from ruart_amd import ops
d = torch.device("cuda:0")
def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (B, T, K, N) in [(64, 100, 1250, 1000), (64, 100, 1800, 250), (64, 100, 300, 1000)]:
    x = torch.randn(B, T, K, device=d); w = torch.randn(N, K, device=d) * 0.05; m = (torch.rand(B, K, device=d) > 0.3).float() / 0.7; kp = (m != 0).view(torch.uint8)
    gy = torch.randn(B * T, N, device=d); x2 = x.view(-1, K)
    print(B, T, K, N)
    print("  fwd  plain %.1f  fused %.1f  (+mul %.1f)" % (timeit(lambda: ops.mm(x2, w.t())), timeit(lambda: ops.mm(x2, w.t(), a_keep=kp, keep_scale=1 / 0.7, rpm=T)), timeit(lambda: x * m.unsqueeze(1))))
    print("  dX   plain %.1f  fused %.1f" % (timeit(lambda: ops.mm(gy, w)), timeit(lambda: ops.mm(gy, w, c_scale=m, rpm=T))))
    print("  dW   plain %.1f  fused %.1f" % (timeit(lambda: ops.mm(gy.t(), x2)), timeit(lambda: ops.mm(gy.t(), x2, b_keep=kp, keep_scale=1 / 0.7, rpm=T))))
