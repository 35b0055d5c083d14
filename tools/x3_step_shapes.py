#!/usr/bin/env python3
"""Every ops.mm call of one training step (shape, operand layouts), timed standalone on ruart_gemm_x3 and on torch.mm."""
import os, sys
from collections import Counter
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth, ops
from ruart_amd.arguments import default_opt

dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
if os.environ.get("RUART_UNLOCK_X3"):              # the fp32-class trainable-encoder graph: every encoder product on ruart_gemm_x3 as well
    opt.pop("LOCK_BERT")
    opt["bert_train_gemm"] = "x3"
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
batches = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)) for i in range(2)]
for i in range(3):
    tr.update(batches[i % 2], i)
calls = Counter()
orig = ops.mm
def spy(a, b, bias=None, mode=None, out=None, **kw):
    M, K = a.shape; N = b.shape[1]
    calls[(M, N, K, int(a.stride(1) == 1), int(b.stride(0) == 1 and b.stride(1) != 1), bias is not None)] += 1
    return orig(a, b, bias, mode, out, **kw)
ops.mm = spy
tr.update(batches[0], 3)
ops.mm = orig
torch.cuda.synchronize()
def timeit(f, n=10):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
rows = []
for (M, N, K, ak, bk, hb), cnt in calls.items():
    a = torch.randn(M, K, device=dev) if ak else torch.randn(K, M, device=dev).t()
    b = torch.randn(N, K, device=dev).t() if bk else torch.randn(K, N, device=dev)
    tx = timeit(lambda: orig(a, b, mode="x3")); tb = timeit(lambda: torch.mm(a, b))
    rows.append((cnt * tx, cnt, M, N, K, "K" if ak else "M", "K" if bk else "N", tx, tb))
tot_x = sum(r[0] for r in rows); tot_b = sum(r[1] * r[8] for r in rows)
for r in sorted(rows, reverse=True)[:40]:
    print("%7.1f us/step  x%2d  M %5d N %5d K %5d  A:%s-contig B:%s-contig  x3 %6.1f us  rocBLAS %6.1f us  (%.0f TF/s-equiv)" %
          (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], 2.0 * r[2] * r[3] * r[4] / r[7] / 1e6))
print("calls %d, unique %d; standalone sum: x3 %.2f ms, rocBLAS %.2f ms" % (sum(calls.values()), len(calls), tot_x / 1e3, tot_b / 1e3))
