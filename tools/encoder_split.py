#!/usr/bin/env python3
"""Experiment: the encoder pass of the bench batch as ONE packed stream vs TWO half streams (split by sequences) running
concurrently on two HIP streams (phase-shifted GEMM rounds, LN / attention of one half beside GEMMs of the other)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth
from ruart_amd.arguments import default_opt
from ruart_amd.bert import PackedTokens, bert_encode, _Buffers

dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
W = tr.network.Bert.weights
q, ocr, od, _, _ = synth.synthetic_batch(opt, 64, seed=7, n_q=30, n_ocr=100, n_od=36)
groups = [(q["bert"], q["bert_mask"]), (ocr["bert"], ocr["bert_mask"]), (od["bert"], od["bert_mask"])]
full = PackedTokens(groups, dev)
def halves(n_parts):
    parts = [[] for _ in range(n_parts)]
    for ids, m in groups:
        n = ids.shape[0]
        cuts = [n * i // n_parts for i in range(n_parts + 1)]
        for p in range(n_parts):
            parts[p].append((ids[cuts[p]:cuts[p + 1]], m[cuts[p]:cuts[p + 1]]))
    return [PackedTokens(g, dev) for g in parts]
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
bf = _Buffers()
print("one stream, one pack (T=%d): %.2f ms" % (full.T, timeit(lambda: bert_encode(W, full, bf))))
for n_parts in (2, 3, 4):
    ps = halves(n_parts)
    bufs = [_Buffers() for _ in ps]
    streams = [torch.cuda.Stream(device=dev) for _ in ps]
    def run():
        cur = torch.cuda.current_stream()
        for p, b, s in zip(ps, bufs, streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                bert_encode(W, p, b)
        for s in streams:
            cur.wait_stream(s)
    def run_serial():
        for p, b in zip(ps, bufs):
            bert_encode(W, p, b)
    print("%d packs (T=%s): concurrent on %d streams %.2f ms; same packs back to back on one stream %.2f ms" %
          (n_parts, [p.T for p in ps], n_parts, timeit(run), timeit(run_serial)))
