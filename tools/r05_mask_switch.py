#!/usr/bin/env python3
"""Does a CU-masked encoder stream created MID-RUN overlap with the trunk like the one created at the first step?  (A trunk stream created
after the masked stream does not: profiles/HISTORY.md round 5 (6b).)  Two batches whose row counts plan different masks; phases: A only, B only
(its stream is created here, mid-run), alternating A / B, A only again.  Prints the median step time of every phase."""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth
from ruart_amd.arguments import default_opt

dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
A = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)) for i in range(2)]
B = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=17 + i, n_q=30, n_ocr=int(os.environ.get("B_OCR", 88)), n_od=36)) for i in range(2)]
bert = tr.network.Bert
for name, bs in (("A", A), ("B", B)):
    p = bs[0][0]["_ruart_index"].packed
    print("batch %s: %d rows -> plan %s CUs" % (name, p.Tp, bert.plan_prefetch_cus(p.Tp)), flush=True)


def phase(name, seq, n=40):
    ts = []
    torch.cuda.synchronize()
    for i in range(n):
        b, nb = seq[i % len(seq)], seq[(i + 1) % len(seq)]
        t0 = time.perf_counter()
        tr.update(b, i, next_batch=nb)
        ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    print("%-28s median %.2f ms  (p90 %.2f, streams alive: %s)" % (name, statistics.median(ts[5:]), sorted(ts[5:])[int(0.9 * (n - 5))],
                                                                   sorted(bert._pf_streams.keys())), flush=True)


if os.environ.get("B_FIRST"):
    A, B = B, A
    print("(order swapped: the first phase runs the smaller batch)")
if os.environ.get("PRECREATE"):      # both masks' streams created before the first step
    from ruart_amd import hip
    for b_ in (A, B):
        c = bert.plan_prefetch_cus(b_[0][0]["_ruart_index"].packed.Tp)
        bert._pf_streams.setdefault(c, hip.cu_masked_stream(c, dev))
    _dummies = [hip.cu_masked_stream(200, dev) for _ in range(int(os.environ.get("PRE_EXTRA", 0)))]     # never used: they only take queue slots
phase("first batch only", A)
phase("second batch only (new stream)", B)
if not os.environ.get("SHORT"):
    phase("alternating", [A[0], B[0], A[1], B[1]])
    phase("first batch only again", A)
    phase("second batch only again", B)
tr.close(final=True)
