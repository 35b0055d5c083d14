#!/usr/bin/env python3
"""Is the product's forward bit-repeatable inside one process?  The same batch through SDNet.forward (training mode, no grad, on the
trainer's step stream, encoder inline) N times, interleaved with a few training steps; prints the max difference between the runs'
scores and against the first."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth
from ruart_amd.arguments import default_opt
import ruart_amd.layers as L

dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
b = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)) for i in range(2)]
L.set_dropout_prob(0.0)


def fwd():
    tr.network.train()
    tr.network.drop_emb = False
    def f():
        with torch.no_grad():
            return tr.network(b[0][0], b[0][1], b[0][2])[0]
    s = tr.on_step_stream(f)
    torch.cuda.synchronize()
    return s.float().cpu()


ref = fwd()
for k in range(int(os.environ.get("N", 12))):
    if os.environ.get("STEPS"):
        for i in range(3):
            tr.update(b[i % 2], i, next_batch=None if os.environ.get("INLINE") else b[(i + 1) % 2])
        # the steps changed the weights: a new reference for this round
        ref = fwd()
    s = fwd()
    print("run %2d: max |s - ref| = %.3e" % (k, float((s - ref).abs().max())), flush=True)
tr.close(final=True)
