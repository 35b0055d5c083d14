#!/usr/bin/env python3
"""Eight training steps from the same seeds in the product's schedule (encoder pass one step ahead on the masked stream, three-stream
trunk, loop on the step stream, loss read back one step late) and fully serialised (encoder inline, one stream, per-step readback):
the parameters must agree to the rounding of the gradient fan-in order - a buffer reused too early or a missing cross-stream
dependency shows as a real difference.  (Round 5: losses agree to 6e-5 after eight steps, the first two to every printed digit.  The
parameter comparison is only indicative: Adamax divides by the running max |g|, so an element whose gradient is at rounding level moves
by +-lr per step whichever sign the rounding gives it.)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth
from ruart_amd.arguments import default_opt

dev = torch.device("cuda:0")
N = int(os.environ.get("N", 8))


def run(pipelined):
    opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64, ruart_streams=pipelined)
    opt["ruart_defer_readback"] = pipelined
    for k in ("DROPOUT", "dropout_emb"):         # no dropout: the two schedules draw their masks in different orders
        opt.pop(k, None)
    opt["DROPOUT"] = 0.0
    tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
    bs = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100 - 7 * i, n_od=36)) for i in range(3)]
    losses = []
    ctx = tr.step_stream() if pipelined else __import__("contextlib").nullcontext()
    with ctx:
        for i in range(N):
            losses.append(tr.update(bs[i % 3], i, next_batch=bs[(i + 1) % 3] if pipelined else None))
        tr.flush_readback()
    torch.cuda.synchronize()
    params = {n: p.detach().float().cpu().clone() for n, p in tr.network.named_parameters()}
    losses = [float(x) for x in losses]
    tr.close()
    return losses, params


la, pa = run(True)
lb, pb = run(False)
print("losses pipelined:", " ".join("%.6f" % x for x in la))
print("losses serial   :", " ".join("%.6f" % x for x in lb))
worst = max(((pa[n] - pb[n]).abs().max().item() / max(1e-12, pb[n].abs().max().item()), n) for n in pa)
print("max |loss difference| %.3e; worst relative parameter difference %.3e (%s)" % (max(abs(x - y) for x, y in zip(la, lb)), worst[0], worst[1]))
