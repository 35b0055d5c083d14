#!/usr/bin/env python3
"""Three-stream against one-stream forward of the SAME network and batch (encoder inline): the kernels and their per-op order are the
same, so the scores must be bit-identical; a difference is a missing cross-stream dependency.  SYNC=<point> adds a device sync at a
named point of SDNet.forward to localise it."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import synth
from ruart_amd.arguments import default_opt
import ruart_amd.layers as L

dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
b = tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7, n_q=30, n_ocr=100, n_od=36))
if os.environ.get("TRAIN_FIRST"):      # what bench.py --no-prefetch does before its parity check: inline training steps, then a SECOND trainer
    b2 = tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=8, n_q=30, n_ocr=100, n_od=36))
    opt_d = dict(opt, ruart_defer_readback=True)
    tr.opt["ruart_defer_readback"] = True
    with tr.step_stream():
        for i in range(int(os.environ["TRAIN_FIRST"])):
            tr.update([b, b2][i % 2], i, next_batch=None)
        tr.flush_readback()
    torch.cuda.synchronize()
    tr, _ = bench.build_trainer(dict(opt, ruart_dp=False), synth.bert_config(), dev)
    b = tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7, n_q=30, n_ocr=100, n_od=36))
L.set_dropout_prob(0.0)
net = tr.network
if os.environ.get("POOLVAR") is not None:
    from ruart_amd import hip as _hip
    _hip.load().ruart_bert_pool_ln_set_variant(int(os.environ["POOLVAR"]))
if os.environ.get("SYNC") == "encoder":            # the inline encoder pass has finished before anything else of the forward is enqueued
    _lf = net.Bert.layers_for
    def _synced(packed):
        out = _lf(packed)
        torch.cuda.synchronize()
        return out
    net.Bert.layers_for = _synced
if os.environ.get("SYNC") == "pool":               # every pooling call followed by a device sync
    from ruart_amd import bert as _b
    _ap = _b._PoolMix.apply
    def _pool(*a):
        out = _ap(*a)
        torch.cuda.synchronize()
        return out
    import ruart_amd.sdnet as _sd
    _sd._PoolMix = type("P", (), {"apply": staticmethod(_pool)})


def fwd(streams):
    net.opt["ruart_streams"] = streams
    net.train()
    net.drop_emb = False
    def f():
        if streams and os.environ.get("PREFETCH"):          # the encoder pass on the CU-masked run-ahead stream instead of inline
            net.Bert.prefetch(net.prepare(b[0], b[1], b[2]).packed)
        with torch.no_grad():
            return net(b[0], b[1], b[2])[0]
    s = tr.on_step_stream(f)
    torch.cuda.synchronize()
    return s.float().cpu()


one = fwd(False)
for k in range(int(os.environ.get("N", 8))):
    three = fwd(True)
    one2 = fwd(False)
    print("run %d: |three - one| = %.3e   |one - one| = %.3e" % (k, float((three - one).abs().max()), float((one2 - one).abs().max())), flush=True)
tr.close(final=True)
