#!/usr/bin/env python3
"""Sub-word pooling + layer mix (ruart_bert_pool_mix / _bwd - or their _ln forms over the pre-LayerNorm rows of a folded encoder pass, the
fp16c default; RUART_LN_FOLD=0 for the plain forms) on the bench batch's three word groups over fp32 layer outputs: device time
per call set, Infinity Cache flushed between calls, GB/s of the algorithmic bytes (12 layers x pieces x 3 KB read + 3 KB per word written)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import hip, synth
from ruart_amd.arguments import default_opt
from ruart_amd.bert import _PoolMix
dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
b = tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7, n_q=30, n_ocr=100, n_od=36))
bi = b[0]["_ruart_index"]
layers = tr.network.Bert.layers_for(bi.packed)
torch.cuda.synchronize()
lw = torch.softmax(torch.randn(12, device=dev), 0).requires_grad_(True)
flush = torch.empty(128 << 20, dtype=torch.float32, device=dev)
def ev(): return torch.cuda.Event(enable_timing=True)
tot_f = tot_b = 0.0; bytes_f = 0
for g in range(3):
    s_, l_, dst, rows, s_last = bi.spans[g]
    pieces = int(l_.sum().item()); W = s_.numel()
    tf, tb = [], []
    for _ in range(12):
        flush.fill_(1.0)
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        ln = getattr(layers, "_ln", None) or (None, None, None)       # a LayerNorm-folded pass (the fp16c default): the kernels normalise the rows they read
        out = _PoolMix.apply(lw, layers, s_, l_, dst, rows, hip.dtype_code(layers), s_last, *ln)
        e1.record()
        go = torch.ones_like(out)
        flush.fill_(2.0)
        e1b = ev(); e1b.record()
        out.backward(go)
        e2.record(); e2.synchronize()
        tf.append(e0.elapsed_time(e1) * 1e3); tb.append(e1b.elapsed_time(e2) * 1e3)
    tf.sort(); tb.sort()
    nb = 12 * pieces * 3072 + W * 3072
    print("group %d: %5d words %6d pieces | forward %6.1f us = %.2f TB/s | backward (incl. zero-fill + partial sums) %6.1f us" % (g, W, pieces, tf[6], nb / tf[6] / 1e6, tb[6]))
    tot_f += tf[6]; tot_b += tb[6]; bytes_f += nb
print("all groups: forward %.1f us = %.2f TB/s, backward %.1f us" % (tot_f, bytes_f / tot_f / 1e6, tot_b))
tr.close(final=True)
