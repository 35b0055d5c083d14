#!/usr/bin/env python3
"""Run the same full-size batch several times (streams on/off, permuted) and report max differences."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import synth
from ruart_amd.arguments import default_opt
from ruart_amd.sdnet import SDNet
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
dev = "cuda:0"
cfg = synth.bert_config(vocab_size=3000)
bw = synth.make_bert_weights(cfg, seed=21)
def make(precision, **extra):
    opt = default_opt(vocab_size=2000, cuda=True, device=dev, bert_precision=precision, max_od_num=36, **extra)
    opt["bert_state"], opt["bert_config"] = bw, cfg
    sw = synth.make_sdnet_weights(opt, seed=21)
    net = SDNet(opt, {"glove_embedding": T(sw["glove_embed.weight"]), "fast_embedding": T(sw["fast_embed.weight"])})
    net.load_state_dict({k: T(v) for k, v in sw.items()})
    net.to(dev).eval(); net.drop_emb = False
    return net, opt
def run(net, b):
    q, ocr, od = [dict(x) for x in b[:3]]
    q.pop("_ruart_index", None)
    with torch.no_grad():
        s, _ = net(q, ocr, od)
    torch.cuda.synchronize()
    return s.float().cpu()
for prec in (os.environ.get("PRECISIONS", "fp32,fp16,fp16c").split(",")):
    for streams in (True, False):
        net, opt = make(prec, ruart_streams=streams)
        B = 64
        batch = synth.synthetic_batch(opt, B, seed=31, n_q=30, n_ocr=100, n_od=36, bert_vocab=3000, ragged=True)
        a = run(net, batch); b = run(net, batch); c = run(net, batch)
        print(prec, "streams", streams, "repeat diff %.3e %.3e" % (float((a - b).abs().max()), float((a - c).abs().max())), flush=True)
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).tolist()
        def permute_items(d):
            starts = np.concatenate([[0], np.cumsum(d["num_cnt"])])
            rows = np.concatenate([np.arange(starts[p], starts[p + 1]) for p in perm])
            out = {}
            for k, v in d.items():
                if k in ("num_cnt", "len_cnt"): out[k] = [v[p] for p in perm]
                elif k == "position": out[k] = v[perm]
                elif isinstance(v, torch.Tensor): out[k] = v[rows]
                else: out[k] = [v[r] for r in rows]
            return out
        q, ocr, od = batch[:3]
        qp = {k: (v[perm] if isinstance(v, torch.Tensor) else [v[p] for p in perm]) for k, v in q.items() if k != "_ruart_index"}
        sp = run(net, (qp, permute_items(ocr), permute_items(od)))
        d = (sp - a[perm]).abs()
        print(prec, "streams", streams, "perm diff %.3e at" % float(d.max()), np.unravel_index(int(d.argmax()), d.shape), "p=", float(a[perm].reshape(-1)[int(d.argmax())]), flush=True)
        del net
