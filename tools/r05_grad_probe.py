import os, sys
import torch
sys.path.insert(0, "/root/repo")
import bench
from ruart_amd import synth
from ruart_amd.arguments import default_opt
import ruart_amd.layers as L
dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
b = tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7, n_q=30, n_ocr=100, n_od=36))
L.set_dropout_prob(0.0)
net = tr.network
def run(streams):
    net.opt["ruart_streams"] = streams
    net.train(); net.drop_emb = False
    net.zero_grad(set_to_none=True)
    def f():
        s = net(b[0], b[1], b[2])[0]
        loss = tr.loss_func(s, b[3])
        loss.backward()
        return s
    s = tr.on_step_stream(f)
    torch.cuda.synchronize()
    return s.detach().float().cpu(), {n: p.grad.detach().float().cpu().clone() for n, p in net.named_parameters() if p.grad is not None}
s1, g1 = run(False)
for k in range(3):
    s3, g3 = run(True)
    worst = max(((g3[n] - g1[n]).abs().max().item(), n) for n in g1)
    nz = sum(1 for n in g1 if not torch.equal(g3[n], g1[n]))
    print("run %d: scores equal %s; grads differing %d of %d, worst %.3e (%s)" % (k, torch.equal(s3, s1), nz, len(g1), worst[0], worst[1]), flush=True)
s1b, g1b = run(False)
print("one-stream repeat: grads differing", sum(1 for n in g1 if not torch.equal(g1b[n], g1[n])))
tr.close(final=True)
