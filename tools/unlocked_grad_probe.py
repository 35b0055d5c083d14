import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_sdnet as t
from ruart_amd import synth
import ruart_amd.layers as L
z = np.load("tests/golden/sdnet_e2e_unlocked.npz")
mode = sys.argv[1] if len(sys.argv) > 1 else "16"
net, opt = t._unlocked(z, "x3", bert_train_gemm=mode)
names = dict(net.named_parameters())
q, ocr, od, gt, _ = synth.synthetic_batch(opt, int(z["B"]), seed=int(z["batch_seed"]), n_q=12, n_ocr=16, n_od=6, bert_vocab=2000, ragged=True)
L.set_dropout_prob(0.0); net.train(); net.drop_emb = False
scores, _ = net(q, ocr, od)
print("max dp", np.abs(scores.detach().cpu().numpy() - z["scores"]).max())
gt = gt.to(scores.device)
loss = torch.nn.functional.binary_cross_entropy_with_logits(scores, gt) * gt.size(1)
loss.backward()
rows = []
for name, ref in zip(z["grad_names"].tolist(), z["grad_norms"].tolist()):
    g = names[name].grad
    if g is None or ref < 0: continue
    got = float(g.double().norm())
    rows.append((abs(got - ref) / max(ref, 1e-4), name, got, ref))
rows.sort(reverse=True)
for r in rows[:25]: print("%.3e %-70s got %.4e ref %.4e" % r)
print("median rel", np.median([r[0] for r in rows]), "n", len(rows))
for k in z.files:
    if k.startswith("grad:Bert"):
        name = k[5:]; ref = z[k]; g = names[name].grad
        got = g[tuple(slice(0, n) for n in ref.shape)].detach().cpu().numpy()
        print(name, "slice max err / max |ref|", np.abs(got - ref).max() / np.abs(ref).max())
