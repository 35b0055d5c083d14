#!/usr/bin/env python3
"""Time the UNMODIFIED reference's CPU path next to the CPU restatement (oracle) on the same synthetic samples, in the build
container (the reference cannot travel to the GPU box; there the oracle is the `cpu_baseline` of bench.py, kind "port").

TEST INFRASTRUCTURE ONLY.   python oracle/time_reference.py [--samples 4] [--threads 8]
Both run forward + BCE_D1 loss + backward once on a B-sample batch of the bench workload (q=30, 100 OCR items, 36 objects,
bert-base with seeded random weights), dropout configured to 0."""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import _refshim  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=4)
ap.add_argument("--threads", type=int, default=8)
a = ap.parse_args()
torch.set_num_threads(a.threads)
_refshim.install()
from oracle import ruart_oracle as O  # noqa: E402
from ruart_amd import synth  # noqa: E402
from ruart_amd.arguments import default_opt  # noqa: E402


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


opt = default_opt(vocab_size=20000, max_od_num=36, DROPOUT=0.0, dropout_emb=0.0)
cfg = synth.bert_config(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
bw = synth.make_bert_weights(cfg, seed=1033, w_std=0.02)
sw = synth.make_sdnet_weights(opt, seed=1033)
batch = synth.synthetic_batch(opt, a.samples, seed=7, n_q=30, n_ocr=100, n_od=36)

# ---- the reference itself -----------------------------------------------------------------------------------------------
from Models.SDNet import SDNet  # noqa: E402
import Models.Layers as L  # noqa: E402
ropt = dict(opt)
ropt["BERT_model_file"] = _refshim.write_bert_dir(cfg, bw)
ropt["datadir"] = ""
net = SDNet(ropt, {"glove_embedding": T(sw["glove_embed.weight"]).clone(), "fast_embedding": T(sw["fast_embed.weight"]).clone()})
net.load_state_dict({k: T(v) for k, v in sw.items()}, strict=False)
L.set_dropout_prob(0.0)
net.train()
net.Bert.bert_model.eval()
net.drop_emb = False
q, ocr, od, gt, _ = batch
t0 = time.perf_counter()
scores, _ = net(q, ocr, od)
loss = torch.nn.functional.binary_cross_entropy_with_logits(scores, gt) * gt.size(1)
loss.backward()
t_ref = time.perf_counter() - t0
ref_scores = scores.detach().clone()

# ---- the restatement ----------------------------------------------------------------------------------------------------
P = {k: T(v).requires_grad_(v.shape != (1, 1, 1)) for k, v in sw.items()}
bwt = {k: T(v) for k, v in bw.items()}
t0 = time.perf_counter()
s2 = O.sdnet_forward(P, opt, bwt, cfg, q, ocr, od)
l2 = O.instance_bce_with_logits(s2, gt)
l2.backward()
t_or = time.perf_counter() - t0
print("reference: %.2f s for %d samples = %.3f samples/s;  oracle: %.2f s = %.3f samples/s  (%d threads);  max |p_ref - p_oracle| = %.2e"
      % (t_ref, a.samples, a.samples / t_ref, t_or, a.samples / t_or, a.threads, float((ref_scores - s2.detach()).abs().max())))
