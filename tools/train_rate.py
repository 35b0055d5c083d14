#!/usr/bin/env python3
"""The whole training loop on real files, as the reference runs it: synthetic preprocessed records (100 OCR items + 36 objects per
sample) written as {train,val}-preprocessed.msgpack + train_meta.msgpack, then ``SDNetTrainer.train()`` - VQA_Dataset in DataLoader
workers, collate with the host-built batch index, pinned batches, ToCUDA, update with the next batch's encoder pass one step ahead.
Prints the steady-state step time per epoch.   python tools/train_rate.py [--workers 8] [--samples 1024] [--epochs 3]"""
import argparse, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--workers", type=int, default=8)
ap.add_argument("--samples", type=int, default=1024)
ap.add_argument("--epochs", type=int, default=3)
ap.add_argument("--batch", type=int, default=64)
a = ap.parse_args()

import msgpack, torch
from ruart_amd import synth
from ruart_amd.arguments import default_opt
from ruart_amd.trainer import SDNetTrainer

g = np.random.default_rng(0)
alpha = "abcdefghijklmnopqrstuvwxyz"
words = ["".join(alpha[int(k)] for k in g.integers(0, 26, size=int(g.integers(2, 9)))) for _ in range(4000)]
pieces = sorted({w[:k] for w in words for k in range(1, len(w) + 1)} | {"##" + w[k:] for w in words for k in range(1, len(w))})
vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + [p for p in pieces if len(p.replace("##", "")) <= 4][:28000]
V = 5000

def annotated(ws):
    return {"word": ws, "wordid": g.integers(5, V, size=len(ws)).tolist(), "pos_id": g.integers(0, 51, size=len(ws)).tolist(),
            "ent_id": g.integers(0, 75, size=len(ws)).tolist()}

def item(kind):
    ws = [words[int(k)] for k in g.integers(0, len(words), size=int(g.integers(1, 4 if kind == "ocr" else 3)))]
    d = {"original": " ".join(ws), "pos": g.random(8).round(4).tolist(), "ANLS": float(g.random() ** 3), "ACC": 0.0, "cnt": 1, "idx": 0}
    d["word" if kind == "ocr" else "object"] = annotated(ws)
    return d

def record(i):
    q = [words[int(k)] for k in g.integers(0, len(words), size=14)]
    return {"question_id": i, "question": " ".join(q), "filename": "x.jpg", "orign_answers": ["stop"], "annotated_question": annotated(q),
            "ocr_PMTD_ASTER": [item("ocr") for _ in range(88)], "ocr_PMTD_ASTER_gram2": [item("ocr") for _ in range(4)],
            "ES_ocr": [item("ocr") for _ in range(10)], "OD_bottom-up": [item("od") for _ in range(35)]}

tmp = tempfile.mkdtemp()
feat = os.path.join(tmp, "source", "data", "synth")
os.makedirs(feat)
with open(os.path.join(tmp, "vocab.txt"), "w") as f:
    f.write("\n".join(vocab) + "\n")
t0 = time.time()
recs = [record(i) for i in range(a.samples)]
with open(os.path.join(feat, "train-preprocessed.msgpack"), "wb") as f:
    msgpack.dump({"data": recs}, f)
with open(os.path.join(feat, "val-preprocessed.msgpack"), "wb") as f:
    msgpack.dump({"data": recs[:a.batch]}, f)
with open(os.path.join(feat, "train_meta.msgpack"), "wb") as f:
    msgpack.dump({"vocab": ["w%d" % i for i in range(V)], "char_vocab": list("abc"),
                  "glove_embedding": g.standard_normal((V, 300)).astype(np.float32).tolist(),
                  "fast_embedding": g.standard_normal((V, 300)).astype(np.float32).tolist()}, f)
print("files written in %.1f s (%d records)" % (time.time() - t0, a.samples), flush=True)

opt = default_opt(cuda=True, datadir=tmp, source_dir="synth", BERT_tokenizer_file="vocab.txt", batch_size=a.batch, max_od_num=36,
                  num_worker=a.workers, epoch=a.epochs, ruart_cache_samples=True)
for k in ("RESUME", "vocab_size"):
    opt.pop(k, None)
cfg = synth.bert_config(vocab_size=len(vocab))
opt["bert_state"], opt["bert_config"] = synth.make_bert_weights(cfg, seed=1033, w_std=0.02), cfg
tr = SDNetTrainer(opt, device="cuda:0")
stamps = []
orig = tr.update
def update(batch, i, next_batch=None):
    r = orig(batch, i, next_batch=next_batch)
    stamps.append(time.perf_counter())
    return r
tr.update = update
t0 = time.time()
tr.train(eval_every=10 ** 9)
per_epoch = a.samples // a.batch
d = np.diff(np.array(stamps)) * 1e3
print("%d steps in %.1f s" % (len(stamps), time.time() - t0))
for e in range(a.epochs):
    seg = d[max(e * per_epoch, 1) - 1:(e + 1) * per_epoch - 1]
    if len(seg):
        print("epoch %d: median step %.1f ms (%.0f samples/s), mean %.1f ms" % (e, np.median(seg), a.batch / np.median(seg) * 1e3, seg.mean()))
