#!/usr/bin/env python3
"""Host side of the training loop without a GPU: records -> VQA_Dataset.__getitem__ (word pieces, labels) -> VQA_collate with
prepare_index=True (padded id matrices + the hot path's batch index), at the bench's sizes (100 OCR items, 36 objects per
sample, B = 64).  Prints ms per sample / per batch and the number of DataLoader workers a 20 ms step needs.
    python tools/loader_rate.py [--batch 64] [--batches 4]"""
import argparse, os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd.arguments import default_opt
from ruart_amd.batch import VQA_collate
from ruart_amd.dataset import VQA_Dataset

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--batches", type=int, default=4)
a = ap.parse_args()
g = np.random.default_rng(0)
alpha = "abcdefghijklmnopqrstuvwxyz"
words = ["".join(alpha[int(k)] for k in g.integers(0, 26, size=int(g.integers(2, 9)))) for _ in range(4000)]
pieces = sorted({w[:k] for w in words for k in range(1, len(w) + 1)} | {"##" + w[k:] for w in words for k in range(1, len(w))})
vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + [p for p in pieces if len(p.replace("##", "")) <= 4][:28000]
tmp = tempfile.mkdtemp()
with open(os.path.join(tmp, "vocab.txt"), "w") as f:
    f.write("\n".join(vocab) + "\n")

def annotated(ws):
    return {"word": ws, "wordid": g.integers(5, 19000, size=len(ws)).tolist(), "pos_id": g.integers(0, 51, size=len(ws)).tolist(),
            "ent_id": g.integers(0, 75, size=len(ws)).tolist()}

def item(kind):
    ws = [words[int(k)] for k in g.integers(0, len(words), size=int(g.integers(1, 4 if kind == "ocr" else 3)))]
    d = {"original": " ".join(ws), "pos": g.random(8).round(4).tolist(), "ANLS": float(g.random() ** 3), "ACC": 0.0, "cnt": 1, "idx": 0}
    d["word" if kind == "ocr" else "object"] = annotated(ws)
    return d

recs = []
for i in range(a.batch * a.batches):
    q = [words[int(k)] for k in g.integers(0, len(words), size=14)]
    recs.append({"question_id": i, "question": " ".join(q), "filename": "x.jpg", "orign_answers": ["stop"], "annotated_question": annotated(q),
                 "ocr_PMTD_ASTER": [item("ocr") for _ in range(88)], "ocr_PMTD_ASTER_gram2": [item("ocr") for _ in range(4)],
                 "ES_ocr": [item("ocr") for _ in range(10)], "OD_bottom-up": [item("od") for _ in range(35)]})
opt = default_opt(datadir="", BERT_tokenizer_file=os.path.join(tmp, "vocab.txt"), max_od_num=36, vocab_size=20000, ruart_cache_samples=True)
ds = VQA_Dataset(recs, opt)
coll = VQA_collate(opt, prepare_index=True).VQA_collate_fun
plain = VQA_Dataset(recs, {k: v for k, v in opt.items() if k != "ruart_cache_samples"})     # full samples (item lists kept)
warm = [ds[i] for i in range(a.batch)]               # first-call costs (thread pools, imports, allocator growth) stay out of the timing
coll(warm)
walk_samples = [{k: v for k, v in plain[i].items() if k != "_flat"} for i in range(a.batch * a.batches)]
coll(walk_samples[:a.batch])
ds._cache.clear()
t0 = time.perf_counter()
samples = [ds[i] for i in range(len(ds))]
t_item = (time.perf_counter() - t0) / len(ds)
t0 = time.perf_counter()
again = [ds[i] for i in range(len(ds))]               # a later epoch: finished samples come from the per-index cache
t_cached = (time.perf_counter() - t0) / len(ds)
def timed(fast):
    best = 1e9
    for _ in range(3):                                # best of three: the first pass after a change of path pays allocator growth
        t0 = time.perf_counter()
        for b in range(a.batches):
            coll((samples if fast else walk_samples)[b * a.batch:(b + 1) * a.batch])
        best = min(best, (time.perf_counter() - t0) / a.batches)
    return best


t_walk = timed(False)
t_coll = timed(True)
print("items per sample: ocr %d, od %d" % (len(plain[0]["ocr"]), len(plain[0]["od"])))
print("__getitem__ %.2f ms/sample first visit, %.3f ms cached;  collate + batch index %.1f ms/batch (%.1f ms walking the item dicts)"
      % (t_item * 1e3, t_cached * 1e3, t_coll * 1e3, t_walk * 1e3))
for label, ti in (("first epoch", t_item), ("later epochs (ruart_cache_samples)", t_cached)):
    per_batch = ti * a.batch + t_coll
    print("%-36s %.1f ms/batch: one worker feeds %.0f samples/s; a 20 ms step (3 200 samples/s) needs %d worker(s)"
          % (label, per_batch * 1e3, a.batch / per_batch, int(np.ceil(per_batch / 0.020))))
