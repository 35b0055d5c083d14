"""Seeded synthetic weights and batches in the reference's layouts.

There is no network on any box, so checkpoints and datasets are synthetic:
  * BERT weights under the HF-0.x names the reference loads
    (``Models/Bert/modeling.py:497-521``: ``bert.`` prefix, ``LayerNorm.gamma/beta``);
  * SDNet weights under the reference's state-dict names;
  * batches in the ``VQA_collate_fun`` 5-tuple layout
    (``Utils/VQA_Dataset.py:448-542``): dicts of int64 id matrices, bool masks,
    python offset / count lists, ``position (B, max_num, 8)`` and ``gt (B, No+1)``.

Everything is drawn from ``numpy.random.default_rng(seed)`` in a fixed order so the
build container (golden generator) and the GPU box regenerate identical arrays.
"""
import math

import numpy as np
import torch

BERT_BASE = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12,
                 num_attention_heads=12, intermediate_size=3072, hidden_act="gelu",
                 hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                 max_position_embeddings=512, type_vocab_size=2, initializer_range=0.02)


def bert_config(**kw):
    cfg = dict(BERT_BASE)
    cfg.update(kw)
    return cfg


def make_bert_weights(cfg, seed=1033, w_std=0.05):
    """name -> float32 array.  Linear / embedding weights N(0, w_std); LayerNorm
    gamma ~ 1 + 0.1 N, beta ~ 0.1 N; biases ~ 0.02 N (non-trivial on purpose so
    every epilogue term is exercised by parity tests)."""
    g = np.random.default_rng(seed)
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    w = {}

    def lin(name, o, i):
        w[name + ".weight"] = (g.standard_normal((o, i)) * w_std).astype(np.float32)
        w[name + ".bias"] = (g.standard_normal(o) * 0.02).astype(np.float32)

    def ln(name):
        w[name + ".gamma"] = (1.0 + 0.1 * g.standard_normal(H)).astype(np.float32)
        w[name + ".beta"] = (0.1 * g.standard_normal(H)).astype(np.float32)

    e = "bert.embeddings."
    w[e + "word_embeddings.weight"] = (g.standard_normal((cfg["vocab_size"], H)) * w_std).astype(np.float32)
    w[e + "position_embeddings.weight"] = (g.standard_normal((cfg["max_position_embeddings"], H)) * w_std).astype(np.float32)
    w[e + "token_type_embeddings.weight"] = (g.standard_normal((cfg["type_vocab_size"], H)) * w_std).astype(np.float32)
    ln(e + "LayerNorm")
    for l in range(cfg["num_hidden_layers"]):
        p = "bert.encoder.layer.%d." % l
        lin(p + "attention.self.query", H, H)
        lin(p + "attention.self.key", H, H)
        lin(p + "attention.self.value", H, H)
        lin(p + "attention.output.dense", H, H)
        ln(p + "attention.output.LayerNorm")
        lin(p + "intermediate.dense", I, H)
        lin(p + "output.dense", H, I)
        ln(p + "output.LayerNorm")
    lin("bert.pooler.dense", H, H)
    return w


def add_bert_outliers(w, cfg, seed=77, n_dims=4, gain=(10.0, 30.0), emb_scale=20.0, n_emb_dims=3):
    """Give seeded random encoder weights the heavy tails a PRETRAINED BERT has and N(0, s) weights lack (in place, deterministic):
    a few hidden dimensions - the same ones in every LayerNorm, as in the published checkpoints, where a handful of dimensions carry
    activations an order of magnitude above the rest - get their gain multiplied by U(gain), and a few columns of the word-embedding
    table are scaled by ``emb_scale``.  Used by the fp16c range test (tests/golden/sdnet_e2e_outliers.npz): LayerNorm outputs of
    30-100 and embedding entries of ~1 push the e4m3 correction operands (csrc/common.h: |v| < 112, |w| < 3.5) to their limits."""
    g = np.random.default_rng(seed)
    H = cfg["hidden_size"]
    dims = g.choice(H, size=n_dims, replace=False)
    for name in sorted(w):
        if name.endswith("LayerNorm.gamma"):
            w[name][dims] *= g.uniform(gain[0], gain[1], size=n_dims).astype(np.float32)
    edims = g.choice(H, size=n_emb_dims, replace=False)
    w["bert.embeddings.word_embeddings.weight"][:, edims] *= np.float32(emb_scale)
    return w


# --------------------------------------------------------------------------------------
# SDNet parameter table (names and shapes follow the reference constructor,
# Models/SDNet.py:21-251 and Models/Layers.py; verified against the instantiated
# reference by oracle/gen_golden.py).
# --------------------------------------------------------------------------------------
def sdnet_dims(opt):
    bert_dim, bert_layers = (1024, 24) if "BERT_LARGE" in opt else (768, 12)
    q_names = opt["q_embedding"].split(",")
    o_names = opt["ocr_embedding"].split(",")

    def width(names):
        n = 0
        n += opt["glove_dim"] if "glove" in names else 0
        n += opt["fast_dim"] if "fasttext" in names else 0
        n += int(opt["phoc_dim"]) if "phoc" in names else 0
        n += bert_dim if "bert" in names else 0
        n += opt["pos_dim"] if "pos" in names else 0
        n += opt["ent_dim"] if "ent" in names else 0
        return n

    x_in = width(o_names) + (300 if "PRE_ALIGN_befor_rnn" in opt else 0)
    q_in = width(q_names)
    h = opt["hidden_size"]
    hh = opt["highlvl_hidden_size"]
    m2o = opt["multi2one_hidden_size"] * (2 if opt["multi2one_bidir"] else 1)
    nl = opt["in_rnn_layers"]
    att_size = 2 * h * nl + m2o
    deep_rnn_in = 2 * h * nl * 2 + 2 * hh
    self_att_in = 2 * hh + deep_rnn_in + m2o
    return dict(bert_dim=bert_dim, bert_layers=bert_layers, x_in=x_in, q_in=q_in, h=h, hh=hh,
                m2o=m2o, nl=nl, att_size=att_size, deep_rnn_in=deep_rnn_in,
                self_att_in=self_att_in, ctx_final=2 * hh, q_final=2 * hh,
                ocr_final=4 * hh if opt["pos_att_merge_mod"] == "cat" else 2 * hh)


def sdnet_param_shapes(opt):
    d = sdnet_dims(opt)
    V = int(opt["vocab_size"])
    s = {}
    s["alphaBERT"] = (d["bert_layers"],)
    s["gammaBERT"] = (1, 1)
    s["fast_embed.weight"] = (V, opt["fast_dim"])
    s["glove_embed.weight"] = (V, opt["glove_dim"])
    if "PHOC" in opt:
        s["phoc_embed.weight"] = (V, int(opt["phoc_dim"]))      # values: phoc_vocab_words() through the PHOC builder, not random

    def attn(name, din, hid, similarity=False):
        s[name + ".scoring.linear.weight"] = (hid, din)
        s[name + ".scoring.diagonal"] = (1, 1, 1) if similarity else (1, 1, hid)

    def lstm(name, din, hid, layers, bidir=True):
        for l in range(layers):
            i = din if l == 0 else hid * (2 if bidir else 1)
            for sfx in ([""] + (["_reverse"] if bidir else [])):
                p = "%s.rnns.%d." % (name, l)
                s[p + "weight_ih_l0" + sfx] = (4 * hid, i)
                s[p + "weight_hh_l0" + sfx] = (4 * hid, hid)
                s[p + "bias_ih_l0" + sfx] = (4 * hid,)
                s[p + "bias_hh_l0" + sfx] = (4 * hid,)

    attn("pre_align", 300, opt["prealign_hidden"], True)
    s["pos_embedding.weight"] = (opt["pos_vocab_size"], opt["pos_dim"])
    s["ent_embedding.weight"] = (opt["ent_vocab_size"], opt["ent_dim"])
    lstm("multi2one", d["x_in"], opt["multi2one_hidden_size"], 1, bool(opt["multi2one_bidir"]))
    lstm("context_rnn", d["m2o"], d["h"], d["nl"])
    lstm("ques_rnn", d["q_in"], d["h"], d["nl"])
    for i in range(d["nl"] + 1):
        attn("deep_attn.int_attn_list.%d" % i, d["att_size"], opt["deep_att_hidden_size_per_abstr"])
    lstm("deep_attn.rnn", d["deep_rnn_in"], d["hh"], 1)
    lstm("high_lvl_ques_rnn", 2 * d["h"] * d["nl"], d["hh"], opt["question_high_lvl_rnn_layers"])
    attn("highlvl_self_att", d["self_att_in"], opt["deep_att_hidden_size_per_abstr"])
    lstm("high_lvl_context_rnn", 4 * d["hh"], d["hh"], 1)
    attn("ques_self_attn", d["q_final"], opt["query_self_attn_hidden_size"])
    attn("od_ocr_attn", d["ctx_final"], d["h"], True)
    attn("position_attn", opt["position_dim"], d["h"], True)
    s["ques_merger.linear.weight"] = (1, d["q_final"])
    s["ques_merger.linear.bias"] = (1,)
    xs, hs = d["ocr_final"], d["q_final"]
    s["get_answer.noanswer_linear.weight"] = (xs, hs)
    s["get_answer.noanswer_linear.bias"] = (xs,)
    s["get_answer.noanswer_w.weight"] = (1, xs)
    s["get_answer.noanswer_w.bias"] = (1,)
    s["get_answer.attn.linear.weight"] = (xs, hs)
    s["get_answer.attn.linear.bias"] = (xs,)
    s["get_answer.rnn.weight_ih"] = (3 * hs, xs)
    s["get_answer.rnn.weight_hh"] = (3 * hs, hs)
    s["get_answer.rnn.bias_ih"] = (3 * hs,)
    s["get_answer.rnn.bias_hh"] = (3 * hs,)
    s["get_answer.attn2.linear.weight"] = (xs, hs)
    s["get_answer.attn2.linear.bias"] = (xs,)
    return s


def make_sdnet_weights(opt, seed=1033):
    """name -> float32 array for every entry of sdnet_param_shapes(opt)."""
    g = np.random.default_rng(seed + 17)
    out = {}
    for name, shape in sdnet_param_shapes(opt).items():
        if name == "alphaBERT":
            a = 1.0 + 0.5 * g.standard_normal(shape)
        elif name == "gammaBERT":
            a = np.full(shape, 0.9)
        elif name.endswith("embed.weight"):
            a = g.standard_normal(shape)
            a[0] = 0.0
        elif name.endswith("scoring.diagonal"):
            a = (np.full(shape, 1.0 / math.sqrt(opt_hidden_for(name, opt))) if shape == (1, 1, 1)
                 else 1.0 + 0.2 * g.standard_normal(shape))
        elif name.endswith("embedding.weight"):
            a = g.standard_normal(shape)
        else:
            fan = shape[-1] if len(shape) > 1 else None
            if "rnn" in name:            # torch LSTM/GRU init: U(-1/sqrt(hidden), +)
                hid = shape[0] // (3 if name.startswith("get_answer.rnn") else 4)
                k = 1.0 / math.sqrt(hid)
            elif fan is not None:
                k = 1.0 / math.sqrt(fan)
            else:
                k = 0.05
            a = g.uniform(-k, k, shape)
        out[name] = np.ascontiguousarray(a, dtype=np.float32)
    return out


def opt_hidden_for(name, opt):
    if name.startswith("pre_align"):
        return opt["prealign_hidden"]
    return opt["hidden_size"]


# --------------------------------------------------------------------------------------
# Batches
# --------------------------------------------------------------------------------------
def _bertify(g, n_words, bert_vocab, max_bert_len, p2=0.4):
    """[CLS] pieces... [SEP]; returns (ids, offsets[[st, ed], ...]) like
    VQA_Dataset.bertify (Utils/VQA_Dataset.py:415-436)."""
    pieces = 1 + (g.random(n_words) < p2).astype(np.int64)
    while pieces.sum() > max_bert_len - 2:
        pieces[np.argmax(pieces)] -= 1
    ids = [101]
    offs = []
    for c in pieces:
        offs.append([len(ids), len(ids) + int(c)])
        ids.extend(int(v) for v in g.integers(1000 if bert_vocab > 1500 else 10, bert_vocab, size=int(c)))
    ids.append(102)
    return ids, offs


def phoc_vocab_words(V, seed=1033):
    """A deterministic spelling for every word id (ids 0..4 = padding / unknown / sentinels get the empty word): the PHOC
    table of a synthetic vocabulary is built from these, on the reference side with its ``build_phoc``, here with
    ``ruart_amd.phoc.phoc_table``."""
    g = np.random.default_rng(seed + 41)
    alpha = "abcdefghijklmnopqrstuvwxyz0123456789"
    return ["" if i < 5 else "".join(alpha[int(k)] for k in g.integers(0, 36, size=int(g.integers(1, 13)))) for i in range(V)]


def synthetic_batch(opt, B, seed=7, n_q=30, n_ocr=100, n_od=None, bert_vocab=30522,
                    ragged=False, targets=True):
    """One batch in VQA_collate_fun layout.  ``n_ocr`` / ``n_od`` count items *including*
    the trailing <OCR>/<OD> sentinel (word id 3 / 4, Utils/VQA_Dataset.py:336-349).
    ``ragged=True`` draws a different item count per sample (>= ES_ocr_len + 2)."""
    g = np.random.default_rng(seed)
    V = int(opt["vocab_size"])
    n_od = opt["max_od_num"] if n_od is None else n_od
    Q, Qb = opt["max_q_len"], opt["max_q_bert_len"]
    n_q = min(n_q, Q)

    def fill(rows, width, dtype=np.int64):
        a = np.zeros((len(rows), width), dtype=dtype)
        for i, r in enumerate(rows):
            a[i, :len(r)] = r
        return a

    # ---- question ----
    q_word, q_pos, q_ent, q_bert, q_off = [], [], [], [], []
    for _ in range(B):
        nq = int(g.integers(max(2, n_q - 6), n_q + 1)) if ragged else n_q
        q_word.append(g.integers(5, V, size=nq))
        q_pos.append(g.integers(0, opt["pos_vocab_size"], size=nq))
        q_ent.append(g.integers(0, opt["ent_vocab_size"], size=nq))
        ids, offs = _bertify(g, nq, bert_vocab, Qb, p2=0.3)
        q_bert.append(ids)
        q_off.append(offs)
    q = {}
    q["glove"] = torch.from_numpy(fill(q_word, Q))
    q["glove_mask"] = ~q["glove"].eq(0)
    q["pos"] = torch.from_numpy(fill(q_pos, Q))
    q["ent"] = torch.from_numpy(fill(q_ent, Q))
    q["bert"] = torch.from_numpy(fill(q_bert, Qb))
    q["bert_mask"] = ~q["bert"].eq(0)
    q["bert_offsets"] = q_off

    def items(n_items, max_num, max_len, max_bert_len, sentinel, max_words):
        word, pos, ent, bert, off = [], [], [], [], []
        num_cnt, len_cnt = [], []
        position = np.zeros((B, max_num, 8), dtype=np.float32)
        for b in range(B):
            if ragged and b > 0:      # sample 0 stays at the maximum item count
                lo = min(max_num, opt.get("ES_ocr_len", 0) + 2) if sentinel == 3 else 1
                n = int(g.integers(lo, min(n_items, max_num) + 1))
            else:
                n = min(n_items, max_num)
            lens = []
            for k in range(n):
                last = (k == n - 1)
                nw = 1 if last else int(g.integers(1, max_words + 1))
                nw = min(nw, max_len)
                word.append([sentinel] if last else g.integers(5, V, size=nw))
                pos.append([0] if last else g.integers(0, opt["pos_vocab_size"], size=nw))
                ent.append([0] if last else g.integers(0, opt["ent_vocab_size"], size=nw))
                ids, offs = _bertify(g, nw, bert_vocab, max_bert_len)
                bert.append(ids)
                off.append(offs)
                lens.append(nw)
                if not last:
                    position[b, k] = g.random(8, dtype=np.float32)
            num_cnt.append(n)
            len_cnt.append(lens)
        d = {}
        d["fasttext"] = torch.from_numpy(fill(word, max_len))
        d["pos"] = torch.from_numpy(fill(pos, max_len))
        d["ent"] = torch.from_numpy(fill(ent, max_len))
        d["bert"] = torch.from_numpy(fill(bert, max_bert_len))
        d["bert_offsets"] = off
        d["position"] = torch.from_numpy(position)
        d["fasttext_mask"] = ~d["fasttext"].eq(0)
        if "phoc" in opt["ocr_embedding"].split(","):            # the PHOC table is indexed by the same word ids
            d["phoc"], d["phoc_mask"] = d["fasttext"].clone(), d["fasttext_mask"].clone()
        d["bert_mask"] = ~d["bert"].eq(0)
        d["num_cnt"] = num_cnt
        d["len_cnt"] = len_cnt
        return d

    ocr = items(n_ocr, opt["max_ocr_num"], opt["max_ocr_len"], opt["max_ocr_bert_len"], 3, 3)
    od = items(n_od, opt["max_od_num"], opt["max_od_len"], opt["max_od_bert_len"], 4, 2)

    gt = None
    if targets:
        gt = torch.zeros(B, opt["max_ocr_num"] + 1)
        for b in range(B):
            gt[b, int(g.integers(0, max(1, ocr["num_cnt"][b] - 1)))] = 1.0
    extra = [{"q_id": b, "answers": None,
              "ocr_list": ["w%d" % k for k in range(ocr["num_cnt"][b] - 1)] + ["<OCR>"],
              "image_path": "synthetic/%d.jpg" % b} for b in range(B)]
    return q, ocr, od, gt, extra
