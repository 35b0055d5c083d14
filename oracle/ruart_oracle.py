"""CPU restatement (fp32, plain torch-CPU tensor ops) of RUArt's hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module, and only as the checker / CPU baseline.
The product (ruart_amd/) never imports it.

Parity status: PINNED.  The reference ships no tests, so this
restatement is pinned against outputs of the reference itself run in the build
container: tests/golden/*.npz written by oracle/gen_golden.py, checked by
tests/test_oracle_golden.py (forward <= 2e-5 abs on probabilities, gradients
<= 1e-4 relative).

Every function cites the reference lines (relative to /root/reference) it follows.
Weights are addressed by the reference's state-dict names.  Dropout is the identity
here (the reference's eval path / p = 0); the variational-dropout contract is tested
separately on the product side.
"""
import math

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------
# BERT encoder  (Models/Bert/modeling.py)
# ------------------------------------------------------------------------------------
def bert_layer_norm(x, gamma, beta, eps=1e-12):
    """modeling.py:164-168 - TF-style LN, eps inside the sqrt, biased variance."""
    u = x.mean(-1, keepdim=True)
    s = ((x - u) ** 2).mean(-1, keepdim=True)
    return gamma * ((x - u) / torch.sqrt(s + eps)) + beta


def gelu_erf(x):
    """modeling.py:52-57 - exact erf form."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def bert_forward(w, cfg, ids, mask):
    """modeling.py:585-614 (+ :185-199 embeddings, :224-250 self-attention,
    :260-264 / :299-303 dense+residual+LN, :286-289 FFN+GELU).  Returns the list of
    all layer outputs, each (N, L, H).  token_type_ids are all zero on this path
    (Models/Bert/Bert.py:136)."""
    H = cfg["hidden_size"]
    nh = cfg["num_attention_heads"]
    dh = H // nh
    N, L = ids.shape
    e = "bert.embeddings."
    x = (w[e + "word_embeddings.weight"][ids]
         + w[e + "position_embeddings.weight"][torch.arange(L)].unsqueeze(0)
         + w[e + "token_type_embeddings.weight"][0])
    x = bert_layer_norm(x, w[e + "LayerNorm.gamma"], w[e + "LayerNorm.beta"])
    add_mask = (1.0 - mask.to(torch.float32))[:, None, None, :] * -10000.0      # :596-604
    outs = []
    for l in range(cfg["num_hidden_layers"]):
        p = "bert.encoder.layer.%d." % l

        def lin(name, t):
            return F.linear(t, w[p + name + ".weight"], w[p + name + ".bias"])

        def heads(t):
            return t.view(N, L, nh, dh).permute(0, 2, 1, 3)

        q, k, v = heads(lin("attention.self.query", x)), heads(lin("attention.self.key", x)), heads(lin("attention.self.value", x))
        s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh) + add_mask
        ctx = torch.matmul(torch.softmax(s, dim=-1), v).permute(0, 2, 1, 3).reshape(N, L, H)
        a = bert_layer_norm(lin("attention.output.dense", ctx) + x,
                            w[p + "attention.output.LayerNorm.gamma"], w[p + "attention.output.LayerNorm.beta"])
        f = gelu_erf(lin("intermediate.dense", a))
        x = bert_layer_norm(lin("output.dense", f) + a,
                            w[p + "output.LayerNorm.gamma"], w[p + "output.LayerNorm.beta"])
        outs.append(x)
    return outs


def pool_subwords(layers, offsets, word_mask):
    """Models/Bert/Bert.py:149-165 - per (item, word): masked word -> zeros;
    span [st, ed): one piece -> copy, several -> mean, empty/inverted -> zeros.
    ``layers``: list of (N, L, H); returns list of (N, Lw, H)."""
    N, Lw = word_mask.shape
    outs = [torch.zeros(N, Lw, l.shape[-1]) for l in layers]
    for i in range(N):
        for j in range(Lw):
            if not bool(word_mask[i, j]):
                continue
            st, ed = offsets[i][j]
            if ed <= st:
                continue
            for o, l in zip(outs, layers):
                o[i, j] = l[i, st:ed].sum(0) / float(ed - st) if ed - st > 1 else l[i, st]
    return outs


def linear_sum(pooled, alpha, gamma):
    """Models/SDNet.py:573-583 (dropout = identity)."""
    a = torch.softmax(alpha, dim=0)
    res = 0
    for i, t in enumerate(pooled):
        res = res + t * a[i] * gamma
    return res


# ------------------------------------------------------------------------------------
# Layers  (Models/Layers.py)
# ------------------------------------------------------------------------------------
def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse=False):
    """One nn.LSTM direction, batch_first, zero initial state; gate order i, f, g, o
    (what Layers.py:137,166 delegates to torch).  Padding is NOT masked."""
    B, T, _ = x.shape
    Hh = w_hh.shape[1]
    h = x.new_zeros(B, Hh)
    c = x.new_zeros(B, Hh)
    xp = F.linear(x, w_ih, b_ih)
    ys = [None] * T
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        g = xp[:, t] + F.linear(h, w_hh, b_hh)
        i, f, gg, o = g.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        ys[t] = h
    return torch.stack(ys, 1)


def whole_tensor_layer_norm(x, eps=1e-5):
    """Layers.py:167-168 - F.layer_norm over the WHOLE (B, L, D) tensor, no affine."""
    m = x.mean()
    v = ((x - m) ** 2).mean()
    return (x - m) / torch.sqrt(v + eps)


def stacked_brnn(P, prefix, x, num_layers, bidirectional=True, LN=False, concat=False):
    """Layers.py:156-180.  Returns (output, [per-layer outputs])."""
    hid = [x]
    for l in range(num_layers):
        p = "%s.rnns.%d." % (prefix, l)
        inp = hid[-1]
        y = lstm_direction(inp, P[p + "weight_ih_l0"], P[p + "weight_hh_l0"], P[p + "bias_ih_l0"], P[p + "bias_hh_l0"])
        if bidirectional:
            yr = lstm_direction(inp, P[p + "weight_ih_l0_reverse"], P[p + "weight_hh_l0_reverse"],
                                P[p + "bias_ih_l0_reverse"], P[p + "bias_hh_l0_reverse"], reverse=True)
            y = torch.cat([y, yr], 2)
        if LN:
            y = whole_tensor_layer_norm(y)
        hid.append(y)
    out = torch.cat(hid[1:], 2) if concat else hid[-1]
    return out, hid[1:]


def attention(x1, x2, x2_mask, W, diag, x3=None):
    """Layers.py:208-245 (correlation_func 3) + :265-295.
    s_ij = (ReLU(W x1_i) * d) . ReLU(W x2_j); masked keys -> -inf; softmax over j; alpha @ x3."""
    a = torch.relu(F.linear(x1, W)) * diag
    b = torch.relu(F.linear(x2, W))
    s = torch.bmm(a, b.transpose(1, 2))
    s = s.masked_fill(x2_mask.eq(0).unsqueeze(1), float("-inf"))
    alpha = torch.softmax(s, dim=2)
    return torch.bmm(alpha, x2 if x3 is None else x3)


def module_attention(P, prefix, x1, x2, x2_mask, x3=None):
    return attention(x1, x2, x2_mask, P[prefix + ".scoring.linear.weight"], P[prefix + ".scoring.diagonal"], x3)


def deep_attention(P, prefix, x1_word, x1_abstr, x2_word, x2_abstr, x2_mask):
    """Layers.py:493-524.  Returns (BiLSTM output, pre-rnn concat)."""
    x1_att = torch.cat(x1_word + x1_abstr, 2)
    x2_att = torch.cat(x2_word + x2_abstr[:-1], 2)
    x1 = torch.cat(x1_abstr, 2)
    for i, x2_i in enumerate(x2_abstr):
        x1 = torch.cat([x1, module_attention(P, "%s.int_attn_list.%d" % (prefix, i), x1_att, x2_att, x2_mask, x3=x2_i)], 2)
    y, _ = stacked_brnn(P, prefix + ".rnn", x1, 1)
    return y, x1


def linear_self_attn(x, x_mask, w, b):
    """Layers.py:328-341."""
    s = F.linear(x, w, b).squeeze(-1)
    s = s.masked_fill(x_mask.eq(0), float("-inf"))
    return torch.softmax(s, dim=1)


def weighted_avg(x, weights):
    """Layers.py:529-534."""
    return torch.bmm(weights.unsqueeze(1), x).squeeze(1)


def bilinear_seq_attn(x, y, x_mask, w, b, mask_flag=True):
    """Layers.py:446-468."""
    s = torch.bmm(x, F.linear(y, w, b).unsqueeze(2)).squeeze(2)
    if mask_flag:
        s = s.masked_fill(x_mask.eq(0), float("-inf"))
    return s


def get_final_scores(P, prefix, x, h0, x_mask, ES_len, mask_flag=True):
    """Layers.py:373-432, useES + no_answer branch.  The GRU step (:395-397) has no
    effect on the result and is omitted; its parameters receive no gradient."""
    x_es, x_ocr = x[:, :ES_len], x[:, ES_len:]
    m_es, m_ocr = x_mask[:, :ES_len], x_mask[:, ES_len:]
    s_ocr = bilinear_seq_attn(x_ocr, h0, m_ocr, P[prefix + ".attn.linear.weight"], P[prefix + ".attn.linear.bias"], mask_flag)
    s_es = bilinear_seq_attn(x_es, h0, m_es, P[prefix + ".attn2.linear.weight"], P[prefix + ".attn2.linear.bias"], mask_flag)
    s = torch.cat([s_es, s_ocr], -1)
    # get_single_score (:421-432): always masked
    wh = F.linear(h0, P[prefix + ".noanswer_linear.weight"], P[prefix + ".noanswer_linear.bias"])
    xwh = torch.bmm(x, wh.unsqueeze(2)).squeeze(2).masked_fill(x_mask.eq(0), float("-inf"))
    pooled = torch.bmm(torch.softmax(xwh, 1).unsqueeze(1), x)
    s_na = F.linear(pooled, P[prefix + ".noanswer_w.weight"], P[prefix + ".noanswer_w.bias"]).squeeze(2)
    return torch.softmax(torch.cat([s, s_na], -1), dim=-1)


def instance_bce_with_logits(scores, labels, d1=True):
    """Models/SDNetTrainer.py:510-518 - the probabilities are fed as *logits* (SURVEY §0.6)."""
    loss = F.binary_cross_entropy_with_logits(scores, labels)
    return loss * labels.size(1) if d1 else loss


# ------------------------------------------------------------------------------------
# SDNet.forward  (Models/SDNet.py:253-437, shipped-conf branches)
# ------------------------------------------------------------------------------------
def _embed(P, opt, bert_w, bert_cfg, items, names, word_key):
    """SDNet.py:439-493 for one of q / ocr / od.  Returns (concat embedding, raw word vectors)."""
    parts = []
    if "phoc" in names:                                  # :441-446, first in the concatenation
        parts.append(P["phoc_embed.weight"][items["phoc"]])
    table = P["fast_embed.weight"] if word_key == "fasttext" else P["glove_embed.weight"]
    wv = table[items[word_key]]
    parts.append(wv)
    if "bert" in names:
        layers = bert_forward(bert_w, bert_cfg, items["bert"], items["bert_mask"])
        pooled = pool_subwords(layers, items["bert_offsets"], items[word_key + "_mask"])
        parts.append(linear_sum(pooled, P["alphaBERT"], P["gammaBERT"]))
    if "pos" in names:
        parts.append(P["pos_embedding.weight"][items["pos"]])
    if "ent" in names:
        parts.append(P["ent_embedding.weight"][items["ent"]])
    return torch.cat(parts, -1), wv


def _prealign(P, wv_items, len_cnt, q_wv, q_mask):
    """SDNet.py:495-551 for one of ocr / od: re-pack each sample's real words into one row,
    attend over the question word vectors, scatter back to (items, Lw, 300)."""
    B = len(len_cnt)
    tmax = max(sum(l) for l in len_cnt)
    packed = wv_items.new_zeros(B, tmax, wv_items.shape[-1])
    idx = 0
    for i in range(B):
        c = 0
        for j in len_cnt[i]:
            packed[i, c:c + j] = wv_items[idx, :j]
            c += j
            idx += 1
    att = module_attention(P, "pre_align", packed, q_wv, q_mask)
    out = torch.zeros_like(wv_items)
    rows, cols, src_b, src_t = [], [], [], []
    idx = 0
    for i in range(B):
        c = 0
        for j in len_cnt[i]:
            for k in range(j):
                rows.append(idx); cols.append(k); src_b.append(i); src_t.append(c + k)
            c += j
            idx += 1
    out = out.index_put((torch.tensor(rows), torch.tensor(cols)), att[torch.tensor(src_b), torch.tensor(src_t)])
    return out, att


def _items_to_samples(m2o, num_cnt, len_cnt, max_num):
    """SDNet.py:288-318: state at each item's last real word -> (B, max_num, D), mask."""
    B = len(num_cnt)
    rows, lasts, bi, ki = [], [], [], []
    idx = 0
    for i in range(B):
        for k, j in enumerate(len_cnt[i]):
            rows.append(idx); lasts.append(j - 1); bi.append(i); ki.append(k)
            idx += 1
    out = m2o.new_zeros(B, max_num, m2o.shape[-1])
    out = out.index_put((torch.tensor(bi), torch.tensor(ki)), m2o[torch.tensor(rows), torch.tensor(lasts)])
    mask = torch.zeros(B, max_num, dtype=torch.uint8)
    for i in range(B):
        mask[i, :len(len_cnt[i])] = 1
    return out, mask


def sdnet_forward(P, opt, bert_w, bert_cfg, q, ocr, od, caps=None):
    """Returns score_s (B, max_ocr_num + 1).  ``caps`` (dict) receives named intermediates."""
    def cap(k, v):
        if caps is not None:
            i = 0
            while "%s#%d" % (k, i) in caps:
                i += 1
            caps["%s#%d" % (k, i)] = v.detach()

    qn = opt["q_embedding"].split(",")
    on = opt["ocr_embedding"].split(",")
    q_in, q_wv = _embed(P, opt, bert_w, bert_cfg, q, qn, opt["q_emb_initial"])
    ocr_in, ocr_wv = _embed(P, opt, bert_w, bert_cfg, ocr, on, opt["ocr_emb_initial"])
    od_in, od_wv = _embed(P, opt, bert_w, bert_cfg, od, on, opt["ocr_emb_initial"])
    for v in (q_in, ocr_in, od_in):
        cap("embed", v)
    q_mask = q[opt["q_emb_initial"] + "_mask"]

    ocr_pa, ocr_att = _prealign(P, ocr_wv, ocr["len_cnt"], q_wv, q_mask)        # :265-268
    od_pa, od_att = _prealign(P, od_wv, od["len_cnt"], q_wv, q_mask)
    cap("pre_align", ocr_att); cap("pre_align", od_att)
    ocr_in = torch.cat([ocr_in, ocr_pa], -1)
    od_in = torch.cat([od_in, od_pa], -1)

    m2o_ocr, _ = stacked_brnn(P, "multi2one", ocr_in, 1, bidirectional=bool(opt["multi2one_bidir"]))   # :269-271
    m2o_od, _ = stacked_brnn(P, "multi2one", od_in, 1, bidirectional=bool(opt["multi2one_bidir"]))
    cap("multi2one", m2o_ocr); cap("multi2one", m2o_od)
    ocr_x, ocr_mask = _items_to_samples(m2o_ocr, ocr["num_cnt"], ocr["len_cnt"], ocr["position"].size(1))
    od_x, od_mask = _items_to_samples(m2o_od, od["num_cnt"], od["len_cnt"], od["position"].size(1))

    nl = opt["in_rnn_layers"]
    _, ocr_l = stacked_brnn(P, "context_rnn", ocr_x, nl, LN=True)               # :338-340
    _, q_l = stacked_brnn(P, "ques_rnn", q_in, nl, LN=True)
    _, od_l = stacked_brnn(P, "context_rnn", od_x, nl, LN=True)
    cap("context_rnn", torch.stack(ocr_l)); cap("ques_rnn", torch.stack(q_l)); cap("context_rnn", torch.stack(od_l))
    q_high, _ = stacked_brnn(P, "high_lvl_ques_rnn", torch.cat(q_l, 2), opt["question_high_lvl_rnn_layers"], LN=True, concat=True)   # :350
    cap("high_lvl_ques_rnn", q_high)
    q_l = q_l + [q_high]

    ocr_h, ocr_pre = deep_attention(P, "deep_attn", [ocr_x], ocr_l, [q_wv], q_l, q_mask)     # :376-377
    od_h, od_pre = deep_attention(P, "deep_attn", [od_x], od_l, [q_wv], q_l, q_mask)
    cap("deep_attn", ocr_h); cap("deep_attn_pre", ocr_pre); cap("deep_attn", od_h); cap("deep_attn_pre", od_pre)

    ocr_sa_in = torch.cat([ocr_h, ocr_pre, ocr_x], 2)                            # :380-381
    od_sa_in = torch.cat([od_h, od_pre, od_x], 2)
    ocr_sa = module_attention(P, "highlvl_self_att", ocr_sa_in, ocr_sa_in, ocr_mask, x3=ocr_h)   # :387-388
    od_sa = module_attention(P, "highlvl_self_att", od_sa_in, od_sa_in, od_mask, x3=od_h)
    cap("highlvl_self_att", ocr_sa); cap("highlvl_self_att", od_sa)
    ocr_hl, _ = stacked_brnn(P, "high_lvl_context_rnn", torch.cat([ocr_h, ocr_sa], 2), 1, LN=True)   # :389-390
    od_hl, _ = stacked_brnn(P, "high_lvl_context_rnn", torch.cat([od_h, od_sa], 2), 1, LN=True)
    cap("high_lvl_context_rnn", ocr_hl); cap("high_lvl_context_rnn", od_hl)

    x_od_ocr = module_attention(P, "od_ocr_attn", ocr_hl, od_hl, od_mask)         # :399-401
    pos_att = module_attention(P, "position_attn", ocr["position"], od["position"], od_mask, x3=od_hl)
    cap("od_ocr_attn", x_od_ocr); cap("position_attn", pos_att)
    ocr_final = torch.cat([ocr_hl, x_od_ocr + pos_att], 2)                        # :404-405

    q_final = module_attention(P, "ques_self_attn", q_high, q_high, q_mask)       # :411-415
    cap("ques_self_attn", q_final)
    w = linear_self_attn(q_final, q_mask, P["ques_merger.linear.weight"], P["ques_merger.linear.bias"])
    cap("ques_merger", w)
    q_merged = weighted_avg(q_final, w)
    score = get_final_scores(P, "get_answer", ocr_final, q_merged, ocr_mask, opt["ES_ocr_len"], mask_flag="mask_score" in opt)   # :428-429
    cap("get_answer", score)
    return score


# ------------------------------------------------------------------------------------
# PHOC word descriptor  (Utils/cphoc.c, Utils/phoc.py)
# ------------------------------------------------------------------------------------
_PHOC_UNIGRAMS = "abcdefghijklmnopqrstuvwxyz0123456789"
_PHOC_BIGRAMS = ("th he in er an re es on st nt en at ed nd to or ea ti ar te ng al it as is ha et se ou of le sa ve ro ra ri hi ne me "
                 "de co ta ec si ll so na li la el").split()


def build_phoc_raw(word):
    """cphoc.c:12-113.  604 floats: 36 unigrams x the 14 regions of pyramid levels 2..5, then 50 bigrams x the 2 regions of
    level 2.  A character (bigram) spanning [lo, hi] of the word's unit interval is counted in a region when
    (min(hi, r1) - max(lo, r0)) / (hi - lo) >= 0.5; every operation is a single-precision one, in the C source's order
    (:33-34, :57-62, :89-98) - boundary cases depend on it.  Raises on a character outside the unigram list (:45-50)."""
    import numpy as np
    f = np.float32
    out = np.zeros(604, dtype=np.float32)
    n = len(word)

    def half_inside(lo, hi, region, level):
        r0, r1 = f(region) / f(level), f(region + 1) / f(level)
        o0, o1 = max(lo, r0), min(hi, r1)
        return (o1 - o0) / (hi - lo) >= f(0.5)

    for i, ch in enumerate(word):
        lo, hi = f(i) / f(n), f(i + 1) / f(n)
        ci = _PHOC_UNIGRAMS.find(ch)
        if ci < 0:
            raise RuntimeError("Error: unigram %s is unknown" % ch)
        for level in range(2, 6):
            base = sum(l for l in range(2, 6) if l < level)
            for region in range(level):
                if half_inside(lo, hi, region, level):
                    out[base * 36 + region * 36 + ci] = 1
    for i in range(n - 1):
        if word[i:i + 2] not in _PHOC_BIGRAMS:
            continue
        bi = _PHOC_BIGRAMS.index(word[i:i + 2])
        lo, hi = f(i) / f(n), f(i + 2) / f(n)
        for region in range(2):
            if half_inside(lo, hi, region, 2):
                out[36 * 14 + region * 50 + bi] = 1
    return out


def build_phoc(token):
    """phoc.py:8-12: lower-case, strip, keep [a-z0-9], then the raw descriptor (as a list of floats)."""
    token = "".join(c for c in token.lower().strip() if c in _PHOC_UNIGRAMS)
    return build_phoc_raw(token).tolist()
