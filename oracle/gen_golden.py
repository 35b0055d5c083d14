#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the UNMODIFIED reference on CPU.

TEST INFRASTRUCTURE ONLY - runs in the build container (where /root/reference is
mounted) and nowhere else.  The fixtures are plain arrays: seeded synthetic inputs
and the reference's outputs / gradients for them.  Weights are NOT stored for the
large configurations; they are regenerated from the same seed by
``ruart_amd.synth`` on whichever box runs the tests (a float64 checksum of every
weight tensor is stored so drift is detected).

    python oracle/gen_golden.py            # writes all fixtures
    python oracle/gen_golden.py layers     # one group: layers | bert | e2e | e2e_full | e2e_full_ragged | e2e_stress | e2e_outliers | e2e_phoc | e2e_unlocked | e2e_unlocked_long | host | dataset | phoc | predict | update

Reference entry points exercised (file:line in /root/reference):
    Models/Bert/modeling.py:585-614   BertModel.forward
    Models/Bert/Bert.py:130-176       Bert.combine_forward
    Models/Layers.py:124-180,182-295,320-341,352-468,471-534
    Models/SDNet.py:253-437           SDNet.forward
    Models/SDNetTrainer.py:510-518    instance_bce_with_logits
    Models/SDNetTrainer.py:296-376    SDNetTrainer.setup_model + update (three optimizer steps), trainer_update.npz
    Models/SDNetTrainer.py:378-451    SDNetTrainer.predict (answer decode + ANLS / ACC), predict_decode.json
    Utils/phoc.py:8-12 (+ cphoc.so)   build_phoc, phoc.npz
    Utils/VQA_Dataset.py:13-437       VQA_Dataset (+ Models/Bert/tokenization.py BertTokenizer), dataset_*.json[.gz] fixtures
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import _refshim  # noqa: E402

_refshim.install()
from ruart_amd import synth  # noqa: E402
from ruart_amd.arguments import default_opt  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def checksum(wdict):
    return np.array([float(np.sum(v.astype(np.float64))) for _, v in sorted(wdict.items())])


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f KB, %d arrays)" % (path, os.path.getsize(path) / 1024, len(arrays)))


# ----------------------------------------------------------------------------------
def gen_layers():
    """Per-op vectors from Models/Layers.py with small random modules."""
    import Models.Layers as L
    L.set_dropout_prob(0.0)
    L.set_seq_dropout(True)
    g = np.random.default_rng(11)
    out = {}

    def rnd(*s, scale=1.0):
        return (g.standard_normal(s) * scale).astype(np.float32)

    # ---- Attention, four variants -------------------------------------------------
    for tag, (B, L1, L2, D, Hh, D3, sim) in {
        "attn_a": (3, 9, 7, 20, 12, 20, False),      # x3 is None
        "attn_b": (2, 17, 5, 24, 16, 10, False),     # x3 given
        "attn_c": (2, 6, 11, 8, 5, 14, True),        # do_similarity (fixed scalar diagonal)
        "attn_d": (2, 100, 36, 48, 25, 30, False),   # conf-like lengths
    }.items():
        m = L.Attention(D, Hh, correlation_func=3, do_similarity=sim)
        W = rnd(Hh, D, scale=0.4)
        m.scoring.linear.weight.data = T(W)
        if not sim:
            m.scoring.diagonal.data = T(1.0 + 0.3 * rnd(1, 1, Hh))
        x1 = T(rnd(B, L1, D)).requires_grad_()
        x2 = T(rnd(B, L2, D)).requires_grad_()
        mask = np.ones((B, L2), dtype=np.uint8)
        for b in range(B):
            mask[b, int(g.integers(1, L2 + 1)):] = 0
        x3 = None if tag == "attn_a" else T(rnd(B, L2, D3)).requires_grad_()
        y = m(x1, x2, T(mask), x3=x3)
        gy = T(rnd(*y.shape))
        y.backward(gy)
        out.update({tag + "_W": W, tag + "_diag": m.scoring.diagonal.detach().numpy().copy(),
                    tag + "_x1": x1.detach().numpy(), tag + "_x2": x2.detach().numpy(),
                    tag + "_mask": mask, tag + "_y": y.detach().numpy(), tag + "_gy": gy.numpy(),
                    tag + "_gx1": x1.grad.numpy(), tag + "_gx2": x2.grad.numpy(),
                    tag + "_gW": m.scoring.linear.weight.grad.numpy()})
        if x3 is not None:
            out[tag + "_x3"] = x3.detach().numpy()
            out[tag + "_gx3"] = x3.grad.numpy()
        if not sim:
            out[tag + "_gdiag"] = m.scoring.diagonal.grad.numpy()

    # ---- StackedBRNN: 2-layer BiLSTM with whole-tensor layer norm -----------------
    for tag, (B, Tn, Din, Hh, nl, bid, ln) in {
        "rnn_a": (3, 11, 10, 6, 2, True, True),
        "rnn_b": (5, 4, 14, 9, 1, False, False),
    }.items():
        m = L.StackedBRNN(Din, Hh, nl, bidirectional=bid)
        for n_, p in m.named_parameters():
            a = (g.uniform(-0.5, 0.5, tuple(p.shape))).astype(np.float32)
            p.data = T(a)
            out[tag + "_w_" + n_] = a
        x = T(rnd(B, Tn, Din)).requires_grad_()
        y, ys = m(x, None, return_list=True, LN=ln)
        gy = T(rnd(*y.shape))
        gy0 = T(rnd(*ys[0].shape))
        (y * gy).sum().add((ys[0] * gy0).sum()).backward()
        out.update({tag + "_x": x.detach().numpy(), tag + "_y": y.detach().numpy(),
                    tag + "_y0": ys[0].detach().numpy(), tag + "_gy": gy.numpy(), tag + "_gy0": gy0.numpy(),
                    tag + "_gx": x.grad.numpy()})
        for n_, p in m.named_parameters():
            out[tag + "_g_" + n_] = p.grad.numpy()

    # ---- LinearSelfAttn + weighted_avg --------------------------------------------
    B, Ln, D = 4, 13, 10
    m = L.LinearSelfAttn(D)
    m.linear.weight.data = T(rnd(1, D, scale=0.5))
    m.linear.bias.data = T(rnd(1))
    x = T(rnd(B, Ln, D)).requires_grad_()
    mask = np.ones((B, Ln), dtype=np.uint8)
    mask[1, 5:] = 0
    mask[3, 1:] = 0
    alpha = m(x, T(mask))
    y = L.weighted_avg(x, alpha)
    gy = T(rnd(*y.shape))
    y.backward(gy)
    out.update(dict(merge_w=m.linear.weight.detach().numpy(), merge_b=m.linear.bias.detach().numpy(),
                    merge_x=x.detach().numpy(), merge_mask=mask, merge_alpha=alpha.detach().numpy(),
                    merge_y=y.detach().numpy(), merge_gy=gy.numpy(), merge_gx=x.grad.numpy(),
                    merge_gw=m.linear.weight.grad.numpy(), merge_gb=m.linear.bias.grad.numpy()))

    # ---- GetFinalScores (useES, no_answer, mask_flag) -------------------------------
    B, Ln, X, Hh, ES = 3, 16, 12, 7, 4
    m = L.GetFinalScores(X, Hh, yesno=False, no_answer=True, useES=True)
    for n_, p in m.named_parameters():
        a = g.uniform(-0.4, 0.4, tuple(p.shape)).astype(np.float32)
        p.data = T(a)
        out["score_w_" + n_] = a
    x = T(rnd(B, Ln, X)).requires_grad_()
    h0 = T(rnd(B, Hh)).requires_grad_()
    mask = np.ones((B, Ln), dtype=np.uint8)
    mask[0, 9:] = 0
    mask[2, 6:] = 0
    y = m(x, h0, T(mask), ES, mask_flag=True)
    gy = T(rnd(*y.shape))
    y.backward(gy)
    out.update(dict(score_x=x.detach().numpy(), score_h0=h0.detach().numpy(), score_mask=mask,
                    score_y=y.detach().numpy(), score_gy=gy.numpy(), score_gx=x.grad.numpy(),
                    score_gh0=h0.grad.numpy(), score_ES=np.array(ES)))
    for n_, p in m.named_parameters():
        if p.grad is not None:
            out["score_g_" + n_] = p.grad.numpy()
    out["score_nograd"] = np.array([n_ for n_, p in m.named_parameters() if p.grad is None])

    # ---- whole-tensor layer norm and the loss -------------------------------------
    x = T(rnd(4, 9, 6) * 3 + 0.7).requires_grad_()
    y = torch.nn.functional.layer_norm(x, x.size())
    gy = T(rnd(4, 9, 6))
    y.backward(gy)
    out.update(dict(wln_x=x.detach().numpy(), wln_y=y.detach().numpy(), wln_gy=gy.numpy(), wln_gx=x.grad.numpy()))
    save("layers", **out)


def gen_layers_extra():
    """Per-op vectors for the two composite steps that had end-to-end coverage only: a9 DeepAttention
    (Models/Layers.py:471-524, forward with return_bef_rnn and every gradient) and a5 SDNet.get_prealign_emb (Models/SDNet.py:495-551:
    the reference's re-packing loops around the pre-align Attention), called on a stand-in object that carries only what the method
    reads (opt, pre_align) - the method itself is the reference's."""
    import types
    import Models.Layers as L
    import Models.SDNet as RS
    L.set_dropout_prob(0.0)
    L.set_seq_dropout(True)
    g = np.random.default_rng(23)
    out = {}

    def rnd(*s, scale=1.0):
        return (g.standard_normal(s) * scale).astype(np.float32)

    # ---- DeepAttention: 2 abstraction levels + the high-level one on the question side -------------------------------------
    B, L1, L2, Wd, Hh, HL, per = 3, 11, 6, 7, 4, 5, 8
    opt = {"hidden_size": Hh, "highlvl_hidden_size": HL}
    m = L.DeepAttention(opt, abstr_list_cnt=2, deep_att_hidden_size_per_abstr=per, correlation_func=3, word_hidden_size=Wd)
    for n_, p in m.named_parameters():
        a = g.uniform(-0.3, 0.3, tuple(p.shape)).astype(np.float32)
        p.data = T(a)
        out["deep_w_" + n_] = a
    x1_word = [T(rnd(B, L1, Wd)).requires_grad_()]
    x1_abstr = [T(rnd(B, L1, 2 * Hh)).requires_grad_() for _ in range(2)]
    x2_word = [T(rnd(B, L2, Wd)).requires_grad_()]
    x2_abstr = [T(rnd(B, L2, 2 * Hh)).requires_grad_() for _ in range(2)] + [T(rnd(B, L2, 2 * HL)).requires_grad_()]   # + high level
    m1 = np.ones((B, L1), dtype=np.uint8)
    m1[1, 8:] = 0
    m2 = np.ones((B, L2), dtype=np.uint8)
    m2[0, 4:] = 0
    m2[2, 2:] = 0
    h, pre = m(x1_word, x1_abstr, x2_word, x2_abstr, T(m1), T(m2), return_bef_rnn=True)
    gh, gpre = T(rnd(*h.shape)), T(rnd(*pre.shape))
    ((h * gh).sum() + (pre * gpre).sum()).backward()
    out.update(deep_dims=np.array([B, L1, L2, Wd, Hh, HL, per]), deep_m1=m1, deep_m2=m2, deep_h=h.detach().numpy(),
               deep_pre=pre.detach().numpy(), deep_gh=gh.numpy(), deep_gpre=gpre.numpy())
    for name, lst in (("x1_word", x1_word), ("x1_abstr", x1_abstr), ("x2_word", x2_word), ("x2_abstr", x2_abstr)):
        for i, t in enumerate(lst):
            out["deep_%s_%d" % (name, i)] = t.detach().numpy()
            out["deep_g_%s_%d" % (name, i)] = t.grad.numpy()
    for n_, p in m.named_parameters():
        out["deep_g_" + n_] = p.grad.numpy()

    # ---- get_prealign_emb on a small ragged batch (the layout VQA_collate_fun produces; word vectors are 300-d in the method) ------
    opt = default_opt(vocab_size=400, cuda=False)
    Bp = 3
    q, ocr, od, _, _ = synth.synthetic_batch(opt, Bp, seed=31, n_q=7, n_ocr=14, n_od=5, bert_vocab=500, ragged=True)
    pa = L.Attention(300, 6, correlation_func=3, do_similarity=True)
    wpa = g.uniform(-0.1, 0.1, tuple(pa.scoring.linear.weight.shape)).astype(np.float32)
    pa.scoring.linear.weight.data = T(wpa)
    stand_in = types.SimpleNamespace(opt=opt, pre_align=pa)
    key_o, key_q = ("fasttext" if "fasttext" in opt["ocr_embedding"] else "glove"), ("fasttext" if "fasttext" in opt["q_embedding"] else "glove")
    embs = {}
    for name, lst, key in (("q", q, key_q), ("ocr", ocr, key_o), ("od", od, key_o)):
        ids = lst[key]
        e = T(rnd(ids.shape[0], ids.shape[1], 300, scale=0.5)).requires_grad_()
        lst[key + "_emb"] = e
        embs[name] = e
    ocr_pa, od_pa = RS.SDNet.get_prealign_emb(stand_in, q, ocr, od, Bp)
    g_ocr, g_od = T(rnd(*ocr_pa.shape)), T(rnd(*od_pa.shape))
    ((ocr_pa * g_ocr).sum() + (od_pa * g_od).sum()).backward()
    out.update(pre_B=np.array(Bp), pre_seed=np.array(31), pre_args=np.array([7, 14, 5, 500, 400]), pre_w=wpa, pre_diag=pa.scoring.diagonal.detach().numpy(),
               pre_q_emb=embs["q"].detach().numpy(), pre_ocr_emb=embs["ocr"].detach().numpy(), pre_od_emb=embs["od"].detach().numpy(),
               pre_ocr_out=ocr_pa.detach().numpy(), pre_od_out=od_pa.detach().numpy(), pre_g_ocr=g_ocr.numpy(), pre_g_od=g_od.numpy(),
               pre_gq=embs["q"].grad.numpy(), pre_gocr=embs["ocr"].grad.numpy(), pre_god=embs["od"].grad.numpy(),
               pre_gw=pa.scoring.linear.weight.grad.numpy(),
               pre_ocr_num_cnt=np.array(ocr["num_cnt"]), pre_od_num_cnt=np.array(od["num_cnt"]))
    save("layers_extra", **out)


# ----------------------------------------------------------------------------------
def gen_bert():
    """BertModel.forward: a small config stored with weights, and bert-base dims
    (weights regenerated from seed; selected layers/rows stored)."""
    from Models.Bert.modeling import BertModel, BertConfig
    g = np.random.default_rng(23)

    def run(cfg, seed, ids, mask):
        w = synth.make_bert_weights(cfg, seed=seed)
        d = _refshim.write_bert_dir(cfg, w)
        model = BertModel.from_pretrained(d)
        model.eval()
        with torch.no_grad():
            layers, _ = model(T(ids), token_type_ids=None, attention_mask=T(mask))
        return w, [l.numpy() for l in layers]

    def make_ids(N, Lmax, vocab, lens):
        ids = np.zeros((N, Lmax), dtype=np.int64)
        for i, l in enumerate(lens):
            ids[i, :l] = g.integers(1, vocab, size=l)
        return ids, (ids != 0)

    # small: 3 layers, hidden 128 (2 heads x 64), ragged lengths incl. 1 and full
    cfg = synth.bert_config(vocab_size=500, hidden_size=128, num_hidden_layers=3, num_attention_heads=2,
                            intermediate_size=256, max_position_embeddings=64)
    lens = [1, 2, 3, 5, 8, 13, 21, 30, 30, 4, 50, 17]
    ids, mask = make_ids(len(lens), 50, 500, lens)
    w, layers = run(cfg, 5, ids, mask)
    arrays = dict(ids=ids, mask=mask, seed=np.array(5), wsum=checksum(w),
                  cfg=np.array([cfg["vocab_size"], cfg["hidden_size"], cfg["num_hidden_layers"],
                                cfg["num_attention_heads"], cfg["intermediate_size"], cfg["max_position_embeddings"]]))
    for i, l in enumerate(layers):
        arrays["layer%d" % i] = l
    save("bert_small", **arrays)

    # bert-base dims (12 x 768, 12 heads, FFN 3072), small vocab table
    cfg = synth.bert_config(vocab_size=2000)
    lens = [30, 4, 7, 12, 3, 9]
    ids, mask = make_ids(len(lens), 30, 2000, lens)
    w, layers = run(cfg, 1033, ids, mask)
    arrays = dict(ids=ids, mask=mask, seed=np.array(1033), wsum=checksum(w),
                  cfg=np.array([cfg["vocab_size"], 768, 12, 12, 3072, 512]))
    for i in (0, 5, 11):
        arrays["layer%d" % i] = layers[i]
    save("bert_base", **arrays)


# ----------------------------------------------------------------------------------
def build_reference_sdnet(opt, bert_cfg, seed):
    from Models.SDNet import SDNet
    import Models.Layers as L
    bw = synth.make_bert_weights(bert_cfg, seed=seed)
    opt = dict(opt)
    opt["BERT_model_file"] = _refshim.write_bert_dir(bert_cfg, bw)
    opt["datadir"] = ""
    sw = synth.make_sdnet_weights(opt, seed=seed)
    emb = {"glove_embedding": T(sw["glove_embed.weight"]).clone(),
           "fast_embedding": T(sw["fast_embed.weight"]).clone()}
    net = SDNet(opt, emb)
    missing, unexpected = net.load_state_dict({k: T(v) for k, v in sw.items()}, strict=False)
    assert not unexpected and all(k.startswith("Bert.") for k in missing), (missing, unexpected)
    return net, opt, bw, sw, L


def gen_e2e():
    """SDNet.forward + loss + backward at the shipped conf sizes, bert-base dims."""
    opt = default_opt(vocab_size=1500)
    bert_cfg = synth.bert_config(vocab_size=2000)
    seed = 1033
    net, opt, bw, sw, L = build_reference_sdnet(opt, bert_cfg, seed)
    B = 2
    batch = synth.synthetic_batch(opt, B, seed=7, n_q=30, n_ocr=100, n_od=30, bert_vocab=2000, ragged=True)
    # one sample at the maximum item counts, one small
    q, ocr, od, gt, extra = batch
    print("num_cnt ocr", ocr["num_cnt"], "od", od["num_cnt"])

    caps = {}

    def put(key, t):
        i = 0
        while "%s#%d" % (key, i) in caps:
            i += 1
        caps["%s#%d" % (key, i)] = t.detach().numpy().copy()

    def hook(name):
        def fn(mod, inp, outp):
            if name in ("context_rnn", "ques_rnn"):      # (output, [per-layer outputs])
                put(name, torch.stack(list(outp[1]), 0))
            elif name == "deep_attn":                    # (rnn output, pre-rnn concat)
                put(name, outp[0])
                put(name + "_pre", outp[1])
            else:
                put(name, outp)
        return fn

    for name in ["multi2one", "context_rnn", "ques_rnn", "high_lvl_ques_rnn", "deep_attn", "highlvl_self_att",
                 "high_lvl_context_rnn", "od_ocr_attn", "position_attn", "ques_self_attn", "ques_merger",
                 "get_answer", "pre_align"]:
        getattr(net, name).register_forward_hook(hook(name))
    orig_emb = net.get_embedding_from_list

    def emb_wrap(item_list, names, initial):
        r = orig_emb(item_list, names, initial)
        i = 0
        while "embed#%d" % i in caps:
            i += 1
        caps["embed#%d" % i] = r.detach().numpy().copy()
        return r
    net.get_embedding_from_list = emb_wrap

    # dropout off everywhere (quirk 3 of SURVEY.md: train() would re-enable BERT dropout)
    L.set_dropout_prob(0.0)
    net.train()
    net.Bert.bert_model.eval()
    net.drop_emb = False
    scores, _ = net(q, ocr, od)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(scores, gt) * gt.size(1)
    loss.backward()

    arrays = dict(seed=np.array(seed), batch_seed=np.array(7), B=np.array(B), vocab_size=np.array(1500),
                  bert_wsum=checksum(bw), sdnet_wsum=checksum(sw),
                  scores=scores.detach().numpy(), loss=np.array(loss.item()), gt=gt.numpy(),
                  ocr_num_cnt=np.array(ocr["num_cnt"]), od_num_cnt=np.array(od["num_cnt"]))
    names, norms = [], []
    for n_, p in net.named_parameters():
        if n_.startswith("Bert."):
            continue
        names.append(n_)
        norms.append(-1.0 if p.grad is None else float(p.grad.double().norm()))
        if p.grad is not None and p.numel() <= 4096:
            arrays["grad:" + n_] = p.grad.numpy().copy()
    arrays["grad_names"] = np.array(names)
    arrays["grad_norms"] = np.array(norms)
    arrays["grad:fast_embed.weight[:64]"] = net.fast_embed.weight.grad[:64].numpy().copy()
    # keep the intermediates compact (strided views; the strides are part of the key)
    for k, v in caps.items():
        if k.startswith("embed#") and v.shape[1] != opt["max_q_len"]:
            arrays["cap:" + k + "[:,:4,::8]"] = v[:, :4, ::8].copy()
        elif k.startswith("embed#"):
            arrays["cap:" + k + "[:,:,::8]"] = v[..., ::8].copy()
        elif k.startswith("multi2one#"):      # only the state at each item's last word is consumed
            lens = np.concatenate([np.array(l) for l in (ocr if k.endswith("#0") else od)["len_cnt"]])
            arrays["cap:" + k + "[last]"] = v[np.arange(v.shape[0]), lens - 1].copy()
        elif k.startswith("deep_attn_pre#"):
            arrays["cap:" + k + "[:,:,::5]"] = v[..., ::5].copy()
        elif k.startswith("pre_align#"):
            arrays["cap:" + k + "[:,:,::4]"] = v[..., ::4].copy()
        else:
            arrays["cap:" + k] = v
    save("sdnet_e2e", **arrays)


def _run_reference_step(net, L, q, ocr, od, gt, bert_eval=True):
    """The reference's forward + loss + backward with every dropout off."""
    L.set_dropout_prob(0.0)
    net.train()
    if bert_eval:
        net.Bert.bert_model.eval()
    net.drop_emb = False
    scores, _ = net(q, ocr, od)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(scores, gt) * gt.size(1)
    loss.backward()
    return scores, loss


def _grad_summary(net, arrays, full_below=4096):
    names, norms = [], []
    for n_, p in net.named_parameters():
        if n_.startswith("Bert."):
            continue
        names.append(n_)
        norms.append(-1.0 if p.grad is None else float(p.grad.double().norm()))
        if p.grad is not None and p.numel() <= full_below:
            arrays["grad:" + n_] = p.grad.numpy().copy()
    arrays["grad_names"] = np.array(names)
    arrays["grad_norms"] = np.array(norms)


def gen_e2e_full(which="bench"):
    """The UNMODIFIED reference at BASELINE.json's full per-GPU size: B = 64, 30 question words, 100 OCR items, 36 objects,
    bert-base - SDNet.forward (Models/SDNet.py:253-437) + loss (Models/SDNetTrainer.py:510-518) + backward.
      bench  : exactly bench.py's workload (rank 0, batch 0: vocab 20000, BERT vocab 30522, weights N(0, 0.02), seed 1033,
               batch seed 7, every sample at the maximum item counts)           -> sdnet_e2e_full.npz
      ragged : the workload of tests/test_gpu_sdnet.py::test_full_size_properties (weights N(0, 0.05), seed 21, batch seed 31,
               ragged item counts)                                              -> sdnet_e2e_full_ragged.npz
    Weights are regenerated from the seed where the tests run; stored: scores (64, 101), loss, every gradient norm, the
    gradients of the small tensors and a slice of one embedding-table gradient."""
    import time
    if which == "bench":
        opt = default_opt(vocab_size=20000, max_od_num=36)
        bert_cfg, seed, bseed, w_std, ragged, name = synth.bert_config(), 1033, 7, 0.02, False, "sdnet_e2e_full"
    else:
        opt = default_opt(vocab_size=2000, max_od_num=36)
        bert_cfg, seed, bseed, w_std, ragged, name = synth.bert_config(vocab_size=3000), 21, 31, 0.05, True, "sdnet_e2e_full_ragged"
    from Models.SDNet import SDNet
    import Models.Layers as L
    bw = synth.make_bert_weights(bert_cfg, seed=seed, w_std=w_std)
    opt = dict(opt)
    opt["BERT_model_file"] = _refshim.write_bert_dir(bert_cfg, bw)
    opt["datadir"] = ""
    sw = synth.make_sdnet_weights(opt, seed=seed)
    emb = {"glove_embedding": T(sw["glove_embed.weight"]).clone(), "fast_embedding": T(sw["fast_embed.weight"]).clone()}
    net = SDNet(opt, emb)
    missing, unexpected = net.load_state_dict({k: T(v) for k, v in sw.items()}, strict=False)
    assert not unexpected and all(k.startswith("Bert.") for k in missing), (missing, unexpected)
    B = 64
    q, ocr, od, gt, _ = synth.synthetic_batch(opt, B, seed=bseed, n_q=30, n_ocr=100, n_od=36, bert_vocab=bert_cfg["vocab_size"],
                                              ragged=ragged)
    t0 = time.perf_counter()
    scores, loss = _run_reference_step(net, L, q, ocr, od, gt)
    print("%s: reference fwd+bwd at B=64 took %.1f s on %d threads; loss %.6f" % (name, time.perf_counter() - t0,
                                                                                 torch.get_num_threads(), loss.item()))
    arrays = dict(seed=np.array(seed), batch_seed=np.array(bseed), B=np.array(B), vocab_size=np.array(opt["vocab_size"]),
                  bert_vocab=np.array(bert_cfg["vocab_size"]), w_std=np.array(w_std), ragged=np.array(ragged),
                  bert_wsum=checksum(bw), sdnet_wsum=checksum(sw), scores=scores.detach().numpy(), loss=np.array(loss.item()),
                  ocr_num_cnt=np.array(ocr["num_cnt"]), od_num_cnt=np.array(od["num_cnt"]))
    _grad_summary(net, arrays)
    arrays["grad:fast_embed.weight[:64]"] = net.fast_embed.weight.grad[:64].numpy().copy()
    save(name, **arrays)


def gen_e2e_outliers():
    """The UNMODIFIED reference on encoder weights with pretrained-like heavy tails (synth.add_bert_outliers: a few LayerNorm gains
    x 10-30 in the same hidden dimensions of every layer, a few word-embedding columns x 20, projection weights N(0, 0.04)): B = 2 at
    the shipped item counts, ragged.  Pins the fp16c mode outside the range its constant e4m3 scales were chosen on
    (Models/Bert/modeling.py:155-168, 445-531: real checkpoints look like this) -> sdnet_e2e_outliers.npz."""
    import time
    opt = default_opt(vocab_size=1500, max_od_num=36)
    bert_cfg, seed, bseed, w_std, oseed = synth.bert_config(vocab_size=2000), 1033, 53, 0.04, 77
    from Models.SDNet import SDNet
    import Models.Layers as L
    bw = synth.add_bert_outliers(synth.make_bert_weights(bert_cfg, seed=seed, w_std=w_std), bert_cfg, seed=oseed)
    opt = dict(opt)
    opt["BERT_model_file"] = _refshim.write_bert_dir(bert_cfg, bw)
    opt["datadir"] = ""
    sw = synth.make_sdnet_weights(opt, seed=seed)
    emb = {"glove_embedding": T(sw["glove_embed.weight"]).clone(), "fast_embedding": T(sw["fast_embed.weight"]).clone()}
    net = SDNet(opt, emb)
    missing, unexpected = net.load_state_dict({k: T(v) for k, v in sw.items()}, strict=False)
    assert not unexpected and all(k.startswith("Bert.") for k in missing), (missing, unexpected)
    B = 2
    q, ocr, od, gt, _ = synth.synthetic_batch(opt, B, seed=bseed, n_q=30, n_ocr=100, n_od=36, bert_vocab=2000, ragged=True)
    # how far out the activations are: the encoder's own layer outputs on this batch
    with torch.no_grad():
        net.Bert.bert_model.eval()
        layers, _ = net.Bert.bert_model(ocr["bert"], token_type_ids=None, attention_mask=ocr["bert_mask"].long())
        amax = [float(x.abs().max()) for x in layers]
    print("outliers: max |layer output| per layer:", ["%.1f" % a for a in amax])
    t0 = time.perf_counter()
    scores, loss = _run_reference_step(net, L, q, ocr, od, gt)
    print("sdnet_e2e_outliers: reference fwd+bwd took %.1f s; loss %.6f" % (time.perf_counter() - t0, loss.item()))
    arrays = dict(seed=np.array(seed), batch_seed=np.array(bseed), B=np.array(B), vocab_size=np.array(1500), bert_vocab=np.array(2000),
                  w_std=np.array(w_std), outlier_seed=np.array(oseed), layer_absmax=np.array(amax), bert_wsum=checksum(bw),
                  sdnet_wsum=checksum(sw), scores=scores.detach().numpy(), loss=np.array(loss.item()),
                  ocr_num_cnt=np.array(ocr["num_cnt"]), od_num_cnt=np.array(od["num_cnt"]))
    _grad_summary(net, arrays)
    save("sdnet_e2e_outliers", **arrays)


def gen_e2e_stress():
    """BASELINE.json's stress shapes on the UNMODIFIED reference: bert-large (24 x 1024, 16 heads, FFN 4096; the reference
    switches on `BERT_LARGE`, Models/Bert/Bert.py:26-33), 300 OCR items, 100 objects, B = 2 (one sample at the maximum item
    counts, one ragged).  Stored like sdnet_e2e_full.npz -> sdnet_e2e_stress.npz."""
    import time
    opt = default_opt(vocab_size=800, BERT_LARGE=True, max_ocr_num=300, max_od_num=100)
    bert_cfg = synth.bert_config(vocab_size=1200, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
    seed, bseed, w_std = 1033, 41, 0.02
    from Models.SDNet import SDNet
    import Models.Layers as L
    bw = synth.make_bert_weights(bert_cfg, seed=seed, w_std=w_std)
    opt = dict(opt)
    opt["BERT_large_model_file"] = _refshim.write_bert_dir(bert_cfg, bw)
    opt["datadir"] = ""
    sw = synth.make_sdnet_weights(opt, seed=seed)
    emb = {"glove_embedding": T(sw["glove_embed.weight"]).clone(), "fast_embedding": T(sw["fast_embed.weight"]).clone()}
    net = SDNet(opt, emb)
    missing, unexpected = net.load_state_dict({k: T(v) for k, v in sw.items()}, strict=False)
    assert not unexpected and all(k.startswith("Bert.") for k in missing), (missing, unexpected)
    B = 2
    q, ocr, od, gt, _ = synth.synthetic_batch(opt, B, seed=bseed, n_q=30, n_ocr=300, n_od=100, bert_vocab=1200, ragged=True)
    print("stress num_cnt ocr", ocr["num_cnt"], "od", od["num_cnt"])
    # Models/SDNet.py:299, 318 keeps a per-sample item count in a ByteTensor that nothing reads (`mask_copy`).  Under the
    # reference's pinned torch 1.0.1 a count above 255 wraps silently; torch 2.x raises.  Harness-side version shim (the reference
    # is not touched): an out-of-range integer stored into a uint8 tensor wraps modulo 256, as it did then.
    orig_setitem = torch.Tensor.__setitem__

    def setitem_wrapping(self, key, value):
        try:
            return orig_setitem(self, key, value)
        except RuntimeError as e:
            if self.dtype == torch.uint8 and isinstance(value, int) and "uint8" in str(e):
                return orig_setitem(self, key, value % 256)
            raise
    torch.Tensor.__setitem__ = setitem_wrapping
    t0 = time.perf_counter()
    try:
        scores, loss = _run_reference_step(net, L, q, ocr, od, gt)
    finally:
        torch.Tensor.__setitem__ = orig_setitem
    print("sdnet_e2e_stress: reference fwd+bwd took %.1f s; loss %.6f" % (time.perf_counter() - t0, loss.item()))
    arrays = dict(seed=np.array(seed), batch_seed=np.array(bseed), B=np.array(B), vocab_size=np.array(800), bert_vocab=np.array(1200),
                  w_std=np.array(w_std), bert_wsum=checksum(bw), sdnet_wsum=checksum(sw), scores=scores.detach().numpy(),
                  loss=np.array(loss.item()), ocr_num_cnt=np.array(ocr["num_cnt"]), od_num_cnt=np.array(od["num_cnt"]))
    _grad_summary(net, arrays)
    save("sdnet_e2e_stress", **arrays)


def gen_e2e_phoc():
    """SDNet.forward + loss + backward with the PHOC table enabled (a PHOC conf: `PHOC`, `phoc_dim 604`, `phoc` in ocr_embedding;
    Models/SDNet.py:26-27, 51-55, 73, 441-446): the OCR / object words carry their 604-d pyramidal character histogram next to
    the fastText vector.  The table is the reference's build_phoc over synth.phoc_vocab_words()."""
    from Utils.phoc import build_phoc
    V = 600
    opt = default_opt(vocab_size=V, PHOC=True, phoc_dim=604, ocr_embedding="fasttext,phoc,pos,ent,bert")
    bert_cfg = synth.bert_config(vocab_size=2000)
    seed = 1033
    from Models.SDNet import SDNet
    import Models.Layers as L
    bw = synth.make_bert_weights(bert_cfg, seed=seed)
    opt = dict(opt)
    opt["BERT_model_file"] = _refshim.write_bert_dir(bert_cfg, bw)
    opt["datadir"] = ""
    sw = synth.make_sdnet_weights(opt, seed=seed)
    table = np.array([build_phoc(w) for w in synth.phoc_vocab_words(V, seed)], dtype=np.float32)
    sw["phoc_embed.weight"] = table
    emb = {"glove_embedding": T(sw["glove_embed.weight"]).clone(), "fast_embedding": T(sw["fast_embed.weight"]).clone(),
           "phoc_embedding": T(table).clone()}
    net = SDNet(opt, emb)
    missing, unexpected = net.load_state_dict({k: T(v) for k, v in sw.items()}, strict=False)
    assert not unexpected and all(k.startswith("Bert.") for k in missing), (missing, unexpected)
    B = 3
    q, ocr, od, gt, extra = synth.synthetic_batch(opt, B, seed=19, n_q=14, n_ocr=24, n_od=7, bert_vocab=2000, ragged=True)
    L.set_dropout_prob(0.0)
    net.train()
    net.Bert.bert_model.eval()
    net.drop_emb = False
    scores, _ = net(q, ocr, od)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(scores, gt) * gt.size(1)
    loss.backward()
    arrays = dict(seed=np.array(seed), batch_seed=np.array(19), B=np.array(B), vocab_size=np.array(V), phoc_ones=table.sum(1).astype(np.int32),
                  scores=scores.detach().numpy(), loss=np.array(loss.item()), ocr_num_cnt=np.array(ocr["num_cnt"]))
    names, norms = [], []
    for n_, p in net.named_parameters():
        if n_.startswith("Bert."):
            continue
        names.append(n_)
        norms.append(-1.0 if p.grad is None else float(p.grad.double().norm()))
    arrays["grad_names"] = np.array(names)
    arrays["grad_norms"] = np.array(norms)
    arrays["grad:multi2one.rnns.0.weight_ih_l0[:8]"] = net.multi2one.rnns[0].weight_ih_l0.grad[:8].numpy().copy()
    save("sdnet_e2e_phoc", **arrays)


def gen_e2e_unlocked(name="sdnet_e2e_unlocked", n_q=12):
    """SDNet.forward + loss + backward WITHOUT `LOCK_BERT` (Models/SDNet.py:88-94): the encoder is part of the graph and every one of
    its parameters gets a gradient.  BERT's own dropout (re-enabled by network.train()) is configured to 0 so
    the pass is deterministic.  Stored: scores, loss, the gradient norm of every parameter, a few gradient slices.
    ``e2e_unlocked_long``: the same with 90-word questions - BERT sequences of more than 64 word pieces (Models/Bert/Bert.py:96-99
    takes up to 512), beyond one window of the 16-bit trainable encoder's attention kernels."""
    opt = default_opt(vocab_size=600)
    opt.pop("LOCK_BERT")
    if n_q > opt["max_q_len"]:                       # room for the long questions (the collate widths; the model itself has no limit)
        opt["max_q_len"], opt["max_q_bert_len"] = n_q + 10, 2 * n_q
    bert_cfg = synth.bert_config(vocab_size=2000, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    seed = 1033
    net, opt, bw, sw, L = build_reference_sdnet(opt, bert_cfg, seed)
    assert all(p.requires_grad for p in net.Bert.parameters())
    B = 2
    q, ocr, od, gt, extra = synth.synthetic_batch(opt, B, seed=23, n_q=n_q, n_ocr=16, n_od=6, bert_vocab=2000, ragged=True)
    L.set_dropout_prob(0.0)
    net.train()
    net.drop_emb = False
    scores, _ = net(q, ocr, od)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(scores, gt) * gt.size(1)
    loss.backward()
    arrays = dict(seed=np.array(seed), batch_seed=np.array(23), B=np.array(B), vocab_size=np.array(600), n_q=np.array(n_q),
                  scores=scores.detach().numpy(), loss=np.array(loss.item()), ocr_num_cnt=np.array(ocr["num_cnt"]))
    names, norms = [], []
    for n_, p in net.named_parameters():
        names.append(n_)
        norms.append(-1.0 if p.grad is None else float(p.grad.double().norm()))
    arrays["grad_names"] = np.array(names)
    arrays["grad_norms"] = np.array(norms)
    pick = {"Bert.bert_model.encoder.layer.0.attention.self.query.weight": (slice(0, 4), slice(0, 64)),
            "Bert.bert_model.encoder.layer.11.output.dense.weight": (slice(0, 4), slice(0, 64)),
            "Bert.bert_model.encoder.layer.5.intermediate.dense.bias": (slice(0, 64),),
            "Bert.bert_model.embeddings.LayerNorm.gamma": (slice(0, 64),),
            "Bert.bert_model.embeddings.position_embeddings.weight": (slice(0, 4), slice(0, 64))}
    prm = dict(net.named_parameters())
    for k, sl in pick.items():
        arrays["grad:" + k] = prm[k].grad[sl].numpy().copy()
    print("bert grads:", sum(1 for n_ in names if n_.startswith("Bert.")), "none:", [n_ for n_, v in zip(names, norms) if v < 0])
    save(name, **arrays)


# ----------------------------------------------------------------------------------
def synthetic_samples(opt, n, seed):
    """Per-sample dicts in the layout VQA_Dataset.__getitem__ emits (Utils/VQA_Dataset.py:145-153)."""
    g = np.random.default_rng(seed)

    def item(nw, nb, sentinel=None):
        words = [sentinel] if sentinel is not None else g.integers(5, 900, size=nw).tolist()
        bert = [101] + g.integers(1000, 2000, size=nb).tolist() + [102]
        offs, cur = [], 1
        for k in range(len(words)):
            c = 1 + (k < nb - len(words))
            offs.append([cur, cur + c])
            cur += c
        return {"fasttext": words, "pos": g.integers(0, 51, size=len(words)).tolist(), "ent": g.integers(0, 75, size=len(words)).tolist(),
                "bert": bert, "bert_offsets": offs, "position": g.random(8).round(4).tolist()}

    out = []
    for i in range(n):
        nq = int(g.integers(3, 12))
        q = {"glove": g.integers(5, 900, size=nq).tolist(), "pos": g.integers(0, 51, size=nq).tolist(),
             "ent": g.integers(0, 75, size=nq).tolist(), "bert": [101] + g.integers(1000, 2000, size=nq + 2).tolist() + [102],
             "bert_offsets": [[1 + k, 2 + k] for k in range(nq)]}
        n_ocr, n_od = int(g.integers(12, 20)), int(g.integers(1, 6))
        ocr = [item(int(g.integers(1, 4)), int(g.integers(3, 7))) for _ in range(n_ocr - 1)] + [item(1, 1, sentinel=3)]
        od = [item(int(g.integers(1, 3)), int(g.integers(2, 4))) for _ in range(n_od - 1)] + [item(1, 1, sentinel=4)]
        gt = torch.zeros(1, opt["max_ocr_num"] + 1)
        gt[0, int(g.integers(0, n_ocr - 1))] = 1.0
        out.append({"q": q, "ocr": ocr, "od": od, "gt": gt, "extra_info": {"q_id": i, "answers": None,
                                                                         "ocr_list": ["w"] * n_ocr, "image_path": "x"}})
    return out


def gen_host():
    """VQA_collate_fun (Utils/VQA_Dataset.py:448-542) and VQA_Sampler (Utils/VQA_Sampler.py) outputs."""
    from Utils.VQA_Dataset import VQA_collate
    from Utils.VQA_Sampler import VQA_Sampler
    opt = default_opt()
    samples = synthetic_samples(opt, 3, seed=21)
    q, ocr, od, gt, extra = VQA_collate(opt).VQA_collate_fun(samples)
    arrays = {"seed": np.array(21), "gt": gt.numpy()}
    for name, d in (("q", q), ("ocr", ocr), ("od", od)):
        for k, v in d.items():
            if isinstance(v, torch.Tensor):
                arrays["%s:%s" % (name, k)] = v.numpy()
        if "num_cnt" in d:
            arrays[name + ":num_cnt"] = np.array(d["num_cnt"])
            arrays[name + ":len_cnt"] = np.concatenate([np.array(l) for l in d["len_cnt"]])
        arrays[name + ":n_offsets"] = np.array([len(o) for o in d["bert_offsets"]])
    data = list(range(23))
    arrays["sampler_train"] = np.array(list(VQA_Sampler(data, 7, 5, True)))
    arrays["sampler_train_resume"] = np.array(list(VQA_Sampler(data, 7, 5, True, batch_st=3)))
    arrays["sampler_epoch"] = np.array(list(VQA_Sampler(data, None, 4, True, epoch=2)))
    arrays["sampler_eval"] = np.array(list(VQA_Sampler(data, 99, 5, False)))
    save("host", **arrays)


# ----------------------------------------------------------------------------------
DATASET_VOCAB = (["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list("abcdefghijklmnopqrstuvwxyz0123456789") +
                 ["##" + c for c in "abcdefghijklmnopqrstuvwxyz0123456789"] +
                 ["the", "what", "is", "sign", "stop", "##ing", "##s", "un", "##aff", "##able", "coca", "cola", "shop", "caf", "##e",
                  "cafe", "exit", "##it", "ex", "north", "##th", "bus", "2", "##nd", "street", "park", "##ed", "!", "?", ".", ",", "-",
                  "'", "$", "\u4e2d", "\u56fd", "naive", "brand", "name", "yes", "no", "write", "written", "on", "##on"])


def synthetic_records(seed):
    """Preprocessed records in the layout CoQAPreprocess writes (annotated_question / detector lists / scores), with the text
    cases the tokenizer has to get right: capitals, accents, punctuation runs, CJK, control characters, an unknown word, a word
    of more than 100 characters, empty word lists, repeated strings, more candidates than max_ocr_num."""
    g = np.random.default_rng(seed)
    texts = ["STOP", "Stopping", "unaffable", "Coca-Cola", "caf\u00e9", "Caf\u00c9!", "exit", "north", "2nd", "bus stop", "park.ed",
             "\u4e2d\u56fd shop", "na\u00efve", "x" * 101, "zz\u0000top", "tab\tsplit", "\u00a0wide\u3000space", "$5,00", "what's", "???",
             "q\u0301", "brand-name", "WRITTEN", "the", "\ufffdsign", "s\u200bt"]

    def words_of(text):
        return text.replace("-", " - ").split() or [text]

    def annotated(text):
        w = words_of(text)
        return {"word": w, "wordid": g.integers(5, 900, size=len(w)).tolist(), "pos_id": g.integers(0, 51, size=len(w)).tolist(),
                "ent_id": g.integers(0, 75, size=len(w)).tolist()}

    def ocr_item(text, i, empty=False):
        a = annotated(text)
        if empty:
            a = {"word": [], "wordid": [], "pos_id": [], "ent_id": []}
        return {"word": a, "original": text, "pos": g.random(8).round(4).tolist(), "ANLS": float(np.round(g.random() ** 2, 3)),
                "ACC": float(g.integers(0, 4)) / 10.0, "cnt": int(g.integers(0, 9)), "idx": int(i)}

    def od_item(text):
        return {"object": annotated(text), "original": text.upper(), "pos": g.random(8).round(4).tolist()}

    recs = []
    for qi in range(7):
        n_ocr = [3, 12, 0, 140, 6, 9, 1][qi]
        picks = g.integers(0, len(texts), size=n_ocr)
        ocr = [ocr_item(texts[j], i, empty=(i % 11 == 5)) for i, j in enumerate(picks)]
        gram = [ocr_item(texts[j] + " " + texts[(j + 1) % len(texts)], i) for i, j in enumerate(picks[:5])]
        es = [ocr_item(texts[j], i) for i, j in enumerate(g.integers(0, len(texts), size=[0, 4, 15, 15, 2, 12, 0][qi]))]
        od_texts = ["STOP", "exit", "north", "bus stop", "the", "2nd", "Caf\u00c9!", "WRITTEN", "\u4e2d\u56fd"]   # <= 8 pieces: the reference's
        od = [od_item(od_texts[j]) for j in g.integers(0, len(od_texts), size=[2, 0, 5, 45, 1, 3, 2][qi])]    # collate rejects longer ones
        q_text = ["What is written on the sign?", "what brand-name is the caf\u00e9", "", "Is the bus parked on 2nd street ?",
                  "what's the exit", "WHAT IS THE \u4e2d\u56fd SHOP", "the?"][qi]
        answers = [["stop"], ["coca cola", "Coca-Cola"], ["x"], ["yes"], [], ["answering does not require reading text in the image"],
                   ["exit", "north exit"]][qi]
        recs.append({"question_id": 1000 + qi, "question": q_text, "filename": "img/%d.jpg" % qi, "orign_answers": answers,
                     "annotated_question": annotated(q_text) if q_text else {"word": [], "wordid": [], "pos_id": [], "ent_id": []},
                     "ocr_PMTD_ASTER": ocr, "ocr_PMTD_ASTER_gram2": gram, "ES_ocr": es, "OD_bottom-up": od})
    return recs


def gen_dataset():
    """Utils/VQA_Dataset.py VQA_Dataset.__getitem__ over synthetic preprocessed records, several label / list configurations,
    followed by the reference's collate.  Inputs and expected outputs are stored as JSON (lists of ints / floats / strings)."""
    import copy
    import json
    import tempfile
    from Utils.VQA_Dataset import VQA_Dataset, VQA_collate
    tmp = tempfile.mkdtemp()
    vocab_path = os.path.join(tmp, "vocab.txt")
    with open(vocab_path, "w", encoding="utf-8") as f:
        f.write("\n".join(DATASET_VOCAB) + "\n")
    recs = synthetic_records(5)
    variants = {
        "base": {},
        "dedup_one": {"remove_same": True, "lable_way": "lable_one_offical"},
        "yesno_relevance": {"label_yesno": True, "ES_sort_way": "relevance", "lable_way": "lable_one"},
        "acc_all_no_es": {"score_name": "ACC", "lable_way": "lable_all", "_drop": ["ES_ocr", "label_no_answer"]},
        "test_mode": {"_mode": "test"},
    }
    expected = {}
    for name, ov in variants.items():
        opt = default_opt(datadir="", BERT_tokenizer_file=vocab_path)
        ov = dict(ov)
        mode = ov.pop("_mode", "train")
        for k in ov.pop("_drop", []):
            opt.pop(k, None)
        opt.update(ov)
        ds = VQA_Dataset(copy.deepcopy(recs), opt, mode=mode)
        samples = [ds[i] for i in range(len(ds))]
        out = []
        for smp in samples:
            out.append({"q": smp["q"], "ocr": smp["ocr"], "od": smp["od"], "gt": smp["gt"].tolist(), "extra_info": smp["extra_info"]})
        q, ocr, od, gt, extra = VQA_collate(opt).VQA_collate_fun(samples[:4])
        coll = {"gt": gt.tolist()}
        for nm, d in (("q", q), ("ocr", ocr), ("od", od)):
            for k, v in d.items():
                coll["%s:%s" % (nm, k)] = v.tolist() if isinstance(v, torch.Tensor) else v
        expected[name] = {"n": len(ds), "samples": out, "collated": coll}
        print(name, "samples", len(ds), "ocr counts", [len(s["ocr"]) for s in out])
    with open(os.path.join(OUT, "dataset_input.json"), "w", encoding="utf-8") as f:
        json.dump({"vocab": DATASET_VOCAB, "records": recs, "variants": variants}, f, ensure_ascii=True)
    import gzip
    with gzip.open(os.path.join(OUT, "dataset_expected.json.gz"), "wt", encoding="utf-8") as f:
        json.dump(expected, f, ensure_ascii=True)
    for fn in ("dataset_input.json", "dataset_expected.json.gz"):
        print("wrote", fn, os.path.getsize(os.path.join(OUT, fn)) // 1024, "KB")


def phoc_words(seed=9, n_random=400):
    """Words for the PHOC fixture: the SURVEY's two known answers, every length 1..24 of a repeated letter and of mixed text
    (region boundaries fall differently for every length), all 50 bigrams at the start / middle / end, strings the wrapper has
    to clean, and seeded random words."""
    g = np.random.default_rng(seed)
    alpha = "abcdefghijklmnopqrstuvwxyz0123456789"
    bigrams = ("th he in er an re es on st nt en at ed nd to or ea ti ar te ng al it as is ha et se ou of le sa ve ro ra ri hi ne me "
               "de co ta ec si ll so na li la el").split()
    words = ["the", "Hello-42", "", "a", "  STOP  ", "caf\u00e9", "x" * 70, "0123456789", "!!!", "th", "the the"]
    words += ["a" * n for n in range(1, 25)] + ["thequickbrownfox0123456789"[:n] for n in range(1, 25)]
    for b in bigrams:
        words += [b, b + "xyz", "xy" + b + "z", "wxyz" + b, b + b + b]
    for _ in range(n_random):
        n = int(g.integers(1, 40))
        words.append("".join(alpha[int(k)] for k in g.integers(0, 36, size=n)))
    return words


def gen_phoc():
    """Utils/phoc.py build_phoc (the prebuilt cphoc.so) over phoc_words(); rows stored bit-packed."""
    from Utils.phoc import build_phoc
    words = phoc_words()
    rows = np.array([build_phoc(w) for w in words], dtype=np.float32)
    assert rows.shape == (len(words), 604) and set(np.unique(rows)) <= {0.0, 1.0}
    save("phoc", words=np.array("\n".join(words)), n=np.array(len(words)), bits=np.packbits(rows.astype(np.uint8), axis=1),
         ones=rows.sum(1).astype(np.int32))


def predict_cases(seed=3):
    """Score matrices for the answer decode: random ones plus the orders that matter - sentinel on top, a padding slot on top,
    the no-answer slot on top / second, a sample whose only item is the sentinel, exact answers, ten-answer (TextVQA style) lists."""
    g = np.random.default_rng(seed)
    n_slots = 13                                   # max_ocr_num 12 + no-answer
    words = ["stop", "exit", "coca cola", "north", "bus", "2nd", "cafe", "park", "shop", "street", "sign"]
    cases = []
    for i in range(24):
        n = [1, 2, 5, 12, 7, 3][i % 6]             # items incl. the trailing <ocr> sentinel
        ocr = [words[int(k)] for k in g.integers(0, len(words), size=n - 1)] + ["<ocr>"]
        p = g.random(n_slots).astype(np.float32)
        mode = i % 8
        if mode == 1:
            p[n - 1] = 2.0                          # sentinel first
        elif mode == 2 and n < 12:
            p[n] = 2.0                              # padding slot first
        elif mode == 3:
            p[-1] = 2.0                             # no-answer first
        elif mode == 4:
            p[n - 1], p[-1] = 3.0, 2.0              # sentinel, then no-answer
        elif mode == 5 and n < 11:
            p[n], p[n + 1] = 3.0, 2.5               # two padding slots first
        answers = [None, [ocr[0]], ["Stop", "stopp"], [w for w in (ocr * 10)[:10]], ["zzz"]][i % 5]
        cases.append({"prob": p.tolist(), "num_cnt": n, "ocr_list": ocr, "answers": answers, "q_id": 500 + i})
    return cases, n_slots


def gen_predict():
    """Models/SDNetTrainer.py:378-451 with the network replaced by a function that returns prepared scores: the reference's own
    decode loop, ANLS / ACC accumulation and result records, with and without the no-answer slot."""
    import json
    import types
    import Models.SDNetTrainer as M
    cases, n_slots = predict_cases()
    out = {"cases": cases, "n_slots": n_slots, "expected": {}}
    for name, no_answer in (("no_answer", True), ("plain", False)):
        width = n_slots if no_answer else n_slots - 1
        scores = torch.tensor([c["prob"][:width] for c in cases])
        opt = {"label_no_answer": True} if no_answer else {}

        class Net:
            drop_emb = False

            def eval(self):
                pass

            def __call__(self, q, ocr, od):
                return scores, None

        fake = types.SimpleNamespace(network=Net(), opt=opt, fixed_answers_len=0, fixed_answers_entry=None,
                                     loss_func=lambda s, g: torch.zeros(()))
        batch = (None, {"num_cnt": [c["num_cnt"] for c in cases]}, None, torch.zeros(len(cases), width),
                 [{"q_id": c["q_id"], "answers": c["answers"], "ocr_list": c["ocr_list"], "image_path": "x"} for c in cases])
        loss, anls, acc, res, save_res = M.SDNetTrainer.predict(fake, batch)
        out["expected"][name] = {"ANLS": float(anls), "ACC": float(acc), "res": res, "save_res": save_res}
        print(name, "ANLS %.4f ACC %.4f" % (anls, acc), [r["idx"] for r in save_res])
    with open(os.path.join(OUT, "predict_decode.json"), "w") as f:
        json.dump(out, f)
    print("wrote predict_decode.json", os.path.getsize(os.path.join(OUT, "predict_decode.json")) // 1024, "KB")


def gen_update():
    """Three calls of the reference's own ``SDNetTrainer.update`` (Models/SDNetTrainer.py:330-376: forward, BCE_D1 loss, backward,
    clip_grad_norm_ 10, Adamax, re-pinning of embedding rows >= tune_partial) on one small batch, dropout configured to 0
    everywhere (conf DROPOUT / dropout_emb and BERT's own, which train() re-enables).  The trainer object is created without its
    __init__ (that one opens the preprocessing artefacts); setup_model and update are the reference's."""
    import Models.SDNetTrainer as M
    opt = default_opt(vocab_size=1500, DROPOUT=0.0, dropout_emb=0.0)
    bert_cfg = synth.bert_config(vocab_size=2000, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    seed = 1033
    bw = synth.make_bert_weights(bert_cfg, seed=seed)
    opt = dict(opt)
    opt["BERT_model_file"] = _refshim.write_bert_dir(bert_cfg, bw)
    opt["datadir"] = ""
    opt["cuda"] = False
    sw = synth.make_sdnet_weights(opt, seed=seed)
    tr = M.SDNetTrainer.__new__(M.SDNetTrainer)
    tr.opt, tr.use_cuda, tr.fixed_answers_len, tr.fixed_answers_entry = opt, False, 0, None
    M.set_dropout_prob(0.0)
    tr.setup_model({"glove_embedding": T(sw["glove_embed.weight"]).clone(), "fast_embedding": T(sw["fast_embed.weight"]).clone()})
    missing, unexpected = tr.network.load_state_dict({k: T(v) for k, v in sw.items()}, strict=False)
    assert not unexpected and all(k.startswith("Bert.") for k in missing)
    # On a GPU `network.cuda()` gives the parameters new storage while `fixed_embedding_fast/glove` (plain attributes, views of the
    # CPU tables, Models/SDNet.py:78-81) keep the initial values - that is what makes the re-pinning of :369-373 work.  Under the
    # harness's `.cuda()` -> identity shim they would alias the live weights and the re-pinning would be a no-op: un-alias them.
    for k in ("fixed_embedding_fast", "fixed_embedding_glove"):
        setattr(tr.network, k, getattr(tr.network, k).clone())
    before = {n: p.detach().clone() for n, p in tr.network.named_parameters() if not n.startswith("Bert.")}
    batch = synth.synthetic_batch(opt, 3, seed=29, n_q=12, n_ocr=20, n_od=6, bert_vocab=2000, ragged=True)
    losses = [tr.update(batch, i) for i in range(3)]
    print("losses", losses)
    arrays = dict(seed=np.array(seed), batch_seed=np.array(29), B=np.array(3), vocab_size=np.array(1500), losses=np.array(losses),
                  lr=np.array(float(opt["lr"])))
    names, dnorm, pnorm = [], [], []
    for n, p in tr.network.named_parameters():
        if n.startswith("Bert."):
            continue
        names.append(n)
        dnorm.append(float((p.detach() - before[n]).double().norm()))
        pnorm.append(float(p.detach().double().norm()))
    arrays["names"], arrays["delta_norms"], arrays["param_norms"] = np.array(names), np.array(dnorm), np.array(pnorm)
    prm = dict(tr.network.named_parameters())
    for k in ("alphaBERT", "gammaBERT", "ques_merger.linear.weight", "get_answer.noanswer_w.weight"):
        arrays["after:" + k] = prm[k].detach().numpy().copy()
    arrays["after:fast_embed.weight[:40,:16]"] = prm["fast_embed.weight"].detach()[:40, :16].numpy().copy()
    arrays["after:fast_embed.weight[1000:1004,:16]"] = prm["fast_embed.weight"].detach()[1000:1004, :16].numpy().copy()
    for k in ("fast_embed.weight", "glove_embed.weight"):          # how far every row of the word tables moved
        arrays["rowdelta:" + k] = (prm[k].detach() - before[k]).double().norm(dim=1).numpy().astype(np.float32)
    save("trainer_update", **arrays)


if __name__ == "__main__":
    which = sys.argv[1:] or ["layers", "layers_extra", "bert", "e2e", "e2e_phoc", "e2e_unlocked", "host", "dataset", "phoc", "predict", "update"]
    if "layers" in which:
        gen_layers()
    if "layers_extra" in which:
        gen_layers_extra()
    if "bert" in which:
        gen_bert()
    if "e2e" in which:
        gen_e2e()
    if "e2e_full" in which:
        gen_e2e_full("bench")
    if "e2e_full_ragged" in which:
        gen_e2e_full("ragged")
    if "e2e_stress" in which:
        gen_e2e_stress()
    if "e2e_outliers" in which:
        gen_e2e_outliers()
    if "e2e_phoc" in which:
        gen_e2e_phoc()
    if "e2e_unlocked" in which:
        gen_e2e_unlocked()
    if "e2e_unlocked_long" in which:
        gen_e2e_unlocked("sdnet_e2e_unlocked_long", n_q=90)
    if "host" in which:
        gen_host()
    if "dataset" in which:
        gen_dataset()
    if "phoc" in which:
        gen_phoc()
    if "predict" in which:
        gen_predict()
    if "update" in which:
        gen_update()
