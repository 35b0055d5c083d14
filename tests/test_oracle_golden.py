"""The CPU oracle (oracle/ruart_oracle.py) against vectors produced by the unmodified
reference (tests/golden/*.npz, generator: oracle/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import ruart_oracle as O
from ruart_amd import synth
from ruart_amd.arguments import default_opt

torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def close(a, b, atol, rtol=0.0, what=""):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    err = np.abs(a - b)
    lim = atol + rtol * np.abs(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.all(err <= lim), "%s: max err %.3e (limit %.1e), max|ref| %.3e" % (what, err.max(), atol, np.abs(b).max())


@pytest.fixture(scope="module")
def layers(golden_dir):
    return np.load(os.path.join(golden_dir, "layers.npz"))


@pytest.mark.parametrize("tag", ["attn_a", "attn_b", "attn_c", "attn_d"])
def test_attention(layers, tag):
    z = layers
    x1, x2 = T(z[tag + "_x1"]).requires_grad_(), T(z[tag + "_x2"]).requires_grad_()
    W, d = T(z[tag + "_W"]).requires_grad_(), T(z[tag + "_diag"]).requires_grad_()
    x3 = T(z[tag + "_x3"]).requires_grad_() if tag + "_x3" in z.files else None
    y = O.attention(x1, x2, T(z[tag + "_mask"]), W, d, x3)
    close(y, z[tag + "_y"], 2e-6, 1e-5, tag + " fwd")
    y.backward(T(z[tag + "_gy"]))
    close(x1.grad, z[tag + "_gx1"], 2e-5, 1e-4, tag + " gx1")
    close(x2.grad, z[tag + "_gx2"], 2e-5, 1e-4, tag + " gx2")
    close(W.grad, z[tag + "_gW"], 5e-5, 1e-4, tag + " gW")
    if x3 is not None:
        close(x3.grad, z[tag + "_gx3"], 2e-5, 1e-4, tag + " gx3")
    if tag + "_gdiag" in z.files:
        close(d.grad, z[tag + "_gdiag"], 5e-5, 1e-4, tag + " gdiag")


@pytest.mark.parametrize("tag,nl,bid,ln", [("rnn_a", 2, True, True), ("rnn_b", 1, False, False)])
def test_stacked_brnn(layers, tag, nl, bid, ln):
    z = layers
    P = {"m." + k[len(tag) + 3:]: T(z[k]).requires_grad_() for k in z.files if k.startswith(tag + "_w_")}
    x = T(z[tag + "_x"]).requires_grad_()
    y, ys = O.stacked_brnn(P, "m", x, nl, bidirectional=bid, LN=ln)
    close(y, z[tag + "_y"], 5e-6, 1e-5, tag + " y")
    close(ys[0], z[tag + "_y0"], 5e-6, 1e-5, tag + " y0")
    ((y * T(z[tag + "_gy"])).sum() + (ys[0] * T(z[tag + "_gy0"])).sum()).backward()
    close(x.grad, z[tag + "_gx"], 2e-5, 1e-4, tag + " gx")
    for k, p in P.items():
        close(p.grad, z[tag + "_g_" + k[2:]], 5e-5, 2e-4, tag + " g " + k)


def test_merge(layers):
    z = layers
    x, w, b = T(z["merge_x"]).requires_grad_(), T(z["merge_w"]).requires_grad_(), T(z["merge_b"]).requires_grad_()
    alpha = O.linear_self_attn(x, T(z["merge_mask"]), w, b)
    y = O.weighted_avg(x, alpha)
    close(alpha, z["merge_alpha"], 1e-6, 1e-5, "alpha")
    close(y, z["merge_y"], 2e-6, 1e-5, "y")
    y.backward(T(z["merge_gy"]))
    close(x.grad, z["merge_gx"], 1e-5, 1e-4, "gx")
    close(w.grad, z["merge_gw"], 1e-5, 1e-4, "gw")
    close(b.grad, z["merge_gb"], 1e-5, 1e-4, "gb")


def test_final_scores(layers):
    z = layers
    P = {"ga." + k[len("score_w_"):]: T(z[k]).requires_grad_() for k in z.files if k.startswith("score_w_")}
    x, h0 = T(z["score_x"]).requires_grad_(), T(z["score_h0"]).requires_grad_()
    y = O.get_final_scores(P, "ga", x, h0, T(z["score_mask"]), int(z["score_ES"]), mask_flag=True)
    close(y, z["score_y"], 1e-6, 1e-5, "scores")
    assert np.allclose(y.detach().sum(1).numpy(), 1.0, atol=1e-6)
    y.backward(T(z["score_gy"]))
    close(x.grad, z["score_gx"], 1e-5, 1e-4, "gx")
    close(h0.grad, z["score_gh0"], 1e-5, 1e-4, "gh0")
    for k, p in P.items():
        name = k[3:]
        if name in set(z["score_nograd"].tolist()):
            assert p.grad is None, name          # the dead GRU step, SURVEY §0.9
        else:
            close(p.grad, z["score_g_" + name], 1e-5, 1e-4, "g " + name)


def test_deep_attention_and_prealign(golden_dir):
    """The oracle's deep_attention (Layers.py:471-524) and _prealign (SDNet.py:495-551) against the reference's own per-op vectors
    (tests/golden/layers_extra.npz): outputs and every input gradient."""
    z = np.load(os.path.join(golden_dir, "layers_extra.npz"))
    P = {"m." + k[len("deep_w_"):]: T(z[k]).requires_grad_() for k in z.files if k.startswith("deep_w_")}
    grab = lambda name, n: [T(z["deep_%s_%d" % (name, i)]).requires_grad_() for i in range(n)]
    x1w, x1a, x2w, x2a = grab("x1_word", 1), grab("x1_abstr", 2), grab("x2_word", 1), grab("x2_abstr", 3)
    h, pre = O.deep_attention(P, "m", x1w, x1a, x2w, x2a, T(z["deep_m2"]))
    assert np.abs(h.detach().numpy() - z["deep_h"]).max() < 2e-6 and np.abs(pre.detach().numpy() - z["deep_pre"]).max() < 2e-6
    ((h * T(z["deep_gh"])).sum() + (pre * T(z["deep_gpre"])).sum()).backward()
    for name, lst in (("x1_word", x1w), ("x1_abstr", x1a), ("x2_word", x2w), ("x2_abstr", x2a)):
        for i, t in enumerate(lst):
            assert np.abs(t.grad.numpy() - z["deep_g_%s_%d" % (name, i)]).max() < 1e-5, (name, i)
    for k, t in P.items():
        if t.grad is not None:
            assert np.abs(t.grad.numpy() - z["deep_g_" + k[2:]]).max() < 1e-5, k
    # pre-align: same synthetic batch layout as the generator's
    from ruart_amd import synth
    from ruart_amd.arguments import default_opt
    nq, nocr, nod, bv, V = [int(v) for v in z["pre_args"]]
    opt = default_opt(vocab_size=V, cuda=False)
    q, ocr, od, _, _ = synth.synthetic_batch(opt, int(z["pre_B"]), seed=int(z["pre_seed"]), n_q=nq, n_ocr=nocr, n_od=nod, bert_vocab=bv, ragged=True)
    assert ocr["num_cnt"] == z["pre_ocr_num_cnt"].tolist()
    Pp = {"pre_align.scoring.linear.weight": T(z["pre_w"]).requires_grad_(), "pre_align.scoring.diagonal": T(z["pre_diag"])}
    qe = T(z["pre_q_emb"]).requires_grad_()
    key_q = "fasttext" if "fasttext" in opt["q_embedding"] else "glove"
    for tag, items in (("ocr", ocr), ("od", od)):
        e = T(z["pre_%s_emb" % tag]).requires_grad_()
        out, _ = O._prealign(Pp, e, items["len_cnt"], qe, q[key_q + "_mask"])
        assert np.abs(out.detach().numpy() - z["pre_%s_out" % tag]).max() < 2e-6, tag
        (out * T(z["pre_g_" + tag])).sum().backward()
        assert np.abs(e.grad.numpy() - z["pre_g" + tag]).max() < 1e-5, tag
    assert np.abs(qe.grad.numpy() - z["pre_gq"]).max() < 1e-5
    assert np.abs(Pp["pre_align.scoring.linear.weight"].grad.numpy() - z["pre_gw"]).max() < 1e-5


def test_whole_tensor_ln(layers):
    z = layers
    x = T(z["wln_x"]).requires_grad_()
    y = O.whole_tensor_layer_norm(x)
    close(y, z["wln_y"], 2e-6, 1e-5, "y")
    y.backward(T(z["wln_gy"]))
    close(x.grad, z["wln_gx"], 2e-6, 1e-4, "gx")


def _bert_case(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    c = z["cfg"]
    cfg = synth.bert_config(vocab_size=int(c[0]), hidden_size=int(c[1]), num_hidden_layers=int(c[2]),
                            num_attention_heads=int(c[3]), intermediate_size=int(c[4]), max_position_embeddings=int(c[5]))
    w = synth.make_bert_weights(cfg, seed=int(z["seed"]))
    wsum = np.array([float(np.sum(v.astype(np.float64))) for _, v in sorted(w.items())])
    assert np.array_equal(wsum, z["wsum"]), "synthetic weights drifted from the ones the golden was made with"
    return z, cfg, {k: T(v) for k, v in w.items()}


@pytest.mark.parametrize("name", ["bert_small", "bert_base"])
def test_bert_forward(golden_dir, name):
    z, cfg, w = _bert_case(golden_dir, name)
    with torch.no_grad():
        layers = O.bert_forward(w, cfg, T(z["ids"]), T(z["mask"]))
    for k in z.files:
        if k.startswith("layer"):
            close(layers[int(k[5:])], z[k], 3e-5, 1e-5, name + " " + k)


def test_sdnet_end_to_end(golden_dir):
    z = np.load(os.path.join(golden_dir, "sdnet_e2e.npz"))
    opt = default_opt(vocab_size=int(z["vocab_size"]))
    cfg = synth.bert_config(vocab_size=2000)
    bw = {k: T(v) for k, v in synth.make_bert_weights(cfg, seed=int(z["seed"])).items()}
    sw = synth.make_sdnet_weights(opt, seed=int(z["seed"]))
    wsum = np.array([float(np.sum(v.astype(np.float64))) for _, v in sorted(sw.items())])
    assert np.array_equal(wsum, z["sdnet_wsum"])
    # the scalar similarity 'diagonal' is a non-trainable Parameter in the reference (Layers.py:197-198)
    P = {k: T(v).requires_grad_(v.shape != (1, 1, 1)) for k, v in sw.items()}
    q, ocr, od, gt, _ = synth.synthetic_batch(opt, int(z["B"]), seed=int(z["batch_seed"]), n_q=30, n_ocr=100,
                                              n_od=30, bert_vocab=2000, ragged=True)
    assert ocr["num_cnt"] == z["ocr_num_cnt"].tolist() and od["num_cnt"] == z["od_num_cnt"].tolist()
    assert np.array_equal(gt.numpy(), z["gt"])
    caps = {}
    scores = O.sdnet_forward(P, opt, bw, cfg, q, ocr, od, caps=caps)
    # intermediates, in call order
    for k in z.files:
        if not k.startswith("cap:"):
            continue
        key = k[4:]
        base = key.split("[")[0]
        if base.startswith("embed") or base.startswith("multi2one"):
            continue                                    # checked below with their views
        v = caps[base].numpy()
        if "[:,:,::5]" in key:
            v = v[..., ::5]
        if "[:,:,::4]" in key:
            v = v[..., ::4]
        close(v, z[k], 5e-5, 1e-4, k)
    for i, items in ((1, ocr), (2, od)):
        close(caps["embed#%d" % i].numpy()[:, :4, ::8], z["cap:embed#%d[:,:4,::8]" % i], 3e-5, 1e-4, "embed")
        lens = np.concatenate([np.array(l) for l in items["len_cnt"]])
        m = caps["multi2one#%d" % (i - 1)].numpy()
        close(m[np.arange(m.shape[0]), lens - 1], z["cap:multi2one#%d[last]" % (i - 1)], 3e-5, 1e-4, "multi2one")
    close(caps["embed#0"].numpy()[..., ::8], z["cap:embed#0[:,:,::8]"], 3e-5, 1e-4, "embed q")
    close(scores, z["scores"], 2e-5, 1e-4, "scores")
    loss = O.instance_bce_with_logits(scores, gt)
    assert abs(loss.item() - float(z["loss"])) < 1e-4
    loss.backward()
    for name, ref_norm in zip(z["grad_names"].tolist(), z["grad_norms"].tolist()):
        g = P[name].grad
        if ref_norm < 0:
            assert g is None or float(g.norm()) == 0.0, name
            continue
        got = float(g.double().norm())
        assert abs(got - ref_norm) <= 2e-4 * max(ref_norm, 1e-3), (name, got, ref_norm)
        if "grad:" + name in z.files:
            close(g, z["grad:" + name], 1e-6 + 2e-4 * float(np.abs(z["grad:" + name]).max()), 0, "grad " + name)
    close(P["fast_embed.weight"].grad[:64], z["grad:fast_embed.weight[:64]"], 1e-6, 1e-3, "fast_embed grad rows")


def _phoc_case(golden_dir):
    z = np.load(os.path.join(golden_dir, "phoc.npz"))
    words = str(z["words"]).split("\n")
    assert len(words) == int(z["n"])
    return words, np.unpackbits(z["bits"], axis=1)[:, :604].astype(np.float32), z["ones"]


def test_phoc_restatement(golden_dir):
    """oracle.build_phoc against the reference's build_phoc (prebuilt cphoc.so) on 700+ words, plus the two known answers the
    survey recorded."""
    words, rows, ones = _phoc_case(golden_dir)
    for w, ref in zip(words, rows):
        got = np.array(O.build_phoc(w), dtype=np.float32)
        assert got.shape == (604,) and np.array_equal(got, ref), w
    assert np.flatnonzero(O.build_phoc("the")).tolist() == [19, 40, 43, 91, 115, 148, 199, 259, 292, 343, 403, 472, 504, 555]
    assert int(sum(O.build_phoc("Hello-42"))) == 30
    assert ones[words.index("the")] == 14 and ones[words.index("")] == 0
    with pytest.raises(RuntimeError):
        O.build_phoc_raw("a-b")


def test_sdnet_end_to_end_with_phoc(golden_dir):
    """A PHOC conf (Models/SDNet.py:26-27, 51-55, 73, 441-446): OCR / object words carry their 604-d descriptor; the table is
    rebuilt here by the restated builder from the same synthetic spellings."""
    z = np.load(os.path.join(golden_dir, "sdnet_e2e_phoc.npz"))
    V = int(z["vocab_size"])
    opt = default_opt(vocab_size=V, PHOC=True, phoc_dim=604, ocr_embedding="fasttext,phoc,pos,ent,bert")
    cfg = synth.bert_config(vocab_size=2000)
    bw = {k: T(v) for k, v in synth.make_bert_weights(cfg, seed=int(z["seed"])).items()}
    sw = synth.make_sdnet_weights(opt, seed=int(z["seed"]))
    table = np.array([O.build_phoc(w) for w in synth.phoc_vocab_words(V, int(z["seed"]))], dtype=np.float32)
    assert np.array_equal(table.sum(1).astype(np.int32), z["phoc_ones"])
    sw["phoc_embed.weight"] = table
    P = {k: T(v).requires_grad_(v.shape != (1, 1, 1)) for k, v in sw.items()}
    q, ocr, od, gt, _ = synth.synthetic_batch(opt, int(z["B"]), seed=int(z["batch_seed"]), n_q=14, n_ocr=24, n_od=7, bert_vocab=2000,
                                              ragged=True)
    assert ocr["num_cnt"] == z["ocr_num_cnt"].tolist() and torch.equal(ocr["phoc"], ocr["fasttext"])
    scores = O.sdnet_forward(P, opt, bw, cfg, q, ocr, od)
    close(scores, z["scores"], 2e-5, 1e-4, "scores")
    loss = O.instance_bce_with_logits(scores, gt)
    assert abs(loss.item() - float(z["loss"])) < 1e-4
    loss.backward()
    for name, ref_norm in zip(z["grad_names"].tolist(), z["grad_norms"].tolist()):
        g = P[name].grad
        if ref_norm < 0:
            assert g is None or float(g.norm()) == 0.0, name
            continue
        got = float(g.double().norm())
        assert abs(got - ref_norm) <= 2e-4 * max(ref_norm, 1e-3), (name, got, ref_norm)
    ref = z["grad:multi2one.rnns.0.weight_ih_l0[:8]"]
    close(P["multi2one.rnns.0.weight_ih_l0"].grad[:8], ref, 1e-6 + 2e-4 * float(np.abs(ref).max()), 0, "multi2one grad rows")


def test_sdnet_end_to_end_unlocked_bert(golden_dir):
    """No LOCK_BERT (Models/SDNet.py:88-94): the restated encoder is differentiated too; every BERT parameter's gradient norm
    (197 tensors + the two never-reached pooler ones) and a few slices against the reference's backward."""
    z = np.load(os.path.join(golden_dir, "sdnet_e2e_unlocked.npz"))
    opt = default_opt(vocab_size=int(z["vocab_size"]))
    cfg = synth.bert_config(vocab_size=2000, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    bw = {k: T(v).requires_grad_(True) for k, v in synth.make_bert_weights(cfg, seed=int(z["seed"])).items()}
    sw = synth.make_sdnet_weights(opt, seed=int(z["seed"]))
    P = {k: T(v).requires_grad_(v.shape != (1, 1, 1)) for k, v in sw.items()}
    q, ocr, od, gt, _ = synth.synthetic_batch(opt, int(z["B"]), seed=int(z["batch_seed"]), n_q=12, n_ocr=16, n_od=6, bert_vocab=2000,
                                              ragged=True)
    assert ocr["num_cnt"] == z["ocr_num_cnt"].tolist()
    scores = O.sdnet_forward(P, opt, bw, cfg, q, ocr, od)
    close(scores, z["scores"], 2e-5, 1e-4, "scores")
    loss = O.instance_bce_with_logits(scores, gt)
    assert abs(loss.item() - float(z["loss"])) < 1e-4
    loss.backward()
    n_bert = 0
    for name, ref_norm in zip(z["grad_names"].tolist(), z["grad_norms"].tolist()):
        g = bw["bert." + name[len("Bert.bert_model."):]].grad if name.startswith("Bert.") else P[name].grad
        if ref_norm < 0:
            assert g is None or float(g.norm()) == 0.0, name
            continue
        n_bert += name.startswith("Bert.")
        got = float(g.double().norm())
        assert abs(got - ref_norm) <= 3e-4 * max(ref_norm, 1e-3), (name, got, ref_norm)
        if "grad:" + name in z.files:
            ref = z["grad:" + name]
            sl = tuple(slice(0, n) for n in ref.shape)
            close(g[sl], ref, 1e-7 + 3e-4 * float(np.abs(ref).max()), 0, "grad " + name)
    assert n_bert == 197
