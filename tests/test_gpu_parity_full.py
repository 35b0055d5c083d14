"""Parity at BASELINE.json's sizes against fixtures written by the UNMODIFIED reference (oracle/gen_golden.py):

  * sdnet_e2e_full.npz        - bench.py's own workload: B = 64, 30 question words, 100 OCR items, 36 objects, bert-base,
                                forward + loss + backward (scores (64, 101), loss, every parameter-gradient norm, small gradients);
  * sdnet_e2e_full_ragged.npz - the same size with ragged item counts and N(0, 0.05) encoder weights (a harder case for 16-bit
                                operands: the sub-layer branches outweigh the residual stream);
  * sdnet_e2e_stress.npz      - BASELINE's stress shapes: bert-large 24 x 1024, 300 OCR items, 100 objects, B = 2;
  * layers.npz                - the per-op vectors of Models/Layers.py (LinearSelfAttn + weighted_avg, GetFinalScores,
                                StackedBRNN + whole-tensor LN) run through the PRODUCT modules on the GPU;
  * predict_decode.json       - the reference's predict loop, decoded here from device tensors.

Tolerance (BASELINE.json north star): every answer probability within 1e-3 of the fp32 CPU reference.  The modes that are
asserted to hold it on EVERY output at full size are the ones bench.py may quote as its headline (ruart_amd.PASSING_PRECISIONS);
the plain 16-bit modes are throughput modes: their error at full size is measured, printed and pinned to the stated bounds.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from ruart_amd import synth                              # noqa: E402
from ruart_amd.arguments import default_opt               # noqa: E402

DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _build(z, precision, cfg, **opt_extra):
    from ruart_amd.sdnet import SDNet
    opt = default_opt(vocab_size=int(z["vocab_size"]), cuda=True, device=DEV, bert_precision=precision, **opt_extra)
    bw = synth.make_bert_weights(cfg, seed=int(z["seed"]), w_std=float(z["w_std"]))
    # the weights are regenerated from the seed: the stored float64 checksums prove they are the reference's
    ws = np.array([float(np.sum(v.astype(np.float64))) for _, v in sorted(bw.items())])
    assert np.allclose(ws, z["bert_wsum"], rtol=0, atol=1e-9), "BERT weights differ from the generator's"
    opt["bert_state"], opt["bert_config"] = bw, cfg
    sw = synth.make_sdnet_weights(opt, seed=int(z["seed"]))
    net = SDNet(opt, {"glove_embedding": T(sw["glove_embed.weight"]), "fast_embedding": T(sw["fast_embed.weight"])})
    missing, unexpected = net.load_state_dict({k: T(v) for k, v in sw.items()}, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return net.to(DEV), opt


def _step_and_compare(net, batch, z, tol_p, tol_g, label):
    """forward + loss + backward; returns the measured errors after asserting the bounds."""
    import ruart_amd.layers as L
    q, ocr, od, gt, _ = batch
    L.set_dropout_prob(0.0)
    net.train()
    net.drop_emb = False
    scores, _ = net(q, ocr, od)
    net.check_nan()
    got = scores.detach().float().cpu().numpy()
    ref = z["scores"]
    assert got.shape == ref.shape and np.allclose(got.sum(1), 1.0, atol=1e-5)
    d = np.abs(got - ref)
    err, frac = float(d.max()), float((d > 1e-3).mean())
    gt = gt.to(scores.device)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(scores, gt) * gt.size(1)
    loss_err = abs(loss.item() - float(z["loss"]))
    loss.backward()
    params = dict(net.named_parameters())
    worst_norm, worst_elem = (0.0, ""), (0.0, "")
    for name, ref_norm in zip(z["grad_names"].tolist(), z["grad_norms"].tolist()):
        g = params[name].grad
        if ref_norm < 0:
            assert g is None, name
            continue
        if g is None:
            assert name == "ques_merger.linear.bias" and ref_norm < 1e-6, (name, ref_norm)
            continue
        rel = abs(float(g.double().norm()) - ref_norm) / max(ref_norm, 1e-4)
        worst_norm = max(worst_norm, (rel, name))
        key = "grad:" + name
        if key in z.files:
            r = z[key]
            e = np.abs(g.detach().cpu().numpy() - r).max() / max(np.abs(r).max(), 1e-5)
            worst_elem = max(worst_elem, (float(e), name))
    print("%s: max |p - p_ref| %.2e over %d outputs (mean %.2e, fraction above 1e-3: %.1e); |loss - ref| %.2e; worst grad-norm "
          "rel err %.2e (%s); worst grad element err / max|g| %.2e (%s)"
          % (label, err, d.size, float(d.mean()), frac, loss_err, worst_norm[0], worst_norm[1], worst_elem[0], worst_elem[1]))
    assert err < tol_p, "%s: max |p - p_ref| = %.3e >= %.1e" % (label, err, tol_p)
    assert loss_err < 20 * tol_p
    assert worst_norm[0] < tol_g, worst_norm
    assert worst_elem[0] < 2 * tol_g, worst_elem
    return err


# precision -> (bound on every probability, bound on gradient-norm relative error).  1e-3 is the north star; the rows with a
# larger bound are the plain 16-bit throughput modes, whose worst output at B = 64 is documented in DESIGN.md section 2.
FULL_BOUNDS = {"fp32": (5e-5, 2e-3), "x3": (1e-3, 2e-2), "fp16c": (1e-3, 3e-2), "fp16": (2e-2, 2e-1), "bf16": (1e-1, 1.0)}


@pytest.mark.parametrize("precision", ["fp32", "x3", "fp16c", "fp16", "bf16"])
def test_full_size_vs_reference(golden_dir, precision):
    """bench.py's workload against the reference's own output for it."""
    import ruart_amd
    z = np.load(os.path.join(golden_dir, "sdnet_e2e_full.npz"))
    cfg = synth.bert_config(vocab_size=int(z["bert_vocab"]))
    net, opt = _build(z, precision, cfg, max_od_num=36)
    batch = synth.synthetic_batch(opt, int(z["B"]), seed=int(z["batch_seed"]), n_q=30, n_ocr=100, n_od=36,
                                  bert_vocab=int(z["bert_vocab"]), ragged=bool(z["ragged"]))
    assert batch[1]["num_cnt"] == z["ocr_num_cnt"].tolist()
    tol_p, tol_g = FULL_BOUNDS[precision]
    if precision in ruart_amd.PASSING_PRECISIONS:
        assert tol_p <= 1e-3                      # a headline-eligible mode is held to the north-star bound, nothing looser
    _step_and_compare(net, batch, z, tol_p, tol_g, "B=64 bench workload, %s" % precision)


@pytest.mark.parametrize("precision", ["x3", "fp16c", "fp16"])
def test_full_size_ragged_vs_reference(golden_dir, precision):
    """Ragged item counts, N(0, 0.05) encoder weights (branches larger than the residual stream)."""
    z = np.load(os.path.join(golden_dir, "sdnet_e2e_full_ragged.npz"))
    cfg = synth.bert_config(vocab_size=int(z["bert_vocab"]))
    net, opt = _build(z, precision, cfg, max_od_num=36)
    batch = synth.synthetic_batch(opt, int(z["B"]), seed=int(z["batch_seed"]), n_q=30, n_ocr=100, n_od=36,
                                  bert_vocab=int(z["bert_vocab"]), ragged=True)
    assert batch[1]["num_cnt"] == z["ocr_num_cnt"].tolist()
    tol_p, tol_g = {"x3": (1e-3, 2e-2), "fp16c": (1e-3, 3e-2), "fp16": (5e-2, 3e-1)}[precision]
    _step_and_compare(net, batch, z, tol_p, tol_g, "B=64 ragged, w_std 0.05, %s" % precision)


@pytest.mark.parametrize("precision,tol_p,tol_g", [("x3", 1e-3, 2e-2), ("fp16c", 1e-3, 3e-2), ("fp16", 5e-3, 2e-1), ("bf16", 3e-2, 1.0)])
def test_stress_config_vs_reference(golden_dir, precision, tol_p, tol_g):
    """BASELINE config 4: bert-large (Models/Bert/Bert.py:26-33), 300 OCR items, 100 objects."""
    z = np.load(os.path.join(golden_dir, "sdnet_e2e_stress.npz"))
    cfg = synth.bert_config(vocab_size=int(z["bert_vocab"]), hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                            intermediate_size=4096)
    net, opt = _build(z, precision, cfg, BERT_LARGE=True, max_ocr_num=300, max_od_num=100, BERT_large_model_file="unused")
    batch = synth.synthetic_batch(opt, int(z["B"]), seed=int(z["batch_seed"]), n_q=30, n_ocr=300, n_od=100,
                                  bert_vocab=int(z["bert_vocab"]), ragged=True)
    assert batch[1]["num_cnt"] == z["ocr_num_cnt"].tolist()
    _step_and_compare(net, batch, z, tol_p, tol_g, "stress config (bert-large, 300 OCR, 100 objects), %s" % precision)


@pytest.mark.parametrize("precision,tol_p,tol_g", [("x3", 1e-3, 2e-2), ("fp16c", 1e-3, 3e-2), ("fp16", 1.0, 10.0)])
def test_outlier_encoder_vs_reference(golden_dir, precision, tol_p, tol_g):
    """Encoder weights with the heavy tails of a pretrained checkpoint (synth.add_bert_outliers: a few LayerNorm gains x 10-30 in the
    same hidden dimensions of every layer, a few word-embedding columns x 20, projections N(0, 0.04)): layer outputs reach |v| ~ 450
    (stored in the fixture), far beyond the N(0, s) goldens.  The modes that may be quoted hold 1e-3 here too; the plain f16 mode is
    run for the record only (its error is printed)."""
    z = np.load(os.path.join(golden_dir, "sdnet_e2e_outliers.npz"))
    assert float(z["layer_absmax"].max()) > 300.0
    cfg = synth.bert_config(vocab_size=int(z["bert_vocab"]))
    from ruart_amd.sdnet import SDNet
    opt = default_opt(vocab_size=int(z["vocab_size"]), cuda=True, device=DEV, bert_precision=precision, max_od_num=36)
    bw = synth.add_bert_outliers(synth.make_bert_weights(cfg, seed=int(z["seed"]), w_std=float(z["w_std"])), cfg, seed=int(z["outlier_seed"]))
    ws = np.array([float(np.sum(v.astype(np.float64))) for _, v in sorted(bw.items())])
    assert np.allclose(ws, z["bert_wsum"], rtol=0, atol=1e-9), "BERT weights differ from the generator's"
    opt["bert_state"], opt["bert_config"] = bw, cfg
    sw = synth.make_sdnet_weights(opt, seed=int(z["seed"]))
    net = SDNet(opt, {"glove_embedding": T(sw["glove_embed.weight"]), "fast_embedding": T(sw["fast_embed.weight"])})
    missing, unexpected = net.load_state_dict({k: T(v) for k, v in sw.items()}, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    net = net.to(DEV)
    batch = synth.synthetic_batch(opt, int(z["B"]), seed=int(z["batch_seed"]), n_q=30, n_ocr=100, n_od=36, bert_vocab=int(z["bert_vocab"]),
                                  ragged=True)
    assert batch[1]["num_cnt"] == z["ocr_num_cnt"].tolist()
    _step_and_compare(net, batch, z, tol_p, tol_g, "pretrained-like outliers (|v| up to %.0f), %s" % (float(z["layer_absmax"].max()), precision))


# ---- per-op goldens of Models/Layers.py through the product modules on the device -----------------------------------------
@pytest.fixture(scope="module")
def layers_golden(golden_dir):
    return np.load(os.path.join(golden_dir, "layers.npz"))


def _close(got, ref, atol, rtol, what):
    got = got.detach().float().cpu().numpy() if isinstance(got, torch.Tensor) else got
    err = np.abs(got - ref).max()
    assert got.shape == ref.shape and err <= atol + rtol * np.abs(ref).max(), "%s: max err %.3e" % (what, err)


@pytest.mark.parametrize("gemm", ["fp32", "x3"])
def test_linear_self_attn_vs_reference(layers_golden, gemm):
    """Layers.py:320-341 + 529-534 (question merge): ``LinearSelfAttn.merge`` = weighted_avg(x, self(x, mask))."""
    import ruart_amd.layers as L
    from ruart_amd import ops
    z = layers_golden
    L.set_dropout_prob(0.0)
    ops.trunk_gemm = gemm
    m = L.LinearSelfAttn(z["merge_x"].shape[2]).to(DEV)
    m.linear.weight.data, m.linear.bias.data = T(z["merge_w"]).to(DEV), T(z["merge_b"]).to(DEV)
    x = T(z["merge_x"]).to(DEV).requires_grad_()
    mask = T(z["merge_mask"]).to(DEV)
    _close(m(x, mask), z["merge_alpha"], 1e-6, 1e-5, "alpha")
    y = m.merge(x, mask)
    _close(y, z["merge_y"], 2e-6, 1e-5, "y")
    y.backward(T(z["merge_gy"]).to(DEV))
    _close(x.grad, z["merge_gx"], 1e-5, 1e-4, "gx")
    _close(m.linear.weight.grad, z["merge_gw"], 1e-5, 1e-4, "gw")


@pytest.mark.parametrize("gemm", ["fp32", "x3"])
def test_get_final_scores_vs_reference(layers_golden, gemm):
    """Layers.py:352-432 (useES, no_answer, mask_flag), forward and every gradient; the dead GRU gets none."""
    import ruart_amd.layers as L
    from ruart_amd import ops
    z = layers_golden
    L.set_dropout_prob(0.0)
    ops.trunk_gemm = gemm
    X, Hh = z["score_x"].shape[2], z["score_h0"].shape[1]
    m = L.GetFinalScores(X, Hh, yesno=False, no_answer=True, useES=True).to(DEV)
    for n_, p in m.named_parameters():
        p.data = T(z["score_w_" + n_]).to(DEV)
    x = T(z["score_x"]).to(DEV).requires_grad_()
    h0 = T(z["score_h0"]).to(DEV).requires_grad_()
    y = m(x, h0, T(z["score_mask"]).to(DEV), int(z["score_ES"]), mask_flag=True)
    tol = 1e-6 if gemm == "fp32" else 2e-5
    _close(y, z["score_y"], tol, 1e-5, "scores")
    y.backward(T(z["score_gy"]).to(DEV))
    _close(x.grad, z["score_gx"], 10 * tol, 1e-4, "gx")
    _close(h0.grad, z["score_gh0"], 10 * tol, 1e-4, "gh0")
    nograd = set(z["score_nograd"].tolist())
    for n_, p in m.named_parameters():
        if n_ in nograd:
            assert p.grad is None, n_
        else:
            _close(p.grad, z["score_g_" + n_], 10 * tol, 1e-4, "g " + n_)


@pytest.mark.parametrize("tag,nl,bid,ln", [("rnn_a", 2, True, True), ("rnn_b", 1, False, False)])
@pytest.mark.parametrize("gemm", ["fp32", "x3"])
def test_stacked_brnn_vs_reference(layers_golden, tag, nl, bid, ln, gemm):
    """Layers.py:124-180: stacked (Bi)LSTM, whole-tensor layer norm after every layer, per-layer outputs and all gradients."""
    import ruart_amd.layers as L
    from ruart_amd import ops
    z = layers_golden
    L.set_dropout_prob(0.0)
    ops.trunk_gemm = gemm
    x0 = z[tag + "_x"]
    Hh = z[tag + "_w_rnns.0.weight_hh_l0"].shape[1]
    m = L.StackedBRNN(x0.shape[2], Hh, nl, bidirectional=bid).to(DEV)
    for n_, p in m.named_parameters():
        p.data = T(z[tag + "_w_" + n_]).to(DEV)
    x = T(x0).to(DEV).requires_grad_()
    y, ys = m(x, None, return_list=True, LN=ln)
    tol = 2e-6 if gemm == "fp32" else 5e-5
    _close(y, z[tag + "_y"], tol, 1e-5, "y")
    _close(ys[0], z[tag + "_y0"], tol, 1e-5, "y0")
    (y * T(z[tag + "_gy"]).to(DEV)).sum().add((ys[0] * T(z[tag + "_gy0"]).to(DEV)).sum()).backward()
    _close(x.grad, z[tag + "_gx"], 20 * tol, 1e-4, "gx")
    for n_, p in m.named_parameters():
        _close(p.grad, z[tag + "_g_" + n_], 20 * tol, 2e-4, "g " + n_)


@pytest.mark.parametrize("gemm", ["fp32", "x3"])
def test_deep_attention_vs_reference(golden_dir, gemm):
    """a9, Layers.py:471-524: the product's DeepAttention module (three fused attentions + BiLSTM, return_bef_rnn) on the reference's own
    per-op vectors - outputs and every input / parameter gradient (tests/golden/layers_extra.npz)."""
    import ruart_amd.layers as L
    from ruart_amd import ops
    z = np.load(os.path.join(golden_dir, "layers_extra.npz"))
    L.set_dropout_prob(0.0)
    ops.trunk_gemm = gemm
    B, L1, L2, Wd, Hh, HL, per = [int(v) for v in z["deep_dims"]]
    m = L.DeepAttention({"hidden_size": Hh, "highlvl_hidden_size": HL}, abstr_list_cnt=2, deep_att_hidden_size_per_abstr=per,
                        correlation_func=3, word_hidden_size=Wd).to(DEV)
    for n_, p in m.named_parameters():
        p.data = T(z["deep_w_" + n_]).to(DEV)
    grab = lambda name, n: [T(z["deep_%s_%d" % (name, i)]).to(DEV).requires_grad_() for i in range(n)]
    x1w, x1a, x2w, x2a = grab("x1_word", 1), grab("x1_abstr", 2), grab("x2_word", 1), grab("x2_abstr", 3)
    h, pre = m(x1w, x1a, x2w, x2a, T(z["deep_m1"]).to(DEV), T(z["deep_m2"]).to(DEV), return_bef_rnn=True)
    tol = 2e-6 if gemm == "fp32" else 5e-5
    _close(h, z["deep_h"], tol, 1e-5, "h")
    _close(pre, z["deep_pre"], tol, 1e-5, "pre")
    ((h * T(z["deep_gh"]).to(DEV)).sum() + (pre * T(z["deep_gpre"]).to(DEV)).sum()).backward()
    for name, lst in (("x1_word", x1w), ("x1_abstr", x1a), ("x2_word", x2w), ("x2_abstr", x2a)):
        for i, t in enumerate(lst):
            _close(t.grad, z["deep_g_%s_%d" % (name, i)], 20 * tol, 1e-4, "g %s %d" % (name, i))
    for n_, p in m.named_parameters():
        _close(p.grad, z["deep_g_" + n_], 20 * tol, 2e-4, "g " + n_)


@pytest.mark.parametrize("gemm", ["fp32", "x3"])
def test_prealign_vs_reference(golden_dir, gemm):
    """a5, SDNet.py:495-551: ``SDNet._prealign`` (scatter the real words of a sample into one row by host-built indices, fused attention
    over the question's word vectors, gather back) against the reference's get_prealign_emb with its four Python loops, on the packed
    word layout the product uses: values at every real word, gradients of the item / question vectors and of the projection."""
    import ruart_amd.layers as L
    from ruart_amd import ops
    from ruart_amd.batch import BatchIndex
    from ruart_amd.sdnet import SDNet
    z = np.load(os.path.join(golden_dir, "layers_extra.npz"))
    nq, nocr, nod, bv, V = [int(v) for v in z["pre_args"]]
    opt = default_opt(vocab_size=V, cuda=True, device=DEV)
    q, ocr, od, _, _ = synth.synthetic_batch(opt, int(z["pre_B"]), seed=int(z["pre_seed"]), n_q=nq, n_ocr=nocr, n_od=nod, bert_vocab=bv, ragged=True)
    assert ocr["num_cnt"] == z["pre_ocr_num_cnt"].tolist()
    L.set_dropout_prob(0.0)
    ops.trunk_gemm = gemm
    pa = L.Attention(300, z["pre_w"].shape[0], correlation_func=3, do_similarity=True).to(DEV)
    pa.scoring.linear.weight.data = T(z["pre_w"]).to(DEV)
    host = type("Host", (), {"pre_align": pa})()                       # _prealign reads nothing else of the SDNet
    bi = BatchIndex(q, ocr, od, opt, torch.device(DEV))
    key_q = "fasttext" if "fasttext" in opt["q_embedding"] else "glove"
    qe = T(z["pre_q_emb"]).to(DEV).requires_grad_()
    q_mask = q[key_q + "_mask"].to(DEV).to(torch.uint8)
    tol = 2e-6 if gemm == "fp32" else 5e-5
    for tag, idx in (("ocr", bi.ocr), ("od", bi.od)):
        emb = T(z["pre_%s_emb" % tag]).to(DEV).requires_grad_()           # (items, Lw, 300) as the reference holds it
        flat = idx.dev["flat_word"]
        words = emb.reshape(-1, 300)[flat]                               # the product's packed (real words, 300) layout
        out = SDNet._prealign(host, words, idx, qe, q_mask)
        ref = T(z["pre_%s_out" % tag]).reshape(-1, 300)[flat.cpu()]
        _close(out, ref.numpy(), tol, 1e-5, tag + " out")
        g = T(z["pre_g_" + tag]).reshape(-1, 300)[flat.cpu()].to(DEV)
        (out * g).sum().backward()
        gref = T(z["pre_g" + tag]).reshape(-1, 300)[flat.cpu()]
        _close(emb.grad.reshape(-1, 300)[flat], gref.numpy(), 20 * tol, 1e-4, tag + " g emb")
    _close(qe.grad, z["pre_gq"], 20 * tol, 1e-4, "g q")
    _close(pa.scoring.linear.weight.grad, z["pre_gw"], 20 * tol, 2e-4, "g w")


@pytest.mark.parametrize("variant", ["no_answer", "plain"])
def test_predict_decode_on_device(golden_dir, variant):
    """``trainer.decode_predictions`` fed DEVICE scores (as ``predict`` feeds it) against the reference's predict loop
    (Models/SDNetTrainer.py:391-450): chosen slot, answer string, score, ANLS and ACC sums."""
    from ruart_amd.trainer import decode_predictions
    with open(os.path.join(golden_dir, "predict_decode.json")) as f:
        z = json.load(f)
    cases, want = z["cases"], z["expected"][variant]
    width = z["n_slots"] if variant == "no_answer" else z["n_slots"] - 1
    scores = torch.tensor([c["prob"][:width] for c in cases], device=DEV)
    extra = [{"q_id": c["q_id"], "answers": c["answers"], "ocr_list": c["ocr_list"], "image_path": "x"} for c in cases]
    opt = {"label_no_answer": True} if variant == "no_answer" else {}
    anls, acc, res, save_res = decode_predictions(scores, [c["num_cnt"] for c in cases], extra, opt)
    assert res == want["res"]
    for got, ref in zip(save_res, want["save_res"]):
        assert got == ref, (got, ref)
    assert abs(anls - want["ANLS"]) < 1e-9 and abs(acc - want["ACC"]) < 1e-9
