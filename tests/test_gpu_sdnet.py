"""End-to-end parity of the product SDNet (HIP path) on a real MI355X against the reference's golden outputs
(tests/golden/sdnet_e2e.npz: scores, loss and gradients of the UNMODIFIED reference on the same seeded inputs),
and trainer-level behaviour.

Tolerance (BASELINE.json north star): answer probabilities within 1e-3 of the fp32 CPU reference.  That bound is met with
16-bit MFMA operands in the f16 form (11 significand bits; same MFMA rate as bf16), which is the default; the bf16 form
(8 significand bits) lands at 2e-3..5e-3 on these seeded random weights and is held to 1e-2; fp32 validation mode to 5e-5."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from ruart_amd import synth                              # noqa: E402
from ruart_amd.arguments import default_opt               # noqa: E402


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def build(z, precision, device="cuda:0", **extra):
    from ruart_amd.sdnet import SDNet
    opt = default_opt(vocab_size=int(z["vocab_size"]), cuda=True, device=device, bert_precision=precision, **extra)
    cfg = synth.bert_config(vocab_size=2000)
    opt["bert_state"] = synth.make_bert_weights(cfg, seed=int(z["seed"]))
    opt["bert_config"] = cfg
    sw = synth.make_sdnet_weights(opt, seed=int(z["seed"]))
    emb = {"glove_embedding": T(sw["glove_embed.weight"]), "fast_embedding": T(sw["fast_embed.weight"])}
    net = SDNet(opt, emb)
    missing, unexpected = net.load_state_dict({k: T(v) for k, v in sw.items()}, strict=False)
    assert not missing and not unexpected, (missing, unexpected)      # state-dict keys == the reference's
    return net.to(device), opt


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "sdnet_e2e.npz"))


@pytest.mark.parametrize("precision,tol_p,tol_g", [("fp32", 5e-5, 2e-3), ("x3", 2e-4, 1e-2), ("fp16c", 2e-4, 1e-2), ("fp16", 1e-3, 1e-1),
                                                   ("bf16", 1e-2, 6e-1)])
def test_sdnet_forward_backward_vs_reference(golden, precision, tol_p, tol_g):
    import ruart_amd.layers as L
    z = golden
    net, opt = build(z, precision)
    q, ocr, od, gt, _ = synth.synthetic_batch(opt, int(z["B"]), seed=int(z["batch_seed"]), n_q=30, n_ocr=100, n_od=30,
                                              bert_vocab=2000, ragged=True)
    assert ocr["num_cnt"] == z["ocr_num_cnt"].tolist()
    L.set_dropout_prob(0.0)
    net.train()
    net.drop_emb = False
    scores, _ = net(q, ocr, od)
    net.check_nan()
    got = scores.detach().cpu().numpy()
    err = np.abs(got - z["scores"]).max()
    assert got.shape == z["scores"].shape and np.allclose(got.sum(1), 1.0, atol=1e-5)
    assert err < tol_p, "max |p - p_ref| = %.3e" % err
    gt = gt.to(scores.device)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(scores, gt) * gt.size(1)
    assert abs(loss.item() - float(z["loss"])) < 10 * tol_p
    loss.backward()
    params = dict(net.named_parameters())
    worst_norm, worst_elem = (0.0, ""), (0.0, "")
    for name, ref_norm in zip(z["grad_names"].tolist(), z["grad_norms"].tolist()):
        g = params[name].grad
        if ref_norm < 0:
            assert g is None, name                    # non-trainable scalar / the dead GRU: no gradient, as in the reference
            continue
        if g is None:
            # ques_merger.linear.bias: a constant added to every key's score; softmax is shift invariant, the reference's
            # gradient for it is rounding noise (~1e-9) and the fused kernel does not materialise it
            assert name == "ques_merger.linear.bias" and ref_norm < 1e-6, (name, ref_norm)
            continue
        rel = abs(float(g.double().norm()) - ref_norm) / max(ref_norm, 1e-4)
        worst_norm = max(worst_norm, (rel, name))
        key = "grad:" + name
        if key in z.files:
            ref = z[key]
            e = np.abs(g.detach().cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-5)
            worst_elem = max(worst_elem, (float(e), name))
    print("precision %s: max |dp| %.2e; worst grad-norm rel err %.2e (%s); worst grad element err / max|g| %.2e (%s)"
          % (precision, err, worst_norm[0], worst_norm[1], worst_elem[0], worst_elem[1]))
    assert worst_norm[0] < tol_g, worst_norm
    assert worst_elem[0] < 2 * tol_g, worst_elem


def test_trainer_update_and_predict(golden):
    from ruart_amd.trainer import SDNetTrainer
    z = golden
    opt = default_opt(vocab_size=1500, cuda=True, DROPOUT=0.0, dropout_emb=0.0)      # deterministic: same batch must improve
    cfg = synth.bert_config(vocab_size=2000)
    opt["bert_state"], opt["bert_config"] = synth.make_bert_weights(cfg, seed=1033), cfg
    sw = synth.make_sdnet_weights(opt, seed=1033)
    tr = SDNetTrainer(opt, device="cuda:0")
    tr.setup_model({"glove_embedding": T(sw["glove_embed.weight"]), "fast_embedding": T(sw["fast_embed.weight"])})
    pinned = tr.network.fast_embed.weight.data[opt["tune_partial"]:].clone()
    batch = synth.synthetic_batch(opt, 4, seed=11, n_q=12, n_ocr=30, n_od=9, bert_vocab=2000, ragged=True)
    losses = [tr.update(tr.ToCUDA(batch), i) for i in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses     # same batch: the loss must go down
    assert torch.equal(tr.network.fast_embed.weight.data[opt["tune_partial"]:], pinned)   # rows >= tune_partial re-pinned
    assert tr.network.get_answer.rnn.weight_ih.grad is None                                # dead GRU never trained
    loss, anls, acc, res, save_res = tr.predict(tr.ToCUDA(batch))
    assert len(res) == 4 and all("answer" in r for r in res)
    # checkpoint round trip with the reference's format
    path = "/tmp/ruart_ckpt_test.pt"
    tr.save_for_predict(path)
    ck = torch.load(path, map_location="cpu")
    assert not any(k.startswith("Bert") for k in ck["state_dict"]["network"])
    before = tr.network.alphaBERT.detach().clone()
    with torch.no_grad():
        tr.network.alphaBERT.add_(1.0)
    tr.load_model(path)
    assert torch.equal(tr.network.alphaBERT.detach(), before)


def test_encoder_prefetch_is_bitwise_equivalent():
    """update(batch, next_batch=...) runs the frozen encoder of the following batch one step ahead on its own CU-masked stream
    (two alternating buffer sets).  The losses and the trained parameters must equal the inline schedule bit for bit, also
    when a step without lookahead, an evaluation pass or a repeated batch falls between prefetched steps."""
    from ruart_amd.trainer import SDNetTrainer

    def run(prefetch):
        opt = default_opt(vocab_size=1500, cuda=True, DROPOUT=0.0, dropout_emb=0.0)
        cfg = synth.bert_config(vocab_size=2000)
        opt["bert_state"], opt["bert_config"] = synth.make_bert_weights(cfg, seed=1033), cfg
        sw = synth.make_sdnet_weights(opt, seed=1033)
        tr = SDNetTrainer(opt, device="cuda:0")
        tr.setup_model({"glove_embedding": T(sw["glove_embed.weight"]), "fast_embedding": T(sw["fast_embed.weight"])})
        bs = [tr.ToCUDA(synth.synthetic_batch(opt, 3 + (i % 2), seed=20 + i, n_q=10, n_ocr=20 + 3 * i, n_od=5, bert_vocab=2000,
                                              ragged=True)) for i in range(4)]
        order = [0, 1, 2, 2, 3, 0, 1]                       # includes a repeated batch
        losses = []
        for k, i in enumerate(order):
            nxt = bs[order[k + 1]] if prefetch and k + 1 < len(order) and k != 3 else None      # step 3: no lookahead
            if prefetch == "staged":                         # the loader hook of train(): runs inside update, its result is kept
                losses.append(tr.update(bs[i], k, next_batch=nxt, stage_next=lambda k=k: ("staged", k)))
                assert tr.staged == ("staged", k)
            else:
                losses.append(tr.update(bs[i], k, next_batch=nxt))
            if k == 4:
                losses.append(tr.predict(bs[1])[0])          # an inline encoder pass between two prefetched steps
                if prefetch:                                 # evaluation with its own lookahead (unmasked run-ahead stream, step stream)
                    losses.append(tr.predict(bs[2], next_batch=bs[3])[0])
                    losses.append(tr.predict(bs[3])[0])      # ... consumes that pass
                else:
                    losses.append(tr.predict(bs[2])[0])
                    losses.append(tr.predict(bs[3])[0])
        return losses, {n: p.detach().clone() for n, p in tr.network.named_parameters() if p.requires_grad}

    la, pa = run(False)
    lb, pb = run(True)
    lc, pc = run("staged")
    assert la == lb == lc, (la, lb, lc)
    for n in pa:
        assert torch.equal(pa[n], pb[n]) and torch.equal(pa[n], pc[n]), n


def test_train_evaluate_train_with_lookahead_is_bitwise_equivalent():
    """A training step's run-ahead pass (CU-masked stream) that evaluation DROPS may still be running when the first evaluation
    lookahead starts on the other (unmasked) stream and writes the same buffer sets: the new pass has to start behind it
    (Bert.prefetch: ``_last_pf_event``).  Result must equal the schedule without any lookahead, bit for bit."""
    from ruart_amd.trainer import SDNetTrainer

    def run(prefetch):
        opt = default_opt(vocab_size=1500, cuda=True, DROPOUT=0.0, dropout_emb=0.0)
        cfg = synth.bert_config(vocab_size=2000)
        opt["bert_state"], opt["bert_config"] = synth.make_bert_weights(cfg, seed=1033), cfg
        sw = synth.make_sdnet_weights(opt, seed=1033)
        tr = SDNetTrainer(opt, device="cuda:0")
        tr.setup_model({"glove_embedding": T(sw["glove_embed.weight"]), "fast_embedding": T(sw["fast_embed.weight"])})
        bs = [tr.ToCUDA(synth.synthetic_batch(opt, 4, seed=40 + i, n_q=10, n_ocr=40 + 5 * i, n_od=8, bert_vocab=2000, ragged=True))
              for i in range(4)]
        out = []
        for rep in range(2):
            out.append(float(tr.update(bs[0], 2 * rep, next_batch=bs[1] if prefetch else None)))     # leaves bs[1]'s pass in flight
            out.append(tr.predict(bs[2], next_batch=bs[3] if prefetch else None)[0])               # drops it; lookahead on the other stream
            out.append(tr.predict(bs[3])[0])
            out.append(float(tr.update(bs[1], 2 * rep + 1, next_batch=bs[0] if prefetch else None)))
        tr.close()
        return out, {n: p.detach().clone() for n, p in tr.network.named_parameters() if p.requires_grad}

    la, pa = run(False)
    lb, pb = run(True)
    assert la == lb, (la, lb)
    for n in pa:
        assert torch.equal(pa[n], pb[n]), n


def test_scorer_without_variational_dropout_takes_the_elementwise_path(golden):
    """Without VARIATIONAL_DROPOUT the reference drops x element-wise inside BilinearSeqAttn (Layers.py:32-39, 454); the fused scorer
    only models the (B, D) variational mask, so such a training configuration must run the op-by-op form (and the default one the
    fused form)."""
    import ruart_amd.layers as L
    net, opt = build(golden, "fp16")
    ga = net.get_answer
    calls = []
    orig = ga._forward_fused
    ga._forward_fused = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    x = torch.randn(2, 30, ga.attn.linear.out_features, device="cuda")
    h0 = torch.randn(2, ga.attn.linear.in_features, device="cuda")
    mask = torch.ones(2, 30, device="cuda")
    seq0, p0 = L.do_seq_dropout, L.dropout_p
    try:
        L.set_dropout_prob(0.3)
        ga.train()
        L.set_seq_dropout(True)
        ga(x, h0, mask, 10)
        assert len(calls) == 1
        L.set_seq_dropout(False)
        out = ga(x, h0, mask, 10)
        assert len(calls) == 1 and out.shape == (2, 31)
        ga.eval()
        ga(x, h0, mask, 10)                         # evaluation: no dropout at all, fused again
        assert len(calls) == 2
    finally:
        L.set_seq_dropout(seq0)
        L.set_dropout_prob(p0)
        ga._forward_fused = orig


def test_host_index_from_collate_gives_the_same_forward(golden):
    """A batch whose index was prepared by VQA_collate(prepare_index=True) (and pickled, as a DataLoader worker would) must
    produce bit-identical scores to one prepared inside ToCUDA."""
    import pickle
    from ruart_amd.batch import BatchIndex
    net, opt = build(golden, "fp16")
    net.eval()
    b1 = synth.synthetic_batch(opt, 3, seed=5, n_q=9, n_ocr=17, n_od=4, bert_vocab=2000, ragged=True)
    b2 = synth.synthetic_batch(opt, 3, seed=5, n_q=9, n_ocr=17, n_od=4, bert_vocab=2000, ragged=True)
    b2[0]["_ruart_host_index"] = pickle.loads(pickle.dumps(BatchIndex(b2[0], b2[1], b2[2], opt)))
    with torch.no_grad():
        s1, _ = net(b1[0], b1[1], b1[2])
        s2, _ = net(b2[0], b2[1], b2[2])
    assert b2[0]["_ruart_index"] is b2[0]["_ruart_host_index"]
    assert torch.equal(s1, s2)


def test_fused_adamax_matches_torch():
    """FusedAdamax.clip_and_step == clip_grad_norm_ + torch.optim.Adamax.step over several steps (same clip coefficient, same
    update to fp32 rounding), including a parameter without gradient and an embedding table whose re-pinned rows are skipped."""
    from ruart_amd.optim import FusedAdamax
    g = torch.Generator().manual_seed(0)
    shapes = [(20000, 300), (1000, 1250), (1000,), (7, 3, 5), (8193,), (250, 1800)]
    ref = [torch.nn.Parameter((torch.randn(*s, generator=g) * 0.1).cuda()) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    fixed = ref[0].detach()[1000:].clone()
    o_ref = torch.optim.Adamax(ref, lr=2e-3)
    o_mine = FusedAdamax(mine, lr=2e-3, pinned={mine[0]: 1000})
    for step in range(5):
        for k, (a, b) in enumerate(zip(ref, mine)):
            if k == 3 and step % 2 == 0:
                a.grad = b.grad = None
                continue
            gr = (torch.randn(a.shape, generator=g) * (5.0 if step == 1 else 0.01)).cuda()     # step 1 is clipped hard
            a.grad, b.grad = gr.clone(), gr.clone()
        norm = torch.nn.utils.clip_grad_norm_(ref, 10.0)
        o_ref.step()
        o_mine.clip_and_step(10.0)
        for t in (ref[0], mine[0]):
            t.data[1000:] = fixed                                             # the trainer's re-pin
        assert abs(float(o_mine.norm_coef[0]) - float(norm)) <= 1e-4 * float(norm)
        for a, b in zip(ref, mine):
            assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max())), (step, a.shape)
    assert torch.equal(mine[0].detach()[1000:], fixed)


def test_variational_dropout_contract():
    """Layers.py:23-30: one mask per (row, feature) shared over time, scaled by 1/(1-p)."""
    import ruart_amd.layers as L
    L.set_seq_dropout(True)
    x = torch.ones(5, 7, 11, device="cuda:0")
    y = L.dropout(x, p=0.4, training=True)
    assert torch.equal(y[:, 0], y[:, 3])
    vals = set(y.unique().cpu().tolist())
    assert vals <= {0.0, 1.0 / 0.6} or all(abs(v) < 1e-6 or abs(v - 1 / 0.6) < 1e-5 for v in vals)
    assert torch.equal(L.dropout(x, p=0.4, training=False), x)


@pytest.mark.parametrize("precision,tol", [("x3", 2e-4), ("fp16c", 2e-4)])
def test_frozen_bert_dropout_is_opt_in(golden, precision, tol):
    """``opt['bert_frozen_dropout']`` reproduces the reference's actual training-mode behaviour (Models/SDNetTrainer.py:332 flips the
    dropout(0.1) layers inside the frozen BERT back on, Models/Bert/modeling.py:198, 244, 263, 302): training passes become random,
    evaluation passes stay the deterministic ones; without the option nothing changes."""
    import ruart_amd.layers as L
    z = golden
    net, opt = build(z, precision, bert_frozen_dropout=True)
    # (fp16c: the training-mode passes run on the 16-bit training kernels, bert_train16.BertModelTrainable16.layers_nograd)
    assert type(net.Bert.__dict__["_dropout_model"]).__name__ == ("BertModelTrainable16" if precision == "fp16c" else "BertModelTrainable")
    q, ocr, od, gt, _ = synth.synthetic_batch(opt, int(z["B"]), seed=int(z["batch_seed"]), n_q=30, n_ocr=100, n_od=30, bert_vocab=2000, ragged=True)
    L.set_dropout_prob(0.0)
    net.drop_emb = False
    assert not any(k.startswith("Bert.") for k in net.state_dict())           # the fp32 copy is not part of the checkpoint
    net.eval()
    with torch.no_grad():
        e = net(q, ocr, od)[0].cpu().numpy()
    assert np.abs(e - z["scores"]).max() < tol                                 # evaluation: the deterministic encoder
    net.train()
    with torch.no_grad():
        a = net(q, ocr, od)[0].cpu().numpy()
        b = net(q, ocr, od)[0].cpu().numpy()
    assert np.abs(a - b).max() > 1e-5                                          # two training passes draw different masks
    assert np.isfinite(a).all() and np.allclose(a.sum(1), 1.0, atol=1e-5)       # still probability rows (how far they move is up to the masks)
    scores, _ = net(q, ocr, od)
    torch.nn.functional.binary_cross_entropy_with_logits(scores, gt.to(scores.device)).backward()
    assert net.alphaBERT.grad is not None and torch.isfinite(net.alphaBERT.grad).all()


def test_no_cpu_fallback():
    from ruart_amd import hip, ops
    with pytest.raises(hip.HipError):
        ops.fused_attention(torch.zeros(1, 2, 3), torch.zeros(1, 2, 3), torch.zeros(1, 2, 3), torch.ones(1, 2))


def test_full_size_properties():
    """BASELINE.json's full per-GPU shape (B=64, q=30, 100 OCR items, 36 objects, bert-base): too big for the CPU oracle in a
    test, so parity is checked through size-independent properties:
      * every score row is a probability vector; nothing is NaN;
      * batch-permutation equivariance: samples only interact through the whole-tensor layer-norm statistics, which are
        permutation invariant - permuting the samples must permute the score rows (up to fp32 summation order);
      * packing invariance: dropping padded word-piece slots (pack=True) vs keeping them with the reference's -10000 mask;
      * the f16 production path against the exact-fp32 validation path at full size: mean, 99th percentile, worst case."""
    from ruart_amd.sdnet import SDNet
    import ruart_amd.layers as L
    dev = "cuda:0"
    cfg = synth.bert_config(vocab_size=3000)
    bw = synth.make_bert_weights(cfg, seed=21)
    B = 64

    def make(precision, **extra):
        opt = default_opt(vocab_size=2000, cuda=True, device=dev, bert_precision=precision, max_od_num=36, **extra)
        opt["bert_state"], opt["bert_config"] = bw, cfg
        sw = synth.make_sdnet_weights(opt, seed=21)
        net = SDNet(opt, {"glove_embedding": T(sw["glove_embed.weight"]), "fast_embedding": T(sw["fast_embed.weight"])})
        net.load_state_dict({k: T(v) for k, v in sw.items()})
        net.to(dev).eval()
        net.drop_emb = False
        return net, opt

    net32, opt = make("fp32")
    batch = synth.synthetic_batch(opt, B, seed=31, n_q=30, n_ocr=100, n_od=36, bert_vocab=3000, ragged=True)

    def run(net, b):
        q, ocr, od = [dict(x) for x in b[:3]]
        q.pop("_ruart_index", None)
        with torch.no_grad():
            s, _ = net(q, ocr, od)
        net.check_nan()
        return s.float().cpu()

    s32 = run(net32, batch)
    assert s32.shape == (B, opt["max_ocr_num"] + 1) and torch.isfinite(s32).all()
    assert float((s32.sum(1) - 1).abs().max()) < 1e-5
    # masked answer slots (beyond the sample's items) carry exactly zero probability
    for b in range(B):
        n = batch[1]["num_cnt"][b]
        assert n == 100 or float(s32[b, n:-1].abs().max()) == 0.0

    # ---- batch permutation (exact-fp32 path: only fp32 reduction orders move) -------------------------------
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).tolist()

    def permute_items(d, perm):
        starts = np.concatenate([[0], np.cumsum(d["num_cnt"])])
        rows = np.concatenate([np.arange(starts[p], starts[p + 1]) for p in perm])
        out = {}
        for k, v in d.items():
            if k in ("num_cnt", "len_cnt"):
                out[k] = [v[p] for p in perm]
            elif k == "position":
                out[k] = v[perm]
            elif isinstance(v, torch.Tensor):
                out[k] = v[rows]
            else:
                out[k] = [v[r] for r in rows]
        return out

    q, ocr, od = batch[:3]
    qp = {k: (v[perm] if isinstance(v, torch.Tensor) else [v[p] for p in perm]) for k, v in q.items() if k != "_ruart_index"}
    pbatch = (qp, permute_items(ocr, perm), permute_items(od, perm))
    sp = run(net32, pbatch)
    assert float((sp - s32[perm]).abs().max()) < 5e-5
    # ---- packing invariance: dropped padding vs the reference's -10000 mask ----------------------------------
    net32_np, _ = make("fp32", bert_no_pack=True)
    s_np = run(net32_np, batch)
    assert float((s_np - s32).abs().max()) < 5e-5
    del net32_np, net32

    # ---- the f16 production path against the exact-fp32 path at full size -----------------------------------
    net16, _ = make("fp16")
    s16 = run(net16, batch)
    d16 = (s16 - s32).abs()
    perm_noise = float((run(net16, pbatch) - s16[perm]).abs().max())
    frac = float((d16 > 1e-3).float().mean())
    print("full size: max |p_f16 - p_fp32| = %.2e, mean %.2e, fraction of the %d outputs above 1e-3: %.1e; "
          "f16 packing-order noise %.2e" % (float(d16.max()), float(d16.mean()), d16.numel(), frac, perm_noise))
    # Statistical bound of the 16-bit path at full size: 16-bit MFMA operands cannot hold a hard 1e-3 on
    # every one of 6 464 outputs of this random-weight model (the trunk amplifies BERT feature noise ~10x); the exact
    # fp32 mode does.  Mean error, the 99th percentile and the worst case are pinned here.
    assert float(d16.mean()) < 1e-4 and frac < 0.02 and float(d16.max()) < 2e-2
    # ---- the middle mode: fp32 storage everywhere, every GEMM (encoder and trunk) as three bf16 products -----------
    net_m, _ = make("x3")
    dm = (run(net_m, batch) - s32).abs()
    print("full size, x3 mode (encoder + trunk): max |p_x3 - p_fp32| = %.2e, mean %.2e" % (float(dm.max()), float(dm.mean())))
    assert float(dm.max()) < 1e-3                                         # the north-star bound on EVERY output at full size
    del net_m
    # ---- the split-bf16 trunk GEMM alone: fp32 encoder, x3 projections vs the library's exact fp32 GEMMs -------------
    net_x3, _ = make("fp32", ruart_trunk_gemm="x3")
    dx3 = (run(net_x3, batch) - s32).abs()
    print("full size, fp32 encoder: max |p_x3 - p_fp32gemm| = %.2e, mean %.2e" % (float(dx3.max()), float(dx3.mean())))
    assert float(dx3.max()) < 1e-3 and float(dx3.mean()) < 5e-6          # ~2^-16 per product: far inside the 1e-3 budget


def test_stress_config_bert_large_and_edge_batches():
    """BASELINE config 4 shapes (300 OCR items, 100 objects, bert-large 24 x 1024) run through the same path, and a degenerate
    batch (B=1, one object) does too.  Functional check: finite probabilities that sum to 1, gradients flow."""
    from ruart_amd.sdnet import SDNet
    import ruart_amd.layers as L
    dev = "cuda:0"
    L.set_dropout_prob(0.0)
    cfg = synth.bert_config(vocab_size=1200, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
    opt = default_opt(vocab_size=800, cuda=True, device=dev, BERT_LARGE=True, max_ocr_num=300, max_od_num=100,
                      BERT_large_model_file="unused")
    opt["bert_state"], opt["bert_config"] = synth.make_bert_weights(cfg, seed=2, w_std=0.03), cfg
    sw = synth.make_sdnet_weights(opt, seed=2)
    assert sw["alphaBERT"].shape == (24,) and sw["multi2one.rnns.0.weight_ih_l0"].shape[1] == 300 + 1024 + 12 + 8 + 300
    net = SDNet(opt, {"glove_embedding": T(sw["glove_embed.weight"]), "fast_embedding": T(sw["fast_embed.weight"])})
    net.load_state_dict({k: T(v) for k, v in sw.items()})
    net.to(dev).train()
    net.drop_emb = False
    q, ocr, od, gt, _ = synth.synthetic_batch(opt, 2, seed=8, n_q=30, n_ocr=300, n_od=100, bert_vocab=1200)
    scores, _ = net(q, ocr, od)
    net.check_nan()
    assert scores.shape == (2, 301) and torch.isfinite(scores).all() and float((scores.sum(1) - 1).abs().max()) < 1e-5
    torch.nn.functional.binary_cross_entropy_with_logits(scores, gt.to(dev)).backward()
    assert torch.isfinite(net.alphaBERT.grad).all() and float(net.multi2one.rnns[0].weight_ih_l0.grad.abs().max()) > 0
    b1 = synth.synthetic_batch(opt, 1, seed=9, n_q=3, n_ocr=12, n_od=1, bert_vocab=1200)
    s1, _ = net(b1[0], b1[1], b1[2])
    assert s1.shape == (1, 301) and torch.isfinite(s1).all()


def test_train_and_predict_from_msgpack(golden_dir, tmp_path):
    """The reference's own entry points, end to end on its on-disk artefacts (Models/SDNetTrainer.py:50-123, 231-251):
    ``train()`` reads train_meta / train / val msgpack, builds ``VQA_Dataset`` + sampler + collate, trains, keeps the best
    checkpoints in ``<datadir>/conf~/run_1``; ``predict_for_test()`` resumes from that checkpoint and writes submission.json.
    BERT comes from a HF-0.x model directory (bert_config.json + pytorch_model.bin) and a vocab.txt, as the shipped conf has it."""
    import json
    import msgpack
    from ruart_amd.trainer import SDNetTrainer
    with open(os.path.join(golden_dir, "dataset_input.json"), encoding="utf-8") as f:
        inp = json.load(f)
    feat = tmp_path / "source" / "data" / "stvqa"
    feat.mkdir(parents=True)
    for split in ("train", "val", "test"):
        with open(feat / (split + "-preprocessed.msgpack"), "wb") as f:
            msgpack.dump({"data": inp["records"]}, f)
    g = np.random.default_rng(3)
    V = 1200
    with open(feat / "train_meta.msgpack", "wb") as f:
        msgpack.dump({"vocab": ["w%d" % i for i in range(V)], "char_vocab": list("abc"),
                      "glove_embedding": g.standard_normal((V, 300)).astype(np.float32).tolist(),
                      "fast_embedding": g.standard_normal((V, 300)).astype(np.float32).tolist()}, f)
    (tmp_path / "vocab.txt").write_text("\n".join(inp["vocab"]) + "\n", encoding="utf-8")
    cfg = synth.bert_config(vocab_size=len(inp["vocab"]), max_position_embeddings=64)      # 12 x 768: the conf's BERT-base
    (tmp_path / "bert").mkdir()
    with open(tmp_path / "bert" / "bert_config.json", "w") as f:
        json.dump(cfg, f)
    torch.save({k: T(v) for k, v in synth.make_bert_weights(cfg, seed=5, w_std=0.05).items()}, tmp_path / "bert" / "pytorch_model.bin")

    def options(**kw):
        opt = default_opt(cuda=True, datadir=str(tmp_path), source_dir="stvqa", BERT_tokenizer_file="vocab.txt",
                          BERT_model_file="bert/", batch_size=4, max_batch_num=5, **kw)
        for k in ("RESUME", "epoch", "vocab_size"):
            opt.pop(k, None)
        return opt

    tr = SDNetTrainer(options(), device="cuda:0")
    tr.train(eval_every=3)
    run = tmp_path / "conf~" / "run_1"
    assert tr.updates == 5 and tr.opt["vocab_size"] == V and np.isfinite(tr.train_loss.avg)
    assert (run / "ANLS_best_model.pt").exists() and (run / "ACC_best_model.pt").exists()
    saved = json.load(open(run / "save_res_last.json"))
    assert len(saved) >= 5 and tr.best_ANLS >= 0                     # 5 val samples (+ the wrap-around of the last batch)

    opt = options(MODEL_PATH="conf~/run_1/ANLS_best_model.pt")
    opt["RESUME"] = True
    te = SDNetTrainer(opt, device="cuda:0")
    te.predict_for_test()
    sub = json.load(open(run / "submission.json"))
    assert len(sub) == 6 and all("answer" in r for r in sub)         # 6 test records: padding of the last batch trimmed
    ck = torch.load(run / "ANLS_best_model.pt", map_location="cpu")["state_dict"]["network"]
    assert torch.equal(te.network.alphaBERT.detach().cpu(), ck["alphaBERT"])


@pytest.mark.parametrize("precision,tol_p,tol_g", [("fp32", 5e-5, 2e-3), ("x3", 2e-4, 1e-2), ("fp16", 3e-3, 1e-1)])
def test_sdnet_with_phoc_features_vs_reference(golden_dir, precision, tol_p, tol_g):
    """A PHOC conf end to end: the 604-d table comes from ``ruart_phoc_table`` on the GPU (same synthetic spellings the reference's
    build_phoc saw), the model looks it up beside the fastText vectors; scores and gradient norms against the reference.
    This batch has only 7-24 candidates per sample, so single probabilities reach 0.3 and the f16 encoder's ~5e-3 logit noise
    shows as up to 1.7e-3 absolute (3e-4 at the 100-candidate bench shape, where the 1e-3 bound is asserted); the split-bf16
    mode (x3) is the one to use when 1e-3 has to hold on any input."""
    import ruart_amd.layers as L
    from ruart_amd.phoc import phoc_table
    from ruart_amd.sdnet import SDNet
    z = np.load(os.path.join(golden_dir, "sdnet_e2e_phoc.npz"))
    V, seed = int(z["vocab_size"]), int(z["seed"])
    opt = default_opt(vocab_size=V, cuda=True, device="cuda:0", bert_precision=precision, PHOC=True, phoc_dim=604,
                      ocr_embedding="fasttext,phoc,pos,ent,bert")
    cfg = synth.bert_config(vocab_size=2000)
    opt["bert_state"], opt["bert_config"] = synth.make_bert_weights(cfg, seed=seed), cfg
    sw = synth.make_sdnet_weights(opt, seed=seed)
    table = phoc_table(synth.phoc_vocab_words(V, seed), "cuda:0")
    assert np.array_equal(table.sum(1).cpu().numpy().astype(np.int32), z["phoc_ones"])
    sw["phoc_embed.weight"] = table.cpu().numpy()
    net = SDNet(opt, {"glove_embedding": T(sw["glove_embed.weight"]), "fast_embedding": T(sw["fast_embed.weight"]),
                      "phoc_embedding": table.cpu()})
    missing, unexpected = net.load_state_dict({k: T(v) for k, v in sw.items()}, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    net = net.to("cuda:0")
    q, ocr, od, gt, _ = synth.synthetic_batch(opt, int(z["B"]), seed=int(z["batch_seed"]), n_q=14, n_ocr=24, n_od=7, bert_vocab=2000,
                                              ragged=True)
    L.set_dropout_prob(0.0)
    net.train()
    net.drop_emb = False
    scores, _ = net(q, ocr, od)
    net.check_nan()
    err = np.abs(scores.detach().cpu().numpy() - z["scores"]).max()
    assert err < tol_p, "max |p - p_ref| = %.3e" % err
    gt = gt.to(scores.device)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(scores, gt) * gt.size(1)
    assert abs(loss.item() - float(z["loss"])) < 50 * tol_p
    loss.backward()
    grads = dict(net.named_parameters())
    for name, ref_norm in zip(z["grad_names"].tolist(), z["grad_norms"].tolist()):
        g = grads[name].grad
        if ref_norm < 0:
            assert g is None or float(g.norm()) == 0.0, name
            continue
        if g is None:                                    # shift-invariant softmax bias: rounding noise in the reference (see above)
            assert name == "ques_merger.linear.bias" and ref_norm < 1e-6, (name, ref_norm)
            continue
        got = float(g.double().norm())
        assert abs(got - ref_norm) <= tol_g * max(ref_norm, 1e-3), (name, got, ref_norm)


def _unlocked(z, precision, **extra):
    from ruart_amd.sdnet import SDNet
    opt = default_opt(vocab_size=int(z["vocab_size"]), cuda=True, device="cuda:0", bert_precision=precision, **extra)
    opt.pop("LOCK_BERT")
    cfg = synth.bert_config(vocab_size=2000, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    opt["bert_state"], opt["bert_config"] = synth.make_bert_weights(cfg, seed=int(z["seed"])), cfg
    sw = synth.make_sdnet_weights(opt, seed=int(z["seed"]))
    net = SDNet(opt, {"glove_embedding": T(sw["glove_embed.weight"]), "fast_embedding": T(sw["fast_embed.weight"])})
    missing, unexpected = net.load_state_dict({k: T(v) for k, v in sw.items()}, strict=False)
    assert not unexpected and all(k.startswith("Bert.bert_model.") for k in missing) and len(missing) == 199
    return net.to("cuda:0"), opt


@pytest.mark.parametrize("precision,tol_p,tol_g,tol_trunk", [("fp32", 5e-5, 2e-3, 2e-3), ("x3", 2e-4, 3e-2, 3e-2), ("x3+16gemm", 3e-3, 1.5e-1, 1.5e-1),
                                                             ("x3+16", 1e-3, 3e-2, 6e-2), ("x3-long", 2e-4, 3e-2, 3e-2), ("x3+16-long", 1e-3, 3e-2, 6e-2)])
def test_unlocked_bert_gradients_vs_reference(golden_dir, precision, tol_p, tol_g, tol_trunk):
    """Conf without LOCK_BERT: the trainable encoder (bert_train.py) under the reference's parameter names; scores, loss and the
    gradient norm of every parameter - 197 BERT tensors included - against the reference's backward, plus gradient slices.
    The exact-fp32 mode pins parity (norms to 3e-5 here).  In the split-bf16 mode (2^-16 per product) this small batch is
    mostly padding, the whole-tensor layer norms (Layers.py:168) see little variance and their backward - a difference of
    tensor-wide means - amplifies the operand error: gradient norms of the tensors around them move by up to ~1.6 % (8 %
    element-wise, identical with a locked encoder: tools-level check in DESIGN.md section 2), so only norms are held, at 3 %."""
    import ruart_amd.layers as L
    # "-long" (round 4): the reference's pass on 90-word questions - BERT sequences of more than 64 word pieces, which the 16-bit encoder
    # used to hand to the fp32-class graph; its attention now runs them as 64-token chunks against the whole sequence
    # (ruart_attn_train_fwd_long / _bwd_long), held to the same bounds
    long_q = precision.endswith("-long")
    precision = precision[:-5] if long_q else precision
    z = np.load(os.path.join(golden_dir, "sdnet_e2e_unlocked_long.npz" if long_q else "sdnet_e2e_unlocked.npz"))
    # "x3+16": the 16-bit trainable encoder (bert_train16.py: one autograd Function over f16 / bf16 kernels, opt['bert_train_gemm'] =
    # '16').  Round 3: a pass without active dropout - this one - runs its FORWARD on the frozen path's fp16c kernels (probabilities within
    # 1e-3, the north-star bound, with the encoder unlocked) and recomputes each layer's activations on the f16 kernels in the backward.  Its 197 BERT tensors are held to the 3 % of the round-1 verdict (measured: 0.7 % worst, 0.14 % median).  The trunk runs the
    # same fp32-class kernels as in "x3"; with the fp16c forward the answer probabilities stay within the 1e-3 asserted here (tol_p;
    # the plain-f16 forward of round 2 moved them by 1.7e-3), and the gradient of the no-answer branch (get_answer.noanswer_*, norm 1e-4) is proportional to
    # p(no answer) - y with p(no answer) ~ 0.04, i.e. it moves by dp / p ~ 4 %.  That is the forward tolerance seen through a small
    # probability, not a backward error, so trunk tensors get their own bound (6 %; every other trunk tensor is within 1.7 %).
    # "x3+16gemm": the fp32-class graph with 16-bit MFMA products for x W^T and dY W
    extra = {"bert_train_gemm": precision.split("+")[1]} if "+" in precision else {}
    if long_q:
        extra.update(max_q_len=int(z["n_q"]) + 10, max_q_bert_len=2 * int(z["n_q"]))
    net, opt = _unlocked(z, precision.split("+")[0], **extra)
    names = dict(net.named_parameters())
    assert set(z["grad_names"].tolist()) == set(names), set(z["grad_names"].tolist()) ^ set(names)   # same state-dict surface
    q, ocr, od, gt, _ = synth.synthetic_batch(opt, int(z["B"]), seed=int(z["batch_seed"]), n_q=int(z["n_q"]) if long_q else 12, n_ocr=16, n_od=6,
                                              bert_vocab=2000, ragged=True)
    L.set_dropout_prob(0.0)
    net.train()
    net.drop_emb = False
    if long_q:
        packed = net.prepare(q, ocr, od).packed
        assert packed.max_len > 64
        if "+16" in precision:                           # ... and the 16-bit encoder takes it (no fall-back to the fp32-class graph)
            from ruart_amd.bert_train16 import BertModelTrainable16
            assert isinstance(net.Bert.bert_model, BertModelTrainable16) and net.Bert.bert_model.supports(packed)
            assert packed.train_plan(packed.ids.device)["n_chunks"] >= 2
    scores, _ = net(q, ocr, od)
    net.check_nan()
    err = np.abs(scores.detach().cpu().numpy() - z["scores"]).max()
    assert err < tol_p, "max |p - p_ref| = %.3e" % err
    gt = gt.to(scores.device)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(scores, gt) * gt.size(1)
    assert abs(loss.item() - float(z["loss"])) < 50 * tol_p
    loss.backward()
    worst, worst_trunk = (0.0, ""), (0.0, "")
    for name, ref_norm in zip(z["grad_names"].tolist(), z["grad_norms"].tolist()):
        g = names[name].grad
        if ref_norm < 0:
            assert g is None or float(g.norm()) == 0.0, name
            continue
        if g is None:
            assert name == "ques_merger.linear.bias" and ref_norm < 1e-6, (name, ref_norm)
            continue
        rel = abs(float(g.double().norm()) - ref_norm) / max(ref_norm, 1e-4)
        if name.startswith("Bert."):
            worst = max(worst, (rel, name))
        else:
            worst_trunk = max(worst_trunk, (rel, name))
        if "grad:" + name in z.files and precision == "fp32":
            ref = z["grad:" + name]
            got = g[tuple(slice(0, n) for n in ref.shape)].detach().cpu().numpy()
            assert np.abs(got - ref).max() <= 2 * tol_g * max(np.abs(ref).max(), 1e-6), name
    print("unlocked %s: max |dp| %.2e, worst grad-norm rel err: BERT %.2e (%s), trunk %.2e (%s)"
          % (precision, err, worst[0], worst[1], worst_trunk[0], worst_trunk[1]))
    assert worst[0] < tol_g, worst
    assert worst_trunk[0] < tol_trunk, worst_trunk


def test_unlocked_16bit_encoder_is_repeatable_and_falls_back(golden_dir):
    """The 16-bit trainable encoder, dropout ON (hash-generated multipliers): two passes from the same generator state give bitwise the
    same scores and gradients (ordered reductions, no atomics), another seed gives other masks; a batch with a sequence longer than one
    64-token attention window stays on the 16-bit kernels (round 4: chunks against the whole sequence) and is just as repeatable; only a
    stream with a key bias still falls back to the fp32-class path of bert_train.py."""
    z = np.load(os.path.join(golden_dir, "sdnet_e2e_unlocked.npz"))
    net, opt = _unlocked(z, "x3", bert_train_gemm="16")
    net.Bert.bert_model.p_hidden = net.Bert.bert_model.p_attn = 0.1
    import ruart_amd.layers as L
    L.set_dropout_prob(0.0)
    net.train()
    net.drop_emb = False
    q, ocr, od, gt, _ = synth.synthetic_batch(opt, int(z["B"]), seed=int(z["batch_seed"]), n_q=12, n_ocr=16, n_od=6, bert_vocab=2000, ragged=True)

    def run(seed):
        torch.manual_seed(seed)
        net.zero_grad(set_to_none=True)
        scores, _ = net(q, ocr, od)
        (torch.nn.functional.binary_cross_entropy_with_logits(scores, gt.to(scores.device)) * gt.size(1)).backward()
        return scores.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}

    s1, g1 = run(5)
    s2, g2 = run(5)
    s3, _ = run(6)
    assert torch.equal(s1, s2) and not torch.equal(s1, s3)
    assert g1.keys() == g2.keys() and all(torch.equal(g1[n], g2[n]) for n in g1), [n for n in g1 if not torch.equal(g1[n], g2[n])][:3]
    assert any(n.startswith("Bert.") for n in g1)
    # a 90-word question: more than 64 word pieces in one sequence
    opt["max_q_len"], opt["max_q_bert_len"] = 100, 180
    ql, ocrl, odl, gtl, _ = synth.synthetic_batch(opt, 2, seed=3, n_q=90, n_ocr=8, n_od=4, bert_vocab=2000, ragged=False)
    from ruart_amd.bert_train16 import BertModelTrainable16
    assert isinstance(net.Bert.bert_model, BertModelTrainable16)
    pk_long = net.prepare(ql, ocrl, odl).packed
    assert pk_long.max_len > 64 and net.Bert.bert_model.supports(pk_long) and pk_long.train_plan(pk_long.ids.device)["n_chunks"] >= 2

    def run_long(seed):
        torch.manual_seed(seed)
        net.zero_grad(set_to_none=True)
        scores, _ = net(ql, ocrl, odl)
        (torch.nn.functional.binary_cross_entropy_with_logits(scores, gtl.to(scores.device)) * gtl.size(1)).backward()
        net.check_nan()
        return scores.detach().clone(), dict(net.named_parameters())["Bert.bert_model.encoder.layer.0.attention.self.key.weight"].grad.detach().clone()

    l1, k1 = run_long(7)
    l2, k2 = run_long(7)
    l3, _ = run_long(8)
    assert torch.equal(l1, l2) and torch.equal(k1, k2) and not torch.equal(l1, l3) and torch.isfinite(k1).all() and float(k1.abs().max()) > 0
    # evaluation: same numbers as a training-mode pass without dropout, nothing kept for a backward pass
    net.Bert.bert_model.p_hidden = net.Bert.bert_model.p_attn = 0.0
    s_train, _ = net(q, ocr, od)
    net.eval()
    with torch.no_grad():
        s_eval, _ = net(q, ocr, od)
    assert torch.equal(s_train.detach(), s_eval) and not s_eval.requires_grad


@pytest.mark.parametrize("train_gemm", ["x3", "16"])
def test_unlocked_bert_trains_with_the_fused_optimizer(train_gemm):
    """The trainer path without LOCK_BERT: BERT's parameters sit in the (fused) Adamax, its own dropout is active in training
    (hidden 0.1 / attention 0.1, modeling.py:198-301) and off in evaluation, the loss on a fixed batch goes down, the
    checkpoint keeps the reference's habit of leaving ``Bert.*`` out.  "16": the 16-bit encoder (bert_train16.py), whose dropout
    multipliers are hash-generated in the kernels and regenerated in the backward pass."""
    from ruart_amd.trainer import SDNetTrainer
    opt = default_opt(vocab_size=600, cuda=True, DROPOUT=0.0, dropout_emb=0.0, lr=2e-4, bert_train_gemm=train_gemm)
    opt.pop("LOCK_BERT")
    cfg = synth.bert_config(vocab_size=2000)
    opt["bert_state"], opt["bert_config"] = synth.make_bert_weights(cfg, seed=7, w_std=0.02), cfg
    sw = synth.make_sdnet_weights(opt, seed=7)
    tr = SDNetTrainer(opt, device="cuda:0")
    tr.setup_model({"glove_embedding": T(sw["glove_embed.weight"]), "fast_embedding": T(sw["fast_embed.weight"])})
    n_bert = sum(p.numel() for n, p in tr.network.named_parameters() if n.startswith("Bert.") and p.requires_grad)
    assert n_bert > 80e6
    w0 = dict(tr.network.named_parameters())["Bert.bert_model.encoder.layer.3.output.dense.weight"].detach().clone()
    batch = tr.ToCUDA(synth.synthetic_batch(opt, 3, seed=31, n_q=10, n_ocr=14, n_od=5, bert_vocab=2000, ragged=True))
    raw = [tr.update(batch, i) for i in range(6)]
    assert type(raw[0]).__name__ == "_PendingLoss"                        # trained encoder: the readback trails by one step
    assert tr.train_loss.count == 5                                       # ... so five steps have been resolved by now
    losses = [float(v) for v in raw]
    assert tr.train_loss.count == 6 and tr.train_loss.val == losses[-1]
    assert all(np.isfinite(losses)), losses
    assert min(losses[3:]) < losses[0], losses                            # the same batch six times: the loss comes down
    if train_gemm == "16":
        assert type(tr.network.Bert.bert_model).__name__ == "BertModelTrainable16"
    w1 = dict(tr.network.named_parameters())["Bert.bert_model.encoder.layer.3.output.dense.weight"].detach()
    assert not torch.equal(w0, w1)                                       # the encoder moved
    a = tr.predict(batch)[0]
    b = tr.predict(batch)[0]
    assert a == b                                                        # evaluation: dropout off, deterministic
    tr.network.train()
    s1, _ = tr.network(batch[0], batch[1], batch[2])
    s2, _ = tr.network(batch[0], batch[1], batch[2])
    assert not torch.equal(s1, s2)                                       # training: BERT dropout on
    path = "/tmp/ruart_ckpt_unlocked.pt"
    tr.save_for_predict(path)
    assert not any(k.startswith("Bert") for k in torch.load(path, map_location="cpu")["state_dict"]["network"])


def test_deferred_readback_changes_nothing_but_the_moment_of_the_asserts():
    """Trained encoder: ``update`` reads the loss and the NaN flag of step t back at the end of step t+1 (trainer._readback_later).
    Same losses, same running mean as with the per-step sync; a NaN still stops the run - one step late, and before predict /
    save can look at the weights."""
    from ruart_amd.trainer import SDNetTrainer
    cfg = synth.bert_config(vocab_size=2000, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    bert_state = synth.make_bert_weights(cfg, seed=7, w_std=0.02)

    def make(defer):
        opt = default_opt(vocab_size=600, cuda=True, DROPOUT=0.0, dropout_emb=0.0, lr=2e-4, bert_train_gemm="16")
        opt.pop("LOCK_BERT")
        opt["bert_state"], opt["bert_config"] = bert_state, cfg
        if defer is not None:
            opt["ruart_defer_readback"] = defer
        sw = synth.make_sdnet_weights(opt, seed=7)
        tr = SDNetTrainer(opt, device="cuda:0")
        tr.setup_model({"glove_embedding": T(sw["glove_embed.weight"]), "fast_embedding": T(sw["fast_embed.weight"])})
        return tr, tr.ToCUDA(synth.synthetic_batch(opt, 3, seed=31, n_q=10, n_ocr=14, n_od=5, bert_vocab=2000, ragged=True))

    tr_a, batch = make(False)
    la = [tr_a.update(batch, i) for i in range(4)]
    assert all(isinstance(v, float) for v in la)
    tr_b, batch_b = make(None)                                            # default with a trained encoder: deferred
    lb = [tr_b.update(batch_b, i) for i in range(4)]
    assert tr_b.train_loss.count == 3
    assert tr_b.flush_readback() == la[3] and tr_b.flush_readback() is None
    assert [float(v) for v in lb] == la and tr_b.train_loss.avg == tr_a.train_loss.avg and tr_b.train_loss.count == 4
    assert "%.4f" % lb[0] == "%.4f" % la[0]
    # a NaN: the step that produces it returns; the next update (or predict, or close) raises
    with torch.no_grad():
        tr_b.network.get_answer.attn.linear.weight[0, 0] = float("nan")
    tr_b.update(batch_b, 4)
    with pytest.raises(AssertionError):
        tr_b.predict(batch_b)
    tr_a.close()


@pytest.mark.parametrize("precision,tol_l,tol_d", [("fp32", 2e-4, 5e-3), ("fp16", 2e-2, 8e-2)])
def test_three_optimizer_steps_vs_reference_update(golden_dir, precision, tol_l, tol_d):
    """``SDNetTrainer.update`` end to end against three calls of the reference's own update() (Models/SDNetTrainer.py:330-376) on
    the same batch: forward, BCE_D1, backward, global-norm clip 10, Adamax, re-pinning of the embedding rows >= tune_partial -
    here through the fused clip + Adamax kernels.  Compared: the three losses, how far every tensor moved (norm of the update),
    and the values of small tensors.  (Adamax's first step is lr * sign(g): entries whose gradient is rounding noise may go
    either way, so element-wise checks allow 2 lr.)"""
    from ruart_amd.trainer import SDNetTrainer
    z = np.load(os.path.join(golden_dir, "trainer_update.npz"))
    opt = default_opt(vocab_size=int(z["vocab_size"]), cuda=True, DROPOUT=0.0, dropout_emb=0.0, bert_precision=precision)
    cfg = synth.bert_config(vocab_size=2000, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    opt["bert_state"], opt["bert_config"] = synth.make_bert_weights(cfg, seed=int(z["seed"])), cfg
    sw = synth.make_sdnet_weights(opt, seed=int(z["seed"]))
    tr = SDNetTrainer(opt, device="cuda:0")
    tr.setup_model({"glove_embedding": T(sw["glove_embed.weight"]), "fast_embedding": T(sw["fast_embed.weight"])})
    tr.network.load_state_dict({k: T(v) for k, v in sw.items()})
    before = {n: p.detach().clone() for n, p in tr.network.named_parameters()}
    batch = tr.ToCUDA(synth.synthetic_batch(opt, int(z["B"]), seed=int(z["batch_seed"]), n_q=12, n_ocr=20, n_od=6, bert_vocab=2000,
                                            ragged=True))
    losses = [tr.update(batch, i) for i in range(3)]
    assert np.abs(np.array(losses) - z["losses"]).max() < tol_l * 70, (losses, z["losses"].tolist())
    prm = dict(tr.network.named_parameters())
    lr = float(z["lr"])
    worst = (0.0, "")
    for name, dref in zip(z["names"].tolist(), z["delta_norms"].tolist()):
        d = float((prm[name].detach() - before[name]).double().norm())
        if dref < 1e-12:
            assert d < 1e-12, name                          # never-updated tensors (dead GRU, fixed scalars) stay put
            continue
        if name == "ques_merger.linear.bias":
            # a constant added to every score of a softmax: its true gradient is zero; the reference's is rounding noise
            # (~1e-9) that Adamax turns into a random walk, the fused kernel does not materialise it - no output depends on it
            assert d == 0.0
            continue
        worst = max(worst, (abs(d - dref) / dref, name))
    assert worst[0] < tol_d, worst
    for k in ("fast_embed.weight", "glove_embed.weight"):          # which rows of the word tables moved, and how far
        ref = z["rowdelta:" + k]
        got = (prm[k].detach() - before[k]).double().norm(dim=1).cpu().numpy()
        assert np.array_equal(got > 0, ref > 0), k             # same rows (none >= tune_partial; the question's padding row 0 too)
        assert np.abs(got - ref).max() < (2e-4 if precision == "fp32" else 4e-3), (k, float(np.abs(got - ref).max()))
    for k in z.files:
        if not k.startswith("after:"):
            continue
        name = k[6:].split("[")[0]
        got = prm[name].detach().cpu().numpy()
        if "[:40,:16]" in k:
            got = got[:40, :16]
        elif "[1000:1004,:16]" in k:
            got = got[1000:1004, :16]                       # rows >= tune_partial: re-pinned to their initial values
            assert np.array_equal(got, sw["fast_embed.weight"][1000:1004, :16])
        frac_off = float((np.abs(got - z[k]) > 2.05 * lr).mean())
        assert frac_off <= (0.0 if precision == "fp32" else 0.02), (k, frac_off, float(np.abs(got - z[k]).max()))
    print("update parity %s: losses %s, worst update-norm rel err %.2e (%s)" % (precision, losses, worst[0], worst[1]))


@pytest.mark.parametrize("precision", ["fp16c", "fp16", "x3"])
def test_last_layer_rows_and_dedup_are_result_neutral(golden, precision):
    """Frozen encoder: (1) computing the last layer only on the rows word spans pool (the [CLS] / [SEP] rows are never read,
    Models/Bert/Bert.py:153-165) changes no bit of the scores or of the alpha / gamma gradients; (2) encoding identical item
    sequences once leaves every probability within fp32 rounding (the sequences land at other offsets of their 64-token attention
    windows, which reorders exact-zero terms inside the MFMA sums)."""
    import ruart_amd.layers as L
    z = golden
    outs = {}
    for tag, extra in (("all", dict(bert_dedup=False, bert_last_rows=False)), ("last", dict(bert_dedup=False)), ("both", {})):
        net, opt = build(z, precision, **extra)
        q, ocr, od, gt, _ = synth.synthetic_batch(opt, int(z["B"]), seed=int(z["batch_seed"]), n_q=30, n_ocr=100, n_od=30,
                                                  bert_vocab=2000, ragged=True)
        for items in (ocr, od):                      # real data spells every sentinel item the same way: make it so here
            sent = np.cumsum(items["num_cnt"]) - 1
            for r in sent[1:]:
                items["bert"][r] = items["bert"][sent[0]]
                items["bert_mask"][r] = items["bert_mask"][sent[0]]
                items["bert_offsets"][r] = items["bert_offsets"][sent[0]]
        L.set_dropout_prob(0.0)
        net.train()
        net.drop_emb = False
        scores, _ = net(q, ocr, od)
        gt = gt.to(scores.device)
        (torch.nn.functional.binary_cross_entropy_with_logits(scores, gt) * gt.size(1)).backward()
        bi = q["_ruart_index"]
        outs[tag] = (scores.detach().clone(), net.alphaBERT.grad.clone(), net.gammaBERT.grad.clone(), bi.packed.T, bi._n_last)
        net.Bert.close()
    assert outs["all"][4] == 0 and 0 < outs["last"][4] < outs["last"][3] and outs["both"][3] < outs["last"][3]
    for k in range(3):
        assert torch.equal(outs["all"][k], outs["last"][k]), k
    assert float((outs["both"][0] - outs["all"][0]).abs().max()) < 2e-5
    # (entries of the alpha gradient that cancel to ~1e-5 of the largest one move by a few 1e-7: absolute bound at 1e-4 of the scale)
    scale = float(outs["all"][1].abs().max())
    assert torch.allclose(outs["both"][1], outs["all"][1], rtol=1e-3, atol=max(1e-5 if precision == "fp16" else 1e-7, 1e-4 * scale))


def test_deferred_weight_gradients_are_bitwise_the_same(golden):
    """opt['ruart_defer_dw'] (opt-in; measured slower in the step, see SDNet.forward): the trunk's weight gradients are recorded during backward and computed by grouped launches at
    its end (ops._flush_weight_grads) - same tiles, same K slices, same summation order as the per-site launches: every gradient
    must be bit-identical, modules used twice (deep attention, the shared RNNs: accumulating second wave) included."""
    import ruart_amd.layers as L
    z = golden
    grads = {}
    for tag, extra in (("site", dict(ruart_defer_dw=False)), ("grouped", dict(ruart_defer_dw=True))):
        net, opt = build(z, "fp16c", **extra)
        q, ocr, od, gt, _ = synth.synthetic_batch(opt, int(z["B"]), seed=int(z["batch_seed"]), n_q=30, n_ocr=100, n_od=30,
                                                  bert_vocab=2000, ragged=True)
        L.set_dropout_prob(0.3)
        torch.manual_seed(11)
        net.train()
        net.drop_emb = True
        scores, _ = net(q, ocr, od)
        gt = gt.to(scores.device)
        (torch.nn.functional.binary_cross_entropy_with_logits(scores, gt) * gt.size(1)).backward()
        torch.cuda.synchronize()
        grads[tag] = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
        net.Bert.close()
    L.set_dropout_prob(0.0)
    from ruart_amd import ops
    assert not ops._deferred
    assert grads["site"].keys() == grads["grouped"].keys()
    bad = [n for n in grads["site"] if not torch.equal(grads["site"][n], grads["grouped"][n])]
    assert not bad, bad[:5]
