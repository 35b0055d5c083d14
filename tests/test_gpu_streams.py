"""Cross-stream safety of the product's forward at the bench size: the three-stream trunk must give the SAME BITS as the one-stream
trunk (same kernels, same per-op order) - also in a process whose device is busy with history: a second trainer built after a first
one has trained.  Round 5: a pooling kernel that read its row statistics back with v_readlane passed every parity test and differed
from run to run by up to 1e-3 in exactly this situation (tools/r05_race_probe.py)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def test_three_stream_forward_is_bit_identical_to_one_stream_after_training():
    import bench
    import ruart_amd.layers as L
    from ruart_amd import synth
    from ruart_amd.arguments import default_opt
    dev = torch.device("cuda:0")
    opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
    tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
    b = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100, n_od=36)) for i in range(2)]
    tr.opt["ruart_defer_readback"] = True
    with tr.step_stream():
        for i in range(8):
            tr.update(b[i % 2], i, next_batch=None)            # encoder inline: the situation of bench.py --no-prefetch
        tr.flush_readback()
    torch.cuda.synchronize()
    tr2, _ = bench.build_trainer(dict(opt, ruart_dp=False), synth.bert_config(), dev)
    try:
        b0 = tr2.ToCUDA(synth.synthetic_batch(opt, 64, seed=7, n_q=30, n_ocr=100, n_od=36))
        L.set_dropout_prob(0.0)
        net = tr2.network

        def fwd(streams):
            net.opt["ruart_streams"] = streams
            net.train()
            net.drop_emb = False

            def f():
                with torch.no_grad():
                    return net(b0[0], b0[1], b0[2])[0]
            s = tr2.on_step_stream(f)
            torch.cuda.synchronize()
            return s.float().cpu()

        one = fwd(False)
        for _ in range(3):
            three = fwd(True)
            assert torch.equal(three, one), "three-stream forward differs from the one-stream forward by %.3e" % float((three - one).abs().max())
            assert torch.equal(fwd(False), one)
    finally:
        L.set_dropout_prob(0.0 if "DROPOUT" not in opt else float(opt["DROPOUT"]))
        tr2.close()
        tr.close()


def _train_arm(pipelined, steps):
    """``steps`` update() calls at the bench size with the one-stream trunk (so the gradient fan-in order is the same in both arms).
    pipelined: the timed schedule of bench.py / train() - the frozen encoder pass of batch t+1 on the CU-masked run-ahead stream beside
    step t, the loop on the step stream, losses read back one step late; otherwise the encoder inline on the step's own stream and the
    reference's per-step readback.  Returns (losses, {name: parameter bits}, {name: Adamax state bits})."""
    import bench
    from ruart_amd import synth
    from ruart_amd.arguments import default_opt
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64, ruart_streams=False)
    opt["ruart_defer_readback"] = bool(pipelined)
    tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
    try:
        torch.manual_seed(4321)                                  # the dropout masks of both arms come from one generator state
        bs = [tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7 + i, n_q=30, n_ocr=100 - 9 * i, n_od=36 - 5 * i)) for i in range(3)]
        losses = []
        import contextlib
        with (tr.step_stream() if pipelined else contextlib.nullcontext()):
            for i in range(steps):
                losses.append(tr.update(bs[i % 3], i, next_batch=bs[(i + 1) % 3] if pipelined else None))
            tr.flush_readback()
        torch.cuda.synchronize()
        masked = sorted(getattr(tr.network.Bert, "_pf_streams", {}))              # CU counts of the run-ahead streams this arm created
        params = {n: p.detach().cpu().clone() for n, p in tr.network.named_parameters()}
        state = {}
        for n, p in tr.network.named_parameters():             # (FusedAdamax keys its moments by id(parameter), torch's Adamax by the tensor)
            st = tr.optimizer.state
            ent = st.get(id(p)) if id(p) in st else (st[p] if any(k is p for k in st) else {})
            for k, v in (ent or {}).items():
                if torch.is_tensor(v):
                    state["%s.%s" % (n, k)] = v.detach().cpu().clone()
        return [float(x) for x in losses], params, state, masked
    finally:
        tr.close()


def test_pipelined_training_is_bitwise_equal_to_inline_training_at_bench_size():
    """VERDICT r05 item 2b.  B = 64, 100 OCR items, 36 objects (three batches of different sizes, so the two buffer sets and the packed
    stream change shape from step to step), seven optimizer steps: the masked run-ahead encoder stream + deferred readback against
    the encoder inline + per-step readback, the trunk on ONE stream in both arms.  Same kernels on the same data in the same per-stream
    order: every loss, every parameter and every optimizer moment must be the same bits.  A buffer set handed over too early, a pickup
    event recorded too early or a cross-stream dependency missing in the timed schedule shows here as a difference."""
    steps = 7
    la, pa, sa, masked = _train_arm(True, steps)
    lb, pb, sb, _ = _train_arm(False, steps)
    assert all(x == x for x in la + lb)
    assert la == lb, "losses differ: %s vs %s" % (la, lb)
    bad = [n for n in pa if not torch.equal(pa[n], pb[n])]
    assert not bad, "%d of %d parameters differ, e.g. %s (max |d| %.3e)" % (len(bad), len(pa), bad[0], float((pa[bad[0]].float() - pb[bad[0]].float()).abs().max()))
    assert set(sa) == set(sb) and len(sa) > 0
    bad = [n for n in sa if not torch.equal(sa[n], sb[n])]
    assert not bad, "%d of %d optimizer state tensors differ, e.g. %s" % (len(bad), len(sa), bad[0])
    print("pipelined == inline over %d steps: %d parameters, %d optimizer tensors bit-equal; run-ahead stream CUs: %s; losses %s"
          % (steps, len(pa), len(sa), masked, " ".join("%.6f" % x for x in la)))
