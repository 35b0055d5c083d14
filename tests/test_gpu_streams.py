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
