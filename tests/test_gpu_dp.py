"""The data-parallel code path on the real device.
  * one RCCL rank: process-group init on the device, the gradient hooks, bucketed asynchronous all-reduce on HIP streams and the
    averaged update; with world size 1 the step equals the plain step (every kernel is deterministic);
  * two ranks with the real SDNet on different shards (tests/_dp_two_rank_worker.py): over RCCL with one GPU per rank whenever
    the box has two GPUs (skipped otherwise), and - the rehearsal a one-GPU box can run - over gloo with both ranks on the same
    GPU: a real peer, the real three-stream hook ordering, gradients bit-equal to the hand average, replicas bit-identical.
(GradSync's rules are also covered on CPU with gloo in test_dp_gloo.py; N = 2..8 scaling runs are the driver's.)"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from ruart_amd import synth                       # noqa: E402
from ruart_amd.arguments import default_opt        # noqa: E402


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _make(opt, cfg, sw, dp):
    from ruart_amd.trainer import SDNetTrainer
    import torch.distributed as dist
    tr = SDNetTrainer(opt, device="cuda:0", process_group=dist.group.WORLD if dp else None)
    if not dp:
        tr.process_group = None
    tr.setup_model({"glove_embedding": T(sw["glove_embed.weight"]), "fast_embedding": T(sw["fast_embed.weight"])})
    tr.network.load_state_dict({k: T(v) for k, v in sw.items()})
    return tr


def test_single_rank_nccl_step_equals_plain_step():
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        opt = default_opt(vocab_size=1500, cuda=True, DROPOUT=0.0, dropout_emb=0.0)
        cfg = synth.bert_config(vocab_size=2000)
        opt["bert_state"], opt["bert_config"] = synth.make_bert_weights(cfg, seed=1033), cfg
        sw = synth.make_sdnet_weights(opt, seed=1033)
        batch = synth.synthetic_batch(opt, 3, seed=5, n_q=10, n_ocr=24, n_od=7, bert_vocab=2000, ragged=True)
        plain = _make(opt, cfg, sw, dp=False)
        assert plain.grad_sync is None
        dp = _make(opt, cfg, sw, dp=True)
        assert dp.grad_sync is not None and len(dp.grad_sync.buckets) >= 1
        names = [n for b in dp.grad_sync.buckets for (n, _, _) in b]
        assert not any(n.startswith("get_answer.rnn") for n in names)

        def grads(tr):
            b = tr.ToCUDA(batch)
            tr.network.train()
            tr.network.drop_emb = True
            scores, _ = tr.network(b[0], b[1], b[2])
            loss = tr.loss_func(scores, b[3])
            tr.optimizer.zero_grad(set_to_none=True)
            loss.backward()
            if tr.grad_sync is not None:
                tr.grad_sync.average_gradients()
            torch.cuda.synchronize()
            return loss.item(), {n: p.grad.clone() for n, p in tr.network.named_parameters() if p.grad is not None}

        lp, gp = grads(plain)
        ld, gd = grads(dp)
        assert lp == ld                                     # the step is deterministic: same loss, same gradients
        for n in gp:
            assert torch.equal(gp[n], gd[n]), n              # all-reduce over one rank then * 1/1 is the identity
        for n in set(gd) - set(gp):                          # a used parameter without a gradient is exchanged as zeros
            assert n == "ques_merger.linear.bias" and float(gd[n].abs().max()) == 0.0, n
        l0 = [plain.update(plain.ToCUDA(batch), i) for i in range(3)]
        l1 = [dp.update(dp.ToCUDA(batch), i) for i in range(3)]
        # default DP mode: the embedding tables are exchanged whole, the clip norm is the exact norm of the averaged gradient
        assert dp.grad_sync.mode == "full"
        assert np.allclose(l0, l1, rtol=2e-6, atol=0), (l0, l1)
        assert abs(float(plain.optimizer.norm_coef[0]) - float(dp.optimizer.norm_coef[0])) < 1e-5 * float(plain.optimizer.norm_coef[0])
        t = torch.ones(4, device="cuda:0")
        dist.all_reduce(t)
        dist.barrier()
        assert float(t.sum()) == 4.0
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_two_ranks(backend, devices, mode="post"):
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dp_two_rank_worker.py")
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, backend, str(devices[r]), mode], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=600)
            outs.append(out)
    finally:
        for p in procs:                      # exact PIDs of the two children only
            if p.poll() is None:
                p.kill()
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, out[-4000:])
        assert "rank %d ok" % r in out, out[-2000:]


@pytest.mark.parametrize("mode", ["post", "overlap"])
def test_two_rank_gloo_step_on_one_gpu(mode):
    """Two processes, both on cuda:0, collectives on gloo (RCCL refuses two ranks on one device).  post: the default exchange after
    backward(); overlap: opt['dp_overlap_backward'], hooks + asynchronous collectives from a communication stream."""
    _run_two_ranks("gloo", (0, 0), mode)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (one RCCL rank per GPU)")
def test_two_rank_rccl_step():
    """The N > 1 path over RCCL / xGMI as the driver's scaling runs use it."""
    _run_two_ranks("nccl", (0, 1))
    _run_two_ranks("nccl", (0, 1), "overlap")
