"""The N>1 path without GPUs: two gloo ranks on CPU exercise dp.GradSync (bucketing, asynchronous all-reduce from
gradient hooks, exclusion of the dead GRU, pinned embedding rows) and the rank-sharded sampler."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


class Toy(nn.Module):
    """Parameter names chosen to hit every GradSync rule."""

    def __init__(self):
        super().__init__()
        self.fast_embed = nn.Embedding(12, 4)
        self.glove_embed = nn.Embedding(12, 4)
        self.body = nn.Linear(4, 3)
        self.get_answer = nn.Module()
        self.get_answer.rnn = nn.GRUCell(3, 3)          # never used in forward, like the reference's dead GRU step
        self.frozen = nn.Parameter(torch.ones(2), requires_grad=False)

    def forward(self, ids):
        return self.body(self.fast_embed(ids) + self.glove_embed(ids)).pow(2).sum()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, skip_pinned, overlap, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ruart_amd.dp import GradSync
    torch.manual_seed(100 + rank)                        # different init per rank: broadcast must fix it
    net = Toy()
    opt = {"TUNE_PARTIAL": True, "tune_partial": 5}
    if skip_pinned:
        opt["dp_skip_pinned_rows"] = True
    gs = GradSync(net, opt, bucket_bytes=64, overlap=overlap)   # tiny buckets => several of them
    gs.broadcast_parameters()
    ids = torch.tensor([[1, 2, 7], [3, 9, 11]]) if rank == 0 else torch.tensor([[0, 4, 8], [10, 2, 6]])
    sgd = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=0.1)
    res = {}
    for step in range(2):                                # two steps: hook bookkeeping must reset
        sgd.zero_grad(set_to_none=True)
        loss = net(ids)
        loss.backward()
        local = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
        gs.average_gradients()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 0.5)
        res["local%d" % step] = local
        res["avg%d" % step] = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
        sgd.step()
    res["params"] = {n: p.detach().clone() for n, p in net.named_parameters()}
    res["n_buckets"] = len(gs.buckets)
    res["names"] = [n for b in gs.buckets for (n, _, _) in b]
    torch.save(res, out % rank)
    dist.destroy_process_group()


@pytest.mark.parametrize("skip_pinned,overlap", [(False, False), (True, False), (False, True), (True, True)])
def test_gradsync_two_ranks(tmp_path, skip_pinned, overlap):
    """overlap=False: the default exchange after backward(); True: opt['dp_overlap_backward'], hooks + asynchronous collectives."""
    world = 2
    out = str(tmp_path / "r%d.pt")
    mp.spawn(_worker, args=(world, _free_port(), skip_pinned, overlap, out), nprocs=world, join=True)
    r = [torch.load(out % i) for i in range(world)]
    assert r[0]["n_buckets"] > 1
    assert not any(n.startswith("get_answer.rnn") for n in r[0]["names"]) and "frozen" not in r[0]["names"]
    # recompute the expected average from the ranks' local gradients of step 0 (identical weights after broadcast)
    for name, g0 in r[0]["local0"].items():
        want = (g0 + r[1]["local0"][name]) / 2
        if skip_pinned and name in ("fast_embed.weight", "glove_embed.weight"):
            want[5:] = 0
        # averaged then clipped by the same coefficient on both ranks
        a0, a1 = r[0]["avg0"][name], r[1]["avg0"][name]
        assert torch.equal(a0, a1), name
        scale = (a0.norm() / want.norm()).item() if want.norm() > 0 else 1.0
        assert torch.allclose(a0, want * scale, atol=1e-6), name
    # replicas stay bit-identical after two optimizer steps
    for name in r[0]["params"]:
        assert torch.equal(r[0]["params"][name], r[1]["params"][name]), name


def _shared_rows_worker(rank, world, port, scalar, overlap, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ruart_amd.dp import GradSync
    torch.manual_seed(3)
    net = Toy()
    gs = GradSync(net, {"TUNE_PARTIAL": True, "tune_partial": 5, "dp_overlap_backward": overlap}, bucket_bytes=64, pinned_scalar=scalar)
    gs.broadcast_parameters()
    # both ranks look up the pinned rows 7 and 9 (frequent out-of-head words): their gradients are strongly correlated
    ids = torch.tensor([[1, 2, 7], [3, 9, 7]]) if rank == 0 else torch.tensor([[0, 7, 7], [9, 9, 6]])
    net(ids).backward()
    local = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    gs.average_gradients()
    res = {"mode": gs.mode, "local": local, "avg": {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None},
           "pinned_sq": None if gs.pinned_sq is None else float(gs.pinned_sq)}
    torch.save(res, out % rank)
    dist.destroy_process_group()


@pytest.mark.parametrize("scalar,overlap", [(False, False), (True, False), (True, True)])
def test_clip_norm_when_ranks_share_pinned_rows(tmp_path, scalar, overlap):
    """Models/SDNetTrainer.py:366 clips by the norm of THE gradient - under data parallelism the averaged one.  Default mode: the
    tables are exchanged whole and the norm every rank computes is exactly that.  opt['dp_pinned_scalar']: the pinned rows enter as
    sum_r |g_r|^2 / world^2, which drops the cross terms between ranks - equal to the exact norm only for disjoint rows; here (shared
    rows, positively correlated gradients) it is an under-estimate, bounded by sqrt(world)."""
    world, tp = 2, 5
    out = str(tmp_path / "s%d.pt")
    mp.spawn(_shared_rows_worker, args=(world, _free_port(), scalar, overlap, out), nprocs=world, join=True)
    r = [torch.load(out % i) for i in range(world)]
    names = list(r[0]["local"])
    exact_sq = sum(float(((r[0]["local"][n] + r[1]["local"][n]) / 2).double().pow(2).sum()) for n in names)
    if not scalar:
        assert r[0]["mode"] == "full" and r[0]["pinned_sq"] is None
        got_sq = sum(float(r[0]["avg"][n].double().pow(2).sum()) for n in names)
        assert abs(got_sq - exact_sq) < 1e-6 * exact_sq
        for n in names:
            assert torch.equal(r[0]["avg"][n], r[1]["avg"][n]), n
        return
    assert r[0]["mode"] == "scalar" and r[0]["pinned_sq"] == r[1]["pinned_sq"]
    tables = ("fast_embed.weight", "glove_embed.weight")
    trained_sq = sum(float((r[0]["avg"][n][:tp] if n in tables else r[0]["avg"][n]).double().pow(2).sum()) for n in names)
    got_sq = trained_sq + r[0]["pinned_sq"]
    formula = sum(float(r[k]["local"][n][tp:].double().pow(2).sum()) for k in range(world) for n in tables) / world ** 2
    assert abs(r[0]["pinned_sq"] - formula) < 1e-5 * formula
    assert got_sq < exact_sq * 0.999                  # the cross terms 2 <g_0, g_1> / world^2 of the shared rows are missing
    assert got_sq * world > exact_sq * 0.999          # ... and the estimate is never more than sqrt(world) too small


# ---- the REAL SDNet parameter list (no forward: gradients filled by hand) -------------------------------------------------------
class _StubBert(nn.Module):
    """Stands in for ruart_amd.bert.Bert: the frozen encoder holds no nn.Parameter, so the DP logic never sees it."""

    def __init__(self, opt, device=None):
        super().__init__()

    def lock(self):
        pass


def _real_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    from ruart_amd import sdnet as sdnet_mod, synth
    from ruart_amd.arguments import default_opt
    from ruart_amd.dp import GradSync
    sdnet_mod.Bert = _StubBert
    opt = default_opt(vocab_size=1500, device="cpu")
    sw = synth.make_sdnet_weights(opt, seed=7)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))      # noqa: E731
    net = sdnet_mod.SDNet(opt, {"glove_embedding": T(sw["glove_embed.weight"]), "fast_embedding": T(sw["fast_embed.weight"])})
    gs = GradSync(net, opt, pinned_scalar=True, bucket_bytes=16 << 20)   # the one-scalar form; several buckets
    g = torch.Generator().manual_seed(1000 + rank)
    local = {}
    for name, p in net.named_parameters():
        if p.requires_grad and not name.startswith("get_answer.rnn."):
            p.grad = torch.randn(p.shape, generator=g)
            local[name] = p.grad.clone()
    gs.average_gradients()                                        # no hook fired: every bucket is launched here, in order
    res = {"names": [n for b in gs.buckets for (n, _, _) in b], "rows": [r for b in gs.buckets for (_, _, r) in b],
           "payload": gs.payload_bytes, "mode": gs.mode, "n_buckets": len(gs.buckets), "local": local,
           "avg": {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None},
           "pinned_sq": float(gs.pinned_sq), "tp": int(opt["tune_partial"]),
           "numel": {n: p.numel() for n, p in net.named_parameters() if p.requires_grad}}
    torch.save(res, out % rank)
    dist.destroy_process_group()


def test_gradsync_on_the_real_sdnet_parameter_list(tmp_path):
    """Bucket layout of the actual model: identical order on both ranks, the dead GRU left out, the re-pinned embedding rows
    represented by one scalar, payload = the ~37 MB class of SURVEY section 8e (scaled to this vocabulary)."""
    world = 2
    out = str(tmp_path / "real%d.pt")
    mp.spawn(_real_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r = [torch.load(out % i) for i in range(world)]
    assert r[0]["names"] == r[1]["names"] and r[0]["rows"] == r[1]["rows"] and r[0]["n_buckets"] == r[1]["n_buckets"] >= 2
    names = r[0]["names"]
    assert r[0]["mode"] == "scalar" and not any(n.startswith("get_answer.rnn.") for n in names)
    assert "alphaBERT" in names and "multi2one.rnns.0.weight_ih_l0" in names and "get_answer.attn.linear.weight" in names
    tp = r[0]["tp"]
    want = 0
    for n in names:
        want += tp * 300 if n in ("fast_embed.weight", "glove_embed.weight") else r[0]["numel"][n]
    assert r[0]["payload"] == 4 * want + 4                        # fp32 elements + the scalar
    assert set(names) == {n for n in r[0]["numel"] if not n.startswith("get_answer.rnn.")}
    sq = 0.0
    for n in names:
        l0, l1 = r[0]["local"][n], r[1]["local"][n]
        a0, a1 = r[0]["avg"][n], r[1]["avg"][n]
        if n in ("fast_embed.weight", "glove_embed.weight"):
            assert torch.allclose(a0[:tp], (l0[:tp] + l1[:tp]) / 2, atol=1e-6) and torch.equal(a0[:tp], a1[:tp]), n
            assert torch.equal(a0[tp:], l0[tp:]) and torch.equal(a1[tp:], l1[tp:]), n      # never exchanged: still local
            sq += float(l0[tp:].double().pow(2).sum() + l1[tp:].double().pow(2).sum())
        else:
            assert torch.allclose(a0, (l0 + l1) / 2, atol=1e-6) and torch.equal(a0, a1), n
    assert abs(r[0]["pinned_sq"] - sq / 4) < 1e-4 * sq / 4 and r[0]["pinned_sq"] == r[1]["pinned_sq"]
