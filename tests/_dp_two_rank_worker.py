"""One rank of the two-rank data-parallel check (started by tests/test_gpu_dp.py, one process per rank).

    python tests/_dp_two_rank_worker.py RANK WORLD PORT BACKEND DEVICE_INDEX

What it checks, with the REAL SDNet (three-stream trunk, frozen encoder one step ahead) and a DIFFERENT shard per rank:
  1. the gradients GradSync leaves in ``p.grad`` equal the hand-averaged gradients of the ranks' shards, bit for bit
     ((g_0 + g_1) * 0.5 in fp32 is what a two-rank sum followed by the 1/world scaling computes, and every kernel is deterministic);
  2. after three optimizer steps (clip by the exact norm of the averaged gradient, fused Adamax, re-pinning) the replicas hold
     bit-identical parameters.
BACKEND nccl = RCCL, one GPU per rank (needs >= 2 GPUs).  BACKEND gloo with both ranks on the same device is the rehearsal a
one-GPU box can run: real peer, real hooks on three streams, the collective itself on gloo."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def main():
    rank, world, port, backend, dev_index = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
    from ruart_amd import dp, synth
    from ruart_amd.arguments import default_opt
    from ruart_amd.trainer import SDNetTrainer
    device = torch.device("cuda", dev_index)
    torch.cuda.set_device(device)
    if backend == "nccl":
        dp.init_process_group(device, "nccl", rank=rank, world_size=world)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)

    opt = default_opt(vocab_size=1500, cuda=True, DROPOUT=0.0, dropout_emb=0.0)
    opt["dp_overlap_backward"] = len(sys.argv) > 6 and sys.argv[6] == "overlap"     # default: exchange after backward (dp.py)
    cfg = synth.bert_config(vocab_size=2000)
    opt["bert_state"], opt["bert_config"] = synth.make_bert_weights(cfg, seed=1033), cfg
    sw = synth.make_sdnet_weights(opt, seed=1033)
    emb = {"glove_embedding": T(sw["glove_embed.weight"]), "fast_embedding": T(sw["fast_embed.weight"])}

    def make(data_parallel):
        o = dict(opt)
        o["ruart_dp"] = data_parallel
        tr = SDNetTrainer(o, device=device, process_group=dist.group.WORLD if data_parallel else None)
        tr.setup_model(emb)
        tr.network.load_state_dict({k: T(v) for k, v in sw.items()})
        return tr

    # shard r: different sizes on purpose (ragged), ids overlap across ranks (shared pinned embedding rows)
    shards = [synth.synthetic_batch(opt, 3 + r, seed=5 + r, n_q=10, n_ocr=24, n_od=7, bert_vocab=2000, ragged=True) for r in range(world)]

    def grads(tr, batch):
        b = tr.ToCUDA(batch)
        tr.network.train()
        tr.network.drop_emb = True
        scores, _ = tr.network(b[0], b[1], b[2])
        loss = tr.loss_func(scores, b[3])
        tr.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        if tr.grad_sync is not None:
            tr.grad_sync.average_gradients()
        torch.cuda.synchronize(device)
        return {n: p.grad.detach().clone() for n, p in tr.network.named_parameters() if p.grad is not None}

    plain = make(False)
    assert plain.grad_sync is None
    local = [grads(plain, shards[r]) for r in range(world)]          # every rank computes every shard's gradients itself
    ddp = make(True)
    assert ddp.grad_sync is not None and ddp.grad_sync.world == world and ddp.grad_sync.mode == "full"
    synced = grads(ddp, shards[rank])
    bad = []
    for n, g in synced.items():
        parts = [l[n] for l in local if n in l]
        if not parts:                                                   # exchanged as zeros (e.g. ques_merger.linear.bias)
            ok = float(g.abs().max()) == 0.0
        else:
            acc = parts[0].clone()
            for p_ in parts[1:]:
                acc += p_
            ok = torch.equal(g, acc * (1.0 / world))
        if not ok:
            bad.append(n)
    assert not bad, "rank %d: averaged gradients differ from the hand average: %s" % (rank, bad[:5])

    # three optimizer steps on different shards, next batch's encoder pass running ahead: replicas must stay bit-identical
    mine = [ddp.ToCUDA(synth.synthetic_batch(opt, 3 + rank, seed=50 + 10 * i + rank, n_q=10, n_ocr=24, n_od=7, bert_vocab=2000, ragged=True))
            for i in range(4)]
    losses = [ddp.update(mine[i], i, next_batch=mine[i + 1]) for i in range(3)]
    assert all(np.isfinite(losses))
    torch.cuda.synchronize(device)
    diff = []
    for n, p in ddp.network.named_parameters():
        t = p.detach().clone()
        dist.broadcast(t, src=0)
        if not torch.equal(t, p.detach()):
            diff.append(n)
    assert not diff, "rank %d: replicas diverged in %s" % (rank, diff[:5])
    coef = ddp.optimizer.norm_coef.detach().clone()
    c0 = coef.clone()
    dist.broadcast(c0, src=0)
    assert torch.equal(c0, coef), "ranks clipped by different coefficients"
    ddp.close()
    plain.close()
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok: %d gradients bit-equal to the hand average, %d parameters identical after 3 steps, losses %s"
          % (rank, len(synced), len(list(ddp.network.parameters())), ["%.4f" % l for l in losses]), flush=True)


if __name__ == "__main__":
    main()
