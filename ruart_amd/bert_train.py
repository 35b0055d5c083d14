"""Trainable BERT encoder for confs without ``LOCK_BERT`` (Models/SDNet.py:88-94: the encoder's parameters then simply join the
trainer's optimizer, SDNetTrainer.py:305-311).

The frozen path (bert.py) is one C call over 16-bit copies of the weights; this one keeps the parameters in fp32 under the
reference's own names (``Bert.bert_model.embeddings.word_embeddings.weight``, ``...encoder.layer.3.attention.self.query.weight``,
``...LayerNorm.gamma``) and builds the autograd graph out of the pieces the trunk already has, on the SAME packed token stream:

  projections   ops.linear  -> ``ruart_gemm_x3`` (fp32 operands, three bf16 MFMA products) for x.W^T, dY.W and dY^T.X
  attention     per group of sequences, padded to the group's longest REAL length: ops.fused_attention -> ``ruart_attn_fwd/bwd``
                (the reference's -10000 key bias and a hard mask agree to the last bit in fp32: exp(-10000 + s - max) == 0),
                attention-probability dropout as a multiplier inside the kernel (``ruart_attn_fwd_pscale``);
                plain torch (bmm, softmax, dropout) only when a sequence exceeds the kernel's 384-key panel
  LayerNorm / GELU / embeddings / dropout   torch (row-wise, HBM-bound; TF-style LN == F.layer_norm with eps 1e-12)

Dropout follows the reference: ``hidden_dropout_prob`` after the embeddings and after both output projections,
``attention_probs_dropout_prob`` on the attention probabilities, in training mode only (modeling.py:198, 244-246, 262, 301).
Every layer's output is returned (modeling.py:326-333), as the linear layer mix needs all of them.

Cost: fp32 storage and 3 MFMA products per GEMM - the accuracy class of the fp32 reference (tests hold parameter gradients to 1e-3
relative), about a quarter of the frozen path's encoder speed.  16-bit backward kernels are the next step."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class _Gather(torch.autograd.Function):
    """y = x[idx] for an index map whose backward is known in closed form: rows of x that ``idx`` duplicates (the padding slots
    all point at row 0) or drops receive no gradient in this graph, so dx is a second gather (``back``: for every row of x one
    row of dy, or -1) instead of the atomic scatter-add autograd would run."""

    @staticmethod
    def forward(ctx, x, idx, back):
        ctx.save_for_backward(back)
        return x.index_select(0, idx)

    @staticmethod
    def backward(ctx, gy):
        (back,) = ctx.saved_tensors
        gx = gy.index_select(0, back.clamp_min(0))
        return gx * (back >= 0).to(gx.dtype).unsqueeze(1), None, None


class _Node(nn.Module):
    """Name-only container: gives parameters the reference's dotted paths."""


def _register(root, dotted, tensor):
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if not hasattr(m, p):
            m.add_module(p, _Node())
        m = getattr(m, p)
    prm = nn.Parameter(tensor)
    m.register_parameter(parts[-1], prm)
    return prm


class BertModelTrainable(nn.Module):
    """State-dict compatible stand-in for the reference's ``BertModel`` (modeling.py:524-614) on the packed token stream."""

    def __init__(self, state, cfg, device, gemm="x3"):
        super().__init__()
        self.cfg = dict(cfg)
        # "x3": every projection and gradient on the split-bf16 kernel (fp32-class accuracy, the default);
        # "16": x W^T in f16 and dY W in bf16 on the frozen path's MFMA GEMM, dW on the split-bf16 kernel (opt['bert_train_gemm'])
        self._lin = ops.linear16 if gemm == "16" else ops.linear
        self.hidden = int(cfg["hidden_size"])
        self.n_layers = int(cfg["num_hidden_layers"])
        self.n_heads = int(cfg["num_attention_heads"])
        self.p_hidden = float(cfg.get("hidden_dropout_prob", 0.1))
        self.p_attn = float(cfg.get("attention_probs_dropout_prob", 0.1))
        self._p = {}
        for k, v in state.items():
            name = k[5:] if k.startswith("bert.") else k
            if name.startswith("cls."):
                continue                                   # pre-training heads: not part of BertModel
            name = name.replace("LayerNorm.weight", "LayerNorm.gamma").replace("LayerNorm.bias", "LayerNorm.beta")
            t = (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))).to(torch.float32)
            self._p[name] = _register(self, name, t.clone().to(device))

    def _ln(self, x, prefix):
        return F.layer_norm(x, (self.hidden,), self._p[prefix + ".gamma"], self._p[prefix + ".beta"], 1e-12)

    def _attention(self, qkv, plan, training):
        """qkv (T, 3H) packed -> context (T, H).  ``plan``: per group (token index matrix padded with 0, key mask)."""
        T = qkv.size(0)
        nh, hd = self.n_heads, self.hidden // self.n_heads
        outs = []
        for idx, mask, back in plan["groups"]:
            N, Lg = idx.shape
            g = _Gather.apply(qkv, idx.reshape(-1), back).view(N, Lg, 3, nh, hd).permute(2, 0, 3, 1, 4)  # (3, N, nh, Lg, hd)
            q, k, v = (g[i].reshape(N * nh, Lg, hd) for i in range(3))
            km = mask.view(N, 1, Lg).expand(N, nh, Lg).reshape(N * nh, Lg)
            drop = training and self.p_attn > 0
            if Lg > 384:
                # Beyond the fused kernel's key panel (385..512 word pieces in ONE sequence: the reference's windows are at most 512,
                # Models/Bert/Bert.py:96-99).  Rare enough to be served slice by slice - but not by the vendor's batched GEMM: every
                # product of a step stays on this repository's kernels (ops.py; the stream-K solutions of the library are what hung
                # round 1's three-stream trunk), so each (sequence, head) takes two ruart_gemm_x3 products around a torch softmax.
                ctxs = []
                for n in range(N * nh):
                    s = ops.matmul2d(q[n], k[n].t()).masked_fill(~km[n].bool().unsqueeze(0), float("-inf"))
                    ctxs.append(ops.matmul2d(F.dropout(torch.softmax(s, dim=-1), self.p_attn, training), v[n]))
                ctx = torch.stack(ctxs, 0)
            else:
                ps = None
                if drop:
                    ps = (torch.rand(N * nh, Lg, Lg, device=q.device) >= self.p_attn).to(q.dtype).mul_(1.0 / (1.0 - self.p_attn))
                ctx = ops.fused_attention(q, k, v, km, prob_scale=ps)
            outs.append(ctx.view(N, nh, Lg, hd).permute(0, 2, 1, 3).reshape(N * Lg, self.hidden))
        return _Gather.apply(torch.cat(outs, 0), plan["token_slot"], plan["slot_token"])[:T]

    def forward(self, packed, training=False):
        """All layer outputs (n_layers, T, H) fp32 for the ``PackedTokens`` stream."""
        P = self._p
        plan = attention_plan(packed)
        T = packed.T
        ids, pos = packed.ids[:T].long(), packed.pos[:T].long()
        x = F.embedding(ids, P["embeddings.word_embeddings.weight"]) + F.embedding(pos, P["embeddings.position_embeddings.weight"]) \
            + P["embeddings.token_type_embeddings.weight"][0]
        x = F.dropout(self._ln(x, "embeddings.LayerNorm"), self.p_hidden, training)
        scale = 1.0 / float(np.sqrt(self.hidden // self.n_heads))
        layers = []
        for l in range(self.n_layers):
            pre = "encoder.layer.%d." % l
            a = pre + "attention.self."
            w_qkv = torch.cat([P[a + "query.weight"] * scale, P[a + "key.weight"], P[a + "value.weight"]], 0)
            b_qkv = torch.cat([P[a + "query.bias"] * scale, P[a + "key.bias"], P[a + "value.bias"]], 0)
            ctx = self._attention(self._lin(x, w_qkv, b_qkv), plan, training)
            o = self._lin(ctx, P[pre + "attention.output.dense.weight"], P[pre + "attention.output.dense.bias"])
            x = self._ln(F.dropout(o, self.p_hidden, training) + x, pre + "attention.output.LayerNorm")
            h = F.gelu(self._lin(x, P[pre + "intermediate.dense.weight"], P[pre + "intermediate.dense.bias"]))
            o = self._lin(h, P[pre + "output.dense.weight"], P[pre + "output.dense.bias"])
            x = self._ln(F.dropout(o, self.p_hidden, training) + x, pre + "output.LayerNorm")
            layers.append(x)
        return torch.stack(layers, 0)


def attention_plan(packed):
    """Device index vectors for the padded-per-group attention, cached on the batch: for every group the (N, Lg) matrix of packed
    token indices (0 where padded) with its key mask, and for every packed token its slot in the concatenation of the groups'
    padded outputs."""
    plan = getattr(packed, "_train_plan", None)
    if plan is not None:
        return plan
    if packed.bias_host is not None:
        raise NotImplementedError("the trainable encoder expects the packed stream (no padded slots)")
    dev = packed.ids.device
    groups, slot, base = [], np.zeros(packed.T, dtype=np.int64), 0
    slot_token = []
    for gidx in packed.group_index:
        real = gidx >= 0
        cols = np.nonzero(real.any(0))[0]
        Lg = int(cols.max()) + 1 if len(cols) else 1       # the reference's column positions, cut after the last real one
        g, m = gidx[:, :Lg], real[:, :Lg]
        n_idx, j_idx = np.nonzero(m)
        slot[g[m]] = base + n_idx * Lg + j_idx
        # backward map of this group's gather: packed token t <- its own padded slot (local to the group), others -1
        back = np.full(packed.T, -1, dtype=np.int64)
        back[g[m]] = n_idx * Lg + j_idx
        groups.append((torch.from_numpy(np.where(m, g, 0).astype(np.int64)).to(dev), torch.from_numpy(m.astype(np.uint8)).to(dev),
                       torch.from_numpy(back).to(dev)))
        slot_token.append(np.where(m, g, -1).reshape(-1))  # padded slot -> the token it holds (or -1)
        base += g.shape[0] * Lg
    plan = {"groups": groups, "token_slot": torch.from_numpy(slot).to(dev),
            "slot_token": torch.from_numpy(np.concatenate(slot_token).astype(np.int64)).to(dev)}
    packed._train_plan = plan
    return plan


def mix_layers(layer_w, layers):
    """sum_l layer_w[l] * layers[l]  (SDNet.py:573-581) - as a broadcast multiply and a reduction over the layer axis: the library
    GEMM an einsum lowers this to (M = 1, K = n_layers, N = T * H) took 6 ms per call at B = 64."""
    return (layers * layer_w.to(layers.dtype).view(-1, 1, 1)).sum(0)


def pool_words(mixed, span_start, span_len, dst_row, n_rows, n_pieces=None):
    """Average every word's piece span of the mixed stream (Bert.py:149-165); rows without a word stay zero.
    ``n_pieces``: sum(span_len) when the host knows it (batch.BatchIndex.span_pieces) - without it ``repeat_interleave`` has to read
    the sum back from the device, a host sync in the middle of the step (7 ms each at B = 64: the host waits for the encoder)."""
    W = span_start.numel()
    out = mixed.new_zeros(n_rows, mixed.size(1))
    if W == 0:
        return out
    ln = span_len.long()
    word_of_piece = torch.repeat_interleave(torch.arange(W, device=ln.device), ln, output_size=n_pieces)
    first = torch.cumsum(ln, 0) - ln
    piece = span_start.long()[word_of_piece] + (torch.arange(word_of_piece.numel(), device=ln.device) - first[word_of_piece])
    rows = mixed.index_select(0, piece) / ln[word_of_piece].unsqueeze(1).to(mixed.dtype)
    return out.index_add(0, dst_row.long()[word_of_piece], rows)


def pool_mix(layer_w, layers, span_start, span_len, dst_row, n_rows):
    """Differentiable form of ``ruart_bert_pool_mix``: mix the layers, then average every word's piece span.  Gradients reach
    both the layer weights and the encoder."""
    return pool_words(mix_layers(layer_w, layers), span_start, span_len, dst_row, n_rows)
