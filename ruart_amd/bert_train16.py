"""Trainable BERT encoder, 16-bit path (``opt['bert_train_gemm'] = '16'`` in a conf without ``LOCK_BERT``).

Same parameters, names and dropout semantics as ``bert_train.BertModelTrainable`` (the fp32-class path that pins parity), but the
whole encoder - forward and backward - is ONE ``torch.autograd.Function`` over hand-written kernels instead of an autograd graph of
torch row-wise ops:

  forward   embeddings (torch gathers) -> ``ruart_ln_train_fwd`` (LayerNorm, then dropout: modeling.py:196-199); per layer the master
            weights become GEMM operands in one pass each (``ruart_weight_prep``: f16 for the forward, transposed bf16 for the backward);
            QKV / attention-output / output projections on ``ruart_gemm_16_nt`` (f16 operands, the frozen path's MFMA kernel), the
            intermediate one on ``ruart_gemm_16_nt_gelu2`` (pre-activation AND its GELU out of one epilogue), ``ruart_attn_train_fwd``
            (MFMA flash attention with hash-generated probability dropout), ``ruart_ln_train_fwd`` for dense -> dropout -> + input ->
            LayerNorm (:260-264, :299-303); the layer mix of ``SDNet.linear_sum`` (Models/SDNet.py:573-581) is taken inside
            (``ruart_mix_rows``), so what leaves is ONE (T, H) fp32 stream instead of twelve layer outputs
  saved     per layer, f16: layer input, QKV rows, context rows, both LayerNorm inputs (+ mean / rstd), the intermediate
            pre-activations: 0.8 GB per layer at the bench shape (9.9 GB for bert-base) - the GELU output and every dropout mask are
            recomputed
  backward  ``ruart_ln_train_bwd`` (residual-stream gradient in fp32, GEMM-bound gradient in bf16 with the dropout multiplier
            regenerated from the seed, gamma / beta / dense-bias partial sums; the layer-mix gradient joins the stream there),
            dX = dY . W on the same MFMA kernel with bf16 operands - the one through the GELU as ``ruart_gemm_16_nt_gelu_bwd`` (gelu'
            applied, gelu(h) re-emitted and the bias sums taken in its epilogue) -, dW = dY^T . X on the TN kernel straight from the
            row-major operands (``ruart_gemm_16_tn_splitk`` + ``ruart_splitk_reduce``, slabs summed in order),
            ``ruart_attn_train_bwd``, embedding tables through the ordered host-sorted sum (``ops.embedding_grad``).
            Every reduction has a fixed order: two passes from the same seed give the same bits.

Accuracy class: mixed-precision training - f16 activations, bf16 gradients, fp32 accumulation, master weights and residual-stream
gradient (tests hold every parameter-gradient norm of the reference's backward to 3 %).  Sequences longer than 64 word pieces (up to the
512 of Models/Bert/Bert.py:96-99) are cut into 64-token chunks that attend to their whole sequence (``ruart_attn_train_fwd_long`` /
``_bwd_long``); only a stream with a key bias (the unpacked -10000 mode) still takes the fp32-class path of bert_train.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import hip
from .bert_train import BertModelTrainable

_LAYER_TENSORS = ("attention.self.query.weight", "attention.self.query.bias", "attention.self.key.weight", "attention.self.key.bias",
                  "attention.self.value.weight", "attention.self.value.bias", "attention.output.dense.weight",
                  "attention.output.dense.bias", "attention.output.LayerNorm.gamma", "attention.output.LayerNorm.beta",
                  "intermediate.dense.weight", "intermediate.dense.bias", "output.dense.weight", "output.dense.bias",
                  "output.LayerNorm.gamma", "output.LayerNorm.beta")
_EMB_TENSORS = ("embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight", "embeddings.token_type_embeddings.weight",
                "embeddings.LayerNorm.gamma", "embeddings.LayerNorm.beta")


def _chk(rc, what):
    hip.check(rc, what)


class _Run:
    """One forward / backward pass: buffers, launches and the saved activations."""

    def __init__(self, model, packed, training, params, keep=True):
        self.m, self.packed, self.training = model, packed, training
        self.keep = keep                                              # False: a pass nobody will differentiate - nothing is saved
        self.P = dict(zip(model._order, params))
        self.lib = hip.load()
        self.dev = packed.ids.device
        self.T, self.Tp = packed.T, packed.Tp
        self.H, self.NL, self.nh = model.hidden, model.n_layers, model.n_heads
        self.I = int(model.cfg["intermediate_size"])
        self.p_h = model.p_hidden if training else 0.0
        self.p_a = model.p_attn if training else 0.0
        import os
        self.fused_gelu_bwd = os.environ.get("RUART_FUSED_GELU_BWD") == "1"     # experiments: the GELU backward in the dX product's epilogue
        # one 31-bit stream id per pass (CPU generator: no device sync); every dropout site adds its own offset
        self.seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if (self.p_h > 0 or self.p_a > 0) else 0
        if self.seed and torch.distributed.is_available() and torch.distributed.is_initialized():
            # the CPU generator is seeded identically on every data-parallel rank (trainer.py): keep the ranks' masks apart
            self.seed = (self.seed + 0x632BE5AB * torch.distributed.get_rank()) & 0x7FFFFFFF

    def _st(self):
        return hip.stream_ptr(self.dev)                             # the stream of the tensors' device, not of the current one

    # -- small launch helpers ----------------------------------------------------------------------------------------------
    def _new(self, rows, cols, dtype, zero=False):
        """(rows, cols) buffer; ``zero``: the pad rows past the last real token are cleared (the kernels behind it write the real ones)"""
        b = torch.empty(rows, cols, dtype=dtype, device=self.dev)
        if zero and rows > self.T:
            b[self.T:].zero_()
        return b

    def _gemm(self, A, W, bias, out, in_dt, act=hip.ACT_NONE, res=None, res_dt=hip.DT_F32):
        """out (Tp, N) = act(A (Tp, K) . W (N, K)^T + bias) + res"""
        M, K = A.shape
        N = W.shape[0]
        out_dt = hip.DT_F32 if out.dtype == torch.float32 else in_dt
        _chk(self.lib.ruart_gemm_16_nt(hip.ptr(A), K, hip.ptr(W), K, hip.ptr(bias), hip.ptr(res), N, res_dt, hip.ptr(out), N, out_dt, M, N, K, act,
                                       in_dt, self._st()), "ruart_gemm_16_nt")
        return out

    def _seed(self, layer, site):
        return (self.seed + 7919 * (4 * layer + site + 1)) & 0x7FFFFFFF

    def _ln_fwd(self, x32, res16, g, b, p, seed, post=0, out=None):
        Tp, H = self.Tp, self.H
        y = self._new(Tp, H, torch.float16) if out is None else out
        pre, st = self._new(Tp, H, torch.float16), self._new(Tp, 2, torch.float32)
        _chk(self.lib.ruart_ln_train_fwd(hip.ptr(x32), H, hip.ptr(res16), H, hip.ptr(g), hip.ptr(b), 1e-12, float(p), int(seed), post, hip.ptr(y),
                                         hip.ptr(pre), hip.ptr(st), H, Tp, H, self._st()), "ruart_ln_train_fwd")
        return y, pre, st

    def _prep_weights(self, l, scale):
        """The layer's four weights as GEMM operands, one pass per master tensor: f16 (N, K) for the forward and, kept for the backward
        pass, bf16 (K, N) - the transpose - for dX = dY . W.  Query rows carry the 1/sqrt(d) of the scores."""
        P, H, I, dev = self.P, self.H, self.I, self.dev
        pre = "encoder.layer.%d." % l
        a = pre + "attention.self."
        wq16 = torch.empty(3 * H, H, dtype=torch.float16, device=dev)
        wqT = torch.empty(H, 3 * H, dtype=torch.bfloat16, device=dev) if self.keep else None
        items = (hip.WPrepItemC * 6)()
        k = 0
        for i, (n, sc) in enumerate((("query", scale), ("key", 1.0), ("value", 1.0))):
            items[k] = hip.WPrepItemC(P[a + n + ".weight"].data_ptr(), wq16[i * H:].data_ptr(), wqT[:, i * H:].data_ptr() if self.keep else None,
                                      H, H, 3 * H, H, H, float(sc))
            k += 1
        out = [wq16]
        back = [wqT]
        for n, rows, cols in (("attention.output.dense", H, H), ("intermediate.dense", I, H), ("output.dense", H, I)):
            w16 = torch.empty(rows, cols, dtype=torch.float16, device=dev)
            wT = torch.empty(cols, rows, dtype=torch.bfloat16, device=dev) if self.keep else None
            items[k] = hip.WPrepItemC(P[pre + n + ".weight"].data_ptr(), w16.data_ptr(), wT.data_ptr() if self.keep else None, cols, cols, rows,
                                      rows, cols, 1.0)
            k += 1
            out.append(w16)
            back.append(wT)
        _chk(self.lib.ruart_weight_prep_batch(items, 6, self._st()), "ruart_weight_prep_batch")      # the layer's six weights, one launch
        self.wT[l] = back
        return out

    def _layer_fwd(self, l, x16, out16):
        """One encoder layer on the f16 kernels: x16 (Tp, H) -> out16; returns what its backward reads (QKV rows, context, both LayerNorm
        inputs + statistics, the post-attention stream, the intermediate pre-activation) and leaves the layer's transposed bf16 weights
        in ``self.wT[l]``.  Called by the forward - or, in recompute mode, by the backward just before the layer's own backward."""
        P, lib, Tp, H, I = self.P, self.lib, self.Tp, self.H, self.I
        pk, st = self.packed, self._st
        scale = 1.0 / float(np.sqrt(H // self.nh))
        pre = "encoder.layer.%d." % l
        a = pre + "attention.self."
        b_qkv = torch.cat([P[a + "query.bias"] * scale, P[a + "key.bias"], P[a + "value.bias"]], 0)
        wq16, wo16, w1_16, w2_16 = self._prep_weights(l, scale)
        qkv = self._gemm(x16, wq16, b_qkv, self._new(Tp, 3 * H, torch.float16), hip.DT_F16)
        ctx = self._new(Tp, H, torch.float16, zero=True)
        plan = pk.train_plan(self.dev)
        if plan["n_win"]:                                            # windows of whole short sequences
            _chk(lib.ruart_attn_train_fwd(hip.ptr(qkv), 3 * H, hip.ptr(ctx), H, H, self.nh, plan["n_win"], hip.ptr(plan["win"][0]),
                                          hip.ptr(plan["win"][1]), hip.ptr(pk.tok_lo), float(self.p_a), self._seed(l, 0), st()), "ruart_attn_train_fwd")
        lse = None
        if plan["n_chunks"]:                                         # sequences longer than one window: <= 64-token chunks against the whole sequence
            c = plan["chunks"]
            lse = torch.empty(self.T, self.nh, dtype=torch.float32, device=self.dev)
            _chk(lib.ruart_attn_train_fwd_long(hip.ptr(qkv), 3 * H, hip.ptr(ctx), H, H, self.nh, plan["n_chunks"], hip.ptr(c[0]), hip.ptr(c[1]),
                                               hip.ptr(c[2]), hip.ptr(c[3]), float(self.p_a), self._seed(l, 0), hip.ptr(lse), st()),
                 "ruart_attn_train_fwd_long")
        ao = self._gemm(ctx, wo16, P[pre + "attention.output.dense.bias"], self._new(Tp, H, torch.float32), hip.DT_F16)
        mid, pre1, st1 = self._ln_fwd(ao, x16, P[pre + "attention.output.LayerNorm.gamma"], P[pre + "attention.output.LayerNorm.beta"],
                                      self.p_h, self._seed(l, 1))
        h16, g16 = self._new(Tp, I, torch.float16), self._new(Tp, I, torch.float16)
        _chk(lib.ruart_gemm_16_nt_gelu2(hip.ptr(mid), H, hip.ptr(w1_16), H, hip.ptr(P[pre + "intermediate.dense.bias"]), hip.ptr(h16), hip.ptr(g16),
                                        I, Tp, I, H, hip.DT_F16, st()), "ruart_gemm_16_nt_gelu2")
        ff = self._gemm(g16, w2_16, P[pre + "output.dense.bias"], ao, hip.DT_F16)           # reuses the fp32 buffer
        del g16
        _, pre2, st2 = self._ln_fwd(ff, mid, P[pre + "output.LayerNorm.gamma"], P[pre + "output.LayerNorm.beta"], self.p_h,
                                    self._seed(l, 2), out=out16)
        return (qkv, ctx, pre1, st1, mid, h16, pre2, st2, lse)

    def _embed_fwd(self):
        """embeddings -> LayerNorm -> dropout on the f16 kernels: (x16, LayerNorm input f16, statistics)"""
        P, T, Tp, H, pk = self.P, self.T, self.Tp, self.H, self.packed
        ids, pos = pk.ids[:T].long(), pk.pos[:T].long()
        e = torch.zeros(Tp, H, dtype=torch.float32, device=self.dev)
        e[:T] = (F.embedding(ids, P["embeddings.word_embeddings.weight"]) + F.embedding(pos, P["embeddings.position_embeddings.weight"])) \
            + P["embeddings.token_type_embeddings.weight"][0]
        return self._ln_fwd(e, None, P["embeddings.LayerNorm.gamma"], P["embeddings.LayerNorm.beta"], self.p_h, self._seed(-1, 0), post=1)

    # -- the accurate forward: fp16c kernels of the frozen path on the live parameters -------------------------------------------
    def _forward_accurate(self, layer_w):
        """A pass WITHOUT active dropout (evaluation, and the parity tests: Models/Bert/modeling.py's dropouts are identities there) is
        the frozen encoder's computation on the current parameters - so it runs on the frozen path's kernels in its fp16c precision
        (f16 MFMA products + fp8 correction, fp32 residual stream): the answer probabilities then hold the 1e-3 of the north star with
        the encoder unlocked too (plain f16 operands: 1.7e-3).  Nothing but the layer outputs is kept (f16, 0.8 GB at the bench shape
        instead of 9.9 GB): the backward recomputes each layer's activations on the f16 training kernels right before it needs them."""
        from .bert import _Buffers, bert_encode
        lib, T, Tp, H, NL = self.lib, self.T, self.Tp, self.H, self.NL
        W = self.m.accurate_weights()
        bufs = self.m.__dict__.setdefault("_acc_buffers", _Buffers())
        layers32 = bert_encode(W, self.packed, bufs)                    # (NL, Tp, H) fp32; aliases a reusable buffer: consumed below
        self.layers = torch.empty(NL, Tp, H, dtype=torch.float16, device=self.dev)
        _chk(lib.ruart_cast_f32_to_16(hip.ptr(layers32), hip.ptr(self.layers), hip.DT_F16, NL * Tp * H, 1.0, self._st()), "ruart_cast_f32_to_16")
        self.recompute = True
        self.saved, self.wT = {}, {}
        if layer_w is None:
            return self.layers
        self.lw = layer_w.detach().to(torch.float32).contiguous()
        mixed = torch.zeros(Tp, H, dtype=torch.float32, device=self.dev)
        for l in range(NL):
            mixed.addcmul_(layers32[l], self.lw[l])
        return mixed[:T]

    # -- forward -------------------------------------------------------------------------------------------------------------
    def forward(self, layer_w):
        P, lib, T, Tp, H, I, NL = self.P, self.lib, self.T, self.Tp, self.H, self.I, self.NL
        st = self._st
        self.recompute = False
        if self.p_h == 0.0 and self.p_a == 0.0 and self.m.accurate_forward:
            return self._forward_accurate(layer_w)
        x16, self.pre_e, self.st_e = self._embed_fwd()
        self.x_in = x16                                               # input of layer 0
        self.layers = torch.empty(NL, Tp, H, dtype=torch.float16, device=self.dev)
        self.saved = {}
        self.wT = {}
        for l in range(NL):
            saved = self._layer_fwd(l, x16, self.layers[l])
            if self.keep:
                self.saved[l] = saved
            x16 = self.layers[l]
        if layer_w is None:
            return self.layers                                        # frozen-encoder use: every layer output, no mix
        self.lw = layer_w.detach().to(torch.float32).contiguous()
        mixed = torch.empty(Tp, H, dtype=torch.float32, device=self.dev)
        _chk(lib.ruart_mix_rows(hip.ptr(self.layers), Tp * H, H, NL, hip.ptr(self.lw), hip.ptr(mixed), H, Tp, H, st()), "ruart_mix_rows")
        return mixed[:T]

    # -- backward ------------------------------------------------------------------------------------------------------------
    def _bf16(self, x16):
        """bf16 copy of a saved f16 activation (the X operand of a weight-gradient product; transient)"""
        out = self.xb[:x16.numel()].view(x16.shape)
        _chk(self.lib.ruart_f16_to_bf16(hip.ptr(x16), hip.ptr(out), x16.numel(), self._st()), "ruart_f16_to_bf16")
        return out

    def _dw(self, dY_bf16, X_bf16, row_scales=None):
        """(N_out, K_in) fp32 = dY^T . X over the token rows, straight from the row-major operands: split over the token rows to fill
        the chip (a 768 x 768 output is 9 tiles), slabs summed in slice order.  ``row_scales``: [(rows, scale), ...] cuts the output
        into separate tensors by row blocks (the fused QKV product -> query / key / value gradients, the query's with the 1/sqrt(d)
        that the forward folded into its weights)."""
        lib, Tp = self.lib, self.Tp
        M, N = dY_bf16.shape[1], X_bf16.shape[1]
        tiles = (M // 256) * (N // 256)
        nz = max(1, min(256 // tiles, Tp // 128))
        tchunk = ((Tp + nz - 1) // nz + 127) // 128 * 128
        nz = (Tp + tchunk - 1) // tchunk
        part = self.part[:nz * M * N]
        _chk(lib.ruart_gemm_16_tn_splitk(hip.ptr(dY_bf16), M, hip.ptr(X_bf16), N, hip.ptr(part), N, M, N, Tp, tchunk, hip.DT_BF16, self._st()),
             "ruart_gemm_16_tn_splitk")
        outs, r0 = [], 0
        for rows, scale in (row_scales or [(M, 1.0)]):
            dW = torch.empty(rows, N, dtype=torch.float32, device=self.dev)
            _chk(lib.ruart_splitk_reduce(hip.ptr(part[r0 * N:]), M * N, nz, hip.ptr(dW), rows * N, float(scale), 0, self._st()),
                 "ruart_splitk_reduce")
            outs.append(dW)
            r0 += rows
        return outs if row_scales else outs[0]

    def _ln_bwd(self, dy, add, add_scale, pre16, stats, gamma, p, seed, post=0):
        Tp, H = self.Tp, self.H
        # one buffer each per backward pass, pad rows cleared once: a LayerNorm's two outputs are dead (every reader enqueued, on this
        # stream) before the next LayerNorm backward writes them again - 46 pad-row fills per step fewer than fresh buffers
        if getattr(self, "_d_res", None) is None:
            self._d_res = self._new(Tp, H, torch.float32, zero=True)
            self._d_gemm = self._new(Tp, H, torch.bfloat16, zero=True)
        d_res = self._d_res
        d_gemm = self._d_gemm if not post else None
        dg, db = torch.empty(H, device=self.dev), torch.empty(H, device=self.dev)
        dbias = torch.empty(H, device=self.dev) if not post else None
        _chk(self.lib.ruart_ln_train_bwd(hip.ptr(dy), H, hip.ptr(add), hip.ptr(add_scale), hip.ptr(pre16), H, hip.ptr(stats), hip.ptr(gamma), float(p),
                                         int(seed), post, hip.ptr(d_res), H, hip.ptr(d_gemm), H, hip.ptr(dg), hip.ptr(db), hip.ptr(dbias), 0,
                                         hip.ptr(self.ln_ws), self.T, H, self._st()), "ruart_ln_train_bwd")
        return d_res, d_gemm, dg, db, dbias

    def backward(self, g_mixed):
        P, lib, T, Tp, H, I, NL = self.P, self.lib, self.T, self.Tp, self.H, self.I, self.NL
        pk = self.packed
        st = self._st
        dev = self.dev
        scale = 1.0 / float(np.sqrt(H // self.nh))
        grads = {}
        G = torch.zeros(Tp, H, dtype=torch.float32, device=dev)
        G[:T] = g_mixed
        d_lw = torch.empty(NL, dtype=torch.float32, device=dev)
        ws = torch.empty(512 * NL, dtype=torch.float32, device=dev)
        _chk(lib.ruart_mix_rows_bwd(hip.ptr(self.layers), Tp * H, H, NL, hip.ptr(G), H, hip.ptr(d_lw), hip.ptr(ws), Tp, H, st()), "ruart_mix_rows_bwd")
        wmax = max(3 * H, I)
        self.xb = torch.empty(Tp * H, dtype=torch.bfloat16, device=dev)
        self.part = torch.empty(max(256, Tp // 128) * 256 * 256 + 4 * wmax * H, dtype=torch.float32, device=dev)
        self.cs_ws = torch.empty(int(lib.ruart_gemm_16_nt_gelu_bwd_ws_floats(Tp, I)), dtype=torch.float32, device=dev)
        self.ln_ws = torch.empty(int(lib.ruart_ln_train_bwd_ws_floats(H)), dtype=torch.float32, device=dev)
        plan = pk.train_plan(dev)
        n_win, n_chunks = plan["n_win"], plan["n_chunks"]
        self.bias_part = torch.empty(max(n_win, 1), 2 * H, dtype=torch.float32, device=dev)
        if n_chunks:
            self.bias_part_long = torch.empty(n_chunks, 2 * H, dtype=torch.float32, device=dev)
            self.delta_ws = torch.empty(T, self.nh, dtype=torch.float32, device=dev)
            self.scale_ws = torch.empty(n_chunks, self.nh, dtype=torch.float32, device=dev)
        dqkv = torch.zeros(Tp, 3 * H, dtype=torch.bfloat16, device=dev)                 # pad rows stay zero
        dX = torch.zeros(Tp, H, dtype=torch.float32, device=dev)
        blk_q0, blk_q1 = plan["win"]
        if self.recompute:                                                               # (accurate forward: only layer outputs were kept)
            self.x_in, self.pre_e, self.st_e = self._embed_fwd()
            scratch = self._new(Tp, H, torch.float16)
        for l in range(NL - 1, -1, -1):
            pre = "encoder.layer.%d." % l
            a = pre + "attention.self."
            x16 = self.layers[l - 1] if l > 0 else self.x_in
            if self.recompute:
                self.saved[l] = self._layer_fwd(l, x16, scratch)      # this layer's activations again, on the f16 kernels
            qkv, ctx, pre1, st1, mid, h16, pre2, st2, lse = self.saved[l]
            # ---- output LayerNorm (+ the layer-mix gradient of this layer's output) and the FFN
            d_res2, d_g2, dg2, db2, dbias2 = self._ln_bwd(dX, G, self.lw[l:l + 1], pre2, st2, P[pre + "output.LayerNorm.gamma"], self.p_h,
                                                          self._seed(l, 2))
            grads[pre + "output.LayerNorm.gamma"], grads[pre + "output.LayerNorm.beta"] = dg2, db2
            grads[pre + "output.dense.bias"] = dbias2
            w_qkv_t, wot, w1t, w2t = self.wT[l]                                                    # (K, N) bf16: dX = dY . W as NT products
            # dY . W2 with the GELU backward in the product's epilogue: d_h, gelu(h) again and the bias partial sums in one kernel
            d_h, g_b = self._new(Tp, I, torch.bfloat16), self._new(Tp, I, torch.bfloat16)
            db1_ff = torch.empty(I, dtype=torch.float32, device=dev)
            if self.fused_gelu_bwd:
                _chk(lib.ruart_gemm_16_nt_gelu_bwd(hip.ptr(d_g2), H, hip.ptr(w2t), H, hip.ptr(h16), I, hip.ptr(d_h), hip.ptr(g_b), I,
                                                   hip.ptr(self.cs_ws), Tp, I, H, st()), "ruart_gemm_16_nt_gelu_bwd")
            else:      # (default, round 5) the plain product, then one elementwise pass: 160 + 190 us against 476 in the fused epilogue
                self._gemm(d_g2, w2t, None, d_h, hip.DT_BF16)
                _chk(lib.ruart_gelu_bwd_rows(hip.ptr(d_h), hip.ptr(h16), I, hip.ptr(g_b), hip.ptr(self.cs_ws), Tp, I, st()), "ruart_gelu_bwd_rows")
            _chk(lib.ruart_colsum_f32_rows(hip.ptr(self.cs_ws), Tp // 128, I, I, hip.ptr(db1_ff), 0, st()), "ruart_colsum_f32_rows")
            grads[pre + "output.dense.weight"] = self._dw(d_g2, g_b)
            del g_b
            grads[pre + "intermediate.dense.bias"] = db1_ff
            grads[pre + "intermediate.dense.weight"] = self._dw(d_h, self._bf16(mid))
            d_mid = self._gemm(d_h, w1t, None, self._new(Tp, H, torch.float32), hip.DT_BF16, res=d_res2)
            del d_h, d_res2
            # ---- attention-output LayerNorm, output projection, attention, QKV projection
            d_res1, d_g1, dg1, db1, dbias1 = self._ln_bwd(d_mid, None, None, pre1, st1, P[pre + "attention.output.LayerNorm.gamma"], self.p_h,
                                                          self._seed(l, 1))
            grads[pre + "attention.output.LayerNorm.gamma"], grads[pre + "attention.output.LayerNorm.beta"] = dg1, db1
            grads[pre + "attention.output.dense.bias"] = dbias1
            grads[pre + "attention.output.dense.weight"] = self._dw(d_g1, self._bf16(ctx))
            d_ctx = self._gemm(d_g1, wot, None, self._new(Tp, H, torch.bfloat16), hip.DT_BF16)
            db = torch.empty(2 * H, dtype=torch.float32, device=dev)          # [query | value] bias gradients: the windows' sums, in order
            if n_win:
                _chk(lib.ruart_attn_train_bwd(hip.ptr(qkv), 3 * H, hip.ptr(d_ctx), H, hip.ptr(dqkv), 3 * H, H, self.nh, n_win, hip.ptr(blk_q0),
                                              hip.ptr(blk_q1), hip.ptr(pk.tok_lo), float(self.p_a), self._seed(l, 0), hip.ptr(self.bias_part), st()),
                     "ruart_attn_train_bwd")
                _chk(lib.ruart_colsum_f32_rows(hip.ptr(self.bias_part), n_win, 2 * H, 2 * H, hip.ptr(db), 0, st()), "ruart_colsum_f32_rows")
            if n_chunks:                                                      # sequences longer than one window
                c = plan["chunks"]
                _chk(lib.ruart_attn_train_bwd_long(hip.ptr(qkv), 3 * H, hip.ptr(d_ctx), H, hip.ptr(dqkv), 3 * H, H, self.nh, n_chunks,
                                                   hip.ptr(c[0]), hip.ptr(c[1]), hip.ptr(c[2]), hip.ptr(c[3]), hip.ptr(c[4]), float(self.p_a),
                                                   self._seed(l, 0), hip.ptr(lse), hip.ptr(self.delta_ws), hip.ptr(self.scale_ws),
                                                   hip.ptr(self.bias_part_long), st()), "ruart_attn_train_bwd_long")
                _chk(lib.ruart_colsum_f32_rows(hip.ptr(self.bias_part_long), n_chunks, 2 * H, 2 * H, hip.ptr(db), 1 if n_win else 0, st()),
                     "ruart_colsum_f32_rows")
            # the key bias shifts every score of a query row by the same q . b_k, which the softmax ignores: its gradient is
            # sum_i q_i sum_j dS_ij with sum_j dS_ij = 0 - exactly zero (the reference's 1e-9 is its own rounding noise)
            grads[a + "query.bias"], grads[a + "key.bias"], grads[a + "value.bias"] = db[:H] * scale, torch.zeros_like(db[:H]), db[H:]
            grads[a + "query.weight"], grads[a + "key.weight"], grads[a + "value.weight"] = self._dw(
                dqkv, self._bf16(x16), row_scales=[(H, scale), (H, 1.0), (H, 1.0)])
            dX = self._gemm(dqkv, w_qkv_t, None, self._new(Tp, H, torch.float32), hip.DT_BF16, res=d_res1)
            self.saved.pop(l, None)                                                       # release this layer's activations
            self.wT.pop(l, None)
        # ---- embeddings: dropout(LayerNorm(word + position + type))
        d_e, _, dge, dbe, _ = self._ln_bwd(dX, None, None, self.pre_e, self.st_e, P["embeddings.LayerNorm.gamma"], self.p_h, self._seed(-1, 0), post=1)
        grads["embeddings.LayerNorm.gamma"], grads["embeddings.LayerNorm.beta"] = dge, dbe
        d_e = d_e[:T]
        from .ops import embedding_grad
        for name, srt in zip(("embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight"), pk.embedding_sorts(dev)):
            grads[name] = embedding_grad(d_e, srt, P[name].shape)
        gt = torch.zeros_like(P["embeddings.token_type_embeddings.weight"])
        gt[0] = d_e.sum(0)
        grads["embeddings.token_type_embeddings.weight"] = gt
        self.xb = self.part = None
        return d_lw, grads


class _Encoder16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, packed, training, keep, layer_w, *params):
        run = _Run(model, packed, training, params, keep)
        ctx.run = run if keep else None
        ctx.lw_dtype = layer_w.dtype
        return run.forward(layer_w)

    @staticmethod
    def backward(ctx, g_mixed):
        run = ctx.run
        d_lw, grads = run.backward(g_mixed.contiguous().to(torch.float32))
        ctx.run = None
        return (None, None, None, None, d_lw.to(ctx.lw_dtype)) + tuple(grads.get(n) for n in run.m._order)


class BertModelTrainable16(BertModelTrainable):
    """``forward_mixed(packed, layer_w, training)`` -> (T, H) fp32 = sum_l layer_w[l] * layer_l, differentiable w.r.t. ``layer_w`` and
    every encoder parameter; ``forward`` (all layer outputs, the fp32-class graph of the parent) stays available."""

    fused_mix = True

    def __init__(self, state, cfg, device):
        super().__init__(state, cfg, device, gemm="x3")
        if self.hidden % 256 or int(cfg["intermediate_size"]) % 256:
            raise ValueError("the 16-bit trainable encoder takes hidden / intermediate sizes that are multiples of 256")
        self._order = [n for n in _EMB_TENSORS] + ["encoder.layer.%d.%s" % (l, t) for l in range(self.n_layers) for t in _LAYER_TENSORS]
        missing = [n for n in self._order if n not in self._p]
        if missing:
            raise ValueError("checkpoint lacks encoder tensors: %s" % missing[:3])

    accurate_forward = True          # passes without active dropout run the frozen path's fp16c kernels (bert.Bert.unlock sets it from opt)

    def accurate_weights(self):
        """The live parameters as the frozen path's operand set (f16 + e4m3 companions), rebuilt only when a parameter has changed."""
        from .bert import BertEncoderWeights
        key = tuple(self._p[n]._version for n in self._order) + tuple(self._p[n].data_ptr() for n in self._order[:2])
        ent = self.__dict__.get("_acc_weights")
        if ent is None or ent[0] != key:
            state = {n: self._p[n] for n in self._p}
            dev = self._p[self._order[0]].device
            ent = (key, BertEncoderWeights(state, self.cfg, dev, "fp16c", check_range=False))
            self.__dict__["_acc_weights"] = ent
        return ent[1]

    def supports(self, packed):
        """every packed stream without a key bias: windows of whole short sequences and - round 4 - <= 64-token chunks of longer ones"""
        if packed.bias_host is not None or packed.Tp % 256 != 0:
            return False
        if packed.n_long_blocks == 0 and packed.max_len <= 64:
            return True
        host_plan = packed.train_plan(packed.ids.device if getattr(packed, "ids", None) is not None else torch.device("cpu"))
        return bool(host_plan["ok"])

    def supports_forward_only(self, packed, training):
        """no gradient wanted and no dropout active: the accurate forward takes any packed stream the frozen fp16c path takes"""
        no_drop = not training or (self.p_hidden == 0.0 and self.p_attn == 0.0)
        return self.accurate_forward and no_drop and packed.n_long_blocks == 0 and packed.bias_host is None and packed.Tp % 256 == 0

    @torch.no_grad()
    def layers_nograd(self, packed, training=False):
        """(n_layers, Tp, H) f16: every layer output of one pass with this module's dropouts on or off and nothing kept for a backward
        pass - the frozen encoder's training-mode pass under ``opt['bert_frozen_dropout']`` (Models/SDNetTrainer.py:332 switches the
        dropout layers inside the locked BERT back on)."""
        if not self.supports(packed):
            return None
        return _Run(self, packed, training, [self._p[n] for n in self._order], keep=False).forward(None)

    def forward_mixed(self, packed, layer_w, training=False):
        params = [self._p[n] for n in self._order]
        keep = torch.is_grad_enabled() and (layer_w.requires_grad or any(p.requires_grad for p in params))
        if not self.supports(packed) and not (not keep and self.supports_forward_only(packed, training)):
            from .bert_train import mix_layers
            return mix_layers(layer_w, self.forward(packed, training=training))
        return _Encoder16.apply(self, packed, training, keep, layer_w, *params)
