"""BERT contextual encoder on MI355X: packing, C-ABI launch, sub-word pooling + layer mix.

Host-side mirror of the reference's ``Models/Bert/Bert.py`` (class ``Bert``: ``forward`` :56-90,
``combine_forward`` :130-176) and of the encoder it wraps (``Models/Bert/modeling.py:585-614``).
Everything numeric happens in libruart_hip.so (include/ruart_hip.h); this file only
  * fuses / casts the HF-0.x checkpoint tensors once (Q rows pre-scaled by 1/sqrt(64), exact),
  * packs the valid word pieces of ALL sequences of a step (question + OCR items + object items)
    into one token stream, so padded slots are never computed (the reference computes 30 slots for
    items that hold 3-8 pieces),
  * builds the int32 descriptors the attention and pooling kernels consume.

MI355X-first differences from the reference, all result-preserving:
  * one encoder pass per training step instead of three (sequences are independent);
  * the -10000 additive mask becomes "the key does not exist" (exp underflows to exactly 0 in fp32);
  * the O(items x words) Python pooling loop with a device sync per word is one kernel that also applies
    the softmax(alpha) * gamma layer mix of ``SDNet.linear_sum`` (Models/SDNet.py:573-581).
"""
import ctypes
import os
from ctypes import c_void_p

import numpy as np
import torch
import torch.nn as nn

from . import hip

LONG_BLOCK = 128       # queries per block of the long-sequence attention kernel (ruart_bert_attention)
ROW_PAD = 256          # GEMM row granularity (ruart_gemm_16_nt: M % 128 == 0; the 256-row tile variant needs 256)


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def split_f16c(x, lo_shift=None, hi_shift=None):
    """fp32 (M, K) -> (f16 (M, K), uint8 (M, 2K)): the "f16 + fp8 correction" operand form of ruart_gemm_16c_nt (csrc/common.h):
    row of the second = [e4m3((x - f16(x)) 2^lo_shift) | e4m3(x 2^hi_shift)].  Weights use shifts (18, 7) and the halves swapped."""
    hi = x.to(torch.float16)
    lo = x - hi.to(torch.float32)
    sa_lo, sa_hi, _, _ = hip.f16c_shifts()
    lo_shift = sa_lo if lo_shift is None else lo_shift
    hi_shift = sa_hi if hi_shift is None else hi_shift
    pair = torch.cat([lo * float(2.0 ** lo_shift), x * float(2.0 ** hi_shift)], 1).clamp_(-448.0, 448.0)
    return hi.contiguous(), pair.to(torch.float8_e4m3fn).view(torch.uint8).contiguous()


class BertEncoderWeights:
    """Device-resident encoder weights in the layout ruart_bert_forward expects."""

    def __init__(self, state, cfg, device, dtype="fp16", check_range=True, ln_fold=False):
        """``check_range`` (fp16c): warn at load time when a projection weight exceeds the fp8 correction operand's range (one host
        sync per matrix; off for the trainable encoder's per-step operand rebuilds).
        ``ln_fold`` (fp16c, hidden and intermediate sizes multiples of 256): prepare the weights for ``ruart_bert_forward_folded`` -
        every LayerNorm of the encoder (modeling.py:164-168) folded into the projection that reads it: the QKV weights of layer l >= 1
        become W' = Wqkv diag(ln2_g[l-1]) 2^-s (s >= 0: keeps |W'| inside the fp8 companion's range), their bias d = b + Wqkv ln2_b[l-1],
        the intermediate dense the same with the layer's attention-output LayerNorm; c = row sums of W' (csrc/gemm_corr.hip, CorrFold)."""
        self._check_range = check_range
        # (round 6: the plain 16-bit modes fold too - ruart_gemm_16_nt_fold; whole-sequence encoding only, the pooling kernels read
        #  pre-LayerNorm rows in fp32)
        self.ln_fold = bool(ln_fold) and dtype in ("fp16c", "fp16", "bf16") and cfg["hidden_size"] % 256 == 0 \
            and cfg["intermediate_size"] % 256 == 0 and cfg["hidden_size"] <= 1024
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        self.precision = dtype
        self.dtype = hip.PRECISION[dtype]              # storage type of the layer outputs
        self.tdtype = hip.TORCH_DTYPE[self.dtype]
        self.corr8 = dtype == "fp16c"                  # f16 MFMA operands + fp8 correction (csrc/gemm_corr.hip)
        wdtype = torch.float16 if self.corr8 else self.tdtype
        H = cfg["hidden_size"]
        nh = cfg["num_attention_heads"]
        if H % nh or H // nh != 64:
            raise ValueError("ruart_amd BERT kernels are built for head_dim 64 (got hidden %d, heads %d)" % (H, nh))
        pre = "bert." if any(k.startswith("bert.") for k in state) else ""

        def f32(name):
            t = state[pre + name]
            if isinstance(t, torch.Tensor) and t.is_cuda:      # live parameters of the trainable encoder: no host round trip
                return t.detach().to(self.device, torch.float32).contiguous()
            return torch.as_tensor(_np(t), dtype=torch.float32).to(self.device).contiguous()

        def gemm_w(t):
            return t.to(wdtype).contiguous()

        def gemm_w8(t):
            """(out, 2 in) e4m3 companion of a weight matrix: [fp8(f16(w) 2^7) | fp8((w - f16(w)) 2^18)] (common.h)."""
            hi = t.to(torch.float16).to(torch.float32)
            if self._check_range:
                self._w_absmax = max(getattr(self, "_w_absmax", 0.0), float(t.abs().max()))      # load time: one sync per matrix
            _, _, sw_hi, sw_lo = hip.f16c_shifts()
            pair = torch.cat([hi * float(2.0 ** sw_hi), (t - hi) * float(2.0 ** sw_lo)], 1).clamp_(-448.0, 448.0)
            return pair.to(torch.float8_e4m3fn).view(torch.uint8).contiguous()

        e = "embeddings."
        self.word_emb = f32(e + "word_embeddings.weight")
        self.pos_emb = f32(e + "position_embeddings.weight")
        self.type_emb = f32(e + "token_type_embeddings.weight")
        self.emb_ln_g = f32(e + "LayerNorm.gamma")
        self.emb_ln_b = f32(e + "LayerNorm.beta")
        keys = ["w_qkv", "b_qkv", "w_ao", "b_ao", "ln1_g", "ln1_b", "w_ff1", "b_ff1", "w_ff2", "b_ff2", "ln2_g", "ln2_b"]
        keys8 = ["w8_qkv", "w8_ao", "w8_ff1", "w8_ff2"] if self.corr8 else []
        keysf = ["fold_c_qkv", "fold_c_ff1"] if self.ln_fold else []
        self._w_absmax = 0.0
        self.layers = {k: [] for k in keys + keys8 + keysf}
        self.fold_scales = {"qkv": [], "ff1": []}
        scale = 1.0 / 8.0                      # 1/sqrt(head_dim = 64): a power of two, exact in fp32 and bf16

        def folded(w, b, g, be):
            """(W' = w diag(g) 2^-s, d = b + w be, c = row sums of W', 2^s) - sums in fp64"""
            wf = w * g[None, :]
            amax = float(wf.abs().max())
            sh = 0
            while self.corr8 and amax * 2.0 ** -sh >= 3.4:     # e4m3(f16(w) 2^7) saturates at 3.5 (csrc/common.h); plain 16-bit: no companion, s = 0
                sh += 1
            wf = wf * float(2.0 ** -sh)
            d = (b.double() + w.double() @ be.double()).float().contiguous()
            # c = row sums of the W' the matrix cores multiply: the 16-bit rounding of W' in the plain modes (mu c then cancels the mean's
            # share of the product exactly), W' itself in fp16c (its f16 + e4m3 representation carries ~16 bits)
            c = (wf if self.corr8 else gemm_w(wf)).double().sum(1).float().contiguous()
            return wf, d, c, float(2.0 ** sh)

        prev_g = prev_b = None
        for l in range(cfg["num_hidden_layers"]):
            p = "encoder.layer.%d." % l
            wq, wk, wv = f32(p + "attention.self.query.weight"), f32(p + "attention.self.key.weight"), f32(p + "attention.self.value.weight")
            bq, bk, bv = f32(p + "attention.self.query.bias"), f32(p + "attention.self.key.bias"), f32(p + "attention.self.value.bias")
            L = self.layers
            w_qkv, b_qkv = torch.cat([wq * scale, wk, wv], 0), torch.cat([bq * scale, bk, bv], 0).contiguous()
            w_ff1, b_ff1 = f32(p + "intermediate.dense.weight"), f32(p + "intermediate.dense.bias")
            g1, b1 = f32(p + "attention.output.LayerNorm.gamma"), f32(p + "attention.output.LayerNorm.beta")
            g2, b2 = f32(p + "output.LayerNorm.gamma"), f32(p + "output.LayerNorm.beta")
            if self.ln_fold:
                if l > 0:
                    w_qkv, b_qkv, c, sc = folded(w_qkv, b_qkv, prev_g, prev_b)
                else:
                    c, sc = torch.zeros(w_qkv.shape[0], dtype=torch.float32, device=self.device), 1.0
                L["fold_c_qkv"].append(c)
                self.fold_scales["qkv"].append(sc)
                w_ff1, b_ff1, c, sc = folded(w_ff1, b_ff1, g1, b1)
                L["fold_c_ff1"].append(c)
                self.fold_scales["ff1"].append(sc)
                prev_g, prev_b = g2, b2
            if self.corr8:
                L["w8_qkv"].append(gemm_w8(w_qkv))
                L["w8_ao"].append(gemm_w8(f32(p + "attention.output.dense.weight")))
                L["w8_ff1"].append(gemm_w8(w_ff1))
                L["w8_ff2"].append(gemm_w8(f32(p + "output.dense.weight")))
            L["w_qkv"].append(gemm_w(w_qkv))
            L["b_qkv"].append(b_qkv)
            L["w_ao"].append(gemm_w(f32(p + "attention.output.dense.weight")))
            L["b_ao"].append(f32(p + "attention.output.dense.bias"))
            L["ln1_g"].append(g1)
            L["ln1_b"].append(b1)
            L["w_ff1"].append(gemm_w(w_ff1))
            L["b_ff1"].append(b_ff1)
            L["w_ff2"].append(gemm_w(f32(p + "output.dense.weight")))
            L["b_ff2"].append(f32(p + "output.dense.bias"))
            L["ln2_g"].append(g2)
            L["ln2_b"].append(b2)
        nl = cfg["num_hidden_layers"]
        if self.ln_fold:       # the output LayerNorms' parameters as two tables: the pooling kernel normalises the rows it reads
            self.ln2_g_all = torch.stack(self.layers["ln2_g"]).contiguous()
            self.ln2_b_all = torch.stack(self.layers["ln2_b"]).contiguous()
        self._arrays = {}
        m = hip.BertModelC()
        m.hidden, m.n_heads, m.n_layers, m.intermediate = H, nh, nl, cfg["intermediate_size"]
        m.dtype, m.ln_eps = (hip.DT_F16 if self.corr8 else self.dtype), 1e-12
        m.f32_gemm = 1 if dtype == "x3" else 0
        m.corr8 = 1 if self.corr8 else 0
        if self.corr8 and self._w_absmax >= 3.5:
            # e4m3(f16(w) 2^7) saturates at |w| = 3.5 (csrc/common.h): beyond it the correction of the ACTIVATION's rounding residual is
            # taken against a clipped weight - the product falls back towards plain-f16 accuracy for those weights, nothing breaks
            import warnings
            warnings.warn("fp16c: a BERT projection weight reaches |w| = %.2f; the fp8 correction operand saturates at 3.5 "
                          "(accuracy of the affected products degrades towards the plain f16 mode)" % self._w_absmax)
        for k in ("word_emb", "pos_emb", "type_emb", "emb_ln_g", "emb_ln_b"):
            setattr(m, k, getattr(self, k).data_ptr())
        for k in keys + keys8 + keysf:
            arr = (c_void_p * nl)(*[t.data_ptr() for t in self.layers[k]])
            self._arrays[k] = arr
            setattr(m, k, ctypes.cast(arr, ctypes.POINTER(c_void_p)))
        if self.ln_fold:
            m.ln_fold = 1
            for k in ("qkv", "ff1"):
                arr = (ctypes.c_float * nl)(*self.fold_scales[k])
                self._arrays["fold_s_" + k] = arr
                setattr(m, "fold_s_" + k, ctypes.cast(arr, ctypes.POINTER(ctypes.c_float)))
        self.c_model = m

    @property
    def hidden(self):
        return self.cfg["hidden_size"]

    @property
    def n_layers(self):
        return self.cfg["num_hidden_layers"]


class PackedTokens:
    """One step's word pieces as a packed stream + the descriptors the kernels need.

    ``groups``: list of (ids (N, L) int64, mask (N, L) bool) CPU tensors - e.g. question, OCR items, object items.
    ``pack=True`` keeps only mask==1 positions; ``pack=False`` keeps every position and turns the mask into the
    reference's additive -10000 key bias (exact reference semantics for arbitrary masks)."""

    def __init__(self, groups, device=None, pack=True, mfma_long=True, window=512, max_positions=None, dedup=False):
        """Host part (numpy only; picklable, so it can run in DataLoader workers) + ``bind(device)`` when a device is given.
        ``dedup`` (frozen encoder only): rows of a group with identical kept (position, id) sequences are encoded ONCE - every
        duplicate's ``group_index`` points at the first occurrence's packed rows, so its word spans pool the same bits.  The
        encoder is a deterministic function of the sequence (no dropout when frozen, block-diagonal attention), which makes this
        bit-neutral; the ``<OCR>`` / ``<OD>`` sentinel items alone are 2 B identical sequences per batch, repeated scene words and
        object classes come on top.
        ``window``: the reference cuts a row longer than 512 word pieces into independent 512-windows, each encoded as its own
        sequence with positions restarting at 0, and concatenates the outputs (Models/Bert/Bert.py:18, 96-99, 133-138): here
        every (row, window) with at least one kept piece is a sequence of the packed stream; a row's pieces stay contiguous, so
        a word whose pieces straddle a window boundary still pools over one contiguous span.  ``max_positions``: size of the
        position table (checked, instead of reading past its end on the device)."""
        ids_l, pos_l, len_l, bias_l = [], [], [], []
        self.group_index = []            # per group: (N, L) int32 packed index of each kept position, -1 if dropped
        base = 0
        for ids, mask in groups:
            ids = _np(ids).astype(np.int64)
            mask = _np(mask).astype(bool)
            N, L = ids.shape
            if max_positions is not None and min(L, window) > max_positions:
                raise ValueError("BERT input rows of %d word pieces need %d position embeddings, the checkpoint has %d"
                                 % (L, min(L, window), max_positions))
            keep = mask if pack else np.ones_like(mask)
            if (keep.sum(1) == 0).any():
                raise ValueError("a BERT input row has no attendable token")
            nw = (L + window - 1) // window
            canon = None
            if dedup and nw == 1 and N > 1:
                key = np.ascontiguousarray(np.where(keep, ids, -1))
                _, first, inv = np.unique(key.view(np.dtype((np.void, key.dtype.itemsize * L))).reshape(-1), return_index=True, return_inverse=True)
                canon = first[inv.reshape(-1)]                       # row -> its first identical row
                if (canon == np.arange(N)).all():
                    canon = None
                else:
                    keep = keep & (canon == np.arange(N))[:, None]   # duplicates contribute no packed rows
            if nw == 1:
                lens = keep.sum(1).astype(np.int64)
                lens = lens[lens > 0] if canon is not None else lens
            else:                        # (row, window) sequences in row-major order, empty windows dropped
                padded = np.zeros((N, nw * window), dtype=bool)
                padded[:, :L] = keep
                lens = padded.reshape(N, nw, window).sum(2).reshape(-1).astype(np.int64)
                lens = lens[lens > 0]
            flat = keep.reshape(-1)
            idx = np.full(N * L, -1, dtype=np.int64)
            idx[flat] = base + np.arange(int(flat.sum()))
            idx = idx.reshape(N, L)
            self.group_index.append(idx if canon is None else idx[canon])
            ids_l.append(ids.reshape(-1)[flat])
            pos_l.append(np.broadcast_to(np.arange(L) % window, (N, L)).reshape(-1)[flat])
            len_l.append(lens)
            if not pack:
                bias_l.append(np.where(mask.reshape(-1), 0.0, -10000.0).astype(np.float32))
            base += int(flat.sum())
        lens = np.concatenate(len_l)
        T = int(lens.sum())
        Tp = (T + ROW_PAD - 1) // ROW_PAD * ROW_PAD
        cu = np.zeros(len(lens) + 1, dtype=np.int64)
        np.cumsum(lens, out=cu[1:])
        seq_of = np.repeat(np.arange(len(lens)), lens)
        blk, lblk = self._plan_blocks(lens, cu, mfma_long)
        nb, nlb = blk.shape[1], lblk.shape[1]
        # one int32 host buffer -> one H2D copy
        host = np.zeros(2 * Tp + 2 * T + 4 * nb + 4 * nlb, dtype=np.int32)
        host[0:T] = np.concatenate(ids_l)
        host[Tp:Tp + T] = np.concatenate(pos_l)
        host[2 * Tp:2 * Tp + T] = cu[seq_of]
        host[2 * Tp + T:2 * Tp + 2 * T] = cu[seq_of + 1]
        host[2 * Tp + 2 * T:2 * Tp + 2 * T + 4 * nb] = blk.reshape(-1)
        host[2 * Tp + 2 * T + 4 * nb:] = lblk.reshape(-1)
        self.T, self.Tp, self.n_blocks, self.n_long_blocks = T, Tp, nb, nlb
        self.n_seq = len(lens)
        self.max_len = int(lens.max())
        self.max_pos = int(host[Tp:Tp + T].max()) + 1 if T else 0      # position-table rows this stream reads
        self.sum_len_sq = float((lens.astype(np.float64) ** 2).sum())
        self.host = host
        self.bias_host = np.concatenate(bias_l) if not pack else None
        self.buf = None
        if device is not None:
            self.bind(device)

    _DEVICE_FIELDS = ("buf", "ids", "pos", "tok_lo", "tok_hi", "blk", "lblk", "key_bias", "c_batch", "_layers", "_event", "_set", "_train_plan",
                      "_emb_sorts", "last_rows")

    def prepare_embedding_sorts(self):
        """Host part of the trainable encoder's embedding gradients (numpy; BatchIndex calls it in the DataLoader worker when the conf has
        no LOCK_BERT): the word-piece ids and the positions of the packed stream, each sorted by batch._sort_ids - the input of
        ops.embedding_grad, which adds the gradient rows of one table row in a fixed order (torch's index_add_ uses atomics: run-to-run
        different bits)."""
        if getattr(self, "_emb_sorts_host", None) is None:
            from .batch import _sort_ids
            T, Tp = self.T, self.Tp
            sorts = [_sort_ids(ids) for ids in (self.host[0:T], self.host[Tp:Tp + T])]
            self._emb_sorts_host = (np.concatenate([a for srt in sorts for a in srt]).astype(np.int32), [[len(a) for a in srt] for srt in sorts])
        return self._emb_sorts_host

    def train_plan(self, device):
        """Attention plan of the trainable 16-bit encoder (bert_train16.py), from this stream's query blocks: ``win`` = (q0, q1) of the
        windows of whole short sequences (ruart_attn_train_fwd / _bwd) and ``chunks`` = (q0, q1, k0, k1, first) of the <= 64-token
        chunks of every sequence longer than one window (ruart_attn_train_fwd_long / _bwd_long: chunks of a sequence are consecutive,
        start at its first token in steps of 64; ``first`` = index of the sequence's first chunk).  int32 device tensors, cached."""
        plan = getattr(self, "_train_plan", None)
        if plan is None or plan["device"] != device:
            T, Tp, nb, nlb = self.T, self.Tp, self.n_blocks, self.n_long_blocks
            o = 2 * Tp + 2 * T
            blk = self.host[o:o + 4 * nb].reshape(4, nb)
            lblk = self.host[o + 4 * nb:o + 4 * nb + 4 * nlb].reshape(4, nlb)
            own = (blk[0] == blk[2]) & (blk[1] == blk[3]) & ((blk[1] - blk[0]) <= 64)
            win = blk[:2, own]
            ch = [tuple(int(v) for v in blk[:, i]) for i in np.nonzero(~own)[0]]
            for i in range(nlb):                                   # 128-query blocks of the frozen path's long-sequence kernel
                q0, q1, k0, k1 = (int(v) for v in lblk[:, i])
                ch += [(a, min(a + 64, q1), k0, k1) for a in range(q0, q1, 64)]
            ch.sort()
            ok = all(q1 - q0 <= 64 and (q0 - k0) % 64 == 0 and k0 <= q0 < q1 <= k1 for q0, q1, k0, k1 in ch)
            idx = {c[0]: i for i, c in enumerate(ch)}
            ok = ok and all(k0 in idx and (q0 - k0) // 64 == i - idx[k0] for i, (q0, q1, k0, k1) in enumerate(ch))
            arr = np.array([c + (idx.get(c[2], 0),) for c in ch], dtype=np.int32).reshape(-1, 5).T.copy()
            both = torch.from_numpy(np.concatenate([win.reshape(-1), arr.reshape(-1)]).astype(np.int32)).to(device)
            nw, nc = win.shape[1], arr.shape[1]
            plan = {"device": device, "ok": bool(ok), "n_win": nw, "n_chunks": nc,
                    "win": [both[i * nw:(i + 1) * nw] for i in range(2)],
                    "chunks": [both[2 * nw + i * nc:2 * nw + (i + 1) * nc] for i in range(5)]}
            self._train_plan = plan
        return plan

    def embedding_sorts(self, device):
        """[sort of the word-piece ids, sort of the positions] as tuples of int32 device tensors (one H2D copy, cached)."""
        if getattr(self, "_emb_sorts", None) is None:
            flat, sizes = self.prepare_embedding_sorts()
            dev = torch.from_numpy(flat).to(device, non_blocking=True)
            out, o = [], 0
            for group in sizes:
                views = []
                for n in group:
                    views.append(dev[o:o + n])
                    o += n
                out.append(tuple(views))
            self._emb_sorts = out
        return self._emb_sorts

    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k not in self._DEVICE_FIELDS}

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.buf = None

    def bind(self, device, host_tensor=None):
        """One H2D copy of the int32 descriptor buffer (``host_tensor``: the same buffer as a - possibly pinned - torch tensor);
        builds the C struct the encoder entry point takes."""
        T, Tp, nb, nlb = self.T, self.Tp, self.n_blocks, self.n_long_blocks
        dev = (host_tensor if host_tensor is not None else torch.from_numpy(self.host)).to(device, non_blocking=True)
        self.buf = dev
        o = 0
        self.ids = dev[o:o + Tp]; o += Tp
        self.pos = dev[o:o + Tp]; o += Tp
        self.tok_lo = dev[o:o + T]; o += T
        self.tok_hi = dev[o:o + T]; o += T
        self.blk = [dev[o + i * nb:o + (i + 1) * nb] for i in range(4)]
        o += 4 * nb
        self.lblk = [dev[o + i * nlb:o + (i + 1) * nlb] for i in range(4)]
        self.key_bias = None
        if self.bias_host is not None:
            self.key_bias = torch.from_numpy(self.bias_host).to(device)
        b = hip.BertBatchC()
        b.n_tokens, b.n_rows, b.n_blocks = T, Tp, nb
        b.ids, b.pos_ids = self.ids.data_ptr(), self.pos.data_ptr()
        b.blk_q0, b.blk_q1, b.blk_k0, b.blk_k1 = [t.data_ptr() for t in self.blk]
        b.tok_lo, b.tok_hi = self.tok_lo.data_ptr(), self.tok_hi.data_ptr()
        b.key_bias = self.key_bias.data_ptr() if self.key_bias is not None else None
        b.n_long_blocks = nlb
        b.lblk_q0, b.lblk_q1, b.lblk_k0, b.lblk_k1 = [t.data_ptr() if nlb else None for t in self.lblk]
        b.n_last_rows, b.last_rows = 0, None
        self.c_batch = b
        return self

    def set_last_rows(self, rows_dev):
        """``rows_dev``: ascending int32 device vector of the packed rows some word span pools (batch.BatchIndex).  The frozen
        encoder then computes its LAST layer on those rows only and leaves that layer's output compacted
        (ruart_bert_batch.last_rows); the pooling kernels take the compacted span starts beside the ordinary ones."""
        self.last_rows = rows_dev
        self.c_batch.n_last_rows = int(rows_dev.numel())
        self.c_batch.last_rows = rows_dev.data_ptr()

    @staticmethod
    def _plan_blocks(lens, cu, mfma_long=True):
        """Query blocks.  Short sequences are packed whole into windows of <= 64 tokens (block-diagonal mask inside the window).
        A sequence longer than 64 is split into chunks that each see the whole sequence as keys: 128-query chunks for the MFMA
        long-sequence kernel (``mfma_long``), or 64-query chunks appended to the short list in fp32 mode (VALU kernel).
        Returns two int arrays (4, n): q0, q1, k0, k1."""
        S = len(lens)
        short, long_ = [], []
        s = 0
        while s < S:
            if lens[s] > 64:
                tgt, step = (long_, LONG_BLOCK) if mfma_long else (short, 64)
                for q0 in range(int(cu[s]), int(cu[s + 1]), step):
                    tgt.append((q0, min(q0 + step, int(cu[s + 1])), int(cu[s]), int(cu[s + 1])))
                s += 1
                continue
            e = int(np.searchsorted(cu, cu[s] + 64, side="right")) - 1      # last boundary within 64 tokens
            while e > s + 1 and (lens[s:e] > 64).any():                      # never swallow a long sequence
                e -= 1
            e = max(e, s + 1)
            short.append((int(cu[s]), int(cu[e]), int(cu[s]), int(cu[e])))
            s = e
        def arr(x):
            return np.array(x, dtype=np.int32).reshape(-1, 4).T.copy()
        return arr(short), arr(long_)


class _Buffers:
    """Grow-only device buffers for the encoder (layer outputs + workspace)."""

    def __init__(self):
        self.layers = None
        self.ws = None
        self.stats = None

    def get(self, w, Tp):
        lib = hip.load()
        es = 4 if w.dtype == hip.DT_F32 else 2
        need_l = w.n_layers * Tp * w.hidden * es                      # bytes
        if self.layers is None or self.layers.numel() < need_l or self.layers.device != w.device:
            self.layers = torch.zeros(need_l, dtype=torch.uint8, device=w.device)
        fold = getattr(w, "ln_fold", False)
        need_w = int((lib.ruart_bert_workspace_bytes_folded if fold else lib.ruart_bert_workspace_bytes)(ctypes.byref(w.c_model), Tp))
        if self.ws is None or self.ws.numel() < need_w or self.ws.device != w.device:
            self.ws = torch.zeros(need_w, dtype=torch.uint8, device=w.device)
        if fold:
            need_s = w.n_layers * Tp * 2
            if self.stats is None or self.stats.numel() < need_s or self.stats.device != w.device:
                self.stats = torch.zeros(need_s, dtype=torch.float32, device=w.device)
        return self.layers[:need_l].view(w.tdtype).view(w.n_layers, Tp, w.hidden), self.ws, need_w

    def stats_for(self, w, Tp):
        return self.stats[:w.n_layers * Tp * 2].view(w.n_layers, Tp, 2)


_buffers = _Buffers()


def bert_encode(weights, packed, buffers=None):
    """Run the encoder; returns all layer outputs as one (n_layers, Tp, H) tensor in the weights' dtype.
    The tensor aliases a reusable buffer: consume it before the next call."""
    lib = hip.load()
    if getattr(packed, "max_pos", 0) > weights.cfg["max_position_embeddings"]:
        raise ValueError("the packed stream addresses %d position embeddings, the checkpoint has %d"
                         % (packed.max_pos, weights.cfg["max_position_embeddings"]))
    buffers = buffers or _buffers
    layers, ws, ws_bytes = buffers.get(weights, packed.Tp)
    if getattr(weights, "ln_fold", False):
        # LayerNorm-folded pass: `layers` holds the PRE-LayerNorm rows of every layer's output, `layers._ln` = ((mu, rstd) per row, the
        # output LayerNorms' gamma / beta tables) what turns them into the layer outputs - on the fly in the pooling kernel
        # (_PoolMix), or materialised by layer_outputs() for tests
        stats = buffers.stats_for(weights, packed.Tp)
        rc = lib.ruart_bert_forward_folded(ctypes.byref(weights.c_model), ctypes.byref(packed.c_batch), hip.ptr(layers), hip.ptr(stats),
                                           hip.ptr(ws), ws_bytes, hip.stream_ptr())
        hip.check(rc, "ruart_bert_forward_folded")
        layers._ln = (stats, weights.ln2_g_all, weights.ln2_b_all)
        return layers
    rc = lib.ruart_bert_forward(ctypes.byref(weights.c_model), ctypes.byref(packed.c_batch), hip.ptr(layers), hip.ptr(ws),
                                ws_bytes, hip.stream_ptr())
    hip.check(rc, "ruart_bert_forward")
    return layers


def layer_outputs(layers, n_last=None):
    """The encoder's layer outputs as the reference keeps them (modeling.py:326-334) from what ``bert_encode`` returned: the tensor itself,
    or - after a LayerNorm-folded pass - (y - mu) rstd gamma + beta of its pre-LayerNorm rows (a new tensor; tests and tools, not the
    product path, which normalises inside the pooling kernel)."""
    ln = getattr(layers, "_ln", None)
    if ln is None:
        return layers
    stats, g, b = ln
    return (layers - stats[:, :, 0:1]) * stats[:, :, 1:2] * g[:, None, :] + b[:, None, :]


from .ops import _ABL_SKIP      # timing diagnostics (refused without RUART_DIAGNOSTICS=1, ops.py)


class _PoolMix(torch.autograd.Function):
    """out[dst_row[w]] = sum_l layer_w[l] * mean(layer_l[span]) ; gradient only w.r.t. layer_w (BERT is locked,
    Models/SDNet.py:91-94)."""

    @staticmethod
    def forward(ctx, layer_w, layers, span_start, span_len, dst_row, n_rows, dtype_code, span_start_last=None, ln_stats=None, ln_g=None,
                ln_b=None):
        """``span_start_last``: the spans' first rows in the LAST layer's matrix when the encoder left it compacted
        (PackedTokens.set_last_rows); None = as in every other layer.  ``ln_stats`` / ``ln_g`` / ``ln_b``: `layers` holds the
        pre-LayerNorm rows of a folded pass (bert_encode) and the kernel normalises what it reads."""
        lib = hip.load()
        NL, Tp, H = layers.shape
        W = span_start.numel()
        out = torch.zeros(n_rows, H, dtype=torch.float32, device=layers.device)
        lw = layer_w.detach().to(torch.float32).contiguous()
        if W > 0 and "pool" not in _ABL_SKIP:          # (timing diagnostics only, see ops._ABL_SKIP: empty in every product run)
            if ln_stats is not None:
                rc = lib.ruart_bert_pool_mix_ln(hip.ptr(layers), Tp * H, H, NL, hip.ptr(ln_stats), Tp, hip.ptr(ln_g), hip.ptr(ln_b),
                                                hip.ptr(span_start), hip.ptr(span_start_last), hip.ptr(span_len), hip.ptr(dst_row), hip.ptr(lw),
                                                hip.ptr(out), H, W, H, hip.stream_ptr())
            else:
                rc = lib.ruart_bert_pool_mix(hip.ptr(layers), Tp * H, H, dtype_code, NL, hip.ptr(span_start), hip.ptr(span_start_last), hip.ptr(span_len),
                                             hip.ptr(dst_row), hip.ptr(lw), hip.ptr(out), H, W, H, hip.stream_ptr())
            hip.check(rc, "ruart_bert_pool_mix")
        ctx.save_for_backward(layers, span_start, span_len, dst_row, span_start_last, ln_stats, ln_g, ln_b)
        ctx.dtype_code = dtype_code
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = hip.load()
        layers, span_start, span_len, dst_row, span_start_last, ln_stats, ln_g, ln_b = ctx.saved_tensors
        NL, Tp, H = layers.shape
        W = span_start.numel()
        g = torch.zeros(NL, dtype=torch.float32, device=layers.device)
        if W > 0 and "pool" not in _ABL_SKIP:
            grad_out = grad_out.contiguous()
            partial = torch.empty(W * NL, dtype=torch.float32, device=layers.device)
            if ln_stats is not None:
                rc = lib.ruart_bert_pool_mix_ln_bwd(hip.ptr(layers), Tp * H, H, NL, hip.ptr(ln_stats), Tp, hip.ptr(ln_g), hip.ptr(ln_b),
                                                    hip.ptr(span_start), hip.ptr(span_start_last), hip.ptr(span_len), hip.ptr(dst_row),
                                                    hip.ptr(grad_out), H, hip.ptr(partial), hip.ptr(g), W, H, hip.stream_ptr())
            else:
                rc = lib.ruart_bert_pool_mix_bwd(hip.ptr(layers), Tp * H, H, ctx.dtype_code, NL, hip.ptr(span_start), hip.ptr(span_start_last), hip.ptr(span_len),
                                                 hip.ptr(dst_row), hip.ptr(grad_out), H, hip.ptr(partial), hip.ptr(g), W, H,
                                                 hip.stream_ptr())
            hip.check(rc, "ruart_bert_pool_mix_bwd")
        return g, None, None, None, None, None, None, None, None, None, None


def word_spans(packed, group, offsets, word_mask, offsets_arr=None):
    """Descriptors for pooling one group: for every (row n, word j) with word_mask[n, j] and a non-empty span
    [st, ed) (Models/Bert/Bert.py:155-165) -> packed start, length, destination row n * Lw + j.
    ``offsets`` is the reference's python list [N][words][2]; ``offsets_arr`` an equivalent (N, Lw, 2) int array."""
    wm = _np(word_mask).astype(bool)
    N, Lw = wm.shape
    if offsets_arr is None:
        offsets_arr = np.zeros((N, Lw, 2), dtype=np.int64)
        for n, row in enumerate(offsets):
            if len(row) and not isinstance(row[0], (list, tuple)):
                continue                                   # the reference's "[1, 1]" for an empty word list
            k = min(len(row), Lw)
            if k:
                offsets_arr[n, :k] = np.asarray(row[:k], dtype=np.int64)
    offsets_arr = _np(offsets_arr)
    st, ed = offsets_arr[..., 0], offsets_arr[..., 1]
    sel = wm & (ed > st)
    n_idx, j_idx = np.nonzero(sel)
    gidx = packed.group_index[group]
    s, e = st[sel], ed[sel]
    start = gidx[n_idx, s]
    last = gidx[n_idx, e - 1]
    if (start < 0).any() or (last - start != e - s - 1).any():
        raise ValueError("a word's piece span touches a masked BERT position; encode this batch with pack=False")
    return start.astype(np.int32), (e - s).astype(np.int32), (n_idx * Lw + j_idx).astype(np.int32), N * Lw


class Bert(nn.Module):
    """Drop-in for the reference's ``Bert`` module (Models/Bert/Bert.py:14-45).  ``opt`` keys used:
    BERT_LARGE, BERT_model_file / BERT_large_model_file, datadir, plus ruart extensions
    ``bert_precision`` ('fp16c' default - ruart_amd.DEFAULT_PRECISION - | 'x3' | 'fp32' | 'fp16' | 'bf16') and ``bert_state`` / ``bert_config`` to pass weights in memory."""

    def __init__(self, opt, device=None):
        super().__init__()
        import json
        import os
        self.opt = opt
        self.linear_combine = "BERT_LINEAR_COMBINE" in opt
        self.bert_dim, self.bert_layer = (1024, 24) if "BERT_LARGE" in opt else (768, 12)
        self._device = torch.device(device if device is not None else "cuda")
        if "bert_state" in opt:
            state, cfg = opt["bert_state"], opt["bert_config"]
        else:
            key = "BERT_large_model_file" if "BERT_LARGE" in opt else "BERT_model_file"
            d = os.path.join(opt.get("datadir", ""), opt[key])
            with open(os.path.join(d, "bert_config.json")) as f:
                cfg = json.load(f)
            state = torch.load(os.path.join(d, "pytorch_model.bin"), map_location="cpu")
        if cfg["hidden_size"] != self.bert_dim or cfg["num_hidden_layers"] != self.bert_layer:
            raise ValueError("BERT checkpoint is %dx%d, conf expects %dx%d" % (cfg["num_hidden_layers"], cfg["hidden_size"],
                                                                              self.bert_layer, self.bert_dim))
        from . import precision_of
        precision = precision_of(opt, cfg["hidden_size"] if cfg["intermediate_size"] % 256 == 0 else 1)
        # LayerNorms folded into the projections (fp16c; csrc/gemm_corr.hip, CorrFold): on unless opt['bert_ln_fold'] / RUART_LN_FOLD say 0
        fold = bool(int(opt.get("bert_ln_fold", os.environ.get("RUART_LN_FOLD", 1))))
        if int(opt.get("bert_tail_cus", os.environ.get("RUART_TAIL_CUS", 0)) or 0) > 0:
            fold = False                     # the tail split exists in the unfolded pass only (ruart_bert_forward_folded refuses it)
        if precision != "fp16c":
            fold = False                     # the sub-word pooling kernels read pre-LayerNorm rows in fp32 only: the plain 16-bit folded pass
                                             # (round 6) serves whole-sequence encoding (bert_encode, bench.py --mode bert512), not this class
        self.weights = BertEncoderWeights(state, cfg, self._device, precision, ln_fold=fold)
        self.bert_model = None               # trainable fp32 encoder (bert_train.BertModelTrainable) once ``unlock`` is called
        self._source = (state, cfg)          # kept until SDNet has decided between the frozen and the trainable path
        self.pack = not opt.get("bert_no_pack", False)
        # CUs the run-ahead encoder pass may occupy.  In the fp16c mode the pass (19 ms) outlasts the trunk's step (15 ms), and a trunk
        # kernel otherwise waits for a GEMM workgroup (which owns a whole CU's registers and LDS for ~45 us) to retire: keeping 16
        # CUs out of the encoder's reach took the step from 26.7 to 25.8 ms.  The plain 16-bit modes gain nothing (round 1).
        # Round 5: 224 - on the bench batch the smallest mask with as few GEMM tile rounds as 240 (plan_prefetch_cus below; 22.9 ms
        # against 23.5), and the better one on a 10 % smaller batch too (21.4 against 22.3 at that batch's own plan, 232).  'auto' plans
        # ONCE, from the first training batch, and keeps that stream: a second CU-masked stream in the same process lands on a hardware
        # queue pipe one of the step's four streams already uses and the two run one after the other (27-29 ms steps,
        # tools/r05_mask_switch.py, profiles/HISTORY.md round 5 (9)).
        self._opt_prefetch_cus = opt.get("bert_prefetch_cus", 224 if precision == "fp16c" else 0)
        self._init_pipeline()
        # Tail split of the encoder GEMMs (csrc/gemm_corr.hip, gemm.hip; opt['bert_tail_cus'] = the CU count the split is planned for, one
        # value per model so that every pass computes the same bits).  OFF by default: measured on the bench batch (round 4, DESIGN.md
        # section 5) it pays on the long-K output dense alone (-8 %) and the whole pass does not get shorter - a nearly empty last round
        # of tiles runs at a higher clock and with the memory system to itself, it costs far less than a full tile time.
        tail = opt.get("bert_tail_cus", os.environ.get("RUART_TAIL_CUS", 0))
        self.weights.c_model.tail_cus = max(0, int(tail))

    def unlock(self):
        """Confs without LOCK_BERT (Models/SDNet.py:88-94): the encoder's parameters become fp32 ``nn.Parameter``s under the
        reference's names (``Bert.bert_model.*``) and every pass goes through the autograd-capable path of bert_train.py."""
        from .bert_train import BertModelTrainable
        if not self.pack:
            raise NotImplementedError("the trainable encoder runs on the packed token stream (bert_no_pack is for the frozen path)")
        state, cfg = self._source
        mode = str(self.opt.get("bert_train_gemm", "x3"))
        if mode == "16" and cfg["hidden_size"] % 256 == 0 and cfg["intermediate_size"] % 256 == 0:
            # the whole encoder, forward and backward, as one autograd Function over 16-bit kernels (bert_train16.py)
            from .bert_train16 import BertModelTrainable16
            self.bert_model = BertModelTrainable16(state, cfg, self._device)
            # passes without active dropout (evaluation, parity tests) on the frozen path's fp16c kernels: 1e-3 with the encoder unlocked
            self.bert_model.accurate_forward = bool(self.opt.get("bert_train_accurate_fwd", True))
        else:
            # "x3": fp32-class graph (pins parity); "16gemm": the same graph with the two row-parallel products of every projection on
            # the frozen path's 16-bit MFMA GEMM
            self.bert_model = BertModelTrainable(state, cfg, self._device, gemm="16" if mode in ("16", "16gemm") else "x3")
        self._source = None

    def lock(self):
        """Frozen encoder (``LOCK_BERT``, Models/SDNet.py:91-94).  With ``opt['bert_frozen_dropout']`` the training-mode passes
        reproduce what the reference really trains with: ``SDNetTrainer.update`` calls ``network.train()``
        (Models/SDNetTrainer.py:332), which switches the dropout(0.1) layers INSIDE the frozen BERT back on
        (Models/Bert/modeling.py:198, 244, 263, 302) although ``Bert.__init__`` had put it in eval mode.  Off by default: the
        deterministic encoder is what that eval call intended, and it is what lets the pass run one step ahead.  When on, training
        passes go through the fp32 module of bert_train.py (the dropouts sit exactly where the reference has them) without
        gradients; evaluation passes stay on the fast frozen path."""
        if self.opt.get("bert_frozen_dropout"):
            from .bert_train import BertModelTrainable
            state, cfg = self._source
            if (self.weights.precision in ("fp16c", "fp16", "bf16") and cfg["hidden_size"] % 256 == 0
                    and cfg["intermediate_size"] % 256 == 0):
                # the 16-bit modes: the dropout pass runs on the 16-bit training kernels (bert_train16.py, forward only); batches it
                # cannot take (a sequence beyond 64 word pieces) use the parent's fp32-class graph
                from .bert_train16 import BertModelTrainable16
                model = BertModelTrainable16(state, cfg, self._device)
            else:
                model = BertModelTrainable(state, cfg, self._device, gemm="x3")
            for p in model.parameters():
                p.requires_grad_(False)
            self.__dict__["_dropout_model"] = model        # not a registered child: the state dict keeps the reference's keys
        self._source = None

    def _frozen_dropout_active(self):
        return self.training and self.__dict__.get("_dropout_model") is not None

    # -- encoder pipelining across steps ---------------------------------------------------------------------
    # BERT is frozen, so the encoder pass of batch t+1 depends on nothing step t produces.  ``prefetch`` launches it on a
    # separate stream (CU-masked to 240 of the 256 CUs) while step t's SDNet trunk (hundreds of small, latency-bound kernels on
    # three streams of the same priority) runs beside it.  Two buffer sets alternate so the layer outputs
    # step t's backward still reads (pool_mix_bwd) are not overwritten.
    def _init_pipeline(self):
        self._bufsets = [_Buffers(), _Buffers()]
        self._pending = None                 # PackedTokens whose prefetched pass has not been consumed yet
        self._in_use = 1                     # set whose layer outputs the current step's forward/backward reads
        # CUs the prefetch pass may occupy; the rest stay free for the trunk (0 = all, the default).
        cus = os.environ.get("RUART_PREFETCH_CUS", self._opt_prefetch_cus)
        self._pf_cus = cus if cus == "auto" else int(cus)
        self._pf_streams = {}                # CU count (0 = all) -> stream

    def plan_prefetch_cus(self, n_rows):
        """CU mask of the run-ahead pass for a batch of ``n_rows`` packed rows.  A GEMM workgroup (one 256 x 256 tile) owns a CU, so a
        product of t tiles takes ceil(t / cus) rounds whatever is left of the last one: among 208 .. 248 CUs the SMALLEST mask with
        the fewest rounds over the layer's four products (weighted by their K) runs the encoder as fast as the largest one and leaves
        the most CUs to the trunk.  Bench batch (167 row tiles): 224, 232, 240 and 248 all take 31 rounds per layer, 216 takes 32 -
        measured 22.94 / 23.22 / 23.47 / 23.59 ms per step for 224 / 232 / 240 / 248 and 23.78 for 216 (profiles/r05_knob_sweep.log).
        The model counts whole rounds only: a product that fits its rounds with no slack (153 row tiles at 232 CUs: 5.94 / 1.98 / 7.91
        rounds) spills as soon as the trunk holds a few CUs, and that batch ran better at 224 - which is why 224 is the default and
        this plan an option ('auto')."""
        cfg = self.weights.cfg
        H, I = cfg["hidden_size"], cfg["intermediate_size"]
        rt = -(-int(n_rows) // 256)
        prods = ((3 * H // 256, 1.0), (H // 256, 1.0), (I // 256, 1.0), (H // 256, I / float(H)))      # (column tiles, relative K)
        best = None
        for c in range(208, 249, 8):
            cost = sum(w * -(-rt * nt // c) for nt, w in prods)
            if best is None or cost < best[0] - 1e-9:
                best = (cost, c)
        return best[1]

    def prefetch_cus(self, packed=None):
        """CUs the run-ahead pass may use NOW: the configured mask (a number - 224 by default in the fp16c schedule -, or 'auto':
        plan_prefetch_cus of the FIRST training batch, kept from then on) in
        training - the trunk is on the device for 85 % of a training step and needs CUs no GEMM workgroup can take - and all of them
        in evaluation, where the trunk's forward is gone after a quarter of the step and the mask only costs (forward-only steps:
        24.0 ms masked, 20.5 unmasked).  RUART_PREFETCH_CUS_EVAL overrides the evaluation value (experiments)."""
        if self.training:
            if self._pf_cus == "auto":
                if packed is None:
                    return 240
                self._pf_cus = self.plan_prefetch_cus(packed.Tp)      # planned once: the process keeps the stream it starts with
            return self._pf_cus
        return int(os.environ.get("RUART_PREFETCH_CUS_EVAL", 0))

    def prefetch(self, packed, after_stream=None):
        """Encode ``packed`` asynchronously into the buffer set the current step does NOT use; ``layers_for`` returns the
        result.  ``after_stream``: the stream whose already enqueued work (the previous consumers of that set) must finish
        first.  Call it after ``layers_for`` of the current batch."""
        if getattr(packed, "_layers", None) is not None or getattr(self, "bert_model", None) is not None or self._frozen_dropout_active():
            return                           # (a trainable encoder changes every step, a dropout pass is drawn per step: nothing to run ahead)
        dev = self._device
        cus = self.prefetch_cus(packed)
        st = self._pf_streams.get(cus)
        if st is None or st.device != dev:
            epr = int(os.environ.get("RUART_ENCODER_PRIORITY", 0))            # experiments: -1 = a HIGH-priority (unmasked) encoder stream
            st = self._pf_streams[cus] = hip.cu_masked_stream(cus, dev) if cus > 0 else torch.cuda.Stream(device=dev, priority=epr)
        # the set being recycled was last read by the step BEFORE the current one: waiting for the point where the current
        # step picked up its own layers (layers_for) is enough, wherever in the step the prefetch is launched
        if after_stream is not None:
            st.wait_stream(after_stream)
        elif getattr(self, "_pickup_event", None) is not None:
            st.wait_event(self._pickup_event)
        else:
            st.wait_stream(torch.cuda.current_stream(dev))
        # the previous run-ahead pass - consumed or dropped, on this stream or on the other CU mask's - wrote the same two buffer
        # sets: this pass starts behind it (a no-op on the same stream; train -> evaluate switches streams, and a dropped pass
        # may still be running on the masked one)
        if getattr(self, "_last_pf_event", None) is not None:
            st.wait_event(self._last_pf_event)
        if self._pending is not None and self._pending is not packed:
            self._pending._layers = None     # only one pass can be in flight: an older unconsumed one is dropped
        self._pending = packed
        with torch.cuda.stream(st):
            packed._set = self._in_use ^ 1
            packed._layers = bert_encode(self.weights, packed, self._bufsets[packed._set])
            packed._event = self._last_pf_event = st.record_event()

    def close(self, destroy=None):
        """End of a training / evaluation session: the pass that is still running ahead is dropped.  ``destroy``: also destroy the
        CU-masked run-ahead stream (hip.destroy_stream; a later prefetch creates a new one).  True at the end of a process
        (``SDNetTrainer.close(final=True)``: bench.py, the tools, __graft_entry__) - a masked stream alive at static destruction takes a
        process with an HSA tool library loaded (rocprofv3 ...) down in __cxa_finalize.  Otherwise (None) the stream is KEPT for the
        next session: a new masked stream lands on another hardware queue slot, and with an evaluation's streams created in between
        it shared a slot with one of the trunk's streams - every later training step took 27.3 ms instead of 22.1
        (tools/r05_two_sessions.py, profiles/HISTORY.md round 5 (9))."""
        if destroy is None:
            destroy = os.environ.get("RUART_DESTROY_STREAMS") == "1"        # (experiments; the product decides through SDNetTrainer.close(final=))
        if self._pending is not None:
            self._pending._layers = None
            self._pending = None
        if not destroy:
            for st in getattr(self, "_pf_streams", {}).values():
                st.synchronize()                 # (the dropped pass has finished: its buffers may be reused at once)
            return
        streams, self._pf_streams = getattr(self, "_pf_streams", {}), {}
        self._last_pf_event = None
        for st in streams.values():
            hip.destroy_stream(st)

    def layers_for(self, packed):
        """All-layer encoder output of ``packed``: the prefetched one (the current stream waits for it) or computed now."""
        if getattr(self, "bert_model", None) is not None:
            return self.bert_model(packed, training=self.training)
        if self._frozen_dropout_active():
            with torch.no_grad():
                model = self.__dict__["_dropout_model"]
                fast = model.layers_nograd(packed, training=True) if hasattr(model, "layers_nograd") else None
                return fast if fast is not None else model(packed, training=True)
        layers = getattr(packed, "_layers", None)
        if layers is not None:
            torch.cuda.current_stream(self._device).wait_event(packed._event)
            packed._layers = None            # consumed: a later forward on the same batch re-encodes
            self._pending = None
            self._in_use = packed._set
            self._pickup_event = torch.cuda.current_stream(self._device).record_event()
            return layers
        pend = self._pending
        if pend is None or getattr(pend, "_layers", None) is None or pend._set != self._in_use ^ 1:
            self._in_use ^= 1                # else: the other set holds a pass still to be consumed - reuse the current one
        self._pickup_event = torch.cuda.current_stream(self._device).record_event()
        return bert_encode(self.weights, packed, self._bufsets[self._in_use])

    # -- fused path used by ruart_amd.SDNet -------------------------------------------------------------
    def encode(self, groups):
        packed = PackedTokens(groups, self._device, pack=self.pack, mfma_long=self.weights.dtype != hip.DT_F32,
                              max_positions=self.weights.cfg["max_position_embeddings"])
        if getattr(self, "bert_model", None) is not None:
            return packed, self.bert_model(packed, training=self.training)
        return packed, bert_encode(self.weights, packed)

    def pool_mix(self, packed, layers, group, offsets, word_mask, layer_w, offsets_arr=None):
        """(N, Lw, H) fp32 = sum_l layer_w[l] * pooled_l  - Bert.py:149-165 + SDNet.py:573-581 in one kernel."""
        s, n, d, rows = word_spans(packed, group, offsets, word_mask, offsets_arr)
        dev = self._device
        desc = torch.from_numpy(np.concatenate([s, n, d])).to(dev)
        W = len(s)
        if getattr(self, "bert_model", None) is not None:
            from .bert_train import pool_mix
            out = pool_mix(layer_w, layers, desc[:W], desc[W:2 * W], desc[2 * W:], rows)
        else:
            ln = getattr(layers, "_ln", None) or (None, None, None)
            out = _PoolMix.apply(layer_w, layers, desc[:W], desc[W:2 * W], desc[2 * W:], rows, hip.dtype_code(layers), None, *ln)
        N, Lw = word_mask.shape
        return out.view(N, Lw, self.weights.hidden)

    # -- reference-compatible call: list of per-layer pooled tensors (Bert.py:56-90, 130-176) -------------
    def forward(self, x_bert, x_bert_mask, x_bert_offset, x_mask, device=None):
        """List of ``bert_layer`` pooled tensors (N, Lw, H), as the reference returns under BERT_LINEAR_COMBINE.  Rows longer than
        512 pieces are windowed (PackedTokens); ``opt['BERT_MAX_BatchSize']`` splits the rows into chunks that are encoded one after
        the other and concatenated (Bert.py:65-85) - result-neutral, it only bounds the size of one pass."""
        bs = getattr(self, "opt", {}).get("BERT_MAX_BatchSize")
        N = x_bert.shape[0]
        if bs and N > bs:
            parts = []
            for st in range(0, N, int(bs)):
                ed = min(st + int(bs), N)
                off = x_bert_offset[st:ed] if x_bert_offset is not None else None
                parts.append(self.forward(x_bert[st:ed], x_bert_mask[st:ed], off, x_mask[st:ed], device=device))
            return [torch.cat([p[i] for p in parts], 0) for i in range(len(parts[0]))]
        packed, layers = self.encode([(x_bert, x_bert_mask)])
        outs = []
        eye = torch.eye(self.weights.n_layers, device=self._device)
        for l in range(self.weights.n_layers):
            outs.append(self.pool_mix(packed, layers, 0, x_bert_offset, x_mask, eye[l]))
        return outs
