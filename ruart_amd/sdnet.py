"""``SDNet`` - drop-in for the reference's model class (Models/SDNet.py:20-437) on MI355X.

Same constructor signature ``SDNet(opt, embedding)``, same ``forward(q_list, ocr_list, od_list, return_score=False)
-> (score_s (B, max_ocr_num + 1), None)``, same sub-module / parameter names (so ``state_dict`` keys match the
reference's checkpoints), same mutable ``drop_emb`` attribute the trainer flips.

What is different underneath (results identical up to fp32 summation order; BERT in bf16 by default):
  * one packed BERT pass per step for question + OCR items + object items (ruart_bert_forward), sub-word pooling and
    the alpha/gamma layer mix fused in one kernel;
  * OCR / object words stay in a packed (real words, D) matrix: embedding concat, pre-align scatter/gather and the
    ``multi2one`` LSTM run over real words only; the reference's four Python loops in ``get_prealign_emb`` and the
    per-item gather loop (:300-318) are index_put / gather with index vectors prepared once per batch (batch.py);
  * every LSTM recurrence, attention score/softmax/context and whole-tensor layer norm is a HIP kernel (ops.py);
  * the reference's per-op ``assert isnan == 0`` device syncs are one flag check per forward.

Supported configuration: the shipped conf and its simple toggles.  Branches the shipped conf does not enable
(img_feature, fixed_answers, ES post_process, ModelParallel, PRE_ALIGN_after_rnn, label_yesno, bidirectional
multi2one) raise NotImplementedError - they are out of the hot path's scope.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip
from . import layers as L
from . import ops
from .batch import BatchIndex
from .bert import Bert, _PoolMix, bert_encode
from . import bert_train
from .layers import Attention, DeepAttention, GetFinalScores, LinearSelfAttn, RNN_from_opt, dropout, row_dropout

# The forward runs its question / object / OCR branches on three streams and several modules (deep attention, the high-level
# RNNs) serve two branches, so a parameter's gradient contributions legitimately arrive from streams other than the one its
# accumulation node was created on; autograd orders them with events.  That is the design, not a leaked graph: silence the hint.
if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)

_HOST_DELAY_US = ops._HOST_DELAY_US          # timing diagnostics only; refused without RUART_DIAGNOSTICS=1 (ops.py)

_UNSUPPORTED = ("img_feature", "fixed_answers", "ModelParallel", "PRE_ALIGN_after_rnn", "label_yesno", "no_Context_Self_Attention",
                "no_DeepAttention")


def _record(t, stream):
    """``t`` (allocated on another stream) is read on ``stream``: tell the caching allocator - except while a graph is being
    captured, where the private pool keeps every block until the graph dies and record_stream is not allowed."""
    if isinstance(t, torch.Tensor) and t.is_cuda and not torch.cuda.is_current_stream_capturing():
        t.record_stream(stream)


def _fork(main, sides, tensors):
    """side streams may start once everything enqueued on ``main`` so far is done; tensors made on ``main`` and read on a
    side stream are registered with the caching allocator."""
    for s in sides:
        if s is main:
            continue
        s.wait_stream(main)
        for t in tensors:
            _record(t, s)


def _join(main, sides, tensors):
    for s in sides:
        if s is not main:
            main.wait_stream(s)
    for t in tensors:
        _record(t, main)


class _Trunk(nn.Module):
    """The fixed-shape part of SDNet.forward (Models/SDNet.py:339-436): question / OCR / object RNN stacks, deep attention,
    self attention, high-level RNNs, object->OCR attention, answer scores.  Every tensor is dense (B, L, D), so for a given
    batch size the whole forward (and its backward) is one fixed launch sequence - which is what lets it be captured in a
    hipGraph.  Holds the SAME submodule objects as the SDNet it was built from (shared parameters); it is never registered as
    a child of that SDNet, so checkpoints keep the reference's keys."""

    def __init__(self, net):
        super().__init__()
        self.opt = net.opt
        for name in ("ques_rnn", "high_lvl_ques_rnn", "ques_self_attn", "ques_merger", "context_rnn", "deep_attn",
                     "highlvl_self_att", "high_lvl_context_rnn", "get_answer"):
            setattr(self, name, getattr(net, name))
        for name in ("od_ocr_attn", "position_attn"):
            if hasattr(net, name):
                setattr(self, name, getattr(net, name))
        self._net_streams = net._side_streams       # eager mode shares the SDNet's two side streams (few HW queues)
        self._net_use_streams = net._use_streams
        self._bank = L.MaskBank()

    def _side_streams(self, dev):
        if torch.cuda.is_current_stream_capturing():
            # side streams of a capture must be fresh ones forked from the capturing stream
            st = self.__dict__.get("_cap_streams")
            if st is None:
                st = self.__dict__["_cap_streams"] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
            return st
        return self._net_streams(dev)

    def forward(self, q_input, q_raw, q_mask, x_ocr, x_od, ocr_mask, od_mask, ocr_pos, od_pos):
        opt = self.opt
        dev = q_input.device
        prev_bank, L.mask_bank = L.mask_bank, self._bank       # masks of the trunk are drawn inside it (capturable)
        try:
            if self.training:
                self._bank.begin_step(dev)
            # The question, OCR and object branches are independent until they meet in deep_attn / od_ocr_attn.  Their kernels
            # are small (B=64 rows), so they run on three HIP streams and overlap on the 256 CUs; autograd replays each backward
            # on its forward's stream, so the backward overlaps the same way.
            main = torch.cuda.current_stream(dev)
            use_streams = self._net_use_streams()
            s_q, s_od = self._side_streams(dev) if use_streams else (main, main)
            _fork(main, (s_q, s_od), [q_input, q_raw, q_mask, x_od, od_mask])

            with torch.cuda.stream(s_q):                                        # ---- question branch (SDNet.py:339, 350)
                _, q_rnn_layers = self.ques_rnn(q_input, q_mask, return_list=True, LN=True)
                q_highlvl = self.high_lvl_ques_rnn(torch.cat(q_rnn_layers, 2), q_mask, LN=True)
                q_rnn_layers = q_rnn_layers + [q_highlvl]
                ev_q_layers = s_q.record_event() if use_streams else None
                q_final = self.ques_self_attn(q_highlvl, q_highlvl, q_mask)    # SDNet.py:411-415
                q_merged = self.ques_merger.merge(q_final, q_mask)
            q_long = [q_raw]

            def context_branch(x, mask, helper=None):
                """context_rnn, deep_attn, self-attention, high-level rnn for OCR tokens or objects"""
                _, rnn_layers = self.context_rnn(x, mask, return_list=True, LN=True)
                if ev_q_layers is not None:
                    torch.cuda.current_stream(dev).wait_event(ev_q_layers)
                    for t in q_rnn_layers:
                        _record(t, torch.cuda.current_stream(dev))
                h, pre = self.deep_attn([x], rnn_layers, q_long, q_rnn_layers, mask, q_mask, return_bef_rnn=True, helper=helper)
                sa_in = torch.cat([h, pre, x], 2)
                sa = self.highlvl_self_att(sa_in, sa_in, mask, x3=h)
                return self.high_lvl_context_rnn(torch.cat([h, sa], 2), mask, LN=True)

            with torch.cuda.stream(s_od):                                       # ---- object branch
                od_hl = context_branch(x_od, od_mask)
            # ---- OCR branch (the longest chain) on the main stream; one of its three independent deep-attention levels borrows
            #      the question stream, which is idle by then
            ocr_hl = context_branch(x_ocr, ocr_mask, helper=s_q if use_streams else None)
            _join(main, (s_od, s_q), [od_hl, q_merged])

            if "position_dim" in opt:
                if opt["position_mod"] == "qk+":
                    x_od_ocr = self.od_ocr_attn(ocr_hl, od_hl, od_mask) + self.position_attn(ocr_pos, od_pos, od_mask, x3=od_hl)
                else:
                    x_od_ocr = self.od_ocr_attn(torch.cat([ocr_hl, ocr_pos], 2), torch.cat([od_hl, od_pos], 2), od_mask)
            mode = opt["pos_att_merge_mod"]
            ocr_final = torch.cat([ocr_hl, x_od_ocr], 2) if mode == "cat" else (x_od_ocr if mode == "atted" else ocr_hl)
            es_len = opt["ES_ocr_len"] if "useES" in opt else None
            return self.get_answer(ocr_final, q_merged, ocr_mask, es_len, mask_flag="mask_score" in opt)
        finally:
            L.mask_bank = prev_bank


class SDNet(nn.Module):
    def __init__(self, opt, embedding):
        super().__init__()
        for k in _UNSUPPORTED:
            if k in opt:
                raise NotImplementedError("conf key %r selects a branch outside the accelerated hot path" % k)
        if "ES_ocr" in opt and opt.get("ES_using_way") == "post_process":
            raise NotImplementedError("ES_using_way post_process is outside the accelerated hot path")
        if opt.get("multi2one_bidir"):
            raise NotImplementedError("bidirectional multi2one is outside the accelerated hot path")
        self.opt = opt
        self.vocab_dim = 300
        self.use_cuda = opt.get("cuda") is True
        self.q_embedding = opt["q_embedding"].split(",")
        self.ocr_embedding = opt["ocr_embedding"].split(",")
        self.drop_emb = False
        L.set_dropout_prob(0.0 if "DROPOUT" not in opt else float(opt["DROPOUT"]))
        L.set_seq_dropout("VARIATIONAL_DROPOUT" in opt)

        x_in = q_in = 0
        for flag, attr, key, dim_key in (("PHOC", "phoc_embed", "phoc_embedding", "phoc_dim"),
                                         ("FastText", "fast_embed", "fast_embedding", "fast_dim"),
                                         ("GLOVE", "glove_embed", "glove_embedding", "glove_dim")):
            if flag in opt:
                self.vocab_size = int(opt["vocab_size"])
                emb = nn.Embedding(self.vocab_size, int(opt[dim_key]), padding_idx=1)
                emb.weight.data = embedding[key].clone()
                setattr(self, attr, emb)
        self.glove_dim = int(opt.get("glove_dim", 0))
        self.fast_dim = int(opt.get("fast_dim", 0))
        self.phoc_dim = int(opt.get("phoc_dim", 0))
        for name, dim in (("glove", self.glove_dim), ("fasttext", self.fast_dim), ("phoc", self.phoc_dim)):
            x_in += dim if name in self.ocr_embedding else 0
            q_in += dim if name in self.q_embedding else 0
        if "TUNE_PARTIAL" in opt:
            if "FastText" in opt:
                self.fixed_embedding_fast = embedding["fast_embedding"][opt["tune_partial"]:]
            if "GLOVE" in opt:
                self.fixed_embedding_glove = embedding["glove_embedding"][opt["tune_partial"]:]
        else:
            if "FastText" in opt:
                self.fast_embed.weight.requires_grad = False
            if "GLOVE" in opt:
                self.glove_embed.weight.requires_grad = False

        if "BERT" in opt:
            self.Bert = Bert(opt, device=opt.get("device", "cuda"))
            if "LOCK_BERT" in opt:
                self.Bert.lock()
            else:
                self.Bert.unlock()           # trainable fp32 encoder, parameters named as in the reference (bert_train.py)
            bert_dim, bert_layers = (1024, 24) if "BERT_LARGE" in opt else (768, 12)
            if "BERT_LINEAR_COMBINE" not in opt:
                raise NotImplementedError("the hot path is BERT_LINEAR_COMBINE (all layers mixed)")
            self.alphaBERT = nn.Parameter(torch.ones(bert_layers))
            self.gammaBERT = nn.Parameter(torch.ones(1, 1))
            x_in += bert_dim if "bert" in self.ocr_embedding else 0
            q_in += bert_dim if "bert" in self.q_embedding else 0
        if "PRE_ALIGN" in opt:
            self.pre_align = Attention(self.vocab_dim, opt["prealign_hidden"], correlation_func=3, do_similarity=True)
            if "PRE_ALIGN_befor_rnn" in opt:
                x_in += self.vocab_dim
        if "pos" in self.q_embedding or "pos" in self.ocr_embedding:
            self.pos_embedding = nn.Embedding(int(opt.get("pos_vocab_size", 51)), opt["pos_dim"])
            x_in += opt["pos_dim"] if "pos" in self.ocr_embedding else 0
            q_in += opt["pos_dim"] if "pos" in self.q_embedding else 0
        if "ent" in self.q_embedding or "pos" in self.ocr_embedding:      # sic: the reference tests 'pos' here (SDNet.py:126)
            self.ent_embedding = nn.Embedding(int(opt.get("ent_vocab_size", 75)), opt["ent_dim"])
            x_in += opt["ent_dim"] if "ent" in self.ocr_embedding else 0
            q_in += opt["ent_dim"] if "ent" in self.q_embedding else 0

        self.multi2one, m2o = RNN_from_opt(x_in, opt["multi2one_hidden_size"], num_layers=1, concat_rnn=opt["concat_rnn"],
                                           bidirectional=opt["multi2one_bidir"])
        self.multi2one_output_size = m2o
        self.context_rnn, ctx_out = RNN_from_opt(m2o, opt["hidden_size"], num_layers=opt["in_rnn_layers"], concat_rnn=opt["concat_rnn"])
        self.ques_rnn, q_out = RNN_from_opt(q_in, opt["hidden_size"], num_layers=opt["in_rnn_layers"], concat_rnn=opt["concat_rnn"])
        word_hidden = 0 if ("GLOVE" not in opt and "FastText" not in opt) else m2o
        self.deep_attn = DeepAttention(opt, abstr_list_cnt=opt["in_rnn_layers"],
                                       deep_att_hidden_size_per_abstr=opt["deep_att_hidden_size_per_abstr"], correlation_func=3,
                                       word_hidden_size=word_hidden)
        self.deep_attn_input_size = self.deep_attn.rnn_input_size
        self.deep_attn_output_size = self.deep_attn.output_size
        self.high_lvl_ques_rnn, hq_out = RNN_from_opt(q_out * opt["in_rnn_layers"], opt["highlvl_hidden_size"],
                                                      num_layers=opt["question_high_lvl_rnn_layers"], concat_rnn=True)
        self.after_deep_attn_size = self.deep_attn_output_size + self.deep_attn_input_size + m2o
        self.self_attn_input_size = self.after_deep_attn_size
        self.highlvl_self_att = Attention(self.self_attn_input_size, opt["deep_att_hidden_size_per_abstr"], correlation_func=3)
        self.high_lvl_context_rnn, ctx_final = RNN_from_opt(self.deep_attn_output_size * 2, opt["highlvl_hidden_size"], num_layers=1,
                                                            concat_rnn=False)
        self.ques_self_attn = Attention(hq_out, opt["query_self_attn_hidden_size"], correlation_func=3)
        q_final = hq_out
        pos_out = 0
        if "position_dim" in opt:
            if opt["position_mod"] == "qk+":
                self.od_ocr_attn = Attention(ctx_final, opt["hidden_size"], correlation_func=3, do_similarity=True)
                self.position_attn = Attention(opt["position_dim"], opt["hidden_size"], correlation_func=3, do_similarity=True)
                pos_out = ctx_final
            elif opt["position_mod"] == "cat":
                self.od_ocr_attn = Attention(ctx_final + opt["position_dim"], opt["hidden_size"], correlation_func=3, do_similarity=True)
                pos_out = ctx_final + opt["position_dim"]
        self.ques_merger = LinearSelfAttn(q_final)
        ocr_final = {"cat": ctx_final + pos_out, "atted": pos_out, "original": ctx_final}[opt["pos_att_merge_mod"]]
        self.get_answer = GetFinalScores(ocr_final, q_final, yesno=False, no_answer="label_no_answer" in opt, useES="useES" in opt)

    # ------------------------------------------------------------------------------------------------------
    @property
    def device(self):
        return self.alphaBERT.device

    def prepare(self, q_list, ocr_list, od_list):
        """Build (or fetch) the per-batch index vectors and the packed BERT stream; cached on the batch dict."""
        bi = q_list.get("_ruart_index")
        if bi is None or bi.device != self.device:
            host = q_list.get("_ruart_host_index")          # built by VQA_collate(prepare_index=True) in a loader worker
            frozen = "LOCK_BERT" in self.opt and not self.opt.get("bert_frozen_dropout")
            want = (self.Bert.pack, self.Bert.weights.dtype != 0, bool(frozen and self.opt.get("bert_dedup", True) and self.Bert.pack))
            if host is not None and getattr(host, "plan", None) == want:
                bi = host.to(self.device)
            else:
                bi = BatchIndex(q_list, ocr_list, od_list, self.opt, self.device, bert=self.Bert)
            q_list["_ruart_index"] = bi
        return bi

    def prefetch_bert(self, q_list, ocr_list, od_list):
        """Ask the next ``forward`` to start the (frozen) BERT pass of this FUTURE batch on the encoder's own stream, right
        after it has picked up its own encoder output (see Bert.prefetch)."""
        self._next_batch = (q_list, ocr_list, od_list)

    def launch_prefetch(self):
        """Start the encoder pass registered by ``prefetch_bert`` (no-op when none is pending)."""
        nxt, self._next_batch = getattr(self, "_next_batch", None), None
        if nxt is not None:
            self.Bert.prefetch(self.prepare(*nxt).packed)

    def _use_streams(self):
        """Question / object / OCR branches on three streams.  With a trainable encoder: only its 16-bit form (bert_train16.py: every
        product on this library's kernels).  The fp32-class graph stays on one stream - its first version (library GEMMs of extreme
        shape on the three streams at once) stopped completing steps at B = 64.  For the 16-bit encoder round
        2 measured no difference (1 189-1 201 samples/s against 1 192-1 205 on one stream) because the trunk phase was bound by the
        host's enqueue rate either way; with the loss readback one step late (trainer._readback_later) the host is ahead of the
        device and the trunk phase is as long as its kernels take - three streams shorten that."""
        if ops.trunk_gemm != "x3":          # exact-fp32 validation mode: its projections are library GEMMs (stream-K solutions that
            return False                     # need all their workgroups resident) - never beside other streams' kernels
        if not bool(self.opt.get("ruart_streams", True)):
            return False
        model = getattr(self.Bert, "bert_model", None) if "BERT" in self.opt else None
        if model is None:
            return True
        return bool(self.opt.get("ruart_streams_trained_encoder", type(model).__name__ == "BertModelTrainable16"))

    def _layer_weights(self):
        """softmax(alpha)_l * gamma - the scalar each BERT layer is mixed with (SDNet.py:574-576)."""
        return F.softmax(self.alphaBERT, dim=0) * self.gammaBERT.view(1)

    def _word_table(self, key):
        return {"fasttext": "fast_embed", "glove": "glove_embed", "phoc": "phoc_embed"}[key]

    def _embed_question(self, q_list, bert_mix):
        dev = self.device
        parts, raw = [], None
        p_emb = self.opt.get("dropout_emb", 0.0)
        bi = q_list.get("_ruart_index")
        srt = bi.emb_sort if bi is not None else {}
        for key in ("phoc", "fasttext", "glove"):
            if key in self.q_embedding:
                e = ops.embedding(getattr(self, self._word_table(key)), q_list[key].to(dev), srt.get(("q", key)))
                if key == self.opt["q_emb_initial"]:
                    raw = e
                parts.append(dropout(e, p=p_emb, training=self.drop_emb) if "dropout_emb" in self.opt else e)
        if "bert" in self.q_embedding:
            parts.append(dropout(bert_mix, p=p_emb, training=self.drop_emb))
        if "pos" in self.q_embedding:
            parts.append(ops.embedding(self.pos_embedding, q_list["pos"].to(dev), srt.get(("q", "pos"))))
        if "ent" in self.q_embedding:
            parts.append(ops.embedding(self.ent_embedding, q_list["ent"].to(dev), srt.get(("q", "ent"))))
        return torch.cat(parts, -1), raw

    def _embed_items(self, items, idx, bert_mix):
        """Packed (W, D) embedding of the real words of an item group (the reference's get_embedding_from_list,
        SDNet.py:439-493, restricted to the rows its consumers read)."""
        dev = self.device
        d = idx.dev
        parts, raw = [], None
        p_emb = self.opt.get("dropout_emb", 0.0)
        for key in ("phoc", "fasttext", "glove"):
            if key in self.ocr_embedding:
                ids = d["ids_" + key] if "ids_" + key in d else items[key].to(dev).reshape(-1)[d["flat_word"]]
                e = ops.embedding(getattr(self, self._word_table(key)), ids, idx.emb_sort.get(key))
                if key == self.opt["ocr_emb_initial"]:
                    raw = e
                parts.append(row_dropout(e, d["item_of_word"], idx.N, p_emb, self.drop_emb) if "dropout_emb" in self.opt else e)
        if "bert" in self.ocr_embedding:
            parts.append(row_dropout(bert_mix, d["item_of_word"], idx.N, p_emb, self.drop_emb))
        if "pos" in self.ocr_embedding:
            ids = d["ids_pos"] if "ids_pos" in d else items["pos"].to(dev).reshape(-1)[d["flat_word"]]
            parts.append(ops.embedding(self.pos_embedding, ids, idx.emb_sort.get("pos")))
        if "ent" in self.ocr_embedding:
            ids = d["ids_ent"] if "ids_ent" in d else items["ent"].to(dev).reshape(-1)[d["flat_word"]]
            parts.append(ops.embedding(self.ent_embedding, ids, idx.emb_sort.get("ent")))
        return torch.cat(parts, -1), raw

    def _prealign(self, raw_words, idx, q_raw, q_mask):
        """SDNet.py:495-551 without the loops: scatter each sample's words into one row, attend over the question's
        raw word vectors, gather back."""
        d = idx.dev
        # scatter and gather by ONE flat row index (every row at most once): index_copy / index_select are plain row copies in both
        # directions, whereas advanced indexing with two index tensors goes through index_put's sort-and-accumulate path (108 us for a
        # 2.7 MB tensor, round 5 op table) and sorts the indices on the device again in every backward
        Tm = max(idx.Tmax, 1)
        flat = d["flat_tok"]
        x1 = raw_words.new_zeros(idx.B * Tm, raw_words.size(1)).index_copy(0, flat, raw_words).view(idx.B, Tm, raw_words.size(1))
        att = self.pre_align(x1, q_raw, q_mask)
        return att.reshape(-1, att.shape[2]).index_select(0, flat)

    def _multi2one_last(self, x_words, idx):
        """``multi2one`` (uni-directional LSTM, SDNet.py:137, 269-271) over real words only, returning the state at
        each item's last word already scattered to (B, max_num, hidden) - what SDNet.py:288-318 builds item by item."""
        d = idx.dev
        rnn = self.multi2one.rnns[0]
        if L.dropout_p > 0:
            x_words = row_dropout(x_words, d["item_of_word"], idx.N, L.dropout_p, self.training)
        xproj = ops.linear(x_words, rnn.weight_ih_l0, rnn.bias_ih_l0 + rnn.bias_hh_l0)
        Hh = rnn.weight_hh_l0.shape[1]
        steps = torch.split(xproj.index_select(0, d["step_rows"]), idx.n_active)      # unique rows: no sort in backward
        h0 = x_words.new_zeros(idx.N, Hh)
        h, _ = L.lstm_cell_steps(steps, rnn.weight_hh_l0, idx.n_active, h0, h0)
        flat = d["flat_slot"]                                               # (distinct rows: a plain row scatter, see _prealign)
        return x_words.new_zeros(idx.B * idx.max_num, Hh).index_copy(0, flat, h).view(idx.B, idx.max_num, Hh)

    # ------------------------------------------------------------------------------------------------------
    def forward(self, q_list, ocr_list, od_list, return_score=False):
        if return_score:
            raise NotImplementedError("att_score output is not part of the hot path")
        opt = self.opt
        dev = self.device
        # trunk projections: split-bf16 MFMA kernel, or the library's exact fp32 GEMM in the fp32 validation mode
        prec = self.Bert.weights.precision if "BERT" in opt else "x3"
        ops.trunk_gemm = opt.get("ruart_trunk_gemm", "fp32" if prec == "fp32" else "x3")
        # gradients of the trunk's projections: three bf16 products like the forward ("x1": one product - measured worth only
        # 0.1-0.2 ms of a 20 ms step, these products are bound by their fp32 operand loads, so it stays an option)
        ops.trunk_grad_gemm = opt.get("ruart_trunk_grad_gemm", "x3")
        # opt['ruart_defer_dw'] (off): record every projection's weight gradient and compute them in grouped launches at the end of
        # backward (ops._flush_weight_grads; bitwise the same gradients, ~80 launches fewer).  Measured SLOWER, 26.9-27.0 against
        # 25.2-25.8 ms per step on one box: the products no longer hide inside backward but land in front of the step's final sync
        ops.defer_weight_grads = bool(self.training and torch.is_grad_enabled() and opt.get("ruart_defer_dw", os.environ.get("RUART_DEFER_DW", "0") == "1")
                                      and not opt.get("ruart_graph_trunk", False))
        bi = self.prepare(q_list, ocr_list, od_list)
        if self.training or self.drop_emb:
            L.mask_bank.begin_step(dev)

        # ---- BERT: one packed pass, then pooled + mixed per group --------------------------------------------
        lw = self._layer_weights()
        trainable = getattr(self.Bert, "bert_model", None) is not None
        fused_mix = trainable and getattr(self.Bert.bert_model, "fused_mix", False)
        if fused_mix:                        # 16-bit trainable encoder: the layer mix happens inside its autograd Function
            layers = None
            mixed = self.Bert.bert_model.forward_mixed(bi.packed, lw, training=self.training)
        else:
            layers = self.Bert.layers_for(bi.packed)
        self.launch_prefetch()               # the following step's encoder pass starts now, beside this step's trunk
        if _HOST_DELAY_US:                   # timing diagnostics only (tools/r04_hostdelay.sh): is the host's enqueue time on the step's critical path?
            import time
            t_end = time.perf_counter() + _HOST_DELAY_US * 1e-6
            while time.perf_counter() < t_end:
                pass
        if "trunk" in ops._ABL_SKIP:          # timing diagnostics only (ops._ABL_SKIP, empty in every product run): no trunk work at all
            Bq = q_list[opt["q_emb_initial"]].shape[0]
            return torch.zeros(Bq, bi.ocr_mask.shape[1] + 1, device=dev) + lw.sum() * 0.0, None
        H = self.Bert.weights.hidden
        Bq, Q = q_list[opt["q_emb_initial"]].shape
        q_mask = q_list[opt["q_emb_initial"] + "_mask"].to(dev).to(torch.uint8)    # once: the attention kernels take uint8
        ocr_mask, od_mask = bi.ocr_mask, bi.od_mask

        # ---- front: variable-size part (real words of this batch): BERT pooling, embeddings, pre-align, multi2one -------
        # Three independent groups on three streams (their backward runs there too): question on s_q, objects on s_od, OCR
        # tokens - the heaviest - on the main stream.  The item groups need the question's raw word vectors for pre-align.
        main = torch.cuda.current_stream(dev)
        use_streams = self._use_streams()
        s_q, s_od = self._side_streams(dev) if use_streams else (main, main)

        if trainable and not fused_mix:
            mixed = bert_train.mix_layers(lw, layers)                         # once for the three groups
        elif not trainable:
            mixed = None

        def pooled(g):
            s_, l_, dst, rows, s_last = bi.spans[g]
            if trainable:
                return bert_train.pool_words(mixed, s_, l_, dst, rows, n_pieces=bi.span_pieces[g])
            ln = getattr(layers, "_ln", None) or (None, None, None)      # a LayerNorm-folded encoder pass: the kernel normalises what it reads
            return _PoolMix.apply(lw, layers, s_, l_, dst, rows, hip.dtype_code(layers), s_last, *ln)

        def front(items, idx, mix):
            words, raw = self._embed_items(items, idx, mix)
            if "PRE_ALIGN_befor_rnn" in opt:
                words = torch.cat([words, self._prealign(raw, idx, q_raw, q_mask)], -1)
            return self._multi2one_last(words, idx)                             # (B, max_num, 300)

        _fork(main, (s_q, s_od), [lw, layers, q_mask] + ([mixed] if trainable else []))
        with torch.cuda.stream(s_q):
            q_input, q_raw = self._embed_question(q_list, pooled(0).view(Bq, Q, H))
            ev_q = s_q.record_event() if use_streams else None
        if "PRE_ALIGN_befor_rnn" in opt:
            q_list[opt["q_emb_initial"] + "_emb"] = q_raw                      # the reference's side effect (SDNet.py:449-459)
        with torch.cuda.stream(s_od):
            mix_od = pooled(2)
            if ev_q is not None:
                s_od.wait_event(ev_q)
                _record(q_raw, s_od)
            x_od = front(od_list, bi.od, mix_od)
        mix_ocr = pooled(1)
        if ev_q is not None:
            main.wait_event(ev_q)
            _record(q_raw, main)
        x_ocr = front(ocr_list, bi.ocr, mix_ocr)
        _join(main, (s_od, s_q), [x_od, q_input, q_raw])

        # ---- trunk: fixed-shape part (B, L, D) - eager, or one hipGraph replay per direction ------------------
        if "position_dim" in opt:
            ocr_pos, od_pos = ocr_list["position"].to(dev), od_list["position"].to(dev)
        else:
            ocr_pos = od_pos = q_mask.new_zeros(1)
        trunk = self._trunk_callable(q_input, q_raw, q_mask, x_ocr, x_od, ocr_mask, od_mask, ocr_pos, od_pos)
        score_s = trunk(q_input, q_raw, q_mask, x_ocr, x_od, ocr_mask, od_mask, ocr_pos, od_pos)
        return score_s, None

    # -- the dense trunk and its graph capture -------------------------------------------------------------------
    def _trunk_module(self):
        t = self.__dict__.get("_trunk")
        if t is None:
            t = _Trunk(self)
            self.__dict__["_trunk"] = t          # not a registered child: the state dict keeps the reference's keys
        return t

    def _trunk_callable(self, *args):
        """Eager trunk, or - opt['ruart_graph_trunk'] - a captured forward/backward pair for this shape signature
        (torch.cuda.make_graphed_callables): the trunk is ~850 small launches whose host enqueue time exceeds their GPU time."""
        trunk = self._trunk_module()
        trunk.train(self.training)
        if not (self.opt.get("ruart_graph_trunk", False) and self.training and torch.is_grad_enabled()):
            return trunk
        key = tuple((tuple(a.shape), a.dtype, a.requires_grad) for a in args)
        cache = self.__dict__.setdefault("_trunk_graphs", {})
        ent = cache.get(key)
        if ent is None:
            ent = cache[key] = {"seen": 0, "fn": None}
        if ent["fn"] is None:
            ent["seen"] += 1
            if ent["seen"] <= int(self.opt.get("ruart_graph_after", 2)) or len([e for e in cache.values() if e["fn"]]) >= 4:
                return trunk                      # eager until the shape has recurred (and for a 5th distinct shape)
            torch.cuda.synchronize(self.device)
            sample = tuple(a.detach().clone().requires_grad_(a.requires_grad) for a in args)
            ent["fn"] = torch.cuda.make_graphed_callables(trunk, sample, num_warmup_iters=3, allow_unused_input=True)
            self.zero_grad(set_to_none=True)
        return ent["fn"]

    # -- stream plumbing -----------------------------------------------------------------------------------
    def trunk_stream_priority(self):
        """Priority of the step's streams (the trainer's step stream and the two branch streams), on HIP's scale (-1 high, 0 normal,
        1 low).  LOW beside a CU-masked encoder stream - the fp16c schedule, where the encoder pass IS the step (its last GEMM ends
        it) and the trunk has ~3 ms of slack: a high-priority trunk delays single tiles of the GEMM grids (25.5 -> 24.5 ms per step
        at normal priority, round 3), and at LOW priority - a level torch's stream pool does not offer (hip.priority_stream) - the
        step is another 0.6-0.7 ms shorter on the pool's slower boxes (26.0 -> 25.3 ms) and unchanged on its fastest (24.2);
        `profiles/r04_trunk_priority.log`.  HIGH otherwise: beside an unmasked encoder stream of equal priority the trunk's small
        kernels wait behind 256-workgroup GEMM rounds (plain f16: 21.2 ms against 18.6).  RUART_TRUNK_PRIORITY overrides."""
        env = os.environ.get("RUART_TRUNK_PRIORITY")
        if env is not None:
            return int(env)
        bert = getattr(self, "Bert", None)
        masked = bert is not None and hasattr(bert, "prefetch_cus") and bert.prefetch_cus() > 0
        return 1 if masked else -1

    def trunk_stream_cus(self):
        """CU mask of the step's three streams (the trainer's step stream and the two branch streams): 0 = none (the default); n > 0 the
        first n CUs; n < 0 the LAST |n| CUs (ruart_stream_create_cu_masked).  opt['ruart_trunk_cus'] / RUART_TRUNK_CUS - an experiment,
        NOT part of the schedule.  Round 6 measured it beside the masked encoder pass (which owns the FIRST 224 CUs): the trunk on the
        LAST 160 - so that 96 CUs never hold a trunk wave and a GEMM workgroup, which needs a whole CU, always finds them free - is
        1.35 % faster in one uninterrupted session (22.52 -> 22.22 ms over eight interleaved pairs), and 1-3 ms SLOWER once a session
        has been closed or an evaluation has run (23.3 / 26.0 / 25.5 ms against 22.2 / 22.3 / 22.3): four masked streams and the
        evaluation's streams share hardware queue slots (profiles/HISTORY.md round 5 (9)), and a process that ends with them dumps core.
        profiles/r06_trunk_mask.log."""
        v = os.environ.get("RUART_TRUNK_CUS", self.opt.get("ruart_trunk_cus"))
        return int(v) if v is not None else 0

    def _side_streams(self, dev):
        pr = self.trunk_stream_priority()                           # same priority as the step stream (trainer.on_step_stream)
        cache = self.__dict__.setdefault("_streams", {})
        st = cache.get(pr)
        if st is None or st[0].device != dev:
            ncu = self.trunk_stream_cus()                           # (an experiment's mask; the step stream takes the same: trainer._step_stream)
            if ncu != 0:
                st = (hip.cu_masked_stream(ncu, dev), hip.cu_masked_stream(ncu, dev))
            elif pr > 0:                                            # LOW priority: not in torch's pool (hip.priority_stream)
                st = (hip.priority_stream(pr, dev), hip.priority_stream(pr, dev))
            else:
                st = (torch.cuda.Stream(device=dev, priority=pr), torch.cuda.Stream(device=dev, priority=pr))
            cache[pr] = st
        return st

    def check_nan(self):
        """One host sync honouring every ``assert torch.sum(torch.isnan(.)) == 0`` of the reference's forward."""
        ops.nan_flag.check_and_clear()
