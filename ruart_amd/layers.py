"""SDNet building blocks with the reference's class names, constructor arguments and parameter names
(Models/Layers.py), so checkpoints written by the reference load unchanged - but every sequential / attention /
normalisation step runs on the hand-written HIP kernels of libruart_hip.so (ruart_amd.ops).

Module-global dropout state mirrors Models/Layers.py:15-21 (``set_dropout_prob`` / ``set_seq_dropout``).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.parameter import Parameter

from . import ops

dropout_p = 0.0
do_seq_dropout = False
# GetFinalScores on the fused kernel (csrc/sdnet_scorer.hip); False / RUART_FUSED_SCORER=0: the op-by-op form (A/B runs, tests)
fused_scorer_enabled = os.environ.get("RUART_FUSED_SCORER", "1") != "0"


def set_dropout_prob(p):
    global dropout_p
    dropout_p = p


def set_seq_dropout(option):
    global do_seq_dropout
    do_seq_dropout = option


class MaskBank:
    """Pre-generated variational-dropout masks.  The reference draws one small Bernoulli tensor per dropout site
    (~50 sites per forward => ~200 tiny launches for fill / bernoulli / scale / multiply).  Here one launch per distinct
    p fills a big buffer of already-scaled masks at the start of the step; a site just takes the next slice.  The
    buffer is sized from the previous step's demand; if it runs dry the site falls back to drawing its own mask."""

    def __init__(self):
        self.buf = {}        # p -> (tensor, offset)
        self.demand = {}     # p -> elements used last step
        self.used = {}

    def begin_step(self, device):
        for p, n in self.used.items():
            self.demand[p] = max(self.demand.get(p, 0), n)
        self.used = {}
        self.buf = {}
        for p, n in self.demand.items():
            if n > 0:
                t = torch.empty(int(n * 1.05) + 1024, device=device, dtype=torch.float32).bernoulli_(1.0 - p).mul_(1.0 / (1.0 - p))
                # the same masks as one byte per element (0 = dropped): what the projections read inside their operand loads (ops._Linear)
                self.buf[p] = [t, 0, (t != 0).view(torch.uint8)]

    def take(self, rows, cols, p, like):
        n = rows * cols
        step = (n + 3) & ~3                     # every mask starts 16-byte aligned: the GEMMs read it with vector loads (ops._Linear)
        self.used[p] = self.used.get(p, 0) + step
        ent = self.buf.get(p)
        if ent is not None and ent[0].device == like.device and ent[1] + n <= ent[0].numel():
            m = ent[0][ent[1]:ent[1] + n].view(rows, cols)
            m.keep = ent[2][ent[1]:ent[1] + n].view(rows, cols)
            m.keep_scale = 1.0 / (1.0 - p)
            ent[1] += step
            return m
        return torch.bernoulli(like.new_full((rows, cols), 1.0 - p)) / (1.0 - p)


mask_bank = MaskBank()


def seq_dropout(x, p=0, training=False):
    """Variational dropout (Layers.py:23-30): one Bernoulli mask per (batch row, feature), shared over time,
    scaled by 1/(1-p)."""
    if not training or p == 0:
        return x
    return mask_bank.take(x.size(0), x.size(2), p, x).unsqueeze(1) * x


def seq_dropout_mask(x, p=0, training=False):
    """The (B, D) mask ``seq_dropout`` would multiply a 3-D x with, or None - for consumers that fuse the multiply
    (ops.linear(mask=)).  Draws from the bank exactly like ``dropout`` would, so the random stream is the same."""
    if not training or p == 0 or not do_seq_dropout or x.dim() != 3:
        return None
    return mask_bank.take(x.size(0), x.size(2), p, x)


def dropout(x, p=0, training=False):
    """Layers.py:32-39."""
    if do_seq_dropout and x.dim() == 3:
        return seq_dropout(x, p=p, training=training)
    return F.dropout(x, p=p, training=training)


def row_dropout(x, rows_of, n_rows, p, training):
    """seq_dropout for a PACKED (words, D) matrix: the mask is drawn per (item, feature) and gathered through
    ``rows_of`` (item index of each word) - the same distribution the reference draws on its padded (items, Lw, D)."""
    if not training or p == 0:
        return x
    if not do_seq_dropout:
        return F.dropout(x, p=p, training=True)
    return ops.rows_scale(x, mask_bank.take(n_rows, x.size(1), p, x), rows_of)


class StackedBRNN(nn.Module):
    """Layers.py:124-180.  ``rnns`` holds nn.LSTM modules purely as parameter containers (identical names, shapes and
    initialisation to the reference); the recurrence itself runs in ruart_lstm_fwd/bwd."""

    def __init__(self, input_size, hidden_size, num_layers, rnn_type=nn.LSTM, concat_layers=False, bidirectional=True,
                 add_feat=0, LN=False, batch_size=None, max_len=None):
        super().__init__()
        if rnn_type is not nn.LSTM:
            raise NotImplementedError("only nn.LSTM is on the hot path")
        self.bidir_coef = 2 if bidirectional else 1
        self.bidirectional = bidirectional
        self.num_layers = num_layers
        self.concat_layers = concat_layers
        self.hidden_size = hidden_size
        self.rnns = nn.ModuleList()
        for i in range(num_layers):
            in_size = input_size if i == 0 else (self.bidir_coef * hidden_size + add_feat if i == 1 else self.bidir_coef * hidden_size)
            self.rnns.append(nn.LSTM(in_size, hidden_size, num_layers=1, bidirectional=bidirectional, batch_first=True))

    @property
    def output_size(self):
        return (self.num_layers if self.concat_layers else 1) * self.bidir_coef * self.hidden_size

    def layer_params(self, i):
        r = self.rnns[i]
        p = [r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0]
        if self.bidirectional:
            p += [r.weight_ih_l0_reverse, r.weight_hh_l0_reverse, r.bias_ih_l0_reverse, r.bias_hh_l0_reverse]
        return p

    def forward(self, x, x_mask, return_list=False, x_additional=None, LN=None):
        hiddens = [x]
        for i in range(self.num_layers):
            rnn_input = hiddens[-1]
            if i == 1 and x_additional is not None:
                rnn_input = torch.cat((rnn_input, x_additional), 2)
            if self.hidden_size <= 128:
                # the input dropout (Layers.py:160-164) is folded into the input projection and its two gradient GEMMs
                mask = seq_dropout_mask(rnn_input, p=dropout_p, training=self.training)
                if mask is None and dropout_p > 0:
                    rnn_input = dropout(rnn_input, p=dropout_p, training=self.training)
                out = ops.lstm_layer(rnn_input, *self.layer_params(i), mask=mask)
            else:
                if dropout_p > 0:
                    rnn_input = dropout(rnn_input, p=dropout_p, training=self.training)
                out = lstm_layer_wide(rnn_input, *self.layer_params(i))
            if LN:
                out = ops.whole_layer_norm(out)
            hiddens.append(out)
        output = torch.cat(hiddens[1:], 2) if self.concat_layers else hiddens[-1]
        return (output, hiddens[1:]) if return_list else output


def lstm_cell_steps(xproj_steps, w_hh, n_active, h0, c0, zero_state=True):
    """Step a (wide) LSTM over a ragged, length-sorted batch: at step s only the first n_active[s] rows are alive.
    xproj_steps[s] is (n_active[s], 4h).  Per step: one GEMM (recurrent product, fused with the add of xproj) and one
    fused HIP cell kernel.  Returns the final (h, c) of every row."""
    h, c = h0, c0
    for s, n in enumerate(n_active):
        # with a zero initial state there is nothing to add at step 0
        pre = xproj_steps[s] if (s == 0 and zero_state) else ops.addmm(xproj_steps[s], h[:n], w_hh)
        h, c = ops.lstm_cell(pre, h, c, n)
    return h, c


def lstm_layer_wide(x, w_ih, w_hh, b_ih, b_hh, *reverse_params):
    """Dense (B, T, D) LSTM for hidden sizes above the persistent kernel's 128 limit (multi2one, hidden 300, when it
    is called through the generic module API).  T is short (<= max words per item)."""
    def run(w_ih, w_hh, b_ih, b_hh, rev):
        B, T, _ = x.shape
        xp = ops.linear(x, w_ih, b_ih + b_hh)
        h = x.new_zeros(B, w_hh.shape[1])
        c = x.new_zeros(B, w_hh.shape[1])
        ys = [None] * T
        for t in (range(T - 1, -1, -1) if rev else range(T)):
            h, c = lstm_cell_steps([xp[:, t]], w_hh, [B], h, c, zero_state=False)
            ys[t] = h
        return torch.stack(ys, 1)
    y = run(w_ih, w_hh, b_ih, b_hh, False)
    if reverse_params:
        y = torch.cat([y, run(*reverse_params, True)], 2)
    return y


class AttentionScore(nn.Module):
    """Layers.py:182-245, correlation_func 3 (the only one SDNet instantiates):  s_ij = (ReLU(W x1_i) * d) . ReLU(W x2_j)."""

    def __init__(self, input_size, hidden_size, correlation_func=1, do_similarity=False):
        super().__init__()
        if correlation_func != 3:
            raise NotImplementedError("correlation_func %d is not on the hot path" % correlation_func)
        self.correlation_func = correlation_func
        self.hidden_size = hidden_size
        self.linear = nn.Linear(input_size, hidden_size, bias=False)
        if do_similarity:
            self.diagonal = Parameter(torch.ones(1, 1, 1) / (hidden_size ** 0.5), requires_grad=False)
        else:
            self.diagonal = Parameter(torch.ones(1, 1, hidden_size), requires_grad=True)

    def project_raw(self, x1, x2):
        """W x1, W x2 (after input dropout); the ReLU and the diagonal are applied inside the fused attention kernel."""
        out = []
        for x in (x1, x2):
            mask = seq_dropout_mask(x, p=dropout_p, training=self.training)
            if mask is None:
                x = dropout(x, p=dropout_p, training=self.training)
            out.append(ops.linear(x, self.linear.weight, mask=mask))
        return out[0], out[1]

    def forward(self, x1, x2):
        p1, p2 = self.project_raw(x1, x2)
        return (F.relu(p1) * self.diagonal).bmm(F.relu(p2).transpose(1, 2))


class Attention(nn.Module):
    """Layers.py:247-295 on the fused HIP kernel (scores, -inf key mask, softmax, alpha . x3 without materialising
    anything but the saved probabilities)."""

    def __init__(self, input_size, hidden_size, correlation_func=1, do_similarity=False):
        super().__init__()
        self.scoring = AttentionScore(input_size, hidden_size, correlation_func, do_similarity)

    def forward(self, x1, x2, x2_mask, x3=None, drop_diagonal=False, return_score=False):
        if drop_diagonal or return_score:
            raise NotImplementedError("drop_diagonal / return_score are not used by SDNet.forward")
        p1, p2 = self.scoring.project_raw(x1, x2)
        return ops.fused_attention(p1, p2, x2 if x3 is None else x3, x2_mask, diag=self.scoring.diagonal, relu=True)


def RNN_from_opt(input_size_, hidden_size_, num_layers=1, concat_rnn=False, add_feat=0, bidirectional=True, rnn_type=nn.LSTM,
                 LN=False, batch_size=None, max_len=None):
    """Layers.py:297-317."""
    rnn = StackedBRNN(input_size=input_size_, hidden_size=hidden_size_, num_layers=num_layers, rnn_type=rnn_type,
                      concat_layers=concat_rnn, bidirectional=bidirectional, add_feat=add_feat)
    out = hidden_size_ * (2 if bidirectional else 1) * (num_layers if concat_rnn else 1)
    return rnn, out


class LinearSelfAttn(nn.Module):
    """Layers.py:320-341: alpha = softmax(mask(x w + b))."""

    def __init__(self, input_size):
        super().__init__()
        self.linear = nn.Linear(input_size, 1)

    def forward(self, x, x_mask):
        x = dropout(x, p=dropout_p, training=self.training)
        scores = ops.linear(x, self.linear.weight, self.linear.bias).squeeze(-1).masked_fill(x_mask.eq(0), float("-inf"))
        return F.softmax(scores, dim=1)

    def merge(self, x, x_mask):
        """weighted_avg(x, self(x, mask)) in one fused-attention launch: a single query row w against keys x.
        Note the reference drops out the scoring copy of x only; the averaged values are the undropped x."""
        xs = dropout(x, p=dropout_p, training=self.training)
        B = x.size(0)
        a = self.linear.weight.view(1, 1, -1).expand(B, 1, -1)
        # the scalar bias shifts every score of a row equally: softmax is invariant to it (its gradient is 0)
        return ops.fused_attention(a, xs, x, x_mask).squeeze(1)


def weighted_avg(x, weights):
    """Layers.py:529-534."""
    return weights.unsqueeze(1).bmm(x).squeeze(1)


class BilinearSeqAttn(nn.Module):
    """Layers.py:435-468: o_i = x_i' (W y + b), masked to -inf when mask_flag."""

    def __init__(self, x_size, y_size, identity=False):
        super().__init__()
        self.linear = nn.Linear(y_size, x_size) if not identity else None

    def forward(self, x, y, x_mask, mask_flag=True):
        x = dropout(x, p=dropout_p, training=self.training)
        y = dropout(y, p=dropout_p, training=self.training)
        Wy = ops.linear(y, self.linear.weight, self.linear.bias) if self.linear is not None else y
        xWy = (x * Wy.unsqueeze(1)).sum(2)           # x_i . (W y): a row-wise dot product, not worth a batched GEMM with N = 1
        if mask_flag:
            xWy = xWy.masked_fill(x_mask.eq(0), float("-inf"))
        return xWy


class GetFinalScores(nn.Module):
    """Layers.py:352-432 (useES / no_answer branches of the shipped conf).  ``rnn`` (GRUCell) is kept so the
    checkpoint keys match; the reference computes one step of it and discards the result (:395-397) - it never
    receives a gradient, and it is not computed here."""

    def __init__(self, x_size, h_size, yesno, no_answer, useES):
        super().__init__()
        if yesno:
            raise NotImplementedError("label_yesno is not in the shipped configuration")
        self.no_answer, self.yesno, self.useES = no_answer, yesno, useES
        if no_answer:
            self.noanswer_linear = nn.Linear(h_size, x_size)
            self.noanswer_w = nn.Linear(x_size, 1, bias=True)
        self.attn = BilinearSeqAttn(x_size, h_size)
        self.rnn = nn.GRUCell(x_size, h_size)
        self.attn2 = BilinearSeqAttn(x_size, h_size)

    def forward(self, x, h0, x_mask, ES_len, mask_flag=None):
        if (self.useES and self.no_answer and x.is_cuda and ops.trunk_gemm == "x3" and x.dim() == 3 and x.size(2) % 4 == 0
                and x.size(1) <= 1024 and 0 < ES_len < x.size(1) and fused_scorer_enabled
                # the fused form models the dropout of x only as the variational (B, D) mask; without VARIATIONAL_DROPOUT the
                # reference drops x element-wise (Layers.py:32-39, 454): those configurations take the op-by-op form below
                and (do_seq_dropout or not self.training or dropout_p == 0)):
            return self._forward_fused(x, h0, x_mask, ES_len, mask_flag)
        if self.useES:
            score_ocr = self.attn(x[:, ES_len:], h0, x_mask[:, ES_len:], mask_flag=mask_flag)
            score_es = self.attn2(x[:, :ES_len], h0, x_mask[:, :ES_len], mask_flag=mask_flag)
            score_s = torch.cat([score_es, score_ocr], dim=-1)
        else:
            score_s = self.attn(x, h0, x_mask, mask_flag=mask_flag)
        if self.no_answer:
            h0 = dropout(h0, p=dropout_p, training=self.training)
            score_s = torch.cat([score_s, self.get_single_score(x, h0, x_mask, self.noanswer_linear, self.noanswer_w)], dim=-1)
        return F.softmax(score_s, dim=-1)

    def _forward_fused(self, x, h0, x_mask, ES_len, mask_flag):
        """The same computation with the (B, L, D) work in ONE kernel per direction (ops.fused_scorer).  The three projections of h0 stay
        small GEMMs; the variational-dropout masks BilinearSeqAttn draws for its x (one (B, D) mask per call, :456-457) are folded into
        the projected vectors - (x o m) . u == x . (m o u) - in the order the unfused code draws them (attn, then attn2)."""
        def proj(attn, x_part):
            m = seq_dropout_mask(x_part, p=dropout_p, training=self.training)          # the call's dropout of x
            u = ops.linear(dropout(h0, p=dropout_p, training=self.training), attn.linear.weight, attn.linear.bias)
            return u if m is None else u * m
        u1 = proj(self.attn, x[:, ES_len:])
        u2 = proj(self.attn2, x[:, :ES_len])
        hd = dropout(h0, p=dropout_p, training=self.training)
        uh = ops.linear(hd, self.noanswer_linear.weight, self.noanswer_linear.bias)
        return ops.fused_scorer(x, u1, u2, uh, self.noanswer_w.weight, self.noanswer_w.bias, x_mask, ES_len, bool(mask_flag))

    def get_single_score(self, x, h, x_mask, linear, w):
        """:421-432: w . (softmax(mask(x . W h)) . x) + b - one fused-attention launch with a single query row."""
        Wh = ops.linear(h, linear.weight, linear.bias).unsqueeze(1)
        return ops.linear(ops.fused_attention(Wh, x, x, x_mask), w.weight, w.bias).squeeze(2)


class DeepAttention(nn.Module):
    """Layers.py:471-524 (history-of-word multi-level inter-attention)."""

    def __init__(self, opt, abstr_list_cnt, deep_att_hidden_size_per_abstr, correlation_func=1, word_hidden_size=None):
        super().__init__()
        if "no_DeepAttention" in opt:
            raise NotImplementedError("no_DeepAttention is not in the shipped configuration")
        word_hidden_size = opt["embedding_dim"] if word_hidden_size is None else word_hidden_size
        abstr_hidden_size = opt["hidden_size"] * 2
        att_size = abstr_hidden_size * abstr_list_cnt + word_hidden_size
        self.int_attn_list = nn.ModuleList(
            [Attention(att_size, deep_att_hidden_size_per_abstr, correlation_func=correlation_func) for _ in range(abstr_list_cnt + 1)])
        rnn_input_size = abstr_hidden_size * abstr_list_cnt * 2 + (opt["highlvl_hidden_size"] * 2)
        self.att_size = att_size
        self.rnn_input_size = rnn_input_size
        self.rnn, self.output_size = RNN_from_opt(rnn_input_size, opt["highlvl_hidden_size"], num_layers=1)
        self.opt = opt

    def forward(self, x1_word, x1_abstr, x2_word, x2_abstr, x1_mask, x2_mask, return_bef_rnn=False, return_score=False, helper=None):
        """``helper``: an otherwise idle stream; the attention levels read the same two inputs and do not depend on each other
        (Layers.py:508-517 only concatenates their outputs), so one of them can run there beside the others."""
        if return_score:
            raise NotImplementedError("return_score is not used by SDNet.forward")
        x1_att = torch.cat(x1_word + x1_abstr, 2)
        x2_att = torch.cat(x2_word + x2_abstr[:-1], 2)
        outs = [None] * len(x2_abstr)
        side = 1 if (helper is not None and len(x2_abstr) >= 2 and x1_att.is_cuda) else -1
        if side >= 0:
            cur = torch.cuda.current_stream(x1_att.device)
            helper.wait_stream(cur)
            with torch.cuda.stream(helper):
                outs[side] = self.int_attn_list[side](x1_att, x2_att, x2_mask, x3=x2_abstr[side])
        for i, x2_i in enumerate(x2_abstr):
            if i != side:
                outs[i] = self.int_attn_list[i](x1_att, x2_att, x2_mask, x3=x2_i)
        if side >= 0:
            cur.wait_stream(helper)
            if not torch.cuda.is_current_stream_capturing():
                for t in (x1_att, x2_att, x2_abstr[side]):
                    t.record_stream(helper)
                outs[side].record_stream(cur)
        x1 = torch.cat(list(x1_abstr) + outs, 2)
        x1_hiddens = self.rnn(x1, x1_mask)
        return (x1_hiddens, x1) if return_bef_rnn else x1_hiddens
