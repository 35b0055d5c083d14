"""Batch layout of the hot path: the reference's ``VQA_collate`` 5-tuple plus precomputed device index tensors.

``VQA_collate`` mirrors ``Utils/VQA_Dataset.py:439-542`` (same method names, same output layout: dicts of int64 id
matrices and bool masks, python offset / count lists, ``position (B, max_num, 8)``, ``gt (B, No+1)``).

``prepare`` is new: it turns the python lists the reference walks with per-item Python loops inside ``forward``
(Models/Bert/Bert.py:153-165, Models/SDNet.py:300-318 and :495-551) into a handful of int64 index vectors, computed
once per batch on the host and shipped in ONE host-to-device copy.  OCR / object words live in a packed
(real words, D) matrix from then on - the padded (items, 20, 1388) tensor of the reference is never materialised.
"""
import itertools

import numpy as np
import torch


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


class Offsets:
    """The per-item ``bert_offsets`` field of a collated batch - in the reference a python list (items) of lists (words) of
    [start, end] pairs - held as two arrays: all pairs back to back (n_words, 2) and the word count per item.  It reads like the
    list (len, indexing, iteration, == with a list) but travels between processes as two buffers: unpickling the ~6 400 small
    lists of one OCR group cost the training process 23 ms per batch, under the GIL, while it should be launching kernels."""

    __slots__ = ("pairs", "lens", "_starts")

    def __init__(self, pairs, lens):
        self.pairs, self.lens = pairs, lens
        self._starts = None

    def _row(self, i):
        if self._starts is None:
            self._starts = np.cumsum(self.lens) - self.lens
        s0 = int(self._starts[i])
        return self.pairs[s0:s0 + int(self.lens[i])].tolist()

    def __len__(self):
        return len(self.lens)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._row(j) for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return self._row(i)

    def __iter__(self):
        return (self._row(i) for i in range(len(self)))

    def tolist(self):
        return [r.tolist() for r in np.split(self.pairs, np.cumsum(self.lens)[:-1])] if len(self.lens) else []

    def __eq__(self, other):
        return self.tolist() == (other.tolist() if isinstance(other, Offsets) else other)

    def __getstate__(self):
        return (self.pairs, self.lens)

    def __setstate__(self, st):
        self.pairs, self.lens = st
        self._starts = None


class VQA_collate:
    def __init__(self, opt, prepare_index=False):
        """``prepare_index``: also build the hot path's host-side batch index (BatchIndex, numpy only) here - i.e. inside the
        DataLoader workers - and attach it as ``q_list['_ruart_host_index']``; ``SDNetTrainer.ToCUDA`` then only copies."""
        self.opt = opt
        self.prepare_index = prepare_index

    def VQA_collate_fun(self, batch):
        o = self.opt
        q_list = self.que_collate([t["q"] for t in batch], o["max_q_len"], o["max_q_bert_len"])
        flats = [t.get("_flat") for t in batch]
        ocr_f = [f["ocr"] for f in flats] if all(f is not None for f in flats) else None
        od_f = [f["od"] for f in flats] if ocr_f is not None else None
        ocr_list = self.item_collate([t.get("ocr") for t in batch], o["max_ocr_len"], o["max_ocr_bert_len"], o["max_ocr_num"], ocr_f)
        od_list = self.item_collate([t.get("od") for t in batch], o["max_od_len"], o["max_od_bert_len"], o["max_od_num"], od_f)
        gt_list = self.gt_collate([t["gt"] for t in batch])
        if self.prepare_index:
            q_list["_ruart_host_index"] = BatchIndex(q_list, ocr_list, od_list, o)
        return q_list, ocr_list, od_list, gt_list, [t["extra_info"] for t in batch]

    def gt_collate(self, gt_list):
        return torch.cat(gt_list, dim=0)

    @staticmethod
    def _pad_rows(rows, width):
        """Ragged python lists -> zero-padded (len(rows), width) int64 - one flat gather instead of a numpy assignment per row
        (a batch has ~9 000 item rows per key).  A row longer than ``width`` is an error, as in the reference (:497)."""
        n = len(rows)
        out = np.zeros((n, width), dtype=np.int64)
        if n == 0:
            return torch.from_numpy(out)
        lens = np.fromiter((len(r) for r in rows), dtype=np.int64, count=n)
        if lens.max() > width:
            bad = int(np.argmax(lens > width))
            raise ValueError("row %d holds %d ids, the padded width is %d" % (bad, int(lens[bad]), width))
        total = int(lens.sum())
        if total:
            flat = np.fromiter(itertools.chain.from_iterable(rows), dtype=np.int64, count=total)
            starts = np.cumsum(lens) - lens
            out[np.repeat(np.arange(n), lens), np.arange(total) - np.repeat(starts, lens)] = flat
        return torch.from_numpy(out)

    @staticmethod
    def _scatter(flat, lens, width, tail=()):
        """Values back to back + a length per row -> zero-padded (rows, width[, *tail]) array."""
        n = len(lens)
        out = np.zeros((n, width) + tuple(tail), dtype=flat.dtype)
        total = int(lens.sum())
        if total:
            if lens.max() > width:
                bad = int(np.argmax(lens > width))
                raise ValueError("row %d holds %d entries, the padded width is %d" % (bad, int(lens[bad]), width))
            starts = np.cumsum(lens) - lens
            out[np.repeat(np.arange(n), lens), np.arange(total) - np.repeat(starts, lens)] = flat
        return out

    def item_collate(self, item_list, max_len, max_bert_len, max_num, flats=None):
        """``flats``: the per-sample flat arrays ``VQA_Dataset`` attaches (``sample['_flat'][group]``); with them the id matrices
        are a concatenation and one scatter per key instead of a walk over every item dict.  Compact (cached) samples carry ONLY
        the flat arrays: ``item_list`` entries are then None and the list-valued fields are rebuilt from the arrays."""
        res = {}
        wk = "fasttext" if "FastText" in self.opt else "glove"
        fast = flats is not None and all(f for f in flats)
        have_lists = item_list is not None and all(s_ is not None for s_ in item_list)
        if not fast and not have_lists:
            raise ValueError("samples carry neither item lists nor flat arrays")
        B = len(flats) if fast else len(item_list)
        flat = [it for sample in item_list for it in sample] if have_lists else None
        keys = list(item_list[0][0].keys()) if have_lists else list(flats[0].keys())
        counts = [len(s_) for s_ in item_list] if have_lists else [len(f[wk][1]) for f in flats]
        for k in keys:
            if "offset" in k:
                if fast:
                    vals = np.concatenate([f[k][0] for f in flats])
                    lens = np.concatenate([f[k][1] for f in flats])
                if fast:                               # list-like view over the two arrays (see Offsets)
                    res[k] = Offsets(vals, lens)
                    res["_" + k + "_arr"] = self._scatter(vals, lens, max_len, tail=(2,)) if lens.max() <= max_len \
                        else offsets_to_array(res[k].tolist(), len(lens), max_len)
                else:
                    res[k] = [it[k] for it in flat]
            elif k == "position":
                pos = torch.zeros(B, max_num, 8)
                for b in range(B):
                    pos[b, :counts[b]] = torch.from_numpy(flats[b][k]) if fast else \
                        torch.tensor([it[k] for it in item_list[b]], dtype=torch.float32)
                res[k] = pos
            else:
                width = max_bert_len if k in ("bert", "bert_only") else max_len
                if fast:
                    res[k] = torch.from_numpy(self._scatter(np.concatenate([f[k][0] for f in flats]),
                                                            np.concatenate([f[k][1] for f in flats]), width))
                else:
                    res[k] = self._pad_rows([it[k] for it in flat], width)
        for k in [k for k in res if k in ("glove", "fasttext", "phoc", "bert", "bert_only")]:
            res[k + "_mask"] = ~res[k].eq(0)
        res["num_cnt"] = counts
        if fast:
            res["len_cnt"] = [f[wk][1].tolist() for f in flats]
        else:
            res["len_cnt"] = [[len(it[wk]) for it in sample] for sample in item_list]
        return res

    def que_collate(self, q_list, max_len, max_bert_len):
        res = {}
        for k in q_list[0].keys():
            if k in ("img_features", "img_spatials"):
                res[k] = torch.cat([t[k] for t in q_list], dim=0)
            elif "offset" in k:
                res[k] = [t[k] for t in q_list]
            else:
                res[k] = self._pad_rows([t[k] for t in q_list], max_bert_len if k in ("bert", "bert_only") else max_len)
                if k in ("fasttext", "glove", "phoc", "bert", "bert_only"):
                    res[k + "_mask"] = ~res[k].eq(0)
        return res


# ---------------------------------------------------------------------------------------------------------
def offsets_to_array(offsets, n_rows, width):
    """python list [rows][words][2] -> (rows, width, 2) int64, zero padded; tolerates the reference's flat
    ``[1, 1]`` for an item without words (Utils/VQA_Dataset.py:426-427)."""
    arr = np.zeros((n_rows, width, 2), dtype=np.int64)
    rows = [() if (len(r) and not isinstance(r[0], (list, tuple))) else (r if len(r) <= width else r[:width]) for r in offsets]
    lens = np.fromiter((len(r) for r in rows), dtype=np.int64, count=len(rows))
    total = int(lens.sum())
    if total:
        flat = np.fromiter(itertools.chain.from_iterable(itertools.chain.from_iterable(rows)), dtype=np.int64, count=2 * total)
        starts = np.cumsum(lens) - lens
        arr[np.repeat(np.arange(len(rows)), lens), np.arange(total) - np.repeat(starts, lens)] = flat.reshape(total, 2)
    return arr


class ItemIndex:
    """Index vectors for one item group (OCR tokens or detected objects).  Host arrays; ``.dev`` holds the int64
    device copies after ``to_device``."""

    def __init__(self, items, word_key, max_num):
        num_cnt = np.asarray(items["num_cnt"], dtype=np.int64)
        lens = np.concatenate([np.asarray(l, dtype=np.int64) for l in items["len_cnt"]]) if len(num_cnt) else np.zeros(0, np.int64)
        if (lens < 1).any():
            raise ValueError("every item must hold at least one word (Utils/VQA_Dataset.py:319-320 drops empty ones)")
        B, N = len(num_cnt), len(lens)
        if (num_cnt > max_num).any():
            raise ValueError("more items than position rows")
        self.B, self.N, self.max_num = B, N, max_num
        self.W = int(lens.sum())
        word_start = np.concatenate([[0], np.cumsum(lens)[:-1]])
        item_start = np.concatenate([[0], np.cumsum(num_cnt)[:-1]])
        self.item_of_word = np.repeat(np.arange(N), lens)
        self.pos_in_item = np.arange(self.W) - word_start[self.item_of_word]
        sample_of_item = np.repeat(np.arange(B), num_cnt)
        slot = np.arange(N) - item_start[sample_of_item]
        self.sample_of_word = sample_of_item[self.item_of_word]
        wps = np.bincount(self.sample_of_word, minlength=B)
        tok_start = np.concatenate([[0], np.cumsum(wps)[:-1]])
        self.tok_in_sample = np.arange(self.W) - tok_start[self.sample_of_word]
        self.Tmax = int(wps.max()) if B else 0
        self.flat_tok = self.sample_of_word * max(self.Tmax, 1) + self.tok_in_sample      # row of a word in the (B * Tmax, D) pre-align matrix
        # multi2one schedule: items sorted by word count (descending, stable); at step s the first n_active[s] are alive
        order = np.argsort(-lens, kind="stable")
        ls = lens[order]
        self.maxlen = int(ls[0]) if N else 0
        self.n_active = [int((ls > s).sum()) for s in range(self.maxlen)]
        self.step_rows = np.concatenate([word_start[order[:n]] + s for s, n in enumerate(self.n_active)]) if N else np.zeros(0, np.int64)
        self.sorted_sample = sample_of_item[order]
        self.sorted_slot = slot[order]
        self.flat_slot = self.sorted_sample * max_num + self.sorted_slot                   # row of an item in the (B * max_num, D) matrix
        self.num_cnt = num_cnt
        Lw = items[word_key].shape[1]
        self.Lw = Lw
        self.flat_word = self.item_of_word * Lw + self.pos_in_item          # index into the flattened (N, Lw) id matrices
        # the real words' ids of every id matrix the embedding layers look up, gathered here on the host: the device side did
        # ``items[key].reshape(-1)[flat_word]`` per table and step (ten small index launches per step, round 5 op table)
        self.word_ids = {}
        for k in ("fasttext", "glove", "phoc", "pos", "ent"):
            v = items.get(k)
            if v is not None and tuple(v.shape) == (N, Lw):
                self.word_ids[k] = _np(v).reshape(-1)[self.flat_word].astype(np.int64)
        mask = np.zeros((B, max_num), dtype=np.uint8)
        mask[np.arange(max_num)[None, :] < num_cnt[:, None]] = 1
        self.mask = mask
        self.dev = None
        self.emb_sort = {}

    _FIELDS = ("item_of_word", "sample_of_word", "tok_in_sample", "step_rows", "sorted_sample", "sorted_slot", "flat_word", "flat_tok",
               "flat_slot")

    def fields(self):
        return list(self._FIELDS) + ["ids_" + k for k in sorted(self.word_ids)]

    def pack_host(self):
        return [getattr(self, f).astype(np.int64) for f in self._FIELDS] + [self.word_ids[k] for k in sorted(self.word_ids)]

    def bind(self, tensors):
        self.dev = dict(zip(self.fields(), tensors))


def _sort_ids(ids, padding_idx=None, max_seg=64):
    """The sort ops.embedding's backward works from, int32: positions grouped by id (stable; the padding row - its gradient is defined as
    zero - left out).  Returns (order, seg_start, seg_row) - seg_start[s] .. seg_start[s+1] is the slice of order[] that hit table row
    seg_row[s] - or, when some row has more than ``max_seg`` occurrences (a frequent word, [CLS] / [SEP], the first positions: one
    workgroup would add thousands of rows one after the other), the two-level form (order, sub_start, row_first, row_id): sub-segments of
    at most max_seg occurrences, and row r = the sum of sub-segments row_first[r] .. row_first[r+1]."""
    ids = np.asarray(ids, dtype=np.int64)
    order = np.argsort(ids, kind="stable")
    sid = ids[order]
    if padding_idx is not None:
        keep = sid != padding_idx
        order, sid = order[keep], sid[keep]
    if len(sid) == 0:
        return np.zeros(0, np.int32), np.zeros(1, np.int32), np.zeros(0, np.int32)
    first = np.concatenate([[True], sid[1:] != sid[:-1]])
    starts = np.flatnonzero(first)
    bounds = np.concatenate([starts, [len(sid)]])
    lens = np.diff(bounds)
    if lens.max() <= max_seg:
        return order.astype(np.int32), bounds.astype(np.int32), sid[starts].astype(np.int32)
    n_sub = (lens + max_seg - 1) // max_seg                           # sub-segments per row
    row_first = np.concatenate([[0], np.cumsum(n_sub)])
    row_of_sub = np.repeat(np.arange(len(lens)), n_sub)
    k_in_row = np.arange(row_first[-1]) - row_first[row_of_sub]
    sub_start = np.concatenate([starts[row_of_sub] + k_in_row * max_seg, [len(sid)]])
    return order.astype(np.int32), sub_start.astype(np.int32), row_first.astype(np.int32), sid[starts].astype(np.int32)


class BatchIndex:
    """Everything ``SDNet.forward`` needs besides the reference's own batch tensors.

    Two stages: the constructor does the HOST work (numpy: index vectors, the packed BERT stream, pooling spans) and leaves a
    picklable object - ``VQA_collate(opt, prepare_index=True)`` runs it inside DataLoader workers; ``to(device)`` ships the
    three flat buffers (three H2D copies) and builds the device views.  Passing ``device`` to the constructor does both."""

    def __init__(self, q_list, ocr_list, od_list, opt, device=None, bert=None, pack=None, mfma_long=None):
        wk_o, wk_q = opt["ocr_emb_initial"], opt["q_emb_initial"]
        self.ocr = ItemIndex(ocr_list, wk_o, ocr_list["position"].size(1))
        self.od = ItemIndex(od_list, wk_o, od_list["position"].size(1))
        self.device = None
        self.ocr_mask = self.od_mask = None
        # BERT: one packed pass over question + OCR items + object items, and the pooling descriptors
        self.packed = None
        self.spans = None
        self._spans_host = None
        if bert is not None:
            pack, mfma_long = bert.pack, bert.weights.dtype != 0
        if "BERT" in opt or bert is not None:
            from .bert import PackedTokens, word_spans
            from . import precision_of
            pack = (not opt.get("bert_no_pack", False)) if pack is None else pack
            # the MFMA long-sequence attention kernel serves the plain 16-bit modes only (fp32 storage modes: 64-query VALU blocks)
            mfma_long = (precision_of(opt) in ("fp16", "bf16")) if mfma_long is None else mfma_long
            # frozen, deterministic encoder: identical sequences are encoded once and the last layer runs on pooled rows only (both
            # bit-neutral; opt['bert_dedup'] / opt['bert_last_rows'] = False switch them off)
            frozen = "LOCK_BERT" in opt and not opt.get("bert_frozen_dropout")
            dedup = frozen and bool(opt.get("bert_dedup", True)) and bool(pack)
            self.plan = (bool(pack), bool(mfma_long), dedup)
            groups = [(q_list["bert"], q_list["bert_mask"]), (ocr_list["bert"], ocr_list["bert_mask"]), (od_list["bert"], od_list["bert_mask"])]
            self.packed = PackedTokens(groups, None, pack=pack, mfma_long=mfma_long, dedup=dedup)
            if "LOCK_BERT" not in opt:
                self.packed.prepare_embedding_sorts()      # trainable encoder: its embedding gradients take the sorted, ordered path
            spans = []
            for g, (items, wk) in enumerate(((q_list, wk_q), (ocr_list, wk_o), (od_list, wk_o))):
                wm = _np(items[wk + "_mask"])
                arr = items.get("_bert_offsets_arr")          # attached by the collate's fast path
                if arr is None:
                    arr = items.get("bert_offsets_arr")
                if arr is None:
                    arr = offsets_to_array(items["bert_offsets"], wm.shape[0], wm.shape[1])
                s, l, d, rows = word_spans(self.packed, g, None, wm, offsets_arr=arr)
                if g > 0:
                    # destination = packed word row (item-major), not the padded n * Lw + j slot
                    idx = self.ocr if g == 1 else self.od
                    lut = np.full(wm.shape[0] * wm.shape[1], -1, dtype=np.int64)
                    lut[idx.flat_word] = np.arange(idx.W)
                    d = lut[d]
                    keep = d >= 0                      # words beyond len_cnt never reach multi2one's consumed state
                    s, l, d, rows = s[keep], l[keep], d[keep].astype(np.int32), idx.W
                spans.append((s, l, d, rows))
            # rows of the packed stream some word span reads (Models/Bert/Bert.py:153-165 pools word pieces, never [CLS] / [SEP]):
            # the only rows the LAST encoder layer has to produce.  last_start[w] = the span's first row after compaction to them.
            T = self.packed.T
            self._n_last = 0
            last_rows = np.zeros(0, dtype=np.int32)
            last_start = [np.zeros(0, dtype=np.int32)] * len(spans)
            # (the C side compacts only when a layer precedes the last one, bert_forward.hip last_layer_rows: same rule here, so the
            # compacted starts are never paired with an uncompacted layer)
            # layer count: the loaded encoder's when the model hands it over, else the conf's (``Bert`` accepts only the 12 / 24 layers
            # its BERT / BERT_LARGE flag names, so a checkpoint read from bert_config.json on disk agrees with this default)
            if bert is not None:
                n_layers = int(bert.weights.n_layers)
            else:
                n_layers = int((opt.get("bert_config") or {}).get("num_hidden_layers", 24 if "BERT_LARGE" in opt else 12))
            if frozen and opt.get("bert_last_rows", True) and T > 0 and n_layers >= 2:
                mark = np.zeros(T + 1, dtype=np.int64)
                for (s_, l_, _, _) in spans:
                    np.add.at(mark, s_, 1)
                    np.add.at(mark, s_.astype(np.int64) + l_, -1)
                covered = np.cumsum(mark[:T]) > 0
                if 0 < covered.sum() < T:
                    last_rows = np.nonzero(covered)[0].astype(np.int32)
                    remap = np.cumsum(covered) - 1
                    last_start = [remap[t[0]].astype(np.int32) for t in spans]
                    self._n_last = len(last_rows)
            if self._n_last == 0:
                last_start = [np.zeros(0, dtype=np.int32)] * len(spans)
            self._spans_host = (np.concatenate([np.concatenate(list(t[:3]) + [ls]) for t, ls in zip(spans, last_start)] + [last_rows]).astype(np.int32),
                                [(len(t[0]), t[3], int(np.asarray(t[1], dtype=np.int64).sum())) for t in spans])
            if "LOCK_BERT" in opt and not opt.get("bert_frozen_dropout"):
                self.packed.group_index = None         # (N, L) maps were only needed for the spans: keep the pickle small
                                                       # (the trainable encoder pads its attention per group from them)
        # sort of every embedding lookup's ids (word / POS / entity tables), for ops.embedding's backward
        self._emb_host = {}
        q_keys = [k for k in ("glove", "fasttext", "phoc", "pos", "ent") if k in q_list and isinstance(q_list[k], torch.Tensor)]
        for k in q_keys:
            self._emb_host[("q", k)] = _sort_ids(_np(q_list[k]).reshape(-1), 1 if k in ("glove", "fasttext", "phoc") else None)
        for name, items, idx in (("ocr", ocr_list, self.ocr), ("od", od_list, self.od)):
            for k in ("glove", "fasttext", "phoc", "pos", "ent"):
                if k in items and isinstance(items[k], torch.Tensor):
                    self._emb_host[(name, k)] = _sort_ids(_np(items[k]).reshape(-1)[idx.flat_word],
                                                          1 if k in ("glove", "fasttext", "phoc") else None)
        self.emb_sort = {}
        if device is not None:
            self.to(device)

    # -- host staging: the five flat buffers ``to`` ships, as torch tensors (so a DataLoader can pin them) -----------------------
    def _staged(self):
        h = self.__dict__.get("_h")
        if h is None:
            items = self.ocr.pack_host() + self.od.pack_host()
            h = {"item_sizes": [len(a) for a in items],
                 "items": torch.from_numpy(np.concatenate(items)) if sum(len(a) for a in items) else torch.zeros(0, dtype=torch.long),
                 "ocr_mask": torch.from_numpy(self.ocr.mask), "od_mask": torch.from_numpy(self.od.mask)}
            if self._emb_host:
                h["emb"] = torch.from_numpy(np.concatenate([np.concatenate(self._emb_host[k]) for k in self._emb_host]).astype(np.int32))
            if self.packed is not None:
                h["packed"] = torch.from_numpy(self.packed.host)
                h["spans"] = torch.from_numpy(self._spans_host[0])
            self._h = h
        return h

    def pin_memory(self):
        """Called by ``DataLoader(pin_memory=True)`` on its pinning thread: page-locks the staged buffers so that ``to`` is five
        asynchronous copies instead of five blocking ones."""
        h = self._staged()
        for k, v in h.items():
            if isinstance(v, torch.Tensor):
                h[k] = v.pin_memory()
        return self

    def to(self, device):
        if self.device is not None and self.device == torch.device(device):
            return self
        self.device = torch.device(device)
        h = self._staged()
        buf = h["items"].to(self.device, non_blocking=True)
        parts = list(torch.split(buf, h["item_sizes"]))
        n = len(self.ocr.fields())
        self.ocr.bind(parts[:n])
        self.od.bind(parts[n:])
        self.ocr_mask = h["ocr_mask"].to(self.device, non_blocking=True)
        self.od_mask = h["od_mask"].to(self.device, non_blocking=True)
        if self._emb_host:
            keys = list(self._emb_host)
            dev = h["emb"].to(self.device, non_blocking=True)
            o = 0
            for k in keys:
                parts = []
                for a in self._emb_host[k]:
                    parts.append(dev[o:o + len(a)])
                    o += len(a)
                self.emb_sort[k] = tuple(parts)
            self.ocr.emb_sort = {k[1]: v for k, v in self.emb_sort.items() if k[0] == "ocr"}
            self.od.emb_sort = {k[1]: v for k, v in self.emb_sort.items() if k[0] == "od"}
        if self.packed is not None:
            self.packed.bind(self.device, h["packed"])
            shapes = self._spans_host[1]
            dev = h["spans"].to(self.device, non_blocking=True)
            o = 0
            self.spans = []
            n_last = getattr(self, "_n_last", 0)
            self.span_pieces = [n_pieces for _, _, n_pieces in shapes]     # host-side sum of each group's span lengths
            for W, rows, _ in shapes:
                last = dev[o + 3 * W:o + 4 * W] if n_last else None
                self.spans.append((dev[o:o + W], dev[o + W:o + 2 * W], dev[o + 2 * W:o + 3 * W], rows, last))
                o += (4 if n_last else 3) * W
            if n_last:
                self.packed.set_last_rows(dev[o:o + n_last])
        return self


def to_device(batch, device):
    """The reference's ``SDNetTrainer.ToCUDA`` (Models/SDNetTrainer.py:208-230) without its per-tensor isnan() sync:
    moves id / mask / position tensors of the three dicts and the targets."""
    keys = {"bert", "bert_only", "bert_mask", "bert_only_mask", "fasttext", "fasttext_mask", "phoc", "phoc_mask", "glove",
            "glove_mask", "ent", "pos", "position", "img_features", "img_spatials"}
    out = []
    for idx, item in enumerate(batch):
        if idx < 3:
            item = dict(item)
            for k in list(item.keys()):
                if k in keys and isinstance(item[k], torch.Tensor):
                    item[k] = item[k].to(device, non_blocking=True)
        elif idx == 3 and item is not None:
            item = item.to(device, non_blocking=True)
        out.append(item)
    return out
