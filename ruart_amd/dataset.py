"""Per-sample producer of the batch API: ``VQA_Dataset`` over the reference's preprocessed records, and the readers of the two
msgpack artefacts its training loop opens (``*-preprocessed.msgpack``, ``train_meta.msgpack``).

Reference: ``Utils/VQA_Dataset.py:13-437`` (dataset), ``Utils/CoQAPreprocess.py:481-501`` (``load_data``),
``Models/SDNetTrainer.py:83-105`` (how the trainer wires them).  A sample is

    {'q': {glove, pos, ent, bert, bert_offsets}, 'ocr': [item...], 'od': [item...], 'gt': FloatTensor(1, max_ocr_num [+1]),
     'extra_info': {q_id, answers, ocr_list, image_path}}            item = {fasttext, pos, ent, bert, bert_offsets, position}

which ``batch.VQA_collate`` turns into the 5-tuple the model takes.  Everything here is integer / string work on the host and is
pinned bit-exact against the reference's own class on a synthetic record set (``tests/golden/dataset_*.json``).

Not carried over (outside SURVEY section 8): the ``DEBUG`` length histograms, grid image features (``img_feature``) and the
fixed-answer vocabulary (``fixed_answers``) - asking for them raises."""
import itertools
import logging
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from .metrics import note_stvqa, note_textvqa
from .tokenization import BertTokenizer

log = logging.getLogger(__name__)

_NOREAD = "answering does not require reading text in the image"


def load_msgpack(path):
    """``msgpack.load(f, encoding='utf8')`` of the reference (SDNetTrainer.py:84-87), in the current msgpack API."""
    import msgpack
    with open(path, "rb") as f:
        return msgpack.load(f, raw=False, strict_map_key=False)


def load_meta(opt, folder=None):
    """``CoQAPreprocess.load_data`` (CoQAPreprocess.py:481-501): vocabulary and the initial word-vector tables from
    ``train_meta.msgpack``; sets ``vocab_size`` / ``vocab_dim`` / ``char_vocab_size`` in ``opt`` the way the reference does."""
    meta = load_msgpack(os.path.join(folder if folder is not None else opt["FEATURE_FOLDER"], "train_meta.msgpack"))
    opt["char_vocab_size"] = len(meta["char_vocab"])
    emb = {}
    for flag, key in (("GLOVE", "glove_embedding"), ("FastText", "fast_embedding")):
        if flag in opt:
            emb[key] = torch.Tensor(meta[key])
            opt["vocab_size"], opt["vocab_dim"] = emb[key].size(0), emb[key].size(1)
    if "PHOC" in opt:
        emb["phoc_embedding"] = torch.Tensor(meta["phoc_embedding"])
    return meta["vocab"], meta["char_vocab"], emb


def flatten_items(items):
    """List of item dicts -> {key: (values back to back, length per item)} for the id keys, ``bert_offsets`` as an (n_words, 2)
    array with the word count per item, ``position`` as (n_items, 8) float32."""
    out = {}
    n = len(items)
    if n == 0:
        return out
    for k, v0 in items[0].items():
        if k == "position":
            out[k] = np.asarray([it[k] for it in items], dtype=np.float32).reshape(n, 8)
        elif "offset" in k:
            rows = [() if (len(it[k]) and not isinstance(it[k][0], (list, tuple))) else it[k] for it in items]
            lens = np.fromiter((len(r) for r in rows), dtype=np.int64, count=n)
            tot = int(lens.sum())
            flat = np.fromiter(itertools.chain.from_iterable(itertools.chain.from_iterable(rows)), dtype=np.int64, count=2 * tot)
            out[k] = (flat.reshape(tot, 2), lens)
        else:
            lens = np.fromiter((len(it[k]) for it in items), dtype=np.int64, count=n)
            flat = np.fromiter(itertools.chain.from_iterable(it[k] for it in items), dtype=np.int64, count=int(lens.sum()))
            out[k] = (flat, lens)
    return out


class VQA_Dataset(Dataset):
    def __init__(self, data, opt, mode="train", image_features=None, fixed_answers_entry=None):
        assert mode in ("train", "dev", "test")
        if "img_feature" in opt or image_features is not None:
            raise NotImplementedError("grid image features are outside the accelerated path (SURVEY section 8)")
        if "fixed_answers" in opt or fixed_answers_entry is not None:
            raise NotImplementedError("the fixed-answer vocabulary is outside the accelerated path (SURVEY section 8)")
        self.opt = opt
        self.mode = mode
        dropped = []
        self.data = []
        for datum in data:                               # VQA_Dataset.py:19-27: no question words, or no answers outside test
            if len(datum["annotated_question"]["word"]) == 0 or (mode != "test" and len(datum["orign_answers"]) == 0):
                dropped.append(datum["question_id"])
            else:
                self.data.append(datum)
        log.info("Remove %d samples for empty question or answers: %s", len(dropped), dropped)
        self.ocr_name_list = opt["ocr_name_list"].split(",")
        self.od_name_list = opt["od_name_list"].split(",")
        self.q_embedding = opt["q_embedding"].split(",")
        self.ocr_embedding = opt["ocr_embedding"].split(",")
        self.score_name = opt["score_name"]
        self.max_ocr_num, self.max_od_num = opt["max_ocr_num"], opt["max_od_num"]
        self.max_ocr_len, self.max_od_len, self.max_q_len = opt["max_ocr_len"], opt["max_od_len"], opt["max_q_len"]
        if "ES_ocr" in opt:                              # the retrieved candidates go first and are cut to ES_ocr_len
            self.ocr_name_list = [opt["ES_ocr"]] + self.ocr_name_list
            self.es_ocr_len = int(opt["ES_ocr_len"])
            self.es_sort_way = opt["ES_sort_way"]
        self.bert_tokenizer = None
        self._bert_memo = {}
        self._cache = {} if opt.get("ruart_cache_samples") else None   # finished (compact) samples by index: epochs revisit them
        if "BERT" in opt:
            key = "BERT_large_tokenizer_file" if "BERT_LARGE" in opt else "BERT_tokenizer_file"
            self.bert_tokenizer = BertTokenizer.from_pretrained(os.path.join(opt["datadir"], opt[key]))
            self._cls, self._sep = self.bert_tokenizer.vocab["[CLS]"], self.bert_tokenizer.vocab["[SEP]"]
        self._word_memo = {}
        # the field set flatten_items would produce for the shipped conf; anything else takes the general route
        self._direct_flat = (self.ocr_embedding == ["fasttext", "pos", "ent", "bert"] and "bert" in self.q_embedding
                             and "bert_only" not in self.q_embedding and self.bert_tokenizer is not None)

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        if self._cache is not None and index in self._cache:
            return self._cache[index]
        datum = self.data[index]
        dedup = "remove_same" in self.opt
        ocr_items = self.get_list_from_datum(datum, self.ocr_name_list, od_ocr="ocr", remove_same=dedup)
        od_items = self.get_list_from_datum(datum, self.od_name_list, od_ocr="od", remove_same=dedup)
        datum["annotated_question"]["original"] = datum["question"].lower()
        q = self.get_item_embedding(datum["annotated_question"], self.q_embedding)
        ocr_items = ocr_items[:self.max_ocr_num]
        od_items = od_items[:self.max_od_num]
        answers = datum.get("orign_answers")
        if self._cache is not None and self._direct_flat:
            # compact samples of the standard conf: the flat arrays are filled straight from the records, no per-item dicts
            sample = {"q": q, "gt": self.get_label(ocr_items, q_id=datum["question_id"], answers=answers),
                      "extra_info": {"q_id": datum["question_id"], "answers": answers,
                                     "ocr_list": [t["original"] for t in ocr_items], "image_path": datum["filename"]},
                      "_flat": {"ocr": self._flat_direct(ocr_items), "od": self._flat_direct(od_items)}}
            self._cache[index] = sample
            return sample
        sample = {"q": q,
                  "ocr": self.get_list_embedding(ocr_items, self.ocr_embedding),
                  "od": self.get_list_embedding(od_items, self.ocr_embedding),
                  "gt": self.get_label(ocr_items, q_id=datum["question_id"], answers=answers),
                  "extra_info": {"q_id": datum["question_id"], "answers": answers,
                                 "ocr_list": [t["original"] for t in ocr_items], "image_path": datum["filename"]}}
        # the same content once more as flat arrays per key (ids of all items back to back + a length per item): what
        # ``VQA_collate`` concatenates instead of walking ~9 000 item dicts per batch.  The reference-format lists above stay.
        sample["_flat"] = {"ocr": flatten_items(sample["ocr"]), "od": flatten_items(sample["od"])}
        if self._cache is not None:
            # compact form: the flat arrays ARE the items (~25 KB per sample instead of ~200 KB of python lists); the collate
            # rebuilds the list-valued fields of the batch from them.  Returned on the first visit too, so that a cached dataset
            # always hands out the same kind of sample.
            sample = {k: v for k, v in sample.items() if k not in ("ocr", "od")}
            self._cache[index] = sample
        return sample

    def _flat_direct(self, items):
        """``flatten_items(self.get_list_embedding(items, ...))`` without building the item dicts (same keys, order, dtypes)."""
        n = len(items)
        wid, pos, ent, bert, offs, position = [], [], [], [], [], []
        len_w, len_b = np.empty(n, np.int64), np.empty(n, np.int64)
        bertify = self.bertify
        for i, item in enumerate(items):
            w = item["object"] if "object" in item else item["word"]
            ids = w["wordid"]
            wid += ids
            pos += w["pos_id"]
            ent += w["ent_id"]
            b, o = bertify(w["word"])
            bert += b
            offs += o
            len_w[i] = len(ids)
            len_b[i] = len(b)
            position.append(item["pos"])
        len_p = np.fromiter((len(it["object"]["pos_id"] if "object" in it else it["word"]["pos_id"]) for it in items), np.int64, n)
        len_e = np.fromiter((len(it["object"]["ent_id"] if "object" in it else it["word"]["ent_id"]) for it in items), np.int64, n)
        len_o = np.fromiter((len(it["object"]["word"] if "object" in it else it["word"]["word"]) for it in items), np.int64, n)
        return {"fasttext": (np.array(wid, dtype=np.int64), len_w), "pos": (np.array(pos, dtype=np.int64), len_p),
                "ent": (np.array(ent, dtype=np.int64), len_e), "bert": (np.array(bert, dtype=np.int64), len_b),
                "bert_offsets": (np.array(offs, dtype=np.int64).reshape(-1, 2), len_o),
                "position": np.asarray(position, dtype=np.float32).reshape(n, 8)}

    # -- candidate lists (VQA_Dataset.py:293-349) ---------------------------------------------------------------
    def get_list_from_datum(self, datum, name_list, od_ocr="ocr", remove_same=False):
        """Concatenate the named detector outputs, drop items without words and (optionally) repeated strings, cut to
        max - 1 and append the ``<OCR>`` / ``<OD>`` sentinel (word id 3 / 4).  As in the reference this normalises the
        record in place: ``original`` is lower-cased, an object's ``word`` aliases its ``object`` entry, and the retrieved
        (``ES_ocr``) list is sorted and truncated."""
        assert od_ocr in ("od", "ocr")
        seen, res = set(), []
        for name in name_list:
            es = "ES_ocr" in self.opt and name == self.opt["ES_ocr"]
            if es:
                if self.es_sort_way == "frequency":
                    datum[name].sort(key=lambda x: x["cnt"], reverse=True)
                elif self.es_sort_way == "relevance":
                    datum[name].sort(key=lambda x: x["idx"])
                else:
                    raise AssertionError("es_sort_way is wrong")
                datum[name] = datum[name][:self.es_ocr_len]
            for item in datum[name]:
                if od_ocr == "od":
                    item["word"] = item["object"]
                if len(item["word"]["word"]) == 0:
                    continue
                k = item["original"].lower()
                item["original"] = k
                if es:
                    res.append(item)
                    continue
                if remove_same and k in seen:
                    continue
                seen.add(k)
                res.append(item)
        limit = (self.max_od_num if od_ocr == "od" else self.max_ocr_num) - 1
        res = res[:limit]
        tok = "<OCR>" if od_ocr == "ocr" else "<OD>"
        res.append({"word": {"word": [tok], "wordid": [3 if od_ocr == "ocr" else 4], "pos_id": [0], "ent_id": [0]},
                    "pos": [0] * 8, "original": tok, "ANLS": 0.0, "ACC": 0.0})
        return res

    # -- soft labels (VQA_Dataset.py:211-290) -------------------------------------------------------------------
    def get_label(self, ocr_list, q_id=None, answers=None):
        if self.score_name not in ocr_list[0]:
            return None, None                            # (sic) the reference returns a pair when the records carry no scores
        gt = [t[self.score_name] for t in ocr_list]
        n_ynu = 0
        if "label_yesno" in self.opt:
            note = note_stvqa if self.score_name == "ANLS" else note_textvqa
            gt = [note(answers, _NOREAD), note(answers, "yes"), note(answers, "no")] + gt
            n_ynu = 3
        best, best_idx = -1, -1
        for i, t in enumerate(gt):                       # first maximum
            if t > best:
                best, best_idx = t, i
        way = self.opt["lable_way"]
        if way == "lable_all_with_threshold":
            gt = [t if t >= self.opt["score_threshold"] else 0 for t in gt]
        elif way == "lable_one_offical":
            floor = {"ANLS": 0.5, "ACC": 0.3}.get(self.score_name)
            if floor is not None:
                gt = [t if i == best_idx and best >= floor else 0 for i, t in enumerate(gt)]
        elif way == "lable_one":
            gt = [t if i == best_idx else 0 for i, t in enumerate(gt)]
        elif way != "lable_all":
            raise AssertionError("lable_way is wrong")
        out = torch.zeros(1, n_ynu + self.max_ocr_num)
        out[0, :len(gt)] = torch.FloatTensor(gt)
        if "label_no_answer" in self.opt:
            out = torch.cat([out, torch.full((1, 1), 1.0 if best < 0.1 else 0.0)], dim=1)
        return out

    # -- id fields (VQA_Dataset.py:353-413) ---------------------------------------------------------------------
    def get_item_embedding(self, item, embedding_list, original=None):
        res = {}
        for name in ("fasttext", "phoc", "glove"):
            if name in embedding_list:
                res[name] = item["wordid"]
        if "pos" in embedding_list:
            res["pos"] = item["pos_id"]
        if "ent" in embedding_list:
            res["ent"] = item["ent_id"]
        if "bert" in self.q_embedding:                   # (sic) the question's list decides for items too
            res["bert"], res["bert_offsets"] = self.bertify(item["word"])
        if "bert_only" in self.q_embedding:
            text = item["original"] if "original" in item else original
            assert text is not None
            res["bert_only"] = self.bertify(text)[0]
        return res

    def get_list_embedding(self, item_list, embedding_list):
        res = []
        for item in item_list:
            words = item["object"] if "object" in item else item["word"]
            tmp = self.get_item_embedding(words, embedding_list, original=item["original"])
            tmp["position"] = item["pos"]
            res.append(tmp)
        return res

    def bertify(self, words):
        """[CLS] + word pieces + [SEP] as ids, and for a word list the [start, end) piece span of every word
        (VQA_Dataset.py:415-436).  Two memo levels - a word's piece ids, and whole word lists (scene-text items repeat across
        candidates and samples); a memoised result is returned as the same list objects again: treat them as read-only."""
        if self.bert_tokenizer is None:
            return None
        tok = self.bert_tokenizer
        if isinstance(words, list):
            key = tuple(words)
            hit = self._bert_memo.get(key)
            if hit is None:
                wmemo = self._word_memo
                ids, offsets, n = [self._cls], [], 1
                for word in words:
                    pieces = wmemo.get(word)
                    if pieces is None:
                        pieces = tok.convert_tokens_to_ids(tok.tokenize(word))
                        if len(wmemo) < 1 << 20:
                            wmemo[word] = pieces
                    m = n + len(pieces)
                    offsets.append([n, m])
                    ids += pieces
                    n = m
                if not words:
                    offsets = [1, 1]
                ids.append(self._sep)
                hit = (ids, offsets)
                if len(self._bert_memo) < 1 << 20:
                    self._bert_memo[key] = hit
            return hit
        if isinstance(words, str):
            bpe = ["[CLS]"] + tok.tokenize(words) + ["[SEP]"]
            return tok.convert_tokens_to_ids(bpe), []
        raise AssertionError("BERT tokenizer is wrong")
