"""Host-side logic (CPU only): conf reader, collate mirror, sampler, batch index vectors, BERT packing plan,
answer metrics.  Collate and sampler are checked against outputs of the reference's own VQA_collate / VQA_Sampler
(tests/golden/host.npz, generator oracle/gen_golden.py::gen_host)."""
import os

import numpy as np
import pytest
import torch

from ruart_amd import synth
from ruart_amd.arguments import Arguments, default_opt
from ruart_amd.batch import ItemIndex, VQA_collate, offsets_to_array
from ruart_amd.metrics import note_stvqa, note_textvqa, stvqa_score
from ruart_amd.sampler import VQA_Sampler


def test_conf_reader_semantics(tmp_path):
    p = tmp_path / "conf"
    p.write_text("A\nB\t3\nC 0.5\n# comment\nD true\nE ./x/y\nB 9\nF 1 2 3\noptimizer   #\n\n")
    opt = Arguments(str(p)).readArguments()
    assert opt == {"A": True, "B": 3, "C": 0.5, "D": True, "E": "./x/y", "optimizer": "#"}
    with pytest.raises(Exception):
        Arguments(str(tmp_path / "missing"))
    d = default_opt()
    assert d["hidden_size"] == 125 and d["LN"] is True and d["optimizer"] == "#" and d["concat_rnn"] is False
    assert d["ES_ocr_len"] == 10 and d["max_ocr_num"] == 100 and "PRE_ALIGN_befor_rnn" in d


def _samples_like_generator(opt, n, seed):
    """must mirror oracle/gen_golden.py::synthetic_samples draw for draw"""
    g = np.random.default_rng(seed)

    def item(nw, nb, sentinel=None):
        words = [sentinel] if sentinel is not None else g.integers(5, 900, size=nw).tolist()
        bert = [101] + g.integers(1000, 2000, size=nb).tolist() + [102]
        offs, cur = [], 1
        for k in range(len(words)):
            c = 1 + (k < nb - len(words))
            offs.append([cur, cur + c])
            cur += c
        return {"fasttext": words, "pos": g.integers(0, 51, size=len(words)).tolist(), "ent": g.integers(0, 75, size=len(words)).tolist(),
                "bert": bert, "bert_offsets": offs, "position": g.random(8).round(4).tolist()}

    out = []
    for i in range(n):
        nq = int(g.integers(3, 12))
        q = {"glove": g.integers(5, 900, size=nq).tolist(), "pos": g.integers(0, 51, size=nq).tolist(),
             "ent": g.integers(0, 75, size=nq).tolist(), "bert": [101] + g.integers(1000, 2000, size=nq + 2).tolist() + [102],
             "bert_offsets": [[1 + k, 2 + k] for k in range(nq)]}
        n_ocr, n_od = int(g.integers(12, 20)), int(g.integers(1, 6))
        ocr = [item(int(g.integers(1, 4)), int(g.integers(3, 7))) for _ in range(n_ocr - 1)] + [item(1, 1, sentinel=3)]
        od = [item(int(g.integers(1, 3)), int(g.integers(2, 4))) for _ in range(n_od - 1)] + [item(1, 1, sentinel=4)]
        gt = torch.zeros(1, opt["max_ocr_num"] + 1)
        gt[0, int(g.integers(0, n_ocr - 1))] = 1.0
        out.append({"q": q, "ocr": ocr, "od": od, "gt": gt, "extra_info": {"q_id": i, "answers": None, "ocr_list": ["w"] * n_ocr,
                                                                         "image_path": "x"}})
    return out


def test_collate_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "host.npz"))
    opt = default_opt()
    q, ocr, od, gt, extra = VQA_collate(opt).VQA_collate_fun(_samples_like_generator(opt, 3, int(z["seed"])))
    assert np.array_equal(gt.numpy(), z["gt"])
    for name, d in (("q", q), ("ocr", ocr), ("od", od)):
        for k in z.files:
            if not k.startswith(name + ":") or k.endswith(("num_cnt", "len_cnt", "n_offsets")):
                continue
            key = k.split(":", 1)[1]
            got = d[key]
            assert got.dtype == {"b": torch.bool, "i": torch.int64, "f": torch.float32}[z[k].dtype.kind], (k, got.dtype)
            assert np.array_equal(got.numpy(), z[k]), k
        if name != "q":
            assert d["num_cnt"] == z[name + ":num_cnt"].tolist()
            assert np.concatenate([np.array(l) for l in d["len_cnt"]]).tolist() == z[name + ":len_cnt"].tolist()
        assert [len(o) for o in d["bert_offsets"]] == z[name + ":n_offsets"].tolist()


def test_sampler_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "host.npz"))
    data = list(range(23))
    assert np.array_equal(np.array(list(VQA_Sampler(data, 7, 5, True))), z["sampler_train"])
    assert np.array_equal(np.array(list(VQA_Sampler(data, 7, 5, True, batch_st=3))), z["sampler_train_resume"])
    assert np.array_equal(np.array(list(VQA_Sampler(data, None, 4, True, epoch=2))), z["sampler_epoch"])
    assert np.array_equal(np.array(list(VQA_Sampler(data, 99, 5, False))), z["sampler_eval"])


def test_sampler_rank_sharding_partitions_the_global_stream():
    data = list(range(50))
    whole = list(VQA_Sampler(data, 6, 8, True))                       # single process, batch 8
    parts = [list(VQA_Sampler(data, 6, 4, True, rank=r, world_size=2)) for r in range(2)]
    for step in range(6):
        merged = sorted(parts[0][step] + parts[1][step])
        assert merged == sorted(whole[step]) and len(parts[0][step]) == 4


def test_metrics_known_answers():
    assert stvqa_score("abc", "bd") == pytest.approx(0.33333333333333337)
    assert note_stvqa(["Coca Cola", "cola"], "coca") == pytest.approx(0.75)
    assert note_textvqa(["a", "A", "b"], "a") == pytest.approx(0.2)
    assert stvqa_score("", "") == 1


def test_item_index_reproduces_the_reference_loops():
    """The index vectors must reproduce Models/SDNet.py:300-318 (last-word gather) and :495-551 (pre-align re-packing)."""
    opt = default_opt(vocab_size=300)
    _, ocr, _, _, _ = synth.synthetic_batch(opt, 3, seed=2, n_q=8, n_ocr=25, n_od=6, bert_vocab=2000, ragged=True)
    idx = ItemIndex(ocr, "fasttext", ocr["position"].size(1))
    N, Lw = ocr["fasttext"].shape
    g = torch.Generator().manual_seed(0)
    dense = torch.randn(N, Lw, 5, generator=g)                      # stands for any per-(item, word) tensor
    packed = dense.reshape(N * Lw, 5)[torch.from_numpy(idx.flat_word)]
    # pre-align packing: the reference's per-sample rows
    ref = torch.zeros(idx.B, idx.Tmax, 5)
    i = 0
    for b in range(idx.B):
        c = 0
        for j in ocr["len_cnt"][b]:
            ref[b, c:c + j] = dense[i, :j]
            c += j
            i += 1
    mine = torch.zeros(idx.B, idx.Tmax, 5).index_put((torch.from_numpy(idx.sample_of_word), torch.from_numpy(idx.tok_in_sample)), packed)
    assert torch.equal(ref, mine)
    # last-word gather through the sorted multi2one schedule: emulate "state = value of the item's last processed word"
    order_state = torch.zeros(N, 5)
    off = 0
    for s, n in enumerate(idx.n_active):
        rows = torch.from_numpy(idx.step_rows[off:off + n])
        order_state = torch.cat([packed[rows], order_state[n:]], 0)
        off += n
    out = torch.zeros(idx.B, idx.max_num, 5).index_put((torch.from_numpy(idx.sorted_sample), torch.from_numpy(idx.sorted_slot)), order_state)
    ref = torch.zeros(idx.B, idx.max_num, 5)
    i = 0
    for b in range(idx.B):
        for k, j in enumerate(ocr["len_cnt"][b]):
            ref[b, k] = dense[i, j - 1]
            i += 1
    assert torch.equal(ref, out)
    assert idx.mask.sum(1).tolist() == ocr["num_cnt"]


def test_offsets_array_tolerates_flat_empty_marker():
    arr = offsets_to_array([[[1, 3], [3, 4]], [1, 1], []], 3, 4)
    assert arr[0, :2].tolist() == [[1, 3], [3, 4]] and arr[1].sum() == 0 and arr[2].sum() == 0


def test_packed_token_plan_on_cpu():
    """PackedTokens runs on the host (numpy) - check the block plan invariants without a GPU."""
    from ruart_amd.bert import PackedTokens
    g = np.random.default_rng(1)
    lens = [1, 2, 64, 65, 3, 200, 7, 7, 7, 64, 1]
    L = 200
    ids = np.zeros((len(lens), L), dtype=np.int64)
    for i, l in enumerate(lens):
        ids[i, :l] = g.integers(1, 99, size=l)
    p = PackedTokens([(torch.from_numpy(ids), torch.from_numpy(ids != 0))], "cpu")
    # short windows (<= 64 tokens) and <= 128-query blocks of long sequences (long-sequence kernel) together tile the stream
    q0, q1, k0, k1 = [np.concatenate([a.numpy(), b.numpy()]) for a, b in zip(p.blk, p.lblk)]
    assert p.n_long_blocks == 1 + 2 and p.n_blocks + p.n_long_blocks == len(q0)      # 65 -> 1 block, 200 -> 2 blocks
    for b in range(p.n_long_blocks):                                                   # a long block sees exactly one sequence
        lq0, lk0, lk1 = p.lblk[0][b].item(), p.lblk[2][b].item(), p.lblk[3][b].item()
        assert p.tok_lo[lq0].item() == lk0 and p.tok_hi[lq0].item() == lk1
    p_valu = PackedTokens([(torch.from_numpy(ids), torch.from_numpy(ids != 0))], "cpu", mfma_long=False)
    assert p_valu.n_long_blocks == 0 and p_valu.n_blocks == p.n_blocks + 2 + 4       # fp32 mode: 64-query chunks, VALU kernel
    assert p.T == sum(lens) and p.Tp % 128 == 0 and p.Tp >= p.T
    covered = np.zeros(p.T, dtype=int)
    lo, hi = p.tok_lo.numpy(), p.tok_hi.numpy()
    for b in range(len(q0)):
        assert 0 < q1[b] - q0[b] <= (64 if b < p.n_blocks else 128)
        covered[q0[b]:q1[b]] += 1
        assert k0[b] <= lo[q0[b]:q1[b]].min() and k1[b] >= hi[q0[b]:q1[b]].max()     # keys of every query are staged
    assert (covered == 1).all()
    assert (hi - lo == np.repeat(lens, lens)).all()
    assert np.array_equal(p.ids.numpy()[:p.T], ids[ids != 0]) and (p.ids.numpy()[p.T:] == 0).all()
    # unpacked mode keeps every slot and carries the -10000 key bias
    p2 = PackedTokens([(torch.from_numpy(ids[:3, :8]), torch.from_numpy(ids[:3, :8] != 0))], "cpu", pack=False)
    assert p2.T == 24 and (p2.key_bias.numpy() == np.where((ids[:3, :8] != 0).reshape(-1), 0, -10000)).all()


def test_train_plan_windows_and_chunks_on_cpu():
    """PackedTokens.train_plan (the attention plan of the 16-bit trainable encoder): whole short sequences in windows, every longer
    sequence in consecutive <= 64-token chunks that start at its first token - the same plan from both block layouts of the stream
    (128-query long blocks of the 16-bit frozen path, 64-query chunks of the fp32 / fp16c path); masked-fill dropout masks carry their
    byte twin."""
    from ruart_amd.bert import PackedTokens
    from ruart_amd import layers as L
    g = np.random.default_rng(2)
    lens = [3, 65, 64, 130, 7, 512, 30, 128, 1]
    ids = np.zeros((len(lens), 512), dtype=np.int64)
    for i, l in enumerate(lens):
        ids[i, :l] = g.integers(1, 99, size=l)
    plans = []
    for mfma_long in (True, False):
        p = PackedTokens([(torch.from_numpy(ids), torch.from_numpy(ids != 0))], "cpu", mfma_long=mfma_long)
        plan = p.train_plan(torch.device("cpu"))
        assert plan["ok"] and plan["n_chunks"] == 2 + 3 + 8 + 2
        q0, q1, k0, k1, first = (t.numpy() for t in plan["chunks"])
        w0, w1 = (t.numpy() for t in plan["win"])
        covered = np.zeros(p.T, dtype=int)
        for a, b in zip(w0, w1):
            assert 0 < b - a <= 64
            covered[a:b] += 1
        lo, hi = p.tok_lo.numpy(), p.tok_hi.numpy()
        for i in range(len(q0)):
            assert 0 < q1[i] - q0[i] <= 64 and (q0[i] - k0[i]) % 64 == 0 and (q0[i] - k0[i]) // 64 == i - first[i]
            assert (lo[q0[i]:q1[i]] == k0[i]).all() and (hi[q0[i]:q1[i]] == k1[i]).all() and q0[first[i]] == k0[i]
            covered[q0[i]:q1[i]] += 1
        assert (covered == 1).all()
        plans.append((sorted(zip(w0.tolist(), w1.tolist())), list(zip(q0.tolist(), q1.tolist(), k0.tolist(), k1.tolist(), first.tolist()))))
    assert plans[0] == plans[1]
    # the mask bank: every mask starts 16-byte aligned and carries keep bytes + scale (what ops._Linear hands to ruart_gemm_x3)
    bank = L.MaskBank()
    like = torch.zeros(1)
    bank.take(3, 5, 0.3, like)
    bank.take(7, 2, 0.3, like)
    torch.manual_seed(0)
    bank.begin_step(torch.device("cpu"))
    m1, m2 = bank.take(3, 5, 0.3, like), bank.take(7, 2, 0.3, like)
    for m in (m1, m2):
        assert m.keep.dtype == torch.uint8 and m.keep.shape == m.shape and torch.equal(m.keep != 0, m != 0)
        assert abs(m.keep_scale - 1 / 0.7) < 1e-12 and m.storage_offset() % 4 == 0
        assert bool(((m == 0) | ((m - m.keep_scale).abs() < 1e-6)).all())


def test_packed_tokens_window_rows_longer_than_512():
    """Models/Bert/Bert.py:18, 96-99, 133-138: a row longer than 512 word pieces is encoded as independent 512-windows whose
    positions restart at 0.  In the packed stream every (row, window) is its own sequence; the row's pieces stay contiguous."""
    from ruart_amd.bert import PackedTokens
    g = np.random.default_rng(2)
    lens = [700, 5, 1100, 512, 513]
    L = 1100
    ids = np.zeros((len(lens), L), dtype=np.int64)
    for i, l in enumerate(lens):
        ids[i, :l] = g.integers(1, 99, size=l)
    p = PackedTokens([(torch.from_numpy(ids), torch.from_numpy(ids != 0))], "cpu", mfma_long=False)
    want = [512, 188, 5, 512, 512, 76, 512, 512, 1]                      # window lengths in row-major order
    lo, hi = p.tok_lo.numpy(), p.tok_hi.numpy()
    starts = np.unique(lo)
    assert p.n_seq == len(want) and [int(hi[s_] - s_) for s_ in starts] == want
    assert p.max_pos == 512 and int(p.pos.numpy()[:p.T].max()) == 511
    # positions restart in every window; the pieces of row 0 sit at packed indices 0..699 in order
    assert np.array_equal(p.pos.numpy()[:700], np.arange(700) % 512)
    assert np.array_equal(p.group_index[0][0, :700], np.arange(700)) and p.group_index[0][1, 0] == 700
    assert np.array_equal(p.ids.numpy()[:p.T], ids[ids != 0])
    with pytest.raises(ValueError):                                          # a position table that is too small is an error,
        PackedTokens([(torch.from_numpy(ids), torch.from_numpy(ids != 0))], "cpu", max_positions=256)     # not an out-of-bounds read


def test_word_spans_follow_reference_pooling_rules():
    from ruart_amd.bert import PackedTokens, word_spans
    ids = np.array([[5, 6, 7, 8, 9, 0, 0], [5, 6, 0, 0, 0, 0, 0]])
    p = PackedTokens([(torch.from_numpy(ids), torch.from_numpy(ids != 0))], "cpu")
    offsets = [[[1, 2], [2, 4], [4, 4]], [1, 1]]                 # single piece, two pieces, empty span; item without words
    wm = np.array([[1, 1, 1, 0], [0, 0, 0, 0]], dtype=bool)
    s, n, d, rows = word_spans(p, 0, offsets, wm)
    assert s.tolist() == [1, 2] and n.tolist() == [1, 2] and d.tolist() == [0, 1] and rows == 8
    with pytest.raises(ValueError):
        word_spans(p, 0, [[[4, 7]], [1, 1]], np.array([[1, 0, 0, 0], [0, 0, 0, 0]], dtype=bool))   # span runs into padding


def test_collate_can_prepare_the_batch_index_in_a_worker():
    """VQA_collate(prepare_index=True) builds the host part of BatchIndex (numpy only): it must survive pickling (DataLoader
    workers hand batches over that way) and equal the index SDNet.prepare would build itself."""
    import pickle
    from ruart_amd.batch import BatchIndex
    opt = default_opt()
    samples = _samples_like_generator(opt, 3, 5)
    q, ocr, od, gt, extra = VQA_collate(opt, prepare_index=True).VQA_collate_fun(samples)
    host = pickle.loads(pickle.dumps(q["_ruart_host_index"]))
    q2, ocr2, od2, _, _ = VQA_collate(opt).VQA_collate_fun(samples)
    ref = BatchIndex(q2, ocr2, od2, opt)
    # default precision fp16c: fp32 QKV rows, 64-query split-f16 attention blocks (the MFMA long-sequence plan is the plain 16-bit modes')
    # (third entry: frozen encoder -> identical sequences are encoded once)
    assert host.plan == ref.plan == (True, False, True)
    assert BatchIndex(q2, ocr2, od2, dict(opt, bert_precision="fp16")).plan == (True, True, True)
    unlocked = {k: v for k, v in opt.items() if k != "LOCK_BERT"}
    assert BatchIndex(q2, ocr2, od2, unlocked).plan == (True, False, False)     # trainable encoder: every sequence keeps its own rows
    assert np.array_equal(host.packed.host, ref.packed.host) and host.packed.T == ref.packed.T
    assert np.array_equal(host._spans_host[0], ref._spans_host[0]) and host._spans_host[1] == ref._spans_host[1]
    for a, b in ((host.ocr, ref.ocr), (host.od, ref.od)):
        for f in a._FIELDS:
            assert np.array_equal(getattr(a, f), getattr(b, f)), f
        assert a.n_active == b.n_active and np.array_equal(a.mask, b.mask)
    dev = host.to("cpu")                                            # binding works on any torch device
    assert dev.packed.ids.numel() == dev.packed.Tp and len(dev.spans) == 3 and dev.ocr.dev["flat_word"].dtype == torch.int64


def test_packed_tokens_dedup_and_last_layer_rows():
    """Frozen encoder: identical sequences of a group are packed once (duplicates point at the first occurrence's rows), and the rows
    the last layer must produce are exactly the rows inside word spans, with span starts remapped to the compacted order."""
    from ruart_amd.batch import BatchIndex
    from ruart_amd.bert import PackedTokens
    ids = np.array([[101, 7, 8, 102, 0, 0],
                    [101, 9, 102, 0, 0, 0],
                    [101, 7, 8, 102, 0, 0],      # duplicate of row 0
                    [101, 9, 102, 0, 0, 0],      # duplicate of row 1
                    [101, 7, 102, 0, 0, 0]], dtype=np.int64)
    g = [(torch.from_numpy(ids), torch.from_numpy(ids != 0))]
    plain, dd = PackedTokens(g, "cpu"), PackedTokens(g, "cpu", dedup=True)
    assert plain.T == 4 + 3 + 4 + 3 + 3 and dd.T == 4 + 3 + 3 and dd.n_seq == 3
    gi = dd.group_index[0]
    assert np.array_equal(gi[2], gi[0]) and np.array_equal(gi[3], gi[1]) and gi[0, 0] == 0 and gi[1, 0] == 4 and gi[4, 0] == 7
    assert np.array_equal(dd.host[:dd.T], [101, 7, 8, 102, 101, 9, 102, 101, 7, 102])
    assert np.array_equal(dd.host[dd.Tp:dd.Tp + dd.T], [0, 1, 2, 3, 0, 1, 2, 0, 1, 2])        # positions restart per sequence
    same = PackedTokens([(torch.from_numpy(ids[[0, 1, 4]]), torch.from_numpy(ids[[0, 1, 4]] != 0))], "cpu", dedup=True)
    assert np.array_equal(same.host[:same.T], dd.host[:dd.T])                                  # no duplicates: nothing changes
    # through BatchIndex: the sentinel items of a synthetic batch are duplicates; [CLS] / [SEP] rows are not in last_rows
    opt = default_opt()
    samples = _samples_like_generator(opt, 3, 5)
    q, ocr, od, _, _ = VQA_collate(opt).VQA_collate_fun(samples)
    bi = BatchIndex(q, ocr, od, opt)
    off = BatchIndex(q, ocr, od, dict(opt, bert_dedup=False, bert_last_rows=False))
    assert bi.packed.T <= off.packed.T and off._n_last == 0
    from ruart_amd import synth
    sq, socr, sod, _, _ = synth.synthetic_batch(opt, 4, seed=5, n_q=8, n_ocr=12, n_od=5, bert_vocab=2000)
    # (the generator spells its <OCR> sentinel items with random pieces; real data tokenizes them identically: make them so)
    sent = np.cumsum(socr["num_cnt"]) - 1
    for r in sent[1:]:
        socr["bert"][r] = socr["bert"][sent[0]]
        socr["bert_mask"][r] = socr["bert_mask"][sent[0]]
        socr["bert_offsets"][r] = socr["bert_offsets"][sent[0]]
    a, b = BatchIndex(sq, socr, sod, opt), BatchIndex(sq, socr, sod, dict(opt, bert_dedup=False))
    assert a.packed.n_seq == b.packed.n_seq - (4 - 1) and a.packed.T < b.packed.T
    W = [n for n, _, _ in bi._spans_host[1]]
    flat = bi._spans_host[0]
    o, covered = 0, np.zeros(bi.packed.T, dtype=bool)
    starts, lasts = [], []
    for w, (_, _, n_pieces) in zip(W, bi._spans_host[1]):
        s_, l_, ls = flat[o:o + w], flat[o + w:o + 2 * w], flat[o + 3 * w:o + 4 * w]
        assert n_pieces == int(l_.sum())               # what the trainable encoder's pooling gets instead of a device readback
        for a, n in zip(s_, l_):
            covered[a:a + n] = True
        starts.append(s_)
        lasts.append(ls)
        o += 4 * w
    rows = flat[o:]
    assert bi._n_last == len(rows) == covered.sum() and np.array_equal(rows, np.nonzero(covered)[0])
    pieces = bi.packed.host[:bi.packed.T]
    # [CLS] = 101 / [SEP] = 102 are never pooled (pieces of words past an item's len_cnt are not pooled either)
    assert not np.isin(pieces[rows], (101, 102)).any() and np.isin(pieces, (101, 102))[~covered].sum() == np.isin(pieces, (101, 102)).sum()
    for s_, ls in zip(starts, lasts):
        assert np.array_equal(rows[ls], s_)                        # compacted row ls holds packed row s_


def _dataset_fixture(golden_dir, tmp_path):
    import gzip
    import json
    with open(os.path.join(golden_dir, "dataset_input.json"), encoding="utf-8") as f:
        inp = json.load(f)
    with gzip.open(os.path.join(golden_dir, "dataset_expected.json.gz"), "rt", encoding="utf-8") as f:
        exp = json.load(f)
    vocab = tmp_path / "vocab.txt"
    vocab.write_text("\n".join(inp["vocab"]) + "\n", encoding="utf-8")
    return inp, exp, str(vocab)


def test_tokenizer_cases(golden_dir, tmp_path):
    """WordPiece behaviour on the cases the scheme is known for (Models/Bert/tokenization.py:164-325)."""
    from ruart_amd.tokenization import BertTokenizer
    inp, _, vocab = _dataset_fixture(golden_dir, tmp_path)
    tok = BertTokenizer.from_pretrained(vocab)
    assert tok.tokenize("unaffable") == ["un", "##aff", "##able"]
    assert tok.tokenize("CafÉ!") == ["cafe", "!"]                          # lower-cased, accent stripped, punctuation split
    assert tok.tokenize("中国 shop") == ["中", "国", "shop"]     # CJK ideographs stand alone
    assert tok.tokenize("x" * 101) == ["[UNK]"] and tok.tokenize("~") == ["[UNK]"]
    assert tok.tokenize("zz\x00top\ttab") == ["z", "##z", "##t", "##o", "##p", "t", "##a", "##b"]   # NUL dropped, tab splits
    assert tok.tokenize("") == [] and tok.tokenize(" 　  ") == []
    assert tok.convert_tokens_to_ids(["[CLS]", "stop", "[SEP]"]) == [2, inp["vocab"].index("stop"), 3]
    assert tok.convert_ids_to_tokens([0, 1]) == ["[PAD]", "[UNK]"]
    with pytest.raises(KeyError):
        tok.convert_tokens_to_ids(["absent"])
    with pytest.raises(ValueError):
        BertTokenizer.from_pretrained(str(tmp_path / "missing.txt"))


@pytest.mark.parametrize("variant", ["base", "dedup_one", "yesno_relevance", "acc_all_no_es", "test_mode"])
def test_dataset_matches_reference(golden_dir, tmp_path, variant):
    """``VQA_Dataset.__getitem__`` (Utils/VQA_Dataset.py:109-153) on synthetic preprocessed records, then the collate: every id,
    offset, position, soft label and string equals what the reference's class produced (oracle/gen_golden.py dataset)."""
    import copy
    from ruart_amd.dataset import VQA_Dataset
    from ruart_amd.batch import VQA_collate
    inp, exp, vocab = _dataset_fixture(golden_dir, tmp_path)
    ov = dict(inp["variants"][variant])
    mode = ov.pop("_mode", "train")
    opt = default_opt(datadir="", BERT_tokenizer_file=vocab)
    for k in ov.pop("_drop", []):
        opt.pop(k, None)
    opt.update(ov)
    ds = VQA_Dataset(copy.deepcopy(inp["records"]), opt, mode=mode)
    want = exp[variant]
    assert len(ds) == want["n"]
    samples = [ds[i] for i in range(len(ds))]
    for got, ref in zip(samples, want["samples"]):
        assert got["q"] == ref["q"]
        assert got["ocr"] == ref["ocr"] and got["od"] == ref["od"]
        assert got["gt"].tolist() == ref["gt"]
        assert got["extra_info"] == ref["extra_info"]
    q, ocr, od, gt, extra = VQA_collate(opt).VQA_collate_fun(samples[:4])
    coll = want["collated"]
    assert gt.tolist() == coll["gt"]
    for nm, d in (("q", q), ("ocr", ocr), ("od", od)):
        for k, v in d.items():
            if k.startswith("_"):
                continue
            ref = coll["%s:%s" % (nm, k)]
            assert (v.tolist() if isinstance(v, torch.Tensor) else v) == ref, (nm, k)


def test_msgpack_readers_round_trip(tmp_path):
    """``load_msgpack`` / ``load_meta`` read what the reference's preprocessing writes (CoQAPreprocess.py:470-501)."""
    import msgpack
    from ruart_amd.dataset import load_meta, load_msgpack
    meta = {"vocab": ["<PAD>", "<UNK>", "a"], "char_vocab": ["a", "b"], "glove_embedding": [[0.0, 1.0], [2.0, 3.0], [4.0, 5.0]],
            "fast_embedding": [[1.0, 1.0], [2.0, 2.0], [3.0, 3.0]]}
    with open(tmp_path / "train_meta.msgpack", "wb") as f:
        msgpack.dump(meta, f)
    with open(tmp_path / "dev-preprocessed.msgpack", "wb") as f:
        msgpack.dump({"data": [{"question_id": 7, "question": "café?"}]}, f)
    opt = {"GLOVE": True, "FastText": True, "FEATURE_FOLDER": str(tmp_path)}
    vocab, char_vocab, emb = load_meta(opt)
    assert vocab == meta["vocab"] and char_vocab == ["a", "b"] and opt["vocab_size"] == 3 and opt["vocab_dim"] == 2
    assert opt["char_vocab_size"] == 2 and emb["glove_embedding"].tolist() == meta["glove_embedding"]
    assert set(emb) == {"glove_embedding", "fast_embedding"}
    assert load_msgpack(str(tmp_path / "dev-preprocessed.msgpack"))["data"][0]["question"] == "café?"


@pytest.mark.parametrize("variant", ["no_answer", "plain"])
def test_predict_decode_matches_reference(golden_dir, variant):
    """``trainer.decode_predictions`` (masked arg-max) against the reference's own ``SDNetTrainer.predict`` decode loop
    (Models/SDNetTrainer.py:391-450, run with a stub network that returns these scores): chosen slot, answer string, score,
    ANLS and ACC sums - sentinel / padding / no-answer slots on top and the nothing-admissible case included."""
    import json
    from ruart_amd.trainer import decode_predictions
    with open(os.path.join(golden_dir, "predict_decode.json")) as f:
        z = json.load(f)
    cases, want = z["cases"], z["expected"][variant]
    width = z["n_slots"] if variant == "no_answer" else z["n_slots"] - 1
    scores = torch.tensor([c["prob"][:width] for c in cases])
    extra = [{"q_id": c["q_id"], "answers": c["answers"], "ocr_list": c["ocr_list"], "image_path": "x"} for c in cases]
    opt = {"label_no_answer": True} if variant == "no_answer" else {}
    anls, acc, res, save_res = decode_predictions(scores, [c["num_cnt"] for c in cases], extra, opt)
    assert res == want["res"]
    assert [r["idx"] for r in save_res] == [r["idx"] for r in want["save_res"]]
    for got, ref in zip(save_res, want["save_res"]):
        assert got == ref, (got, ref)
    assert abs(anls - want["ANLS"]) < 1e-9 and abs(acc - want["ACC"]) < 1e-9
    with pytest.raises(NotImplementedError):
        decode_predictions(scores, [c["num_cnt"] for c in cases], extra, {"label_yesno": True})


def test_collate_fast_path_equals_item_walk(golden_dir, tmp_path):
    """Samples from ``VQA_Dataset`` carry flat arrays (``_flat``) that let ``VQA_collate`` concatenate instead of walking every
    item dict; both routes must give identical batches - id matrices, masks, counts, positions, offsets (list and array form) -
    and the batch index built from them must agree too.  Cached samples (``ruart_cache_samples``) are the same objects again."""
    import copy
    from ruart_amd.batch import BatchIndex
    from ruart_amd.dataset import VQA_Dataset
    inp, _, vocab = _dataset_fixture(golden_dir, tmp_path)
    opt = default_opt(datadir="", BERT_tokenizer_file=vocab)
    full = [s_ for s_ in VQA_Dataset(copy.deepcopy(inp["records"]), opt)]
    cached = VQA_Dataset(copy.deepcopy(inp["records"]), dict(opt, ruart_cache_samples=True))
    samples = [cached[i] for i in range(len(cached))]
    assert all("_flat" in s_ and "ocr" not in s_ for s_ in samples) and cached[1] is samples[1]      # compact, and cached
    assert all("_flat" in s_ and "ocr" in s_ for s_ in full)
    assert cached._direct_flat                                                                         # filled straight from the records
    for c_, f_ in zip(samples, full):
        for grp in ("ocr", "od"):
            assert list(c_["_flat"][grp]) == list(f_["_flat"][grp])
            for k, v in f_["_flat"][grp].items():
                got = c_["_flat"][grp][k]
                assert all(np.array_equal(x, y) and x.dtype == y.dtype for x, y in zip(got, v)) if isinstance(v, tuple) \
                    else (np.array_equal(got, v) and got.dtype == v.dtype), (grp, k)
    fast = VQA_collate(opt).VQA_collate_fun(samples)                                                  # from the flat arrays alone
    mixed = VQA_collate(opt).VQA_collate_fun(full)                                                    # lists + flat arrays
    slow = VQA_collate(opt).VQA_collate_fun([{k: v for k, v in s_.items() if k != "_flat"} for s_ in full])   # the item walk
    for a, b in zip(mixed[:3], fast[:3]):
        assert set(a) == set(b)
        for k, v in b.items():
            assert torch.equal(a[k], v) if isinstance(v, torch.Tensor) else (np.array_equal(a[k], v) if isinstance(v, np.ndarray) else a[k] == v), k
    for a, b in zip(fast[:3], slow[:3]):
        assert set(a) - {"_bert_offsets_arr"} == set(b)
        for k, v in b.items():
            assert torch.equal(a[k], v) if isinstance(v, torch.Tensor) else a[k] == v, k
    assert torch.equal(fast[3], slow[3]) and fast[4] == slow[4]
    for g in (1, 2):
        want = offsets_to_array(slow[g]["bert_offsets"], slow[g]["fasttext"].shape[0], slow[g]["fasttext"].shape[1])
        assert np.array_equal(fast[g]["_bert_offsets_arr"], want)
    bf, bs = BatchIndex(fast[0], fast[1], fast[2], opt), BatchIndex(slow[0], slow[1], slow[2], opt)
    assert np.array_equal(bf._spans_host[0], bs._spans_host[0]) and np.array_equal(bf.packed.host, bs.packed.host)
    assert np.array_equal(bf.ocr.flat_word, bs.ocr.flat_word) and np.array_equal(bf.od.step_rows, bs.od.step_rows)


def test_sort_ids_two_level_form():
    """batch._sort_ids: rows with more than max_seg occurrences are cut into sub-segments; the pieces tile order[] exactly and every
    row's sub-segments are consecutive."""
    from ruart_amd.batch import _sort_ids
    rng = np.random.default_rng(3)
    ids = rng.integers(0, 40, 5000)
    ids[::3] = 7                                                    # one very frequent row
    ids[5::11] = 1                                                  # the padding row: left out
    flat = _sort_ids(ids, padding_idx=None, max_seg=10 ** 6)
    assert len(flat) == 3
    order, sub_start, row_first, row_id = _sort_ids(ids, padding_idx=1, max_seg=64)
    assert 1 not in row_id and sorted(row_id.tolist()) == sorted(set(ids.tolist()) - {1})
    assert sub_start[0] == 0 and sub_start[-1] == len(order) == int((ids != 1).sum())
    assert (np.diff(sub_start) > 0).all() and np.diff(sub_start).max() <= 64
    assert row_first[0] == 0 and row_first[-1] == len(sub_start) - 1 and (np.diff(row_first) > 0).all()
    for r, row in enumerate(row_id):
        occ = order[sub_start[row_first[r]]:sub_start[row_first[r + 1]]]
        assert (ids[occ] == row).all() and len(occ) == int((ids == row).sum())
        assert (np.diff(occ) > 0).all()                             # stable: occurrences in their original order
    short = _sort_ids(np.array([3, 3, 5]), None)
    assert len(short) == 3 and short[1].tolist() == [0, 2, 3] and short[2].tolist() == [3, 5]


def test_stream_settings_follow_the_schedule(monkeypatch):
    """The encoder pass that runs ahead is CU-masked only in training, and the step's streams take normal priority only beside a masked
    pass (sdnet.SDNet.trunk_stream_priority, bert.Bert.prefetch_cus) - checked on bare objects, no device needed."""
    from ruart_amd.bert import Bert
    from ruart_amd.sdnet import SDNet
    for k in ("RUART_TRUNK_PRIORITY", "RUART_PREFETCH_CUS_EVAL", "RUART_TRUNK_CUS"):
        monkeypatch.delenv(k, raising=False)

    class FakeBert:
        prefetch_cus = Bert.prefetch_cus
        plan_prefetch_cus = Bert.plan_prefetch_cus

        def __init__(self, cus, training):
            self._pf_cus, self.training = cus, training
            self.weights = type("W", (), {"cfg": {"hidden_size": 768, "intermediate_size": 3072}})()

    class FakeNet:
        trunk_stream_priority = SDNet.trunk_stream_priority
        trunk_stream_cus = SDNet.trunk_stream_cus

        def __init__(self, bert, opt=None):
            self.Bert, self.opt = bert, opt or {}

    assert FakeBert(240, True).prefetch_cus() == 240 and FakeBert(240, False).prefetch_cus() == 0
    # 'auto': the smallest mask with the fewest GEMM tile rounds for the batch's rows (167 / 168 row tiles of the bench batches: 224)
    auto = FakeBert("auto", True)
    pk = lambda rows: type("P", (), {"Tp": rows})()
    assert auto.prefetch_cus() == 240 and FakeNet(auto).trunk_stream_priority() == 1            # (before the first batch is known)
    assert auto.prefetch_cus(pk(42752)) == 224 and auto.prefetch_cus(pk(39168)) == 224        # planned once, then kept (39168 alone: 232)
    assert auto.plan_prefetch_cus(43008) == 224 and auto.plan_prefetch_cus(39168) == 232
    assert all(208 <= auto.plan_prefetch_cus(r) <= 248 for r in range(256, 60000, 256))
    assert FakeBert("auto", False).prefetch_cus(pk(42752)) == 0
    assert FakeNet(FakeBert(240, True)).trunk_stream_priority() == 1          # fp16c schedule, training: LOW beside the masked pass
    assert FakeNet(FakeBert(240, False)).trunk_stream_priority() == -1        # evaluation: unmasked pass, trunk first
    assert FakeNet(FakeBert(0, True)).trunk_stream_priority() == -1           # plain 16-bit modes: never masked
    # the step's streams are never CU-masked by default (SDNet.trunk_stream_cus: an experiment's knob; option and environment set it)
    assert FakeNet(FakeBert(240, True)).trunk_stream_cus() == 0 and FakeNet(FakeBert(240, False)).trunk_stream_cus() == 0
    assert FakeNet(FakeBert(240, True), {"ruart_trunk_cus": -160}).trunk_stream_cus() == -160
    monkeypatch.setenv("RUART_TRUNK_CUS", "-128")
    assert FakeNet(FakeBert(240, True), {"ruart_trunk_cus": 0}).trunk_stream_cus() == -128
    monkeypatch.setenv("RUART_TRUNK_PRIORITY", "-1")
    assert FakeNet(FakeBert(240, True)).trunk_stream_priority() == -1


def test_readback_is_deferred_inside_the_training_loop_only():
    """SDNetTrainer._defer_readback: a bare update() keeps the reference's per-step float and asserts (Models/SDNetTrainer.py:376); inside
    train()'s own loop, and always with a trained encoder, the loss is read back one step late; opt['ruart_defer_readback'] overrides;
    never on the CPU.  step_stream() on the CPU is a null context.  Bare objects, no device needed."""
    import contextlib
    import torch
    from ruart_amd.trainer import SDNetTrainer

    class Fake:
        _defer_readback = SDNetTrainer._defer_readback
        step_stream = SDNetTrainer.step_stream
        _step_stream = SDNetTrainer._step_stream

        def __init__(self, opt, dev, trained=False, in_loop=False):
            self.opt, self.device, self._in_train_loop = opt, torch.device(dev), in_loop
            bert = type("B", (), {"bert_model": object() if trained else None})()
            self.network = type("N", (), {"Bert": bert, "train": lambda self_: None, "trunk_stream_priority": lambda self_: 1,
                                 "trunk_stream_cus": lambda self_: 0})()

    assert Fake({}, "cuda").__class__._defer_readback(Fake({}, "cuda")) is False
    assert Fake({}, "cuda", in_loop=True)._defer_readback() is True
    assert Fake({}, "cuda", trained=True)._defer_readback() is True
    assert Fake({"ruart_defer_readback": False}, "cuda", trained=True, in_loop=True)._defer_readback() is False
    assert Fake({"ruart_defer_readback": True}, "cuda")._defer_readback() is True
    assert Fake({"ruart_defer_readback": True}, "cpu", in_loop=True)._defer_readback() is False
    assert isinstance(Fake({}, "cpu").step_stream(), contextlib.nullcontext)


def test_close_final_decides_the_masked_streams_destruction(monkeypatch):
    """Round 6: SDNetTrainer.close(final=True) destroys the CU-masked run-ahead stream, close() keeps it for the next session - the
    caller decides, no environment variable is inspected for a profiler's name (RUART_DESTROY_STREAMS=1 stays as an experiment knob).
    Bare objects, no device needed."""
    from ruart_amd.bert import Bert
    from ruart_amd.trainer import SDNetTrainer
    seen = []

    class FakeBert:
        def close(self, destroy=None):
            seen.append(destroy)

    class Fake:
        close = SDNetTrainer.close
        _destroy_masked_streams = SDNetTrainer._destroy_masked_streams
        flush_readback = lambda self: None

        def __init__(self):
            self.network = type("N", (), {"Bert": FakeBert()})()

    Fake().close()
    Fake().close(final=True)
    assert seen == [None, True]
    # an experiment's CU-masked step streams (SDNet.trunk_stream_cus) go with final=True too; torch's own streams are left alone
    import ruart_amd.hip as hip_mod
    gone = []
    monkeypatch.setattr(hip_mod, "destroy_stream", lambda st: gone.append(st))
    masked, plain = type("S", (), {"_ruart_masked": True})(), type("S", (), {})()
    f = Fake()
    f._step_streams = {1: masked, -1: plain}
    f.network._streams = {1: (masked, masked), -1: (plain, plain)}
    f.close()
    assert gone == [] and len(f._step_streams) == 2
    f.close(final=True)
    assert gone == [masked] * 3 and f._step_streams == {-1: plain} and f.network._streams == {-1: (plain, plain)}
    # Bert.close(None) keeps the streams whatever tool library the environment names
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    monkeypatch.delenv("RUART_DESTROY_STREAMS", raising=False)
    kept = []

    class FakeStream:
        def synchronize(self):
            kept.append("sync")

    b = type("B", (), {"close": Bert.close})()
    b._pending, b._pf_streams = None, {224: FakeStream()}
    b.close()
    assert kept == ["sync"] and 224 in b._pf_streams


def test_train_hands_the_frozen_generation_back():
    """Round 6 (advisor): train() freezes the collector's permanent generation after its set-up and unfreezes it on every exit path."""
    import gc
    from ruart_amd.trainer import SDNetTrainer

    class Fake:
        train = SDNetTrainer.train
        closed = 0

        def __init__(self, fail):
            self.fail = fail

        def _train(self, *a):
            gc.freeze()
            self._gc_frozen = True
            if self.fail:
                raise RuntimeError("loader died")

        def close(self, final=False):
            self.closed += 1

    gc.unfreeze()
    for fail in (False, True):
        t = Fake(fail)
        try:
            t.train()
        except RuntimeError:
            assert fail
        assert gc.get_freeze_count() == 0 and t.closed == 1 and not t._in_train


def test_mm_row_split_keeps_the_b_side_mask(monkeypatch):
    """Round 6 (advisor): when an operand spans >= 2^30 elements ops.mm splits the product over its rows; the halves must carry the
    b-side dropout mask, its scale and rpm.  The recursion is observed on a stub (no device, no 4 GB tensor)."""
    import torch
    from ruart_amd import ops
    calls = []
    real_mm = ops.mm
    monkeypatch.setattr(ops, "_span", lambda t: (1 << 30) if t.shape[0] == 8 and t.shape[1] == 4 else 1)
    monkeypatch.setattr(ops, "_one_unit_stride", lambda t: t)

    def spy(a, b, bias=None, mode=None, out=None, a_keep=None, b_keep=None, keep_scale=1.0, c_scale=None, rpm=1, residual=None):
        if a.shape[0] == 8:
            return real_mm(a, b, bias, mode, out, a_keep, b_keep, keep_scale, c_scale, rpm, residual)
        calls.append((a.shape[0], b_keep is not None, keep_scale, rpm))
        return out

    monkeypatch.setattr(ops, "mm", spy)

    class T:                       # the few tensor attributes mm() touches before it recurses
        is_cuda, dtype, device = True, torch.float32, "cuda:0"

        def __init__(self, r, c):
            self.shape = (r, c)

        def __getitem__(self, s):
            n = len(range(*s.indices(self.shape[0])))
            return T(n, self.shape[1])

        def numel(self):
            return 1 << 31

        def stride(self, i):
            return (self.shape[1], 1)[i]

    keep = object()
    monkeypatch.setattr(torch, "empty", lambda *a, **k: T(a[0], a[1]))
    spy(T(8, 4), T(4, 6), mode="x3", b_keep=keep, keep_scale=2.0, rpm=2)
    assert calls == [(4, True, 2.0, 2), (4, True, 2.0, 2)]
