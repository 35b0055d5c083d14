"""WordPiece tokenizer with the behaviour of the reference's ``Models/Bert/tokenization.py`` (the 2018 Google BERT scheme):

  text -> drop NUL / U+FFFD / control characters, every whitespace to ' '  (``_clean_text``, :252-263)
       -> CJK ideographs isolated by spaces                                 (``_tokenize_chinese_chars``, :217-250)
       -> whitespace split; lower-case + NFD with combining marks removed   (``BasicTokenizer.tokenize``, :164-195)
       -> every punctuation character becomes its own token                 (``_run_split_on_punc``, :197-215; ``_is_punctuation``, :352-365)
       -> greedy longest-prefix match against the vocabulary, continuation pieces spelled '##x';
          a word of more than 100 characters or without a full cover is '[UNK]'   (``WordpieceTokenizer``, :266-325)

It is the host-side producer of the ``bert`` / ``bert_offsets`` fields of a sample (``dataset.VQA_Dataset.bertify``); the results
are integer ids, pinned bit-exact against the reference in ``tests/test_host_logic.py``."""
import collections
import os
import unicodedata

_CJK_RANGES = ((0x4E00, 0x9FFF), (0x3400, 0x4DBF), (0x20000, 0x2A6DF), (0x2A700, 0x2B73F), (0x2B740, 0x2B81F),
               (0x2B820, 0x2CEAF), (0xF900, 0xFAFF), (0x2F800, 0x2FA1F))


def _is_space(ch):
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_control(ch):
    return ch not in "\t\n\r" and unicodedata.category(ch).startswith("C")


def _is_punct(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:      # all ASCII non-alphanumerics, incl. ^ $ `
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp):
    return any(lo <= cp <= hi for lo, hi in _CJK_RANGES)


def load_vocab(vocab_file):
    """One token per line, id = line number (tokenization.py:62-74)."""
    vocab = collections.OrderedDict()
    with open(vocab_file, "r", encoding="utf-8") as f:
        for index, line in enumerate(f):
            vocab[line.strip()] = index
    return vocab


class BasicTokenizer:
    def __init__(self, do_lower_case=True):
        self.do_lower_case = do_lower_case

    def tokenize(self, text):
        if isinstance(text, bytes):
            text = text.decode("utf-8", "ignore")
        chars = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            if _is_space(ch):
                chars.append(" ")
            elif _is_cjk(cp):
                chars.extend((" ", ch, " "))
            else:
                chars.append(ch)
        out = []
        for word in "".join(chars).split():
            if self.do_lower_case:
                word = "".join(c for c in unicodedata.normalize("NFD", word.lower()) if unicodedata.category(c) != "Mn")
            cur = []
            for ch in word:
                if _is_punct(ch):
                    if cur:
                        out.append("".join(cur))
                        cur = []
                    out.append(ch)
                else:
                    cur.append(ch)
            if cur:
                out.append("".join(cur))
        # the reference re-splits the joined tokens on whitespace: stripping accents can leave an empty token behind
        return " ".join(out).split()


class WordpieceTokenizer:
    def __init__(self, vocab, unk_token="[UNK]", max_input_chars_per_word=100):
        self.vocab = vocab
        self.unk_token = unk_token
        self.max_input_chars_per_word = max_input_chars_per_word

    def tokenize(self, text):
        out = []
        for word in text.split():
            n = len(word)
            if n > self.max_input_chars_per_word:
                out.append(self.unk_token)
                continue
            pieces, start = [], 0
            while start < n:
                end = n
                piece = None
                while end > start:
                    cand = word[start:end] if start == 0 else "##" + word[start:end]
                    if cand in self.vocab:
                        piece = cand
                        break
                    end -= 1
                if piece is None:
                    pieces = None
                    break
                pieces.append(piece)
                start = end
            if pieces is None:
                out.append(self.unk_token)
            else:
                out.extend(pieces)
        return out


class BertTokenizer:
    """``BertTokenizer(vocab_file)`` / ``BertTokenizer.from_pretrained(path)`` (tokenization.py:86-150): ``path`` is the vocabulary
    FILE (the shipped conf points ``BERT_tokenizer_file`` at ``.../vocab.txt``); there is no download here."""

    def __init__(self, vocab_file, do_lower_case=True):
        if not os.path.isfile(vocab_file):
            raise ValueError("Can't find a vocabulary file at path '{}'".format(vocab_file))
        self.vocab = load_vocab(vocab_file)
        self.ids_to_tokens = collections.OrderedDict((i, t) for t, i in self.vocab.items())
        self.basic_tokenizer = BasicTokenizer(do_lower_case=do_lower_case)
        self.wordpiece_tokenizer = WordpieceTokenizer(vocab=self.vocab)
        self._memo = {}                      # text -> pieces: scene-text words repeat across items and samples

    def tokenize(self, text):
        hit = self._memo.get(text)
        if hit is None:
            hit = [p for w in self.basic_tokenizer.tokenize(text) for p in self.wordpiece_tokenizer.tokenize(w)]
            if len(self._memo) < 1 << 20 and len(text) <= 64:
                self._memo[text] = hit
        return list(hit)

    def convert_tokens_to_ids(self, tokens):
        return [self.vocab[t] for t in tokens]           # KeyError on a token outside the vocabulary, as in the reference

    def convert_ids_to_tokens(self, ids):
        return [self.ids_to_tokens[i] for i in ids]

    @classmethod
    def from_pretrained(cls, pretrained_model_name, do_lower_case=True):
        return cls(pretrained_model_name, do_lower_case)
