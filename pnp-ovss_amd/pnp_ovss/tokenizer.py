"""Tokenizer surface the reference driver touches (PnP_OVSS_0514_updated_segmentation.py:271,317,
608,813; blip_image_text_matching.py:230-239): `tok(list[str], padding=, max_length=,
truncation=, return_tensors="pt")` -> object with `.input_ids`, `.attention_mask`, `.to(dev)`;
`tok.decode([id])`; `tok.enc_token_id`, `tok.pad_token_id`.

`bert-base-uncased` vocab.txt is not on disk here or on the GPU box, so two back ends exist:
  * WordPieceTokenizer(vocab_path): greedy longest-match word-piece over a real vocab file
    (LAVIS `BlipBase.init_tokenizer` adds "[DEC]" and "[ENC]" after the 30522 base entries);
  * SynthTokenizer(vocab_size): deterministic stand-in with the same special ids
    ([PAD]=0, [CLS]=101, [SEP]=102, [ENC]=vocab-1) that splits long words into "##" pieces so the
    word-piece merge path (PnP…segmentation.py:810-853) is exercised.
"""
import zlib

import torch


class Encoding(dict):
    def __init__(self, input_ids, attention_mask):
        super().__init__(input_ids=input_ids, attention_mask=attention_mask)
        self.input_ids = input_ids
        self.attention_mask = attention_mask

    def to(self, device):
        return Encoding(self.input_ids.to(device), self.attention_mask.to(device))


class _Base:
    pad_token_id = 0
    cls_token_id = 101
    sep_token_id = 102

    def _pieces(self, word):
        raise NotImplementedError

    def _piece_id(self, piece):
        raise NotImplementedError

    def tokenize(self, text):
        out = []
        for w in text.lower().split():
            out.extend(self._pieces(w))
        return out

    def __call__(self, captions, padding="longest", max_length=None, truncation=False,
                 return_tensors="pt"):
        if isinstance(captions, str):
            captions = [captions]
        rows = []
        for c in captions:
            ids = [self.cls_token_id] + [self._piece_id(p) for p in self.tokenize(c)] + [self.sep_token_id]
            if truncation and max_length is not None and len(ids) > max_length:
                ids = ids[: max_length - 1] + [self.sep_token_id]
            rows.append(ids)
        L = max_length if padding == "max_length" else max(len(r) for r in rows)
        ids = torch.full((len(rows), L), self.pad_token_id, dtype=torch.long)
        att = torch.zeros((len(rows), L), dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, : len(r)] = torch.tensor(r, dtype=torch.long)
            att[i, : len(r)] = 1
        return Encoding(ids, att)


class SynthTokenizer(_Base):
    def __init__(self, vocab_size=30524, max_piece=5):
        self.vocab_size = vocab_size
        self.enc_token_id = vocab_size - 1
        self.max_piece = max_piece
        self._id2piece = {0: "[PAD]", 101: "[CLS]", 102: "[SEP]", self.enc_token_id: "[ENC]"}
        self._piece2id = {}

    def _pieces(self, word):
        if len(word) <= self.max_piece + 2:
            return [word]
        out = [word[: self.max_piece]]
        rest = word[self.max_piece:]
        while rest:
            out.append("##" + rest[: self.max_piece])
            rest = rest[self.max_piece:]
        return out

    def _piece_id(self, piece):
        if piece in self._piece2id:
            return self._piece2id[piece]
        lo, span = 110, self.vocab_size - 112
        i = lo + zlib.crc32(piece.encode()) % span
        while i in self._id2piece and self._id2piece[i] != piece:
            i = lo + (i - lo + 1) % span
        self._id2piece[i] = piece
        self._piece2id[piece] = i
        return i

    def decode(self, ids):
        return " ".join(self._id2piece.get(int(i), "[UNK]") for i in ids)


def _is_punct(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    import unicodedata
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp):
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
            or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class WordPieceTokenizer(_Base):
    """BertTokenizer("bert-base-uncased") as LAVIS' BlipBase.init_tokenizer sets it up (blip_image_text_matching.py:42):
    BERT basic tokenisation (clean control characters, whitespace split, lower-case, NFD accent stripping, every
    punctuation character its own token, CJK characters spaced) followed by greedy longest-match-first word-piece
    (words over 100 characters or without a decomposition -> [UNK]) over a vocab.txt, plus the two added tokens
    "[DEC]" (bos) and "[ENC]" after the base entries.  Pinned against HF BertTokenizer on
    tests/golden/tokenizer_cases.json."""
    max_chars_per_word = 100

    def __init__(self, vocab_path):
        with open(vocab_path, encoding="utf-8") as f:
            toks = [l.rstrip("\n") for l in f]
        while toks and toks[-1] == "":
            toks.pop()
        toks += ["[DEC]", "[ENC]"]
        self._piece2id = {t: i for i, t in enumerate(toks)}
        self._id2piece = toks
        self.vocab_size = len(toks)
        self.enc_token_id = self._piece2id["[ENC]"]
        self.unk_token_id = self._piece2id.get("[UNK]", 100)
        self.pad_token_id = self._piece2id.get("[PAD]", 0)
        self.cls_token_id = self._piece2id.get("[CLS]", 101)
        self.sep_token_id = self._piece2id.get("[SEP]", 102)

    def _basic(self, text):
        import unicodedata
        out = []
        for ch in text:                                   # clean_text + CJK spacing
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or (ch not in "\t\n\r" and unicodedata.category(ch) in ("Cc", "Cf")):
                continue
            if _is_cjk(cp):
                out.append(" " + ch + " ")
            elif ch in " \t\n\r" or unicodedata.category(ch) == "Zs":
                out.append(" ")
            else:
                out.append(ch)
        words = []
        for tok in "".join(out).split():
            tok = "".join(c for c in unicodedata.normalize("NFD", tok.lower()) if unicodedata.category(c) != "Mn")
            cur = ""
            for ch in tok:                                # punctuation splits
                if _is_punct(ch):
                    if cur:
                        words.append(cur)
                        cur = ""
                    words.append(ch)
                else:
                    cur += ch
            if cur:
                words.append(cur)
        return words

    def tokenize(self, text):
        out = []
        for w in self._basic(text):
            out.extend(self._pieces(w))
        return out

    def _pieces(self, w):
        if len(w) > self.max_chars_per_word:
            return ["[UNK]"]
        start, sub = 0, []
        while start < len(w):
            end = len(w)
            piece = None
            while start < end:
                s = w[start:end]
                if start > 0:
                    s = "##" + s
                if s in self._piece2id:
                    piece = s
                    break
                end -= 1
            if piece is None:
                return ["[UNK]"]
            sub.append(piece)
            start = end
        return sub

    def _piece_id(self, piece):
        return self._piece2id.get(piece, self.unk_token_id)

    def decode(self, ids):
        return " ".join(self._id2piece[int(i)] for i in ids)
