"""Host-side text tokenizer of the front-end: the reference's `TextTokenizer` (indextts/utils/front.py:231-343)
over SentencePiece, with the same method names and results.

    tok = TextTokenizer("checkpoints/bpe.model", normalizer)      # normalizer: any object with .normalize(str) -> str, or None
    tokens = tok.tokenize(text); ids = tok.convert_tokens_to_ids(tokens); segments = tok.split_segments(tokens, 120)

The text normaliser of the reference (front.py:11-229) wraps WeTextProcessing / wetext grammars that are not in this image; it is
pluggable here (pass the reference's own `TextNormalizer()` where those packages exist).  The pipeline of `encode` is the reference's:
normalise -> split CJK characters apart and upper-case (common.py:29-51) -> SentencePiece.  `decode` undoes the CJK spacing
(common.py:54-82).  tests/test_tokenizer_cpu.py pins all of it to the reference class on a small SentencePiece model.
"""
from __future__ import annotations

import os
import re
from typing import List, Union

from . import segmenter

# CJK ranges of common.py:46-49 (from nltk's tokenize.util)
_CJK = re.compile("([\u1100-\u11ff\u2e80-\ua4cf\ua840-\uD7AF\uF900-\uFAFF\uFE30-\uFE4F\uFF65-\uFFDC\U00020000-\U0002FFFF])")
_ENGLISH_RUN = re.compile(r"([A-Z]+(?:[\s-][A-Z-]+)*)", re.IGNORECASE)
_SENT_PLACEHOLDER = re.compile(r"^.*?(<sent_(\d+)>)")


def tokenize_by_cjk_char(line: str, do_upper_case: bool = True) -> str:
    """"你好世界是 hello world 的中文" -> "你 好 世 界 是 HELLO WORLD 的 中 文"."""
    parts = _CJK.split(line.strip())
    return " ".join(w.strip().upper() if do_upper_case else w.strip() for w in parts if w.strip())


def de_tokenized_by_cjk_char(line: str, do_lower_case: bool = False) -> str:
    """The inverse: drops the spaces between CJK characters and keeps English runs together."""
    runs = _ENGLISH_RUN.findall(line)
    for i, sent in enumerate(runs):
        line = line.replace(sent, f"<sent_{i}>")
    words = line.split()
    for i in range(len(words)):
        m = _SENT_PLACEHOLDER.match(words[i])
        if m:
            words[i] = words[i].replace(m.group(1), runs[int(m.group(2))])
            if do_lower_case:
                words[i] = words[i].lower()
    return "".join(words)


class TextTokenizer:
    def __init__(self, vocab_file: str, normalizer=None):
        if vocab_file is None:
            raise ValueError("vocab_file is None")
        if not os.path.exists(vocab_file):
            raise ValueError(f"vocab_file {vocab_file} does not exist")
        from sentencepiece import SentencePieceProcessor
        self.vocab_file = vocab_file
        self.normalizer = normalizer
        if normalizer is not None and hasattr(normalizer, "load"):
            normalizer.load()
        self.sp_model = SentencePieceProcessor(model_file=vocab_file)

    vocab_size = property(lambda self: self.sp_model.GetPieceSize())
    unk_token, pad_token, bos_token, eos_token = "<unk>", None, "<s>", "</s>"
    pad_token_id, bos_token_id, eos_token_id = -1, 0, 1
    unk_token_id = property(lambda self: self.sp_model.unk_id())

    @property
    def special_tokens_map(self):
        return {"unk_token": self.unk_token, "pad_token": self.pad_token, "bos_token": self.bos_token, "eos_token": self.eos_token}

    def get_vocab(self):
        return {self.convert_ids_to_tokens(i): i for i in range(self.vocab_size)}

    def convert_ids_to_tokens(self, ids: Union[List[int], int]):
        return self.sp_model.IdToPiece(ids)

    def convert_tokens_to_ids(self, tokens: Union[List[str], str]) -> List[int]:
        if isinstance(tokens, str):
            tokens = [tokens]
        return [self.sp_model.PieceToId(t) for t in tokens]

    def tokenize(self, text: str) -> List[str]:
        return self.encode(text, out_type=str)

    def _pre(self, text: str) -> str:
        if self.normalizer is not None:
            text = self.normalizer.normalize(text)
        return tokenize_by_cjk_char(text)

    def encode(self, text: str, **kwargs):
        if len(text) == 0:
            return []
        if len(text.strip()) == 1:                       # a single character skips normaliser and CJK splitting (front.py:319-320)
            return self.sp_model.Encode(text, out_type=kwargs.pop("out_type", int), **kwargs)
        return self.sp_model.Encode(self._pre(text), out_type=kwargs.pop("out_type", int), **kwargs)

    def batch_encode(self, texts: List[str], **kwargs):
        return self.sp_model.Encode([self._pre(t) for t in texts], out_type=kwargs.pop("out_type", int), **kwargs)

    def decode(self, ids: Union[List[int], int], do_lower_case: bool = False, **kwargs):
        if isinstance(ids, int):
            ids = [ids]
        return de_tokenized_by_cjk_char(self.sp_model.Decode(ids, out_type=kwargs.pop("out_type", str), **kwargs), do_lower_case=do_lower_case)

    split_segments_by_token = staticmethod(segmenter.split_segments_by_token)

    def split_segments(self, tokenized: List[str], max_text_tokens_per_segment: int = 120, quick_streaming_tokens: int = 0) -> List[List[str]]:
        return segmenter.split_segments(tokenized, max_text_tokens_per_segment, quick_streaming_tokens)
