"""CPU: the tokenizer mirror (indextts_amd/tokenizer.py) against fixtures the reference's own TextTokenizer produced
(tests/golden/make_golden.py::make_tokenizer) over the small SentencePiece model committed beside them."""
import json
import os
import warnings

import pytest

from indextts_amd.tokenizer import TextTokenizer, de_tokenized_by_cjk_char, tokenize_by_cjk_char

HERE = os.path.join(os.path.dirname(__file__), "golden")
G = json.load(open(os.path.join(HERE, "tokenizer.json"), encoding="utf-8"))


def test_tokenizer_matches_reference():
    tok = TextTokenizer(os.path.join(HERE, "tiny_bpe.model"))
    assert tok.vocab_size == G["meta"]["vocab_size"] and tok.unk_token_id == G["meta"]["unk_token_id"]
    assert [tok.convert_ids_to_tokens(i) for i in range(12)] == G["meta"]["vocab_head"]
    assert tok.batch_encode([c["text"] for c in G["cases"][:3]]) == G["meta"]["batch_encode"]
    assert (tok.bos_token_id, tok.eos_token_id, tok.pad_token_id) == (0, 1, -1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for c in G["cases"]:
            tokens = tok.tokenize(c["text"])
            assert tokens == c["tokens"], c["text"]
            ids = tok.convert_tokens_to_ids(tokens)
            assert ids == c["ids"] and tok.encode(c["text"]) == c["encode"]
            if ids:
                assert tok.decode(ids) == c["decoded"] and tok.decode(ids, do_lower_case=True) == c["decoded_lower"]
            assert tok.split_segments(tokens, 8) == c["segments_8"]
            assert tok.split_segments(tokens, 20, quick_streaming_tokens=4) == c["segments_20_q4"]


def test_cjk_helpers_and_normalizer_hook():
    assert tokenize_by_cjk_char("你好世界是 hello world 的中文") == "你 好 世 界 是 HELLO WORLD 的 中 文"
    assert de_tokenized_by_cjk_char("你 好 世 界 是 HELLO WORLD 的 中 文") == "你好世界是HELLO WORLD的中文"
    assert de_tokenized_by_cjk_char("SEE YOU!", do_lower_case=True) == "see you!"

    class Upper:                                   # any object with .normalize(); .load() is called when present (front.py:240-241)
        loaded = False
        def load(self): self.loaded = True
        def normalize(self, t): return t.replace("&", " and ")
    n = Upper()
    tok = TextTokenizer(os.path.join(HERE, "tiny_bpe.model"), n)
    assert n.loaded and tok.tokenize("fox & dog") == TextTokenizer(os.path.join(HERE, "tiny_bpe.model")).tokenize("fox  and  dog")
    with pytest.raises(ValueError):
        TextTokenizer(os.path.join(HERE, "missing.model"))
