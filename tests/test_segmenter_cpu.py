"""CPU: the token-level segment splitter (indextts_amd/segmenter.py) against fixtures produced by the reference's own
TextTokenizer.split_segments_by_token (tests/golden/make_golden.py::make_segments; front.py:345-422)."""
import json
import os
import warnings

from indextts_amd import segmenter


def test_split_segments_matches_reference_fixtures(golden_dir):
    with open(os.path.join(golden_dir, "segments.json"), encoding="utf-8") as f:
        g = json.load(f)
    assert g["punctuation"] == segmenter.PUNCTUATION_MARKS_TOKENS
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i, c in enumerate(g["cases"]):
            got = segmenter.split_segments_by_token(c["tokens"], g["punctuation"], c["limit"], c["quick"])
            assert got == c["segments"], (i, c["limit"], c["quick"])
    # the default wrapper = TextTokenizer.split_segments
    c = next(c for c in g["cases"] if c["limit"] == 120 and c["quick"] == 0 and len(c["tokens"]) > 100)
    assert segmenter.split_segments(c["tokens"]) == c["segments"]


def test_split_segments_with_ids():
    """Ids instead of SentencePiece strings: the special tokens are passed as ids."""
    toks = [5, 6, 7, 1, 8, 9, 10, 2, 11, 12, 13, 14, 1]
    out = segmenter.split_segments_by_token(toks, [1], 6, comma_tokens=[2], dash_token=3, apostrophe_tokens=[4])
    assert [t for seg in out for t in seg] == toks and all(len(s) <= 6 for s in out)
    assert segmenter.split_segments_by_token([], [1], 6) == []
