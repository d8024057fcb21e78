"""Token-level segment splitter of the text front-end: how `infer()` cuts a tokenised text into the
segments its loop synthesises one by one (reference: TextTokenizer.split_segments_by_token / split_segments,
indextts/utils/front.py:345-436; call site infer_v2.py:700).  Host-side list logic, no tokenizer needed: tokens are any
hashable values (SentencePiece strings in the reference; ids work when the special tokens are given as ids).

Rules restated from the reference:
  * walk the tokens, cut after a split token once the running segment has more than 2 tokens (an apostrophe token right
    after the split token is glued to the segment -- the reference also leaves it at the head of the next one, see below);
  * a running segment that outgrows the limit is re-split on commas (once, when commas were not already the split set),
    else on "-", else chopped into limit-sized pieces with a RuntimeWarning;
  * afterwards neighbours are merged while the pair fits the limit (and the tokens seen so far exceed
    quick_streaming_tokens), or fits half the limit.
"""
from __future__ import annotations

import warnings
from typing import Hashable, List, Sequence

PUNCTUATION_MARKS_TOKENS = [".", "!", "?", "▁.", "▁?", "▁..."]          # front.py:424-432
COMMA_TOKENS = [",", "▁,"]
DASH_TOKEN = "-"
APOSTROPHE_TOKENS = ["'", "▁'"]


def split_segments_by_token(tokenized: Sequence[Hashable], split_tokens: Sequence[Hashable], max_text_tokens_per_segment: int,
                            quick_streaming_tokens: int = 0, comma_tokens: Sequence[Hashable] = COMMA_TOKENS,
                            dash_token: Hashable = DASH_TOKEN, apostrophe_tokens: Sequence[Hashable] = APOSTROPHE_TOKENS
                            ) -> List[list]:
    if len(tokenized) == 0:
        return []
    limit = max_text_tokens_per_segment
    commas_are_split = any(c in split_tokens for c in comma_tokens)
    segments: List[list] = []
    cur: list = []
    for i, token in enumerate(tokenized):
        cur.append(token)
        n = len(cur)
        if not commas_are_split and any(c in cur for c in comma_tokens):
            # the reference re-splits the running segment on commas as soon as it contains one (front.py:366-370)
            sub = split_segments_by_token(cur, list(comma_tokens), limit, quick_streaming_tokens, comma_tokens, dash_token, apostrophe_tokens)
        elif dash_token not in split_tokens and dash_token in cur:
            sub = split_segments_by_token(cur, [dash_token], limit, quick_streaming_tokens, comma_tokens, dash_token, apostrophe_tokens)
        elif n <= limit:
            if token in split_tokens and n > 2:
                if i < len(tokenized) - 1 and tokenized[i + 1] in apostrophe_tokens:
                    # glued to this segment; the reference's `i += 1` does not advance its for-loop, so the apostrophe is
                    # ALSO the first token of the next segment -- kept, parity over taste
                    cur.append(tokenized[i + 1])
                segments.append(cur)
                cur = []
            continue
        else:
            sub = [cur[j:j + limit] for j in range(0, len(cur), limit)]
            warnings.warn(f"The tokens length of segment exceeds limit: {limit}, Tokens in segment: {cur}.Maybe unexpected behavior",
                          RuntimeWarning)
        segments.extend(sub)
        cur = []
    if cur:
        assert len(cur) <= limit
        segments.append(cur)
    merged: List[list] = []
    total = 0
    for seg in segments:
        total += len(seg)
        if not seg:
            continue
        if not merged:
            merged.append(seg)
        elif len(merged[-1]) + len(seg) <= limit and total > quick_streaming_tokens:
            merged[-1] = merged[-1] + seg
        elif len(merged[-1]) + len(seg) <= limit / 2:
            merged[-1] = merged[-1] + seg
        else:
            merged.append(seg)
    return merged


def split_segments(tokenized: Sequence[Hashable], max_text_tokens_per_segment: int = 120, quick_streaming_tokens: int = 0) -> List[list]:
    """TextTokenizer.split_segments (front.py:433-436)."""
    return split_segments_by_token(tokenized, PUNCTUATION_MARKS_TOKENS, max_text_tokens_per_segment, quick_streaming_tokens)
