"""Deterministic synthetic tensors (weights and inputs) for benchmarks and parity tests.

There are no IndexTTS-2 checkpoints offline (SURVEY.md "Facts established": gpt.pth,
s2mel.pth and bigvgan_generator.pt are absent), so every benchmark and parity test runs on
random-initialised weights of the reference architecture.  The generator below is a
counter-based integer hash (FNV-1a of the tensor name -> splitmix64 of the element index)
mapped to a uniform float32 in [-1, 1): integer arithmetic plus one exact float multiply, so
the SAME bits come out in this container, on the GPU box, in the golden-fixture generator
and in the tests.  Nothing here depends on numpy's or torch's RNG streams.
"""
from __future__ import annotations

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(name: str, shape, scale: float = 1.0, offset: float = 0.0) -> np.ndarray:
    """float32 tensor, elements uniform in [offset-scale, offset+scale), keyed by `name`."""
    shape = tuple(int(s) for s in np.atleast_1d(shape)) if not isinstance(shape, tuple) else shape
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(_fnv1a64(name))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + seed
    h = _splitmix64(idx)
    u = (h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))  # [0,1), exact
    out = (u * np.float32(2.0) - np.float32(1.0)) * np.float32(scale) + np.float32(offset)
    return out.astype(np.float32).reshape(shape)


def integers(name: str, shape, low: int, high: int) -> np.ndarray:
    """int64 tensor with elements in [low, high), keyed by `name`."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(_fnv1a64(name))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + seed
    h = _splitmix64(idx) >> np.uint64(11)
    return (low + (h % np.uint64(high - low)).astype(np.int64)).reshape(shape)


def fan_in_uniform(name: str, shape, fan_in: int, gain: float = 1.0) -> np.ndarray:
    """Uniform init with variance gain^2 / fan_in (keeps activations O(1) through a layer)."""
    return uniform(name, shape, scale=gain * float(np.sqrt(3.0 / max(1, fan_in))))
