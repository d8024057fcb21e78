"""Data-parallel utterance sharding over the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).  The path
shards with NO collective inside the hot loop: every rank holds a full weight replica and synthesises its own
contiguous slice of the utterance list.  The two exchange steps are
  * before: the prompt-conditioning bundle, computed on rank 0, is broadcast as ONE packed buffer (~6 MB fp32);
  * after:  per-utterance sample counts are all-gathered, then the padded waveforms are gathered to rank 0.
Payloads are tiny against 7 x 153 GB/s of xGMI links, so both are single one-shot collectives, not rings
of buckets.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .infer_v2 import PromptConditioning


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous split of n utterances; the first n % world ranks take one extra."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sort_by_length(lengths: Sequence[int]) -> List[int]:
    """Indices sorted by text length so that neighbouring utterances (same shard, same decode batch) have similar step counts."""
    return sorted(range(len(lengths)), key=lambda i: (lengths[i], i))


def broadcast_conditioning(cond: Optional[PromptConditioning], shapes, device, src: int = 0) -> PromptConditioning:
    """Rank `src` passes its bundle, the others pass None; `shapes` is static metadata every rank knows
    (PromptConditioning.shapes())."""
    n = sum(int(torch.tensor(s).prod()) for s in shapes)
    if dist.get_rank() == src:
        flat = cond.pack().to(device, torch.float32).contiguous()
        assert flat.numel() == n
    else:
        flat = torch.empty(n, device=device, dtype=torch.float32)
    dist.broadcast(flat, src=src)
    return PromptConditioning.unpack(flat, shapes)


def gather_waveforms(wavs: List[torch.Tensor], dst: int = 0, device=None, same_count: bool = False) -> Optional[List[List[torch.Tensor]]]:
    """wavs: this rank's list of [1, n_i] waveforms -- the count may differ between ranks (`shard_bounds` hands out uneven
    shards) and may be zero (then pass `device`).  Returns, on rank `dst`, a list over ranks of lists of waveforms trimmed to
    their true lengths; None elsewhere.  Three one-shot collectives: all_gather of the counts, all_gather of the sample
    counts (padded to the largest shard), gather of the padded waveforms; `same_count=True` (every rank holds len(wavs)
    utterances, e.g. fixed batches) skips the first."""
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = wavs[0].device if wavs else torch.device(device if device is not None else "cpu")
    if same_count:
        counts = [len(wavs)] * world
    else:
        count = torch.tensor([len(wavs)], device=dev, dtype=torch.int64)
        counts = [torch.empty_like(count) for _ in range(world)]
        dist.all_gather(counts, count)
        counts = [int(c.item()) for c in counts]
    cmax = max(counts)
    if cmax == 0:
        return [[] for _ in range(world)] if rank == dst else None
    lens = torch.tensor([w.shape[-1] for w in wavs] + [0] * (cmax - len(wavs)), dtype=torch.int64).to(dev)
    all_lens = [torch.empty_like(lens) for _ in range(world)]
    dist.all_gather(all_lens, lens)
    nmax = max(1, int(max(int(l.max()) for l in all_lens)))
    padded = torch.zeros(cmax, nmax, device=dev, dtype=torch.float32)
    for i, w in enumerate(wavs):
        padded[i, : w.shape[-1]] = w.reshape(-1)
    out = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, out, dst=dst)
    if rank != dst:
        return None
    return [[out[r][i, : int(all_lens[r][i])].unsqueeze(0) for i in range(counts[r])] for r in range(world)]
