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


class ShardedSynthesizer:
    """The product-level data-parallel call (BASELINE configs[3]: 256 utterances over 8 GPUs): one list of
    utterances on rank `src` in, their waveforms in the SAME order on rank `dst` out.

        sh = ShardedSynthesizer(tts, batch_size=16, pipeline=BatchPipeline(tts))      # every rank
        wavs = sh.synthesize(texts if rank == 0 else None, cond if rank == 0 else None, shapes, max_mel_tokens=...)

    Steps (collectives only in 1, 2 and 5 -- nothing inside the hot loop):
      1. the token ids travel as ONE packed int64 buffer [n, Lmax + 1] (length in the last column) behind a 2-element header;
      2. the conditioning bundle is broadcast (`broadcast_conditioning`);
      3. every rank sorts the list by text length (neighbouring utterances decode for a similar number of steps) and takes its
         contiguous slice of the SORTED list (`shard_bounds`): shards differ by at most one utterance;
      4. the slice is synthesised in batches of `batch_size` -- through the rank's `BatchPipeline` when one is given (decode chains
         of several batches in flight), else batch after batch;
      5. `gather_waveforms` to rank `dst`, which restores the original order.
    `begin()` / `finish()` split the call at the device work, so a serving loop can have several calls in flight (bench.py).
    """

    def __init__(self, tts, batch_size: int = 16, pipeline=None, src: int = 0, dst: int = 0):
        self.tts, self.batch_size, self.pipeline, self.src, self.dst = tts, int(batch_size), pipeline, src, dst
        self.device = torch.device(tts.device)

    def _broadcast_texts(self, texts) -> List[List[int]]:
        rank = dist.get_rank()
        head = torch.zeros(2, dtype=torch.int64, device=self.device)
        if rank == self.src:
            head[0], head[1] = len(texts), max((len(t) for t in texts), default=0)
        dist.broadcast(head, src=self.src)
        n, lmax = int(head[0]), int(head[1])
        buf = torch.zeros(n, lmax + 1, dtype=torch.int64, device=self.device)
        if rank == self.src and n:
            host = torch.zeros(n, lmax + 1, dtype=torch.int64)
            for i, t in enumerate(texts):
                host[i, : len(t)] = torch.as_tensor(list(t), dtype=torch.int64)
                host[i, lmax] = len(t)
            buf.copy_(host)
        if n:
            dist.broadcast(buf, src=self.src)
        host = buf.cpu()
        return [host[i, : int(host[i, lmax])].tolist() for i in range(n)]

    def begin(self, texts: Optional[Sequence[Sequence[int]]], cond: Optional[PromptConditioning], shapes, max_mel_tokens: int = 1500,
              noise_fn=None, **kw) -> dict:
        """Collectives 1-2 and the submission of this rank's batches.  noise_fn(indices) -> CFM noise [len(indices), C, T] or None."""
        world, rank = dist.get_world_size(), dist.get_rank()
        texts = self._broadcast_texts(texts)
        cond = broadcast_conditioning(cond, shapes, self.device, src=self.src)
        order = sort_by_length([len(t) for t in texts])
        lo, hi = shard_bounds(len(texts), world, rank)
        mine = order[lo:hi]
        stop = self.tts.cfg.gpt.stop_text_token
        jobs = []
        for b0 in range(0, len(mine), self.batch_size):
            idx = mine[b0:b0 + self.batch_size]
            L = max(len(texts[i]) for i in idx)
            toks = torch.full((len(idx), L), stop, dtype=torch.long)
            for r, i in enumerate(idx):
                toks[r, : len(texts[i])] = torch.as_tensor(texts[i], dtype=torch.long)
            noise = noise_fn(idx) if noise_fn is not None else None
            if self.pipeline is not None:
                jobs.append(self.pipeline.submit(toks, cond, max_mel_tokens=max_mel_tokens, noise=noise, **kw))
            else:
                jobs.append(self.tts.synthesize_batch(toks, cond, max_mel_tokens=max_mel_tokens, noise=noise, **kw))
        return {"jobs": jobs, "order": order, "n": len(texts)}

    def finish(self, handle: dict) -> Optional[List[torch.Tensor]]:
        """Waits for this rank's batches, gathers (collective 5); on rank `dst` the waveforms in the caller's original order."""
        wavs = []
        for j in handle["jobs"]:
            wavs.extend(j.result() if hasattr(j, "result") else j)
        handle["local"] = wavs                                # this rank's own utterances (sorted order), for its callers
        per_rank = gather_waveforms(wavs, dst=self.dst, device=self.device)
        if dist.get_rank() != self.dst:
            return None
        flat = [w for r in per_rank for w in r]            # sorted order: rank 0's slice first
        assert len(flat) == handle["n"]
        out: List[Optional[torch.Tensor]] = [None] * handle["n"]
        for pos, i in enumerate(handle["order"]):
            out[i] = flat[pos]
        return out

    def synthesize(self, texts, cond, shapes, **kw) -> Optional[List[torch.Tensor]]:
        return self.finish(self.begin(texts, cond, shapes, **kw))


def synthesize_sharded(tts, texts, cond, shapes, batch_size: int = 16, pipeline=None, **kw) -> Optional[List[torch.Tensor]]:
    """One-shot form of ShardedSynthesizer.synthesize()."""
    return ShardedSynthesizer(tts, batch_size=batch_size, pipeline=pipeline).synthesize(texts, cond, shapes, **kw)
