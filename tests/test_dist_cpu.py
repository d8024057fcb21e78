"""CPU, world_size 2, gloo: the data-parallel exchange steps (conditioning broadcast, waveform gather) and sharding."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from indextts_amd.config import PipelineConfig


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from indextts_amd.dist import broadcast_conditioning, gather_waveforms, shard_bounds
    from indextts_amd.infer_v2 import PromptConditioning
    cfg = PipelineConfig.tiny()
    ref = PromptConditioning.synthetic(cfg, prompt_frames=13, tag="dist/prompt")
    shapes = ref.shapes()
    got = broadcast_conditioning(ref if rank == 0 else None, shapes, torch.device("cpu"))
    ok = all(torch.equal(getattr(got, f), getattr(ref, f)) for f in PromptConditioning.FIELDS)
    lo, hi = shard_bounds(5, world, rank)
    wavs = [torch.full((1, 10 + 3 * i + rank), float(100 * rank + i)) for i in range(3)]
    res = gather_waveforms(wavs, dst=0)
    if rank == 0:
        for r in range(world):
            for i in range(3):
                w = res[r][i]
                ok = ok and w.shape == (1, 10 + 3 * i + r) and bool((w == 100 * r + i).all())
    else:
        ok = ok and res is None
    # uneven shards: 5 utterances over 2 ranks (3 + 2), then 1 over 2 (1 + 0: an empty shard)
    for total in (5, 1):
        lo2, hi2 = shard_bounds(total, world, rank)
        mine = [torch.full((1, 7 + i), float(i)) for i in range(lo2, hi2)]
        res = gather_waveforms(mine, dst=0, device="cpu")
        if rank == 0:
            flat = [w for r in range(world) for w in res[r]]
            ok = ok and len(flat) == total and all(w.shape == (1, 7 + i) and bool((w == i).all()) for i, w in enumerate(flat))
        else:
            ok = ok and res is None
    q.put((rank, ok, (lo, hi)))
    dist.destroy_process_group()


def test_broadcast_and_gather_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in results] == [True, True]
    assert [r[2] for r in results] == [(0, 3), (3, 5)]


def test_shard_bounds_cover_everything():
    from indextts_amd.dist import shard_bounds, sort_by_length
    for n in (0, 1, 7, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
    assert sort_by_length([5, 2, 9, 2]) == [1, 3, 0, 2]


def test_conditioning_pack_roundtrip():
    from indextts_amd.infer_v2 import PromptConditioning
    c = PromptConditioning.synthetic(PipelineConfig.tiny(), prompt_frames=9)
    r = PromptConditioning.unpack(c.pack(), c.shapes())
    assert all(torch.equal(getattr(c, f), getattr(r, f)) for f in PromptConditioning.FIELDS)


def test_bench_self_launch_command():
    """`python bench.py --gpus N` starts N ranks under torch.distributed.run on 127.0.0.1, the command shape the driver uses,
    and forwards its own flags unchanged."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cmd = bench.launcher_command(["--gpus", "8", "--steps", "5", "--warmup", "1"], 8, 29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=8" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(root, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "1"]


class _StubTTS:
    """Stands in for IndexTTS2 on the CPU: the 'waveform' of an utterance encodes its tokens, so order and content can be checked."""

    def __init__(self):
        self.device = torch.device("cpu")
        self.cfg = PipelineConfig.tiny()
        self.calls = []

    def synthesize_batch(self, toks, cond, max_mel_tokens=1500, noise=None, **kw):
        stop = self.cfg.gpt.stop_text_token
        self.calls.append(tuple(toks.shape))
        out = []
        for row in toks:
            ids = row[row != stop]
            n = int(ids.numel())
            out.append((ids.float().sum() + cond.style.sum()).reshape(1, 1).expand(1, 20 + 3 * n).clone())
        return out


def _sharded_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from indextts_amd.dist import ShardedSynthesizer
    from indextts_amd.infer_v2 import PromptConditioning
    cfg = PipelineConfig.tiny()
    cond = PromptConditioning.synthetic(cfg, prompt_frames=11, tag="dist/prompt2")
    g = torch.Generator().manual_seed(5)
    ok, shapes_seen = True, []
    for n in (7, 1, 0, 12):            # uneven shards, a single utterance (one rank idle), nothing at all, several batches per rank
        lens = torch.randint(1, 9, (n,), generator=g).tolist()
        texts = [torch.randint(2, 200, (L,), generator=g).tolist() for L in lens]
        tts = _StubTTS()
        sh = ShardedSynthesizer(tts, batch_size=4)
        res = sh.synthesize(texts if rank == 0 else None, cond if rank == 0 else None, cond.shapes(), max_mel_tokens=16)
        shapes_seen.append(tts.calls)
        if rank == 0:
            ok = ok and len(res) == n
            for t, w in zip(texts, res):           # original order restored, each waveform from its own tokens
                ok = ok and w.shape == (1, 20 + 3 * len(t)) and abs(float(w[0, 0]) - (sum(t) + float(cond.style.sum()))) < 1e-3
        else:
            ok = ok and res is None
    q.put((rank, ok, shapes_seen))
    dist.destroy_process_group()


def test_synthesize_sharded_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in results] == [True, True]
    # 7 utterances -> 4 + 3 (batches 4 | 3), 1 -> 1 + 0, 0 -> nothing, 12 -> 6 + 6 (batches 4, 2 each); sorted by length inside
    r0, r1 = results[0][2], results[1][2]
    assert [len(c) for c in r0] == [1, 1, 0, 2] and [len(c) for c in r1] == [1, 0, 0, 2]
    assert [c[0] for c in r0[3]] == [4, 2] and r0[0][0][0] == 4 and r1[0][0][0] == 3


def test_merge_keeps_kernels_rule():
    """serving.merge_keeps_kernels: a decode merge is allowed only where every request runs the same kernels merged as alone."""
    from indextts_amd.serving import merge_keeps_kernels as ok
    # real requests: 16 utterances x (34 + 128 + 3) prompt rows -- far above the 256-row GEMM threshold
    assert ok([16, 16, 16], 165, True, False, 17)                     # fp32 weights: one decode family
    assert not ok([16, 16, 16], 165, True, True, 17)                  # compact weights: 16 alone = fp32-MFMA GEMV, 48 merged = plane GEMV
    assert ok([16, 16, 16], 165, True, True, 5)                       # plane GEMV from 5 rows on: same family alone and merged
    assert ok([4, 4, 4], 165, True, True, 17) and not ok([4, 4, 4, 8], 165, True, True, 17)     # 12 rows stay below 17, 20 do not
    assert ok([32, 32], 165, True, True, 17)                          # both sides on the plane GEMV already
    # round 3's toy requests: 2-4 utterances x 19 prompt rows = 38-76 GEMM rows each
    assert not ok([2, 3, 4], 19, True, False, 17)                     # split-bf16 mode: merged rows would cross 256
    assert ok([2, 3, 4], 19, False, False, 17)                        # exact mode: no row-count dispatch
    assert ok([7], 19, True, True, 17)
