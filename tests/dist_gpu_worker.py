"""Worker of tests/test_dist_gpu.py: one rank of a two-rank job on ONE GPU (gloo for the collectives, both ranks on cuda:0).
The product-level sharded call -- ShardedSynthesizer over the real HIP pipeline, each rank with its own BatchPipeline -- on an
uneven utterance list of ragged texts; rank 0 compares the gathered waveforms, in the caller's order, with its own single-process
synthesis of the same batches."""
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "index-tts_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist


def main() -> int:
    from indextts_amd import synth, weights
    from indextts_amd.config import PipelineConfig
    from indextts_amd.dist import ShardedSynthesizer
    from indextts_amd.infer_v2 import IndexTTS2, PromptConditioning
    from indextts_amd.serving import BatchPipeline
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = PipelineConfig.tiny()
    wg = weights.synth_gpt_weights(cfg.gpt, tag="t/dist/gpt")
    wg["mel_head.bias"] = wg["mel_head.bias"].copy()
    wg["mel_head.bias"][cfg.gpt.stop_mel_token] = -1e4
    tts = IndexTTS2.from_state_dicts(cfg, wg, weights.synth_s2mel_weights(cfg.s2mel, tag="t/dist/s2mel"),
                                     weights.synth_bigvgan_weights(cfg.bigvgan, tag="t/dist/voc"), device=dev)
    cond = PromptConditioning.synthetic(cfg, prompt_frames=30, tag="t/dist/prompt")
    shapes = cond.shapes()
    n, M, BS = 7, 12, 2                                # 7 utterances over 2 ranks: shards of 4 and 3, batches of 2 (the last ones short)
    lens = [5, 9, 3, 7, 9, 4, 6]
    texts = [synth.integers(f"t/dist/text{i}", (lens[i],), 2, cfg.gpt.number_text_tokens).tolist() for i in range(n)]
    Tg = int(M * cfg.code_to_frame)
    noise_all = torch.from_numpy(synth.uniform("t/dist/noise", (n, cfg.s2mel.in_channels, 30 + Tg), 1.0)).to(dev)
    warnings.simplefilter("ignore")
    with BatchPipeline(tts, decode_lanes=2) as pipe:
        sh = ShardedSynthesizer(tts, batch_size=BS, pipeline=pipe)
        got = sh.synthesize(texts if rank == 0 else None, cond.to(dev) if rank == 0 else None, shapes, max_mel_tokens=M,
                            noise_fn=lambda idx: noise_all[idx])
    ok = True
    if rank == 0:
        assert got is not None and len(got) == n
        # the single-process result: the same batches (sorted by length, contiguous shards, batches of BS, padded with the stop token --
        # what ShardedSynthesizer documents), one after the other on this rank
        from indextts_amd.dist import shard_bounds, sort_by_length
        stop = cfg.gpt.stop_text_token
        order = sort_by_length(lens)
        want = [None] * n
        for r in range(world):
            lo, hi = shard_bounds(n, world, r)
            mine = order[lo:hi]
            for b0 in range(0, len(mine), BS):
                idx = mine[b0:b0 + BS]
                L = max(lens[i] for i in idx)
                toks = torch.full((len(idx), L), stop, dtype=torch.long)
                for row, i in enumerate(idx):
                    toks[row, : lens[i]] = torch.tensor(texts[i], dtype=torch.long)
                for i, w in zip(idx, tts.synthesize_batch(toks, cond.to(dev), max_mel_tokens=M, noise=noise_all[idx])):
                    want[i] = w
        for i in range(n):
            same = got[i].shape == want[i].shape and torch.equal(got[i].to(dev), want[i])
            if not same:
                print(f"utterance {i}: sharded result differs from the single-process call", flush=True)
            ok = ok and same
        print("DIST_GPU_OK" if ok else "DIST_GPU_MISMATCH", flush=True)
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
