#!/usr/bin/env python3
"""Benchmark of the MI355X-native IndexTTS-2 hot path.

    python bench.py --gpus N --steps K --warmup W          (N > 1: re-launches itself, one rank per GPU, before touching the GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): synthesised audio seconds per wall-second (whole job, all GPUs) and RTF.
A "step" = one pass of the hot path over one batch of synthetic inputs already resident in HBM.
Workloads (BASELINE.json `configs`):
  pipeline  configs[2] (default): IndexTTS-2 full pipeline GPT decode -> latent pass -> s2mel (CFM 20 steps, cfg 0.7) ->
            BigVGAN, batch 16 utterances per GPU, 128 text tokens, fixed 512 codes/utterance (10.2 s audio each,
            163.5 s per step per GPU), prompt Tp = 689 frames, greedy decode with repetition penalty 10.
            With --gpus 8 the default is configs[3]: 256 utterances sharded 32 per GPU, same per-utterance recipe.
  vocoder   configs[1]: BigVGAN-only mel->wav, batch 8 of 80x800 mels per GPU (74.3 s audio / step / GPU)
One process per GPU; utterance batches shard data-parallel (weak scaling: every rank synthesises its own batch, full
weight replica); exchange steps of the path: conditioning broadcast from rank 0 before, waveform gather to rank 0 after
(RCCL over xGMI), nothing inside the hot loop.
Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed on its launch stream) and
`cpu_baseline` (the CPU oracle timed on this box's host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "index-tts_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0 # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md "HBM3E peak BW" (spec)
# Implementation noise a decode mode may add to a logit against the fp32 CPU oracle running the same (rounded) model -- the bounds
# tests/test_fullsize_gpu.py asserts (logit std ~ 1 on these random-init weights): fp32 KV cache = summation order only; bf16 KV cache =
# a key / value whose fp32 value differs in the last bit between two summation orders may round to the other bf16 neighbour.
LOGIT_NOISE_BOUND = {"f32": 1e-4, "bf16": 1.5e-2}
# Kernel families are named after the kernel function they time, as rocprofv3 prints it (csrc/prof.h): a family's line below matches
# the rows of profiles/rNN_*_kernel_stats.csv whose name contains it.
MFMA_KERNELS = ("conv1d_mfma_kernel", "conv1d_bf16x3_kernel", "gemm_tn_kernel", "gemm_bf16x3", "flash_attn")
SPLIT_BF16_KERNELS = ("conv1d_bf16x3_kernel", "gemm_bf16x3", "flash_attn_bf16x3_kernel", "flash_attn_planes_kernel")
DECODE_KERNELS = ("gemv_fx_kernel", "gemv_pl_kernel", "gemv_fx_combine_kernel", "decode_attn_kernel", "decode_attn16_kernel", "sample_greedy_kernel", "sample_warp_kernel", "embed_step_kernel",
                  "advance_state_kernel", "beam_")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def roofline_from_profile(prof, steps, workload="pipeline", split_bf16=True, default_config=True, overhead_ms=0.0):
    # an event pair around a launch reads `overhead_ms` even for an empty kernel: families are ranked (dominant kernel, time
    # shares) by their time net of that fixed cost, or thousands of 9-us launches outweigh a few hundred-us ones by accounting
    # alone (rocprofv3's kernel durations, profiles/, do not carry it).  `achieved` of the dominant kernel uses the raw reading.
    for v in prof.values():
        v["net_ms"] = max(v["ms"] - v["launches"] * overhead_ms, 0.05 * v["ms"])
    tot_ms = sum(v["net_ms"] for v in prof.values())
    for name, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
        log(f"[bench] kernel {name:22s} launches/step {v['launches'] // steps:6d}  {v['ms'] / steps:9.3f} ms/step raw, {v['net_ms'] / steps:9.3f} net "
            f"({100 * v['net_ms'] / tot_ms:5.1f}%)  {v['flops'] / max(v['ms'], 1e-9) / 1e9:8.2f} TFLOP/s  "
            f"{v['bytes'] / max(v['ms'], 1e-9) / 1e6:8.1f} GB/s(alg)")
    dom_name, dom = max(prof.items(), key=lambda kv: kv[1]["net_ms"])
    # `achieved` and the per-launch duration use the time net of the bracketing's fixed cost: for a 7-us decode GEMV the raw
    # reading is a third too long, and it is the net figure that agrees with rocprofv3's kernel durations (profiles/)
    dom_ms = dom["net_ms"]
    if dom_name.startswith(MFMA_KERNELS):
        achieved = dom["flops"] / dom_ms / 1e9
        if dom_name.startswith(SPLIT_BF16_KERNELS):
            # split-bf16: every algorithmic multiply-add is executed as three bf16 MFMA products, so the dense bf16 peak,
            # expressed in ALGORITHMIC flops, is 2500/3 TFLOP/s
            peak, note = PEAK_BF16_MFMA_TFLOPS / 3.0, "dense bf16 MFMA peak 2500 TFLOP/s / 3 products per fp32-class product"
        else:
            peak, note = PEAK_F32_MFMA_TFLOPS, "f32-input MFMA peak"
        r = {"kernel": dom_name, "bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
             "frac": round(achieved / peak, 4), "traffic": None, "peak_note": note,
             "alg_flops_per_launch": dom["flops"] / dom["launches"]}
    else:
        achieved = dom["bytes"] / dom_ms / 1e6
        r = {"kernel": dom_name, "bound": "hbm", "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
             "frac": round(achieved / PEAK_HBM_GBS, 4), "traffic": None, "alg_bytes_per_launch": dom["bytes"] / dom["launches"]}
    # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (separate --pmc FETCH_SIZE /
    # WRITE_SIZE runs, gfx950 correction applied; profiles/README.md) -- null when no pass is on file for this kernel
    for rnd in ("r04", "r03"):      # the newest PMC passes on file
        try:
            with open(os.path.join(ROOT, "profiles", f"traffic_{rnd}.json")) as f:
                tr = json.load(f)["kernels"].get(dom_name)
            if tr and workload == "pipeline" and default_config:      # the PMC passes on file are of the default configs[2] command
                r["traffic"] = tr["hbm_bytes_per_launch"]
                r["traffic_source"] = f"profiles/traffic_{rnd}.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bytes = (2*FETCH+WRITE)*1024)"
                break
        except (OSError, KeyError, ValueError):
            pass
    r["launches_per_step"] = dom["launches"] // steps
    r["avg_launch_ms"] = round(dom_ms / dom["launches"], 5)
    r["avg_launch_ms_raw"] = round(dom["ms"] / dom["launches"], 5)
    r["event_overhead_us_per_launch"] = round(1000 * overhead_ms, 3)
    r["kernel_time_share"] = {k: round(v["net_ms"] / tot_ms, 4) for k, v in prof.items()}
    r["kernel_ms_per_step"] = {k: round(v["net_ms"] / steps, 3) for k, v in prof.items()}
    r["kernel_ms_per_step_raw"] = {k: round(v["ms"] / steps, 3) for k, v in prof.items()}
    r["kernel_launches_per_step"] = {k: v["launches"] // steps for k, v in prof.items()}
    top = {}
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["net_ms"])[:5]:      # the five largest families, each against its own roof
        if k.startswith(MFMA_KERNELS):
            pk = PEAK_BF16_MFMA_TFLOPS / 3.0 if k.startswith(SPLIT_BF16_KERNELS) else PEAK_F32_MFMA_TFLOPS
            a = v["flops"] / v["net_ms"] / 1e9
            top[k] = {"bound": "mfma", "achieved": round(a, 1), "peak": round(pk, 1), "unit": "TFLOP/s", "frac": round(a / pk, 4), "time_share": round(v["net_ms"] / tot_ms, 4)}
        else:
            a = v["bytes"] / v["net_ms"] / 1e6
            top[k] = {"bound": "hbm", "achieved": round(a, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(a / PEAK_HBM_GBS, 4), "time_share": round(v["net_ms"] / tot_ms, 4)}
    r["top_kernels"] = top
    r["kernel_names"] = "kernel function names as rocprofv3 prints them (csrc/prof.h): match profiles/*_kernel_stats.csv rows by substring"
    return r


def _lib_geometry():
    from indextts_amd import _lib
    return _lib.get_decode_geometry()


def _lib_plane_rows():
    from indextts_amd import _lib
    return _lib.get_decode_plane_rows()


def stage_rooflines(stages, prof, nprof, B, L, M, Tp, Tg, n_cfm, w_bytes_per_param, kv_bytes=4):
    """One roofline per stage of the step (the step's time is split three ways: HBM-bound decode, MFMA-bound s2mel and vocoder).
    Work units are SURVEY.md 8(d)'s: GPT decode bytes/step = W + B*S*2*24*1280*sizeof(kv), W = 482.76 M params; s2mel flops =
    N*2B*T*(148.4e6 + 26624*T) (what the reference executes; the solver here skips the prompt frames' post-transformer part, so it
    EXECUTES less); BigVGAN 1.80e9 flop per mel frame.  Split-bf16 stages are priced against 2500 / 3 TFLOP/s."""
    out = {}
    P = L + 36 + 1                      # [pad | 34 cond | start, text, stop] + start_mel
    dec_bytes = sum(482.76e6 * w_bytes_per_param + B * (P + n) * 2 * 24 * 1280 * kv_bytes for n in range(M))
    t = stages.get("gpt_gen_time")
    if t:
        dec_launches = sum(v["launches"] for k, v in prof.items() if k.startswith(DECODE_KERNELS)) / max(nprof, 1)
        # The floor of one decode chain: its bytes at the ACHIEVABLE HBM rate (6.3 TB/s, MI355X_MICROARCH.md) plus one dependent kernel
        # boundary per launch (1.45 us between trivial kernels, 1.7-1.9 between streaming ones: the guide's price list) -- the launches of a
        # token form a dependency chain, so the boundaries do not overlap with anything of the same chain.
        lpt = dec_launches / M
        floor_us = 1e6 * (dec_bytes / M) / 6.3e12 + lpt * 1.5
        out["gpt_decode"] = {"bound": "hbm", "alg_bytes_per_step": dec_bytes, "seconds": t, "achieved": round(dec_bytes / t / 1e9, 1), "peak": PEAK_HBM_GBS,
                             "unit": "GB/s", "frac": round(dec_bytes / t / 1e9 / PEAK_HBM_GBS, 4), "launches_per_token": round(lpt, 1),
                             "us_per_token": round(1e6 * t / M, 1), "floor_us_per_token": round(floor_us, 1), "times_the_floor": round(1e6 * t / M / floor_us, 2),
                             "floor_note": "bytes per token / 6.3 TB/s (achievable HBM rate) + launches per token x 1.5 us (dependent kernel boundary)",
                             "kv_cache": "bf16" if kv_bytes == 2 else "fp32", "note": "gpt_gen_time includes the prefill (165 tokens, MFMA)"}
    T = Tp + Tg
    s2_flops = n_cfm * 2 * B * T * (148.4e6 + 26624.0 * T)
    t = stages.get("s2mel_time")
    if t:
        out["s2mel"] = {"bound": "mfma", "alg_flops_per_step": s2_flops, "seconds": t, "achieved": round(s2_flops / t / 1e12, 1),
                        "peak": round(PEAK_BF16_MFMA_TFLOPS / 3, 1), "unit": "TFLOP/s", "frac": round(s2_flops / t / 1e12 / (PEAK_BF16_MFMA_TFLOPS / 3), 4),
                        "note": "reference-algorithm flops (all frames); split-bf16 = 3 bf16 MFMAs per product"}
    v_flops = 1.80e9 * B * Tg
    t = stages.get("bigvgan_time")
    if t:
        out["bigvgan"] = {"bound": "mfma", "alg_flops_per_step": v_flops, "seconds": t, "achieved": round(v_flops / t / 1e12, 1),
                          "peak": round(PEAK_BF16_MFMA_TFLOPS / 3, 1), "unit": "TFLOP/s", "frac": round(v_flops / t / 1e12 / (PEAK_BF16_MFMA_TFLOPS / 3), 4),
                          "note": "stages with >= 192 channels are MFMA-bound, the narrow stages and the anti-alias activations HBM-bound (per-kernel lines)"}
    return out


def cpu_threads():
    # the box's CPU share, not the host's core count (oversubscribing a cgroup quota stalls for minutes)
    return max(1, min(16, len(os.sched_getaffinity(0)), os.cpu_count() or 1))


# ------------------------------------------------------------------------------------------------------
def build_vocoder(args, world, rank, dev):
    from indextts_amd import weights
    from indextts_amd.config import BigVGANConfig
    from indextts_amd.vocoder import BigVGAN
    cfg = BigVGANConfig()
    w = weights.synth_bigvgan_weights(cfg, tag="bench/bigvgan")
    voc = BigVGAN(w, cfg)
    B, Tm = args.batch or 8, args.frames
    mel = torch.from_numpy(weights.synth_mel(f"bench/mel/rank{rank}", B, cfg.num_mels, Tm)).to(dev)
    n_samples = Tm * cfg.total_upsample
    gathered = [torch.empty(B, 1, n_samples, device=dev) for _ in range(world)] if (world > 1 and rank == 0) else None

    def step():
        wav = voc(mel)
        if world > 1:
            dist.gather(wav, gathered, dst=0)
        return wav

    def profiled():
        voc(mel)

    def cpu_leg():
        from oracle import vocoder as ov
        cores = cpu_threads()
        torch.set_num_threads(cores)
        cmel = mel[:1].cpu()
        wt = {k: torch.from_numpy(v) for k, v in w.items()}
        with torch.no_grad():
            ov.bigvgan_forward(wt, cfg, cmel[:, :, :32])
            c0 = time.perf_counter()
            cw = ov.bigvgan_forward(wt, cfg, cmel)
            cdt = time.perf_counter() - c0
        caudio = cw.shape[0] * cw.shape[-1] / cfg.sampling_rate
        err = (voc(mel[:1].contiguous()).cpu() - cw).abs().max().item()
        return {"value": round(caudio / cdt, 4), "unit": "audio_s/s", "cores": cores, "kind": "port",
                "sample": f"oracle/vocoder.py BigVGAN fp32, 1 x [80 x {Tm}] mel of the same batch ({caudio:.2f} s audio)",
                "max_abs_diff_vs_gpu": err}

    desc = {"workload": f"configs[1]: BigVGAN-only mel->wav (bigvgan_v2_22khz_80band_256x, 112M params), batch {B} x [80 x {Tm}] "
                        "mels per GPU", "batch_per_gpu": B, "mel_frames": Tm}
    return step, profiled, cpu_leg, None, B * n_samples / cfg.sampling_rate, desc


def build_pipeline(args, world, rank, dev):
    from indextts_amd import synth, weights
    from indextts_amd.config import PipelineConfig
    from indextts_amd.dist import ShardedSynthesizer, shard_bounds
    from indextts_amd.infer_v2 import IndexTTS2, PromptConditioning
    cfg = PipelineConfig()
    B, L, M, Tp = args.batch or 16, args.text_tokens, args.codes, args.prompt_frames
    t0 = time.time()
    wg = weights.synth_gpt_weights(cfg.gpt, tag="bench/gpt")
    wg["mel_head.bias"][cfg.gpt.stop_mel_token] = -1e4      # fixed-length synthetic batches: never emit EOS -> exactly M codes
    ws = weights.synth_s2mel_weights(cfg.s2mel, tag="bench/s2mel")
    wv = weights.synth_bigvgan_weights(cfg.bigvgan, tag="bench/bigvgan")
    log(f"[bench] rank {rank}: synthetic weights in {time.time() - t0:.1f}s")
    compact = args.gpt_weights != "f32"
    tts = IndexTTS2.from_state_dicts(cfg, wg, ws, wv, device=dev, gpt_weight_format=args.gpt_weights, gpt_kv_format=args.gpt_kv,
                                     keep_effective_gpt=compact and rank == 0 and not args.no_cpu_baseline)
    kv16 = tts.gpt.kv_format == "bf16"
    if tts.gpt.effective_state_dict is not None:      # the CPU leg runs the SAME (rounded) model the kernels run
        wg = tts.gpt.effective_state_dict
    cond0 = PromptConditioning.synthetic(cfg, prompt_frames=Tp, tag="bench/prompt")
    shapes = cond0.shapes()
    # The job's utterance list: B per GPU (rank 0 holds all of it: ShardedSynthesizer scatters the ids and gathers the waveforms);
    # utterance i has its own token ids and CFM noise, the same whichever rank synthesises it.
    Tg = int(M * cfg.code_to_frame)
    n_all = B * world
    lo_r, hi_r = shard_bounds(n_all, world, rank)
    text_all = torch.cat([torch.from_numpy(synth.integers(f"bench/text/rank{r}", (B, L), 2, cfg.gpt.number_text_tokens)) for r in range(world)])
    text = text_all[lo_r:hi_r]
    noise = torch.from_numpy(synth.uniform(f"bench/noise/rank{rank}", (B, cfg.s2mel.in_channels, Tp + Tg), 1.7)).to(dev)
    cond_dev = cond0.to(dev)
    audio_s = B * Tg * cfg.bigvgan.total_upsample / cfg.bigvgan.sampling_rate
    warnings.filterwarnings("ignore", category=RuntimeWarning)

    def noise_rows(idx):        # CFM noise of the utterances `idx` (global indices; all inside this rank's shard)
        return noise[[i - lo_r for i in idx]]

    def step_sequential(last=False):
        if world > 1:
            sh = ShardedSynthesizer(tts, batch_size=16)
            h = sh.begin(text_all.tolist() if rank == 0 else None, cond_dev if rank == 0 else None, shapes, max_mel_tokens=M, noise_fn=noise_rows)
            sh.finish(h)
            return h["local"][0]
        return tts.synthesize_batch(text, cond_dev, max_mel_tokens=M, noise=noise)[0]

    # Software pipeline ACROSS steps (default; indextts_amd/serving.py): `--decode-lanes` decode chains of consecutive batches in
    # flight at once -- latency-bound chains of 125 small launches per token that leave most CUs idle and interleave almost for
    # free, each on its own stream and host thread -- and ONE s2mel + vocoder stage (MFMA-bound) at a time behind them on a further
    # stream.  Every step's whole work stays inside the timed region (flush() joins every batch before the closing barrier), each
    # batch is computed exactly as the sequential loop computes it (checked after the run: `outputs_equal_sequential`), and all
    # collectives stay on the main thread in a fixed order.  --no-overlap runs the steps strictly one after the other.
    # (CU-masked streams, hipExtStreamCreateWithCUMask, are accepted but have no effect for an unprivileged user on this pool.)
    from indextts_amd.serving import BatchPipeline
    lanes = max(1, args.decode_lanes)
    pipe = None if args.no_overlap else BatchPipeline(tts, decode_lanes=lanes, acoustic_workers=max(1, args.acoustic_workers),
                                                      coalesce=max(1, args.coalesce), lane_priority=args.lane_priority,
                                                      acoustic_coalesce=max(1, args.acoustic_coalesce), exclusive=args.exclusive)
    # batches submitted and not yet retired: every lane full plus what one acoustic batch takes
    in_flight = args.in_flight if args.in_flight > 0 else lanes * max(1, args.coalesce) + max(1, args.acoustic_coalesce)
    if pipe is not None and args.trace_jobs:
        pipe.trace = []
    pending = []

    sharder = ShardedSynthesizer(tts, batch_size=16, pipeline=pipe) if (world > 1 and pipe is not None) else None
    per_step = max(1, (hi_r - lo_r + 15) // 16)       # batches of 16 one step puts into this rank's pipeline

    def retire(job):
        if sharder is not None:
            sharder.finish(job)                        # gather to rank 0 + original order (collectives on the main thread)
            return job["local"][0]
        return job.result()[0]

    def step_pipelined(last=False):
        out = retire(pending.pop(0)) if len(pending) * per_step >= in_flight else None
        if sharder is not None:
            pending.append(sharder.begin(text_all.tolist() if rank == 0 else None, cond_dev if rank == 0 else None, shapes,
                                         max_mel_tokens=M, noise_fn=noise_rows))
        else:
            pending.append(pipe.submit(text, cond_dev, max_mel_tokens=M, noise=noise))
        return out

    flushed = []

    def flush():
        out = None
        while pending:
            out = retire(pending.pop(0))
            flushed.append(out)
        return out

    step = step_sequential if args.no_overlap else step_pipelined

    def profiled():
        tts.synthesize_batch(text, cond_dev, max_mel_tokens=M, noise=noise)

    def stage_times():
        tts.synthesize_batch(text, cond_dev, max_mel_tokens=M, noise=noise, sync_timers=True)
        return dict(tts.last_stage_times)

    def cpu_leg():
        from oracle import pipeline as op
        cores = cpu_threads()
        torch.set_num_threads(cores)
        Lc, Mc = 32, args.cpu_codes
        ctext = text[:1, :Lc].clone()
        Tgc = int(Mc * cfg.code_to_frame)
        cnoise = noise[:1, :, : Tp + Tgc].cpu()
        if cnoise.shape[-1] < Tp + Tgc:      # the bounded sample is longer than the run's utterances (--cpu-codes > --codes): its own draw
            cnoise = torch.from_numpy(synth.uniform("bench/noise/cpu-leg", (1, cfg.s2mel.in_channels, Tp + Tgc), 1.7))
        twg = {k: torch.from_numpy(v) for k, v in wg.items()}
        tws = {k: torch.from_numpy(v) for k, v in ws.items()}
        log(f"[bench] cpu baseline: oracle pipeline, 1 utterance, {Lc} text tokens, {Mc} codes, Tp={Tp}, "
            f"{cfg.diffusion_steps} CFM steps, {cores} threads ...")
        with torch.no_grad():
            c0 = time.perf_counter()
            ctimers = {}
            r = op.synthesize_one(twg, tws, wv, cfg, ctext, cond0, cnoise, Mc, kv_round=kv16, timers=ctimers)
            cdt = time.perf_counter() - c0
        caudio = r["wav"].shape[-1] / cfg.bigvgan.sampling_rate
        wavs, mid = tts.synthesize_batch(ctext, cond_dev, max_mel_tokens=Mc, noise=cnoise.to(dev), return_intermediates=True)
        got, ref = mid["codes"][0].cpu().numpy(), r["codes"][0].numpy()
        codes_equal = bool(np.array_equal(got, ref))
        extra = {}
        if not codes_equal:
            # Reduced-precision STORAGE (bf16 KV cache: a one-ulp difference between the two fp32 summation orders can move a key's bf16
            # rounding by 2^-9) can flip a near-tie of the argmax; from there the two sequences are different utterances.  Report where
            # and how close the tie was, and compare everything behind the decode on the ORACLE's codes.
            n = min(len(got), len(ref))
            sdiff = int(np.nonzero(got[:n] != ref[:n])[0][0]) if n and (got[:n] != ref[:n]).any() else n
            from oracle import gpt as og
            with torch.no_grad():
                conds = og.conds_latent(twg, cfg.gpt, cond0.spk_cond_latent, cond0.emo_vec)
                _, lg = og.generate_greedy(twg, cfg.gpt, conds, ctext, sdiff + 1, 10.0, return_logits=True, kv_round=kv16)
                fake = og.prepare_gpt_inputs(twg, cfg.gpt, conds, ctext)[0]
                ids = torch.cat([fake, torch.from_numpy(ref[:sdiff].astype(np.int64))[None]], dim=1)
                sc = og.repetition_penalty(ids, lg[:, sdiff].float(), 10.0)[0]
            margin = float(sc[int(ref[sdiff])] - sc[int(got[sdiff])]) if sdiff < n else None
            extra = {"codes_first_difference_step": sdiff, "oracle_score_margin_at_that_step": margin, "oracle_score_std": float(sc.std()),
                     "stages_behind_the_decode_compared_on": "the oracle's codes (teacher-forced)"}
            st = tts.gpt_stage(ctext, cond_dev, max_mel_tokens=Mc, codes=r["codes"])
            wavs, mid = tts.acoustic_stage(st, noise=cnoise.to(dev), return_intermediates=True)
            log(f"[bench] cpu oracle: greedy codes differ from step {sdiff} of {n} on (oracle's score margin between the two tokens there: {margin:.3e}, "
                f"score std {float(sc.std()):.2f}); comparing the stages behind the decode on the oracle's codes")
        mel_l1 = (mid["mel"][0].cpu() - r["mel"][0]).abs().mean().item()
        wav_err = (wavs[0].cpu() - r["wav"]).abs().max().item() / 32767.0
        log(f"[bench] cpu oracle: {caudio:.2f}s audio in {cdt:.1f}s; gpu-vs-cpu on that utterance: greedy codes equal={codes_equal}, "
            f"mel L1={mel_l1:.2e}, max|wav| diff={wav_err:.2e} (full scale)")
        # What "greedy codes equal" can mean in this mode, measured (oracle/parity.py): the first PB utterances' prefixes decoded by the
        # oracle, the HIP path TEACHER-FORCED on the oracle's codes (logit noise, argmax match rate, the oracle's margin wherever the
        # argmax differs) and free-running (where it leaves the oracle's sequence) -- in the run's own mode and, when that stores the
        # KV cache as bf16, in the exact mode (fp32 cache) as well.
        from oracle import parity
        PB = min(8, len(text))
        lat, emo = cond0.spk_cond_latent.expand(PB, -1, -1).contiguous(), cond0.emo_vec.expand(PB, -1).contiguous()
        par, own = {}, tts.gpt.kv_format
        c1 = time.perf_counter()
        for mode in ([own] + (["f32"] if own == "bf16" else [])):
            tts.gpt.set_kv_format(mode)
            pr = parity.decode_parity(tts.gpt, twg, cfg.gpt, lat, emo, text[:PB, :Lc].clone(), Mc)
            bound = LOGIT_NOISE_BOUND[mode]
            pr["logit_bound"] = bound
            pr["within_bound"] = bool(pr["max_abs_logit_diff"] <= bound and pr["worst_oracle_margin_at_a_mismatch"] <= 2 * bound
                                      and pr["first_difference_step_free_running"] == pr["first_mismatch_step_teacher_forced"])
            par[mode] = pr
            log(f"[bench] decode parity, KV {mode}: {PB} utterances x {pr['steps']} steps teacher-forced on the oracle's codes: max |dlogit| "
                f"{pr['max_abs_logit_diff']:.2e} (bound {bound:.1e}, logit std {pr['logit_std']:.2f}), argmax match rate {pr['codes_match_rate_teacher_forced']:.4f}, "
                f"{len(pr['mismatching_steps'])} differing step(s), worst oracle margin there {pr['worst_oracle_margin_at_a_mismatch']:.2e}; free-running: first "
                f"difference per utterance {pr['first_difference_step_free_running']}; within bound: {pr['within_bound']}")
        tts.gpt.set_kv_format(own)
        extra["decode_parity"] = par
        extra["decode_parity_seconds"] = round(time.perf_counter() - c1, 1)
        extra["codes_match_rate"] = par[own]["codes_match_rate_teacher_forced"]
        extra["codes_match_rate_note"] = (f"{PB} utterances x {par[own]['steps']} steps, HIP decode teacher-forced on the oracle's codes (KV {own}); "
                                          "free-running prefix match rate and the exact mode under decode_parity")
        # BASELINE.md 3's B = 16 leg, from the measured B = 1 stage times: the oracle's greedy decode streams the 1.9 GB of GPT weights once per
        # step whatever the number of rows (its step time at 16 rows is taken as the 1-row time: a LOWER bound on the time, an upper bound on the
        # rate), s2mel and BigVGAN work is per utterance (x 16)
        b16_s = ctimers.get("gpt", 0.0) + 16.0 * (ctimers.get("s2mel", 0.0) + ctimers.get("bigvgan", 0.0))
        extra["stage_seconds"] = {k: round(v, 2) for k, v in ctimers.items()}
        extra["batch16_estimate"] = {"value": round(16.0 * caudio / b16_s, 4) if b16_s > 0 else None, "unit": "audio_s/s",
                                     "how": "16 x this utterance: decode time of ONE row (weight-stream-bound on the CPU: an upper bound on the rate) "
                                            "+ 16 x (s2mel + BigVGAN) of the measured stage times; the full 16 x 512-code batch would take ~20 minutes"}
        return {"value": round(caudio / cdt, 4), "unit": "audio_s/s", "cores": cores, "kind": "port",
                "sample": f"oracle/pipeline.py (fp32 torch CPU): 1 utterance, {Lc} text tokens, {Mc} codes ({caudio:.2f} s audio), "
                          f"Tp={Tp}, {cfg.diffusion_steps} CFM steps, full-size weights (the B = 16 x 512-code batch of the GPU run would "
                          f"take the oracle ~20 minutes: BASELINE.md 3's B = 16 leg does not fit the bounded sample; batch16_estimate derives it "
                          f"from this sample's stage times)",
                **extra,
                "greedy_codes_equal_vs_gpu": codes_equal, "mel_l1_vs_gpu": mel_l1, "wav_max_abs_diff_vs_gpu_fullscale": wav_err}

    desc = {"workload": f"{'configs[4] (long-form, emotion vector, fp8 GPT weights, graph-replayed decode)' if args.longform else 'configs[2]'}: IndexTTS-2 full pipeline (gpt 472M + s2mel 98M + BigVGAN 112M params), batch {B} utterances per "
                        f"GPU, {L} text tokens, {M} codes ({Tg} mel frames, {audio_s / B:.2f} s) each, prompt {Tp} frames, "
                        f"{cfg.diffusion_steps} CFM steps cfg {cfg.cfg_rate}, greedy decode rep-penalty 10"
                        + ("" if not compact else f", GPT linear weights stored as {args.gpt_weights} (rounded once at load; fp32 arithmetic)")
                        + ("" if not kv16 else ", KV cache stored as bf16 (keys / values rounded once when produced; fp32 arithmetic)"),
            "gpt_weights": args.gpt_weights, "gpt_kv_cache": tts.gpt.kv_format,
            "decode_geometry": "narrow (512-thread GEMV workgroups)" if _lib_geometry() else "wide (1024-thread GEMV workgroups)",
            "decode_plane_rows": _lib_plane_rows(),
            "batch_per_gpu": B, "text_tokens": L, "codes": M, "prompt_frames": Tp, "diffusion_steps": cfg.diffusion_steps}
    desc["step_overlap"] = ("none" if args.no_overlap else
                            f"software pipeline across steps: {lanes} decode chain(s) in flight (one stream + host thread each"
                            + (f", a free lane decodes up to {args.coalesce} waiting 16-utterance requests as one batch" if args.coalesce > 1 else "")
                            + f"), {max(1, args.acoustic_workers)} s2mel+vocoder stage(s) at a time behind them"
                            + (f" (a free one takes up to {args.acoustic_coalesce} decoded requests as one batch)" if args.acoustic_coalesce > 1 else "")
                            + ("; decode jobs and acoustic jobs take turns on the chip (exclusive)" if args.exclusive else "")
                            + "; every batch inside the timed region")
    step.flush = (lambda: None) if args.no_overlap else flush
    nref = 16 if world > 1 else len(text)        # multi-rank: a shard goes through the pipeline in batches of 16
    step.reference = lambda: tts.synthesize_batch(text[:nref], cond_dev, max_mel_tokens=M, noise=noise[:nref])[0]
    step.flushed = flushed
    step.pipe = pipe
    step.tts = tts
    return step, profiled, cpu_leg, stage_times, audio_s, desc


def _prompt_audio(tag, sr, seconds):
    from indextts_amd import synth
    n = int(sr * seconds)
    t = np.arange(n) / sr
    return (0.4 * np.sin(2 * np.pi * (180 + 40 * np.sin(2 * np.pi * 1.3 * t)) * t) + 0.1 * np.sin(2 * np.pi * 1900 * t)
            + 0.05 * synth.uniform(tag, (n,), 1.0)).astype(np.float32)


def build_prompt_or_infer(args, world, rank, dev):
    """The two things the prompt block and the reference-default decoding mode (8f2) exist for:
      prompt         15 s speaker prompt + 15 s emotion prompt -> PromptConditioning (w2v-bert 17 layers, semantic codec, mel, CAMPPlus,
                     length regulator, conformer + perceiver + emotion vector; infer_v2.py:618-666, 681-696, 748-754): the cache-miss
                     latency of a request.  value = prompt encodes per second, ms_per_step = the latency.
      infer_default  IndexTTS2.infer() as the reference runs it by default (infer_v2.py:714-722: do_sample, num_beams = 3, top_p .8, top_k 30,
                     temperature .8, repetition_penalty 10) on ONE sentence, B = 1, warm prompt cache (cold = + the prompt workload's latency).
                     value = audio seconds per wall second."""
    from indextts_amd import synth, weights
    from indextts_amd.config import CamPPlusConfig, PipelineConfig, RepCodecConfig, W2VBertConfig
    from indextts_amd.infer_v2 import IndexTTS2, PromptConditioning
    from indextts_amd.prompt import PromptAudio, PromptEncoders
    cfg = PipelineConfig()
    t0 = time.time()
    wg = weights.synth_gpt_weights(cfg.gpt, tag="bench/gpt")
    wg.update(weights.synth_gpt_cond_weights(cfg.gpt, tag="bench/gpt"))
    M = args.codes
    wg["mel_head.bias"][cfg.gpt.stop_mel_token] = -1e4      # fixed length: exactly M codes
    ws = weights.synth_s2mel_weights(cfg.s2mel, tag="bench/s2mel")
    wv = weights.synth_bigvgan_weights(cfg.bigvgan, tag="bench/bigvgan")
    wcfg, ccfg, pcfg = W2VBertConfig(), RepCodecConfig(), CamPPlusConfig()
    wc = weights.synth_repcodec_weights(ccfg, tag="bench/codec")
    for k in ("codebook.weight", "out_project.weight", "out_project.bias"):
        ws[f"semantic_codec.quantizer.quantizers.0.{k}"] = wc[f"quantizer.quantizers.0.{k}"]
    tts = IndexTTS2.from_state_dicts(cfg, wg, ws, wv, device=dev, gpt_weight_format=args.gpt_weights, gpt_kv_format=args.gpt_kv)
    tts.prompt_encoders = PromptEncoders(weights.synth_w2vbert_weights(wcfg, tag="bench/w2v"), wc, weights.synth_campplus_weights(pcfg, tag="bench/campplus"),
                                         tts.s2mel, device=dev, w2vbert_cfg=wcfg, codec_cfg=ccfg, campplus_cfg=pcfg)
    log(f"[bench] synthetic weights + contexts in {time.time() - t0:.1f}s")
    secs = args.prompt_seconds
    spk = PromptAudio(_prompt_audio("bench/p16", 16000, secs), _prompt_audio("bench/p22", 22050, secs))
    emo = PromptAudio(_prompt_audio("bench/e16", 16000, secs))
    warnings.filterwarnings("ignore", category=RuntimeWarning)
    if args.workload == "prompt":
        def step():
            feats = tts.prompt_encoders.encode(spk, emo)
            c = PromptConditioning.from_features(tts.gpt, feats, emo_alpha=0.7)
            return c.spk_cond_latent
        desc = {"workload": f"prompt block (SURVEY 8f1): {secs:.0f} s speaker prompt + {secs:.0f} s emotion prompt -> PromptConditioning, full-size encoders "
                            "(w2v-bert-2.0 17 layers, RepCodec, CAMPPlus, mel, length regulator, conformer 6x512 + perceiver, emotion pair)",
                "prompt_seconds": secs, "unit_note": "value = prompt encodes per second; ms_per_step = cache-miss latency of one request"}
        return step, step, None, None, 1.0, desc
    seg = synth.integers("bench/sentence", (1, args.text_tokens), 2, cfg.gpt.number_text_tokens).tolist()
    G = dict(max_mel_tokens=M)          # everything else: the reference's defaults (beam-sample, 3 beams)
    audio = {}

    def step():
        sr, a = tts.infer(spk, seg, None, emo_audio_prompt=emo, emo_alpha=0.7, **G)
        audio["s"] = a.shape[0] / sr
        return torch.from_numpy(a.astype(np.float32))
    step()      # cold call: fills the prompt caches (its latency = the prompt workload's + one warm call)
    desc = {"workload": f"IndexTTS2.infer() reference defaults (infer_v2.py:714-722: beam-sample, num_beams 3, top_p 0.8, top_k 30, temperature 0.8, "
                        f"rep-penalty 10): ONE sentence of {args.text_tokens} text tokens, B = 1, {M} codes, warm prompt cache",
            "text_tokens": args.text_tokens, "codes": M}
    return step, step, None, None, audio["s"], desc


def launcher_command(argv, n_gpus: int, port: int):
    """The command `python bench.py --gpus N ...` re-launches itself as when it is started as a plain single process:
    one rank per GPU under torch.distributed.run on 127.0.0.1 (the same shape the driver uses)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(argv, n_gpus: int) -> int:
    """Parent of a multi-rank run.  It has not touched the GPU (no HIP call, no torch.cuda.is_available()), starts the ranks as
    child processes, lets their output through (rank 0 prints the one JSON line) and returns their exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = launcher_command(argv, n_gpus, port)
    log("[bench] launching", " ".join(cmd))
    return subprocess.run(cmd, env=env).returncode


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8, help="timed steps (default 8: the pipeline's fill and drain are inside the timed region, "
                                                          "so very short runs under-state the steady rate)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--prompt-seconds", type=float, default=15.0, help="prompt / infer_default workloads: length of the speaker and emotion prompts")
    ap.add_argument("--workload", default="pipeline", choices=["pipeline", "vocoder", "longform", "prompt", "infer_default"],
                    help="pipeline = BASELINE configs[2] (the metric's configuration); vocoder = configs[1]; longform = configs[4]: ONE "
                         "utterance of 1500 codes (30 s), emotion vector mixed in, fp8 GPT weights, graph-replayed decode")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--frames", type=int, default=800, help="vocoder workload: mel frames")
    ap.add_argument("--text-tokens", type=int, default=128)
    ap.add_argument("--codes", type=int, default=0, help="codes per utterance (default 512; longform 1500)")
    ap.add_argument("--prompt-frames", type=int, default=689)
    ap.add_argument("--cpu-codes", type=int, default=96, help="codes of the bounded CPU-baseline utterance")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-exact-mode", action="store_true", help="pipeline workload: skip the second timed run with the fp32 KV cache")
    ap.add_argument("--decode-lanes", type=int, default=3,
                    help="pipeline workload: decode chains of consecutive batches in flight at once (each on its own stream and host thread)")
    ap.add_argument("--acoustic-workers", type=int, default=1, help="pipeline workload: s2mel + vocoder stages of different batches in flight at once")
    ap.add_argument("--lane-priority", default="high", choices=["high", "normal"], help="pipeline workload: HIP stream priority of the decode lanes")
    ap.add_argument("--acoustic-coalesce", type=int, default=1, help="pipeline workload: a free acoustic worker takes up to this many decoded requests as one s2mel + vocoder batch")
    ap.add_argument("--coalesce", type=int, default=1,
                    help="pipeline workload: a free decode lane takes up to this many waiting 16-utterance requests and decodes them as one batch")
    ap.add_argument("--in-flight", type=int, default=0, help="pipeline workload: batches submitted and not yet retired (0 = decode lanes x merged requests + one acoustic batch)")
    ap.add_argument("--exclusive", action="store_true",
                    help="pipeline workload: a decode job and an acoustic job never share the chip (BatchPipeline(exclusive=True)): the "
                         "pipeline only merges and re-orders")
    ap.add_argument("--s2mel-overlap", action="store_true",
                    help="the CFM solver's two CFG halves on two streams (idxtts_s2mel_set_overlap): faster alone, slower beside decode lanes")
    ap.add_argument("--trace-jobs", action="store_true", help="pipeline workload: log the host-side start / end of every decode and acoustic job")
    ap.add_argument("--no-overlap", action="store_true",
                    help="pipeline workload: run the K steps strictly one after the other (default: the decode chains of the next "
                         "--decode-lanes steps overlap the s2mel + vocoder stage of the current one)")
    ap.add_argument("--gpt-weights", default=None, choices=["f32", "bf16", "fp8"],
                    help="storage of the GPT linear weights: bf16 (default for the pipeline workload: what BASELINE configs[2] names), "
                         "fp8-e4m3 with a power-of-two scale per output channel (default for longform = configs[4]), or f32 (the reference's "
                         "own weights, bit for bit); arithmetic stays fp32; the CPU baseline / parity leg runs the same rounded model")
    ap.add_argument("--decode-geometry", default="auto", choices=["auto", "wide", "narrow"],
                    help="decode GEMVs as 1024-thread (wide) or 512-thread (narrow) workgroups (idxtts_set_decode_geometry): auto = narrow for the "
                         "pipelined run, whose decode launches have to find room beside the acoustic stage's kernels, wide with --no-overlap")
    ap.add_argument("--plane-rows", type=int, default=0, help="decode rows from which the plane GEMV runs the decode step (idxtts_set_decode_plane_rows; "
                                                              "0 = the library default 17, or 5 with --coalesce > 1)")
    ap.add_argument("--gpt-kv", default=None, choices=["f32", "bf16"],
                    help="storage of the GPT's KV cache: default bf16 with compact weights (the reference's use_fp16 halves weights and cache "
                         "together), f32 with --gpt-weights f32; keys / values are rounded once when produced, arithmetic stays fp32, and the "
                         "CPU baseline / parity leg rounds the same way")
    ap.add_argument("--gemm", default="bf16x3", choices=["bf16x3", "f32"],
                    help="arithmetic of the GEMM-shaped passes (s2mel, latent pass): split-bf16 (default) or exact fp32 MFMA")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(sys.argv[1:], args.gpus)
    longform = args.workload == "longform"
    if longform:
        args.workload = "pipeline"
        args.batch = args.batch or 1
    side = args.workload in ("prompt", "infer_default")
    if args.workload == "infer_default" and args.text_tokens == 128:
        args.text_tokens = 40                      # one sentence
    args.codes = args.codes or (1500 if longform else (200 if side else 512))
    # configs[2] names bf16 ("IndexTTS-2 full pipeline ... batch=16, 1xMI355X, bf16, greedy decode"), configs[4] fp8: the GPT's linear
    # weights are STORED in that format (rounded once at load), the arithmetic stays fp32, and the parity leg runs the same rounded
    # model on the CPU oracle.  --gpt-weights f32 keeps the reference's own weights bit for bit.
    args.gpt_weights = args.gpt_weights or ("fp8" if longform else ("bf16" if args.workload == "pipeline" else "f32"))
    args.longform = longform

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: the rank count of the launcher wins")
    if args.workload == "pipeline" and not longform and not args.batch and world >= 8:
        args.batch = 32       # BASELINE configs[3]: 256 utterances sharded data-parallel over 8 GPUs
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback for the HIP path)"
    # rehearsal knobs for a one-GPU box (the multi-rank code path end to end, ranks sharing the card over gloo):
    #   IDXTTS_DIST_BACKEND=gloo IDXTTS_FORCE_DEVICE=0 python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2
    backend = os.environ.get("IDXTTS_DIST_BACKEND", "nccl")
    dev_index = int(os.environ.get("IDXTTS_FORCE_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    from indextts_amd import _lib
    _lib.load()
    _lib.set_gemm_mode(0 if args.gemm == "f32" else 1)
    _lib.set_s2mel_overlap(bool(args.s2mel_overlap))
    narrow = args.decode_geometry == "narrow" or (args.decode_geometry == "auto" and args.workload == "pipeline" and not args.no_overlap)
    _lib.set_decode_geometry(narrow)
    if args.plane_rows:
        _lib.set_decode_plane_rows(args.plane_rows)
    elif args.coalesce > 1:
        _lib.set_decode_plane_rows(5)      # merged decodes run on the plane GEMV: the 16-row requests have to run on it alone too (serving.merge_keeps_kernels)
    t0 = time.time()
    build = build_pipeline if args.workload == "pipeline" else (build_prompt_or_infer if side else build_vocoder)
    step, profiled, cpu_leg, stage_times_fn, audio_s_per_step_per_gpu, desc = build(args, world, rank, dev)
    log(f"[bench] rank {rank}: model + inputs ready in {time.time() - t0:.1f}s")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    flush = getattr(step, "flush", lambda: None)
    for i in range(args.warmup):
        step(last=True) if args.workload == "pipeline" else step()
        flush()
        torch.cuda.synchronize()
        log(f"[bench] rank {rank}: warmup {i + 1}/{args.warmup} done")
    barrier()
    t_start = time.perf_counter()
    out, outs = None, []
    for k in range(args.steps):
        o = step(last=(k == args.steps - 1)) if args.workload == "pipeline" else step()
        out = o if o is not None else out
        if o is not None:
            outs.append(o)
    o = flush()                     # the batches still in flight: inside the timed region
    out = o if o is not None else out
    barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert torch.isfinite(out).all()
    log(f"[bench] rank {rank}: {args.steps} steps in {elapsed:.3f}s")
    if getattr(getattr(step, "pipe", None), "trace", None):
        for kind, a, b, rows in sorted(step.pipe.trace, key=lambda r: r[1]):
            if b >= t_start:
                log(f"[bench] job {kind:8s} {rows:3d} rows  start {a - t_start:8.3f} s  end {b - t_start:8.3f} s  ({b - a:.3f} s)")
    equal_seq = None
    if hasattr(step, "reference"):      # every step has the same inputs: each retired batch must equal the sequential call bit for bit
        ref = step.reference()
        outs = outs + list(getattr(step, "flushed", []))[-args.steps:] + [out]
        equal_seq = bool(all(torch.equal(o, ref) for o in outs))
        log(f"[bench] rank {rank}: {len(outs)} pipelined outputs equal the sequential synthesize_batch bit for bit: {equal_seq}")
        assert equal_seq, "a pipelined batch differs from the sequential result"

    # The same pipelined run in the EXACT decode mode (fp32 KV cache: nothing is rounded between the weights and the logits, so the
    # greedy codes are the oracle's up to fp32 summation order) -- reported beside the headline when the headline stores the cache as bf16.
    exact_mode = None
    tts_obj = getattr(step, "tts", None)
    if (args.workload == "pipeline" and not args.no_exact_mode and tts_obj is not None and tts_obj.gpt.kv_format == "bf16" and not args.no_overlap
            and not args.longform):
        k2 = max(2, min(args.steps, 6))
        tts_obj.gpt.set_kv_format("f32")
        step(last=True); flush(); torch.cuda.synchronize()
        barrier()
        t2 = time.perf_counter()
        for k in range(k2):
            step(last=(k == k2 - 1))
        flush()
        barrier()
        e2 = time.perf_counter() - t2
        if world > 1:
            tm2 = torch.tensor([e2], device=dev, dtype=torch.float64)
            dist.all_reduce(tm2, op=dist.ReduceOp.MAX)
            e2 = float(tm2.item())
        tts_obj.gpt.set_kv_format("bf16")
        exact_mode = {"gpt_kv": "f32", "gpt_weights": args.gpt_weights, "value": round(audio_s_per_step_per_gpu * world * k2 / e2, 2), "unit": "audio_s/s",
                      "steps": k2, "ms_per_step": round(1000 * e2 / k2, 3),
                      "note": "the same pipelined run with the KV cache stored fp32 (bench.py --gpt-kv f32): the mode whose greedy codes equal the "
                              "oracle's up to fp32 summation order; its parity figures: cpu_baseline.decode_parity.f32"}
        log(f"[bench] exact mode (fp32 KV cache): {k2} steps in {e2:.3f}s = {exact_mode['value']} audio-s/s")

    roofline = stages = roofline_stages = None
    if rank == 0 and not args.no_roofline:
        # same work again with per-launch HIP events on the launch stream (graph replay is bypassed while profiling)
        empty_ms = _lib.profile_event_overhead(400)
        _lib.profile_enable(True)
        nprof = min(args.steps, 2)
        for _ in range(nprof):
            profiled()
        torch.cuda.synchronize()
        prof = _lib.profile_read()
        _lib.profile_enable(False)
        default_cfg = (not args.longform and args.gpt_weights == "bf16" and args.gemm == "bf16x3" and (args.batch or 16) == 16
                       and args.codes == 512 and args.text_tokens == 128 and args.prompt_frames == 689)
        if stage_times_fn is not None:   # device-synchronised timers behind the reference's four stage names (infer_v2.py:895-901)
            stages = {k: round(v, 4) for k, v in stage_times_fn().items()}
            log(f"[bench] stage seconds (synchronised, one step): {stages}")
        # What bracketing every launch with an event pair adds per launch: the bracketed readings of a step sum to more than the
        # same step takes un-bracketed (device-synchronised stage timers); the excess, spread over the launches, is the method's
        # fixed cost -- never more than what the pair reads around an empty kernel.
        overhead_ms = 0.0
        if stages:
            raw_ms = sum(v["ms"] for v in prof.values()) / nprof
            n_launch = sum(v["launches"] for v in prof.values()) / nprof
            overhead_ms = min(empty_ms, max(0.0, (raw_ms - 1000.0 * sum(stages.values())) / max(1.0, n_launch)))
        log(f"[bench] event-pair reading around an empty launch {1000 * empty_ms:.2f} us; per-launch cost of the bracketing {1000 * overhead_ms:.2f} us")
        roofline = roofline_from_profile(prof, nprof, args.workload, split_bf16=args.gemm != "f32", default_config=default_cfg,
                                         overhead_ms=overhead_ms)
        if stages and args.workload == "pipeline":
            wb = {"f32": 4, "bf16": 2, "fp8": 1}[args.gpt_weights]
            roofline_stages = stage_rooflines(stages, prof, nprof, desc["batch_per_gpu"], desc["text_tokens"], desc["codes"], desc["prompt_frames"],
                                              int(desc["codes"] * 1.72), desc["diffusion_steps"], wb, kv_bytes=2 if desc.get("gpt_kv_cache") == "bf16" else 4)

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and cpu_leg is not None:
        cpu_baseline = cpu_leg()

    if rank == 0:
        if _lib.get_gemm_mode() == 0:
            dtype = "f32"
        elif args.workload == "vocoder":
            dtype = "f32 (split-bf16 convolutions: fp32 operands as hi+lo bf16, 3 bf16 MFMAs per product, fp32 accumulate; activations exact fp32)"
        else:
            dtype = ("f32 (greedy decode + its prefill: exact fp32 MFMA; fp32 accumulate and fp32 activations everywhere) + split-bf16 "
                     "(fp32 operands as hi+lo bf16, 3 bf16 MFMAs per product) GEMMs, DiT attention and convolutions in s2mel, the latent pass and the vocoder")
        if args.workload == "pipeline" and args.gpt_weights != "f32":
            dtype += f"; GPT linear weights STORED as {args.gpt_weights} (rounded once at load, widened to fp32 in registers)"
        if args.workload in ("pipeline", "longform") and desc.get("gpt_kv_cache") == "bf16":
            dtype += ("; KV cache STORED as bf16 (rounded once when a key / value is produced, widened to fp32 in registers; the prefill that "
                      "fills it runs on the split-bf16 GEMMs, the one that fills an fp32 cache on the exact fp32 MFMA)")
        audio_total = audio_s_per_step_per_gpu * world * args.steps
        value = audio_total / elapsed
        cfgd = dict(desc)
        cfgd.update({"parallelism": f"dp{world}",
                     "exchange": ("ShardedSynthesizer: broadcast(token ids + conditioning) from rank 0, sort by length, contiguous shards, "
                                  "gather(waveforms) -> rank 0 in the original order") if world > 1 else "none",
                     "collective_backend": (dist.get_backend() if world > 1 else "none"),
                     "collective_ranks": (dist.get_world_size() if world > 1 else 1),
                     "timed_region": "inputs (token ids, conditioning, CFM noise) already resident in HBM; waveforms stay on the device "
                                     "(rank 0 after the gather): H2D of the inputs and the final wav.cpu() of infer_v2.py:892 are outside "
                                     "(14 MB / step at 16 utterances, < 0.4 ms over PCIe)"})
        if equal_seq is not None:
            cfgd["outputs_equal_sequential"] = equal_seq
        if world > 1 and dist.get_backend() == "nccl":
            cfgd["rccl_ranks"] = dist.get_world_size()
        if world >= 8 and cfgd.get("batch_per_gpu") == 32 and not longform:
            cfgd["workload"] = cfgd["workload"].replace("configs[2]", f"configs[3] ({32 * world} utterances sharded data-parallel, 32 per GPU)")
        res = {
            "metric": ("prompt encodes per second (speaker + emotion prompt -> PromptConditioning)" if args.workload == "prompt" else
                       "synthesised audio seconds per second (IndexTTS-2 infer_v2 hot path), whole job"),
            "value": round(value, 2), "unit": "prompts/s" if args.workload == "prompt" else "audio_s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype,
            "data": "synthetic (seeded random-init weights of the full architecture, synthetic prompt features and token ids)",
            "config": cfgd, "audio_s_per_s_per_gpu": round(value / world, 2), "rtf": round(elapsed / audio_total, 6),
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        if exact_mode:
            if cpu_baseline and "decode_parity" in cpu_baseline and "f32" in cpu_baseline["decode_parity"]:
                pe = cpu_baseline["decode_parity"]["f32"]
                exact_mode.update({"greedy_codes_equal_vs_gpu": pe["free_running_codes_equal"], "codes_match_rate": pe["codes_match_rate_teacher_forced"],
                                   "max_abs_logit_diff": pe["max_abs_logit_diff"], "logit_bound": pe["logit_bound"]})
            res["exact_mode"] = exact_mode
        if roofline_stages:
            res["roofline_stages"] = roofline_stages
        if stages:
            res["stage_seconds"] = stages
            ntok = cfgd["batch_per_gpu"] * cfgd["codes"]
            res["decode"] = {"tokens_per_s": round(ntok / stages["gpt_gen_time"], 1), "ms_per_token_step": round(1000 * stages["gpt_gen_time"] / cfgd["codes"], 4),
                             "note": "gpt_gen_time includes the prefill; one step = one token for every utterance of the batch"}
            if roofline_stages and "gpt_decode" in roofline_stages:
                res["decode"].update({k: roofline_stages["gpt_decode"][k] for k in ("launches_per_token", "us_per_token", "floor_us_per_token", "times_the_floor")})
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
