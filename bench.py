#!/usr/bin/env python3
"""Benchmark of the MI355X-native IndexTTS-2 hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): synthesised audio seconds per wall-second (whole job, all GPUs) and RTF.
A "step" = one pass of the hot path over one batch of synthetic inputs already resident in HBM.
Workloads (BASELINE.json `configs`):
  vocoder   configs[1]: BigVGAN-only mel->wav, batch 8 of 80x800 mels per GPU (74.3 s audio / step / GPU)
One process per GPU; utterance batches are sharded data-parallel (weak scaling: every rank gets its own
batch); the only exchange step is the gather of waveforms to rank 0 over RCCL/xGMI.
Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed) and `cpu_baseline`
(the CPU oracle timed on this box's host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "index-tts_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md "HBM3E peak BW" (spec)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="vocoder", choices=["vocoder"])
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=800)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=1, help="batch rows of the workload the CPU baseline runs (bounded sample)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch with torch.distributed.run for N>1")
        if world == 1 and args.gpus > 1:
            return 2
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback for the HIP path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    from indextts_amd import _lib, weights
    from indextts_amd.config import BigVGANConfig
    from indextts_amd.vocoder import BigVGAN

    _lib.load()
    cfg = BigVGANConfig()
    t0 = time.time()
    w = weights.synth_bigvgan_weights(cfg, tag="bench/bigvgan")   # same random-init weights on every rank
    voc = BigVGAN(w, cfg)
    B, Tm = args.batch, args.frames
    mel = torch.from_numpy(weights.synth_mel(f"bench/mel/rank{rank}", B, cfg.num_mels, Tm)).to(dev)
    n_samples = Tm * cfg.total_upsample
    audio_s_per_step_per_gpu = B * n_samples / cfg.sampling_rate
    gathered = [torch.empty(B, 1, n_samples, device=dev) for _ in range(world)] if (world > 1 and rank == 0) else None
    log(f"[bench] rank {rank}: model + inputs ready in {time.time() - t0:.1f}s")

    def step():
        wav = voc(mel)
        if world > 1:   # the path's one exchange step: waveforms to rank 0 (north_star: "gather of waveforms")
            dist.gather(wav, gathered, dst=0)
        return wav

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        wav = step()
    barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert torch.isfinite(wav).all()

    # ---- roofline leg: same steps again with per-launch HIP events (own stream = torch's current stream)
    roofline = None
    if rank == 0:
        _lib.profile_enable(True)
        for _ in range(args.steps):
            voc(mel)
        torch.cuda.synchronize()
        prof = _lib.profile_read()
        _lib.profile_enable(False)
        tot_ms = sum(v["ms"] for v in prof.values())
        for name, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
            log(f"[bench] kernel {name:22s} launches/step {v['launches'] // args.steps:4d}  "
                f"{v['ms'] / args.steps:8.3f} ms/step ({100 * v['ms'] / tot_ms:5.1f}%)  "
                f"{v['flops'] / v['ms'] / 1e9:8.2f} TFLOP/s  {v['bytes'] / v['ms'] / 1e6:8.1f} GB/s(alg)")
        dom_name, dom = max(prof.items(), key=lambda kv: kv[1]["ms"])
        if dom_name.startswith("conv1d_mfma"):
            achieved = dom["flops"] / dom["ms"] / 1e9
            roofline = {"kernel": dom_name, "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                        "launches_per_step": dom["launches"] // args.steps,
                        "avg_launch_ms": round(dom["ms"] / dom["launches"], 4),
                        "alg_flops_per_launch": dom["flops"] / dom["launches"]}
        else:
            achieved = dom["bytes"] / dom["ms"] / 1e6
            roofline = {"kernel": dom_name, "bound": "hbm", "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS,
                        "unit": "GB/s", "frac": round(achieved / PEAK_HBM_GBS, 4), "traffic": None,
                        "launches_per_step": dom["launches"] // args.steps,
                        "avg_launch_ms": round(dom["ms"] / dom["launches"], 4),
                        "alg_bytes_per_launch": dom["bytes"] / dom["launches"]}
        roofline["kernel_time_share"] = {k: round(v["ms"] / tot_ms, 4) for k, v in prof.items()}

    # ---- CPU baseline: the oracle (a port), on this box's host cores, bounded sample, rank 0 at N=1 only
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import vocoder as ov
        # the box's CPU share, not the host's core count (oversubscribing a cgroup quota stalls for minutes)
        cores = max(1, min(16, len(os.sched_getaffinity(0)), os.cpu_count() or 1))
        torch.set_num_threads(cores)
        rows = max(1, min(B, args.cpu_rows))
        log(f"[bench] cpu baseline: oracle BigVGAN on [{rows},80,{Tm}] with {cores} threads ...")
        cmel = mel[:rows].cpu()
        wt = {k: torch.from_numpy(v) for k, v in w.items()}
        with torch.no_grad():
            ov.bigvgan_forward(wt, cfg, cmel[:, :, :32])   # warm the thread pool
            c0 = time.perf_counter()
            cw = ov.bigvgan_forward(wt, cfg, cmel)
            cdt = time.perf_counter() - c0
        caudio = cw.shape[0] * cw.shape[-1] / cfg.sampling_rate
        # cross-check the GPU result on the same first row while we have it
        gw = voc(mel[:rows].contiguous()).cpu()
        err = (gw - cw).abs().max().item()
        log(f"[bench] cpu oracle: {caudio:.2f}s audio in {cdt:.2f}s on {cores} threads; max|gpu-cpu| on that sample = {err:.2e}")
        cpu_baseline = {"value": round(caudio / cdt, 4), "unit": "audio_s/s", "cores": cores, "kind": "port",
                        "sample": f"oracle/vocoder.py BigVGAN fp32, {rows} x [80 x {Tm}] mels of the same batch ({caudio:.2f} s audio), "
                                  f"torch CPU {cores} threads", "max_abs_diff_vs_gpu": err}

    if rank == 0:
        audio_total = audio_s_per_step_per_gpu * world * args.steps
        value = audio_total / elapsed
        out = {
            "metric": "synthesised audio seconds per second (IndexTTS-2 infer_v2 hot path), whole job",
            "value": round(value, 2), "unit": "audio_s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (seeded random-init weights, log-mel-range inputs)",
            "config": {"workload": "configs[1]: BigVGAN-only mel->wav (bigvgan_v2_22khz_80band_256x, 112M params), "
                                   f"batch {B} x [80 x {Tm}] mels per GPU", "batch_per_gpu": B, "mel_frames": Tm,
                       "parallelism": f"dp{world}", "exchange": "gather(waveforms)->rank0" if world > 1 else "none"},
            "audio_s_per_s_per_gpu": round(value / world, 2), "rtf": round(elapsed / audio_total, 6),
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
