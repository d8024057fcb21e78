"""Batch pipeline for serving loops: several decode chains in flight, the acoustic stages behind them.

The reference synthesises one request after the other (infer_v2.py:732, serve_tars.py's single worker).  On an MI355X the two
halves of a batch behave differently: the autoregressive decode (`IndexTTS2.gpt_stage`) is a chain of ~125 small dependent
launches per token that leaves most CUs idle at any instant, s2mel + vocoder (`IndexTTS2.acoustic_stage`) are MFMA- / HBM-bound
and fill the chip.  `BatchPipeline` keeps `decode_lanes` decode chains of consecutive batches running at once -- each on its own
HIP stream, driven by its own host thread (the C call releases the GIL), with its own KV-cache workspace -- and runs their
acoustic stages on `acoustic_workers` further streams.  Every batch is computed exactly as `synthesize_batch` computes it (same
kernels, same order inside the batch): results are bit-identical, only the interleaving on the device changes.

Two schedules (measured on MI355X at configs[2], profiles/README.md "Round 3"):
  * `exclusive=False` (default): decode chains and acoustic stages share the device.  A decode launch that needs 240-320
    one-round workgroups waits for the 256x256-tile GEMMs of an acoustic stage to give CUs back (their workgroups own a CU's
    whole register file), so the two kinds of work largely take turns anyway: 165-167 audio-s/s.
  * `exclusive=True`: explicit turns.  Up to `decode_lanes` decode chains run together with nothing else on the device (1.18 s
    for three chains of 512 tokens); when they have finished, their acoustic stages run (`acoustic_workers` at a time: stages of
    different batches are at different kernels at any instant, so one stage's HBM-bound GEMM epilogues overlap another's
    MFMA-bound main loops: 3 workers 163 audio-s/s, 1 worker 151); then the next group of decodes.  No better than sharing.

    pipe = BatchPipeline(tts, decode_lanes=3)
    futs = [pipe.submit(text_k, cond, max_mel_tokens=..., noise=noise_k) for text_k in batches]
    wavs = [f.result() for f in futs]        # lists of [1, n] waveforms, as synthesize_batch returns them
    pipe.close()
"""
from __future__ import annotations

import concurrent.futures
import threading
import time
from typing import Optional

import torch


class _Turns:
    """Host-side turn taking between the two kinds of jobs.  A job calls enter(kind) before it touches the device and leave(kind)
    once its stream has drained; `decode` jobs run together up to their limit, then every `acoustic` job they produced, then the
    next group.  Purely a scheduling device: it never changes what a job computes."""

    def __init__(self, limits: dict):
        self.limits = dict(limits)
        self.cv = threading.Condition()
        self.mode = "decode"
        self.running = 0
        self.started = 0                       # jobs started in the current turn
        self.pending = {"decode": 0, "acoustic": 0}

    def announce(self, kind: str) -> None:
        with self.cv:
            self.pending[kind] += 1
            self._maybe_switch()
            self.cv.notify_all()

    def _maybe_switch(self) -> None:
        if self.running:
            return
        other = "acoustic" if self.mode == "decode" else "decode"
        if self.mode == "decode":
            # hand over once this turn's decodes are done and have produced acoustic work; keep decoding while fewer than a
            # full group have started and more decodes are waiting
            if self.pending["acoustic"] and (self.started >= self.limits["decode"] or not self.pending["decode"]):
                self.mode, self.started = other, 0
            elif not self.pending["acoustic"]:
                self.started = 0               # (a group whose decodes all failed leaves nothing to hand over)
        elif not self.pending["acoustic"]:
            self.mode, self.started = other, 0

    def enter(self, kind: str) -> None:
        with self.cv:
            while True:
                self._maybe_switch()
                group_open = kind != "decode" or self.started < self.limits["decode"]
                if self.mode == kind and self.running < self.limits[kind] and group_open:
                    break
                self.cv.wait(timeout=0.5)
            self.running += 1
            self.started += 1
            self.pending[kind] -= 1

    def cancel(self, kind: str) -> None:
        """An announced job that will never enter (it failed first)."""
        with self.cv:
            self.pending[kind] -= 1
            self._maybe_switch()
            self.cv.notify_all()

    def leave(self, kind: str) -> None:
        with self.cv:
            self.running -= 1
            self._maybe_switch()
            self.cv.notify_all()


class BatchPipeline:
    def __init__(self, tts, decode_lanes: int = 3, acoustic_workers: int = 1, exclusive: bool = False):
        if decode_lanes < 1 or acoustic_workers < 1:
            raise ValueError("decode_lanes and acoustic_workers must be >= 1")
        self.tts = tts
        self.device = torch.device(tts.device)
        self.decode_lanes = decode_lanes
        self.acoustic_workers = acoustic_workers
        self.exclusive = exclusive
        lo_pri, hi_pri = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
        self._pri = {"decode": hi_pri, "acoustic": lo_pri}
        self._tls = threading.local()          # one stream per worker THREAD: two jobs never share a stream (= a workspace)
        self._lanes = concurrent.futures.ThreadPoolExecutor(max_workers=decode_lanes, thread_name_prefix="idxtts-decode")
        self._acoustic = concurrent.futures.ThreadPoolExecutor(max_workers=acoustic_workers, thread_name_prefix="idxtts-acoustic")
        self._turns = _Turns({"decode": decode_lanes, "acoustic": acoustic_workers}) if exclusive else None
        self.trace = None                      # set to a list to record (kind, start, end) host times of every job (time.perf_counter)
        tts.gpt.MAX_WORKSPACES = max(tts.gpt.MAX_WORKSPACES, decode_lanes + 2)

    def _stream(self, kind: str) -> torch.cuda.Stream:
        s = getattr(self._tls, "stream", None)
        if s is None:
            s = self._tls.stream = torch.cuda.Stream(device=self.device, priority=self._pri[kind])
        return s

    def submit(self, text_tokens: torch.Tensor, cond, max_mel_tokens: int = 1500, noise: Optional[torch.Tensor] = None,
               repetition_penalty: float = 10.0, sampling: Optional[dict] = None) -> concurrent.futures.Future:
        """Queue one batch; returns a Future of the list of waveforms `synthesize_batch` would return.  Inputs produced on the
        caller's current stream are safe to use (an event recorded here is waited for on the lane's stream), and the waveforms
        are safe to read on that stream (they are tied to it with record_stream before the Future resolves)."""
        caller = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(caller)
        done: concurrent.futures.Future = concurrent.futures.Future()
        turns = self._turns

        def acoustic_job(st):
            entered = False
            try:
                torch.cuda.set_device(self.device)
                if turns:
                    turns.enter("acoustic")
                entered = True
                try:
                    t0 = time.perf_counter()
                    sa = self._stream("acoustic")
                    with torch.cuda.stream(sa):
                        wavs = self.tts.acoustic_stage(st, noise=noise)
                        sa.synchronize()               # the state's tensors may be released once this returns
                    if self.trace is not None:
                        self.trace.append(("acoustic", t0, time.perf_counter()))
                    for w in wavs:                      # allocated on the worker's stream, consumed on the caller's
                        w.record_stream(caller)
                finally:
                    if turns:
                        turns.leave("acoustic")
                done.set_result(wavs)
            except BaseException as e:                  # noqa: BLE001 -- handed to the caller through the future
                if turns and not entered:
                    turns.cancel("acoustic")
                done.set_exception(e)

        def lane_job():
            entered = False
            try:
                torch.cuda.set_device(self.device)
                if turns:
                    turns.enter("decode")
                entered = True
                handed_over = False
                try:
                    t0 = time.perf_counter()
                    sg = self._stream("decode")
                    sg.wait_event(ready)
                    with torch.cuda.stream(sg):
                        st = self.tts.gpt_stage(text_tokens, cond, max_mel_tokens=max_mel_tokens, repetition_penalty=repetition_penalty,
                                                sampling=sampling)
                    # The lane waits for its own stream on the host (the decode has synchronised already, what is left is the
                    # latent pass): a device-side event wait from the acoustic stream is not an option -- HIP refuses to wait on an
                    # event whose stream is capturing, and this lane may be capturing the next batch's decode step by then.
                    sg.synchronize()
                    if self.trace is not None:
                        self.trace.append(("decode", t0, time.perf_counter()))
                    if turns:
                        turns.announce("acoustic")
                    handed_over = True
                finally:
                    if turns:
                        turns.leave("decode")
                if handed_over:
                    self._acoustic.submit(acoustic_job, st)
            except BaseException as e:                  # noqa: BLE001
                if turns and not entered:
                    turns.cancel("decode")
                done.set_exception(e)

        if turns:
            turns.announce("decode")
        self._lanes.submit(lane_job)
        return done

    def close(self):
        self._lanes.shutdown(wait=True)
        self._acoustic.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
