"""Batch pipeline for serving loops: several decode chains in flight, one acoustic stage behind them.

The reference synthesises one request after the other (infer_v2.py:732, serve_tars.py's single worker).  On an MI355X the two
halves of a batch behave differently: the autoregressive decode (`IndexTTS2.gpt_stage`) is a chain of ~125 small dependent
launches per token that leaves most CUs idle at any instant, s2mel + vocoder (`IndexTTS2.acoustic_stage`) are MFMA-bound and
fill the chip.  `BatchPipeline` therefore keeps `decode_lanes` decode chains of consecutive batches running at once -- each on
its own HIP stream, driven by its own host thread (the C call releases the GIL), with its own KV-cache workspace -- and feeds
their results to ONE acoustic worker on a further stream.  Every batch is computed exactly as `synthesize_batch` computes it
(same kernels, same order inside the batch): results are bit-identical, only the interleaving on the device changes.

    pipe = BatchPipeline(tts, decode_lanes=3)
    futs = [pipe.submit(text_k, cond, max_mel_tokens=..., noise=noise_k) for text_k in batches]
    wavs = [f.result() for f in futs]        # lists of [1, n] waveforms, as synthesize_batch returns them
    pipe.close()
"""
from __future__ import annotations

import concurrent.futures
import threading
from typing import Optional

import torch


class BatchPipeline:
    def __init__(self, tts, decode_lanes: int = 3):
        if decode_lanes < 1:
            raise ValueError("decode_lanes must be >= 1")
        self.tts = tts
        self.device = torch.device(tts.device)
        self.decode_lanes = decode_lanes
        lo_pri, hi_pri = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
        self._hi_pri = hi_pri
        self._tls = threading.local()          # one decode stream per lane THREAD: two jobs never share a stream (= a workspace)
        self._lanes = concurrent.futures.ThreadPoolExecutor(max_workers=decode_lanes, thread_name_prefix="idxtts-decode")
        self._acoustic = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="idxtts-acoustic")
        self._ac_stream = torch.cuda.Stream(device=self.device, priority=lo_pri)
        tts.gpt.MAX_WORKSPACES = max(tts.gpt.MAX_WORKSPACES, decode_lanes + 2)

    def _lane_stream(self) -> torch.cuda.Stream:
        s = getattr(self._tls, "stream", None)
        if s is None:
            s = self._tls.stream = torch.cuda.Stream(device=self.device, priority=self._hi_pri)
        return s

    def submit(self, text_tokens: torch.Tensor, cond, max_mel_tokens: int = 1500, noise: Optional[torch.Tensor] = None,
               repetition_penalty: float = 10.0, sampling: Optional[dict] = None) -> concurrent.futures.Future:
        """Queue one batch; returns a Future of the list of waveforms `synthesize_batch` would return.  Inputs produced on the
        caller's current stream are safe to use (an event recorded here is waited for on the lane's stream)."""
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        done: concurrent.futures.Future = concurrent.futures.Future()

        def acoustic_job(st):
            try:
                torch.cuda.set_device(self.device)
                with torch.cuda.stream(self._ac_stream):
                    wavs = self.tts.acoustic_stage(st, noise=noise)
                    self._ac_stream.synchronize()      # the state's tensors may be released once this returns
                done.set_result(wavs)
            except BaseException as e:                  # noqa: BLE001 -- handed to the caller through the future
                done.set_exception(e)

        def lane_job():
            try:
                torch.cuda.set_device(self.device)
                sg = self._lane_stream()
                sg.wait_event(ready)
                with torch.cuda.stream(sg):
                    st = self.tts.gpt_stage(text_tokens, cond, max_mel_tokens=max_mel_tokens, repetition_penalty=repetition_penalty,
                                            sampling=sampling)
                # The lane waits for its own stream on the host (the decode has synchronised already, what is left is the latent
                # pass): a device-side event wait from the acoustic stream is not an option -- HIP refuses to wait on an event
                # whose stream is capturing, and this lane may be capturing the next batch's decode step by then.
                sg.synchronize()
                self._acoustic.submit(acoustic_job, st)
            except BaseException as e:                  # noqa: BLE001
                done.set_exception(e)

        self._lanes.submit(lane_job)
        return done

    def close(self):
        self._lanes.shutdown(wait=True)
        self._acoustic.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
