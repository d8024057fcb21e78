"""Batch pipeline for serving loops: several decode chains in flight, the acoustic stages behind them.

The reference synthesises one request after the other (infer_v2.py:732, serve_tars.py's single worker).  On an MI355X the two
halves of a batch behave differently: the autoregressive decode (`IndexTTS2.gpt_stage`) is a chain of ~125 small dependent
launches per token that leaves most CUs idle at any instant, s2mel + vocoder (`IndexTTS2.acoustic_stage`) are MFMA- / HBM-bound
and fill the chip.  `BatchPipeline` keeps `decode_lanes` decode chains running at once -- each on its own HIP stream, driven by its
own host thread (the C call releases the GIL), with its own KV-cache workspace -- and runs the acoustic stages on
`acoustic_workers` further streams.

`acoustic_coalesce` > 1 lets a free acoustic worker take that many decoded requests (same prompt, explicit CFM noise) as ONE s2mel +
vocoder batch; `coalesce` > 1 is dynamic batching of the decode: a lane that becomes free takes up to `coalesce` waiting requests of the same
prompt and text width and decodes them as ONE batch (a decode step streams the 965 MB of GPT weights once whatever the number of rows, so two
16-utterance requests decoded together cost little more than one), then hands every request's rows to its own acoustic job.

Every request is computed exactly as `synthesize_batch` computes it: the kernels treat the rows of a batch independently (same
arithmetic and summation order per row whatever the batch size: tests/test_serving_gpu.py, and bench.py compares every retired
batch with the sequential call bit for bit), so only the interleaving on the device changes.  Two launches pick their KERNEL by
row count, though, and a merge must not move a request across either threshold (`merge_keeps_kernels`, enforced in `_take`):
the GEMM-shaped passes run split-bf16 from 256 rows up (a request of a few dozen prefill rows would change kernels when merged --
what made `test_batch_pipeline_equals_sequential[2-3-2]` differ from the sequential call at the 1e-4 level in round 3, on toy
requests of 38-76 rows; any real utterance is hundreds of rows on its own), and the decode step runs on the plane GEMV from
`idxtts_set_decode_plane_rows()` rows on (default 17) -- so 16-utterance requests merge only where the plane GEMV also decodes
them alone (`_lib.set_decode_plane_rows(5)`, once per process), or where the merged batch stays below the threshold.

Measured at configs[2] (profiles/README.md "Round 3"): decode chains and acoustic stages SHARING the device give the same
throughput as strict turns (3 decodes together, then their 3 acoustic stages: 163 vs 165 audio-s/s) -- a decode launch that needs
240-320 one-round workgroups waits for the 256x256-tile GEMMs of an acoustic stage to give CUs back anyway (their workgroups own
a CU's whole register file) -- so there is one schedule: sharing.

    _lib.set_decode_geometry(True)           # once per process, before generating: decode GEMVs as 512-thread workgroups, which find room
                                             # beside the acoustic stage's kernels (+1.5 % here, 7 % slower for a decode alone)
    pipe = BatchPipeline(tts, decode_lanes=3)
    futs = [pipe.submit(text_k, cond, max_mel_tokens=..., noise=noise_k) for text_k in batches]
    wavs = [f.result() for f in futs]        # lists of [1, n] waveforms, as synthesize_batch returns them
    pipe.close()
"""
from __future__ import annotations

import collections
import concurrent.futures
import contextlib
import threading
import time
from typing import Optional

import torch


def merge_keeps_kernels(rows_each, prefix_len: int, split_bf16_gemm: bool, compact_weights: bool, plane_rows: int) -> bool:
    """May requests of `rows_each` utterances (same text width; `prefix_len` = conditioning + text rows of one utterance's prompt) be
    decoded as ONE batch and still equal their own sequential calls bit for bit?  Rows are independent inside every kernel; what a merge
    can change is WHICH kernel runs:
      * GEMM-shaped passes (prefill, latent pass): exact fp32 below 256 rows, split-bf16 from 256 rows on (csrc/gemm.hip dispatch) --
        every request must be on the split-bf16 side by itself (its prefill alone has >= 256 rows), unless the exact mode is selected;
      * the decode step with compact weight streams: fp32-MFMA GEMV below `plane_rows` rows, plane GEMV from there on
        (idxtts_set_decode_plane_rows) -- all requests and the merged batch must fall on the same side."""
    rows_each = [int(r) for r in rows_each]
    if len(rows_each) <= 1:
        return True
    if split_bf16_gemm and any(r * prefix_len < 256 for r in rows_each):
        return False
    if compact_weights:
        total = sum(rows_each)
        if not (total < plane_rows or min(rows_each) >= plane_rows):
            return False
    return True


class _Request:
    __slots__ = ("text", "cond", "max_mel_tokens", "noise", "repetition_penalty", "sampling", "ready", "caller", "done")

    def compatible(self, other: "_Request") -> bool:
        # same prompt and settings, greedy (sampling draws are per call) and the SAME text width: a wider neighbour would left-pad this
        # request's prompts further, which moves the key-tile boundaries of the prefill attention (a different fp32 summation order)
        return (self.cond is other.cond and self.max_mel_tokens == other.max_mel_tokens and self.repetition_penalty == other.repetition_penalty
                and self.sampling is None and other.sampling is None and int(self.text.shape[1]) == int(other.text.shape[1]))


class BatchPipeline:
    def __init__(self, tts, decode_lanes: int = 3, acoustic_workers: int = 1, coalesce: int = 1, lane_priority: str = "high",
                 acoustic_coalesce: int = 1, exclusive: bool = False):
        if decode_lanes < 1 or acoustic_workers < 1 or coalesce < 1 or acoustic_coalesce < 1:
            raise ValueError("decode_lanes, acoustic_workers, coalesce and acoustic_coalesce must be >= 1")
        self.tts = tts
        self.device = torch.device(tts.device)
        self.decode_lanes = decode_lanes
        self.acoustic_workers = acoustic_workers
        self.coalesce = coalesce
        self.acoustic_coalesce = acoustic_coalesce
        self._aq = collections.deque()         # decoded requests waiting for an acoustic worker: (request, state)
        lo_pri, hi_pri = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
        self._pri = {"decode": hi_pri if lane_priority == "high" else lo_pri, "acoustic": lo_pri}
        self._tls = threading.local()          # one stream per worker THREAD: two jobs never share a stream (= a workspace)
        self._lanes = concurrent.futures.ThreadPoolExecutor(max_workers=decode_lanes, thread_name_prefix="idxtts-decode")
        self._acoustic = concurrent.futures.ThreadPoolExecutor(max_workers=acoustic_workers, thread_name_prefix="idxtts-acoustic")
        self._queue = collections.deque()
        self._qlock = threading.Lock()
        self._streams = []                     # every worker thread's stream (handed back to the library in close())
        self._running = 0                      # requests taken by a lane and not yet retired
        self._groups_taken = 0                 # decode groups formed since the pipeline was last idle
        # exclusive: a decode job and an acoustic job never share the chip (each takes this lock for the whole of its GPU work) -- the
        # pipeline then only re-orders and merges the work; measured against the overlapped schedule in profiles/README.md "Round 4"
        self._turn = threading.Lock() if exclusive else contextlib.nullcontext()
        self.trace = None                      # set to a list to record (kind, start, end, rows) host times of every job (time.perf_counter)
        tts.gpt.MAX_WORKSPACES = max(tts.gpt.MAX_WORKSPACES, decode_lanes + 2)

    def _stream(self, kind: str) -> torch.cuda.Stream:
        s = getattr(self._tls, "stream", None)
        if s is None:
            s = self._tls.stream = torch.cuda.Stream(device=self.device, priority=self._pri[kind])
            with self._qlock:
                self._streams.append(s)
        return s

    def submit(self, text_tokens: torch.Tensor, cond, max_mel_tokens: int = 1500, noise: Optional[torch.Tensor] = None,
               repetition_penalty: float = 10.0, sampling: Optional[dict] = None) -> concurrent.futures.Future:
        """Queue one batch; returns a Future of the list of waveforms `synthesize_batch` would return.  Inputs produced on the
        caller's current stream are safe to use (an event recorded here is waited for on the lane's stream), and the waveforms
        are safe to read on that stream (they are tied to it with record_stream before the Future resolves)."""
        r = _Request()
        r.text, r.cond, r.max_mel_tokens, r.noise = text_tokens, cond, max_mel_tokens, noise
        r.repetition_penalty, r.sampling = repetition_penalty, sampling
        r.caller = torch.cuda.current_stream(self.device)
        r.ready = torch.cuda.Event()
        r.ready.record(r.caller)
        r.done = concurrent.futures.Future()
        with self._qlock:
            self._queue.append(r)
        self._lanes.submit(self._lane_job)      # one drain per request: a drain that finds the queue empty (its request was merged) returns
        return r.done

    def _take(self):
        with self._qlock:
            if not self._queue:
                return []
            # Slow start: a pipeline that was idle takes its first requests one by one (the first acoustic stage can begin after ONE
            # 16-row decode, 0.6 s, instead of after a merged 48-row one, 1.2-2.3 s with other lanes beside it), then two, then
            # `coalesce` at a time -- the merged decodes' better aggregate rate matters once the acoustic stage is the bottleneck.
            if self._running == 0:
                self._groups_taken = 0
            limit = min(self.coalesce, 1 + self._groups_taken // self.decode_lanes)
            group = [self._queue.popleft()]
            while len(group) < limit and self._queue and group[0].compatible(self._queue[0]) and self._keeps_kernels(group + [self._queue[0]]):
                group.append(self._queue.popleft())
            self._groups_taken += 1
            self._running += len(group)
            return group

    def _keeps_kernels(self, group) -> bool:
        from . import _lib
        g = self.tts.cfg.gpt
        prefix = g.cond_latents + 2 + int(group[0].text.shape[1]) + 2 + 1      # [cond | start, text, stop] + start_mel (gpt.py::prepare_gpt_inputs)
        return merge_keeps_kernels([r.text.shape[0] for r in group], prefix, _lib.get_gemm_mode() == _lib.GEMM_BF16X3,
                                   self.tts.gpt.weight_format != "f32", _lib.get_decode_plane_rows())

    def _lane_job(self):
        group = self._take()
        if not group:
            return
        handed = 0
        try:
            torch.cuda.set_device(self.device)
            t0 = time.perf_counter()
            sg = self._stream("decode")
            for r in group:
                sg.wait_event(r.ready)
            text = group[0].text if len(group) == 1 else torch.cat([torch.as_tensor(r.text).cpu() for r in group])      # equal widths
            subs = []
            with self._turn, torch.cuda.stream(sg):
                st = self.tts.gpt_stage(text, group[0].cond, max_mel_tokens=group[0].max_mel_tokens,
                                        repetition_penalty=group[0].repetition_penalty, sampling=group[0].sampling)
                # every request's rows as its own state, cut on the LANE's stream: the slicing copies below are launches like any
                # other, and the acoustic worker reads their results on a stream of its own -- they must be inside what the
                # host-side wait below covers (and allocated from this stream's pool)
                a = 0
                for r in group:
                    b = a + int(r.text.shape[0])
                    n = max(st["code_lens"][a:b])
                    subs.append({"cond": st["cond"], "B": b - a, "codes": st["codes"][a:b, :n].contiguous(), "code_lens": st["code_lens"][a:b],
                                 "code_lens_t": st["code_lens_t"][a:b].clone(), "latent": st["latent"][a:b, :n].contiguous(),
                                 "times": dict(st["times"])})
                    a = b
                # The lane waits for its own stream on the host (the decode has synchronised already, what is left is the latent
                # pass and the slices): a device-side event wait from the acoustic stream is not an option -- HIP refuses to wait on an
                # event whose stream is capturing, and this lane may be capturing the next batch's decode step by then.
                sg.synchronize()
            if self.trace is not None:
                self.trace.append(("decode", t0, time.perf_counter(), int(text.shape[0])))
            for r, sub in zip(group, subs):      # every request's rows go to its own acoustic job
                with self._qlock:
                    self._aq.append((r, sub))
                handed += 1
                self._acoustic.submit(self._acoustic_drain)      # one drain per request: a drain that finds nothing (merged away) returns
        except BaseException as e:                  # noqa: BLE001 -- handed to the callers through their futures
            with self._qlock:
                self._running -= len(group) - handed      # (requests already handed to an acoustic job are retired there)
            for r in group:
                if not r.done.done():
                    r.done.set_exception(e)

    def _take_acoustic(self):
        """Up to `acoustic_coalesce` decoded requests of one prompt, each with its own CFM noise (a merged draw would consume the
        generator differently from the separate calls)."""
        with self._qlock:
            if not self._aq:
                return []
            group = [self._aq.popleft()]
            while (len(group) < self.acoustic_coalesce and self._aq and group[0][0].noise is not None and self._aq[0][0].noise is not None
                   and self._aq[0][0].cond is group[0][0].cond):
                group.append(self._aq.popleft())
            return group

    @staticmethod
    def _merge_states(group):
        """Several decoded batches as ONE acoustic batch: rows are independent in s2mel and the vocoder (ragged lengths: every row is
        padded at its own end), so stacking them only changes how the launches fill the chip.  (Measured neutral at configs[2]:
        181.8 vs 182.3 audio-s/s, profiles/README.md "Round 3"; useful where single requests are small.)"""
        sts = [st for _, st in group]
        n = max(st["codes"].shape[1] for st in sts)
        codes = torch.cat([torch.nn.functional.pad(st["codes"], (0, n - st["codes"].shape[1])) for st in sts])
        latent = torch.cat([torch.nn.functional.pad(st["latent"], (0, 0, 0, n - st["latent"].shape[1])) for st in sts])
        T = max(r.noise.shape[-1] for r, _ in group)
        noise = torch.cat([torch.nn.functional.pad(r.noise, (0, T - r.noise.shape[-1])) for r, _ in group])
        st = {"cond": sts[0]["cond"], "B": sum(st["B"] for st in sts), "codes": codes, "code_lens": [c for st in sts for c in st["code_lens"]],
              "code_lens_t": torch.cat([st["code_lens_t"] for st in sts]), "latent": latent, "times": dict(sts[0]["times"])}
        return st, noise

    def _acoustic_drain(self):
        group = self._take_acoustic()
        if not group:
            return
        try:
            torch.cuda.set_device(self.device)
            t0 = time.perf_counter()
            sa = self._stream("acoustic")
            with self._turn, torch.cuda.stream(sa):
                if len(group) == 1:
                    st, noise = group[0][1], group[0][0].noise
                else:
                    st, noise = self._merge_states(group)
                wavs = self.tts.acoustic_stage(st, noise=noise)
                sa.synchronize()                 # the states' tensors may be released once this returns
            if self.trace is not None:
                self.trace.append(("acoustic", t0, time.perf_counter(), st["B"]))
            a = 0
            for r, sub in group:
                mine = wavs[a:a + sub["B"]]
                a += sub["B"]
                for w in mine:                    # allocated on the worker's stream, consumed on the caller's
                    w.record_stream(r.caller)
                r.done.set_result(mine)
        except BaseException as e:                  # noqa: BLE001
            for r, _ in group:
                if not r.done.done():
                    r.done.set_exception(e)
        finally:
            with self._qlock:
                self._running -= len(group)

    def close(self):
        self._lanes.shutdown(wait=True)
        self._acoustic.shutdown(wait=True)
        from . import _lib
        with torch.cuda.device(self.device):
            for s in self._streams:          # the library's per-stream scratch goes with the worker threads' streams
                s.synchronize()
                _lib.release_stream(s)
        self._streams = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
