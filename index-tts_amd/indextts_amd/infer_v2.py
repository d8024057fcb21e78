"""Host-side mirror of `indextts.infer_v2.IndexTTS2` for the hot path (reference infer_v2.py:69-937).

Scope: the three hot stages -- GPT decode + latent pass, s2mel, BigVGAN --, the GPT's own prompt conditioning
(conformer + perceiver + emotion vector, model_v2.py:627-671, 897-910; hoisted out of the segment loop), the audio-side prompt
encoders (w2v-bert, RepCodec, CAMPPlus, mel; infer_v2.py:618-696 -> indextts_amd/prompt.py), the text front-end (tokenizer.py,
segmenter.py) and the segment loop that sequences them (infer_v2.py:732-881), with the reference's return contract (905-937) and
generation defaults (714-722: beam-sample, num_beams=3).  `IndexTTS2(cfg_path, model_dir)` assembles all of it from a
checkpoint directory as the reference's constructor does; `infer()` takes a WAV path + a string, or any of the intermediate
forms (`PromptAudio`, `PromptFeatures`, `PromptConditioning`; token-id segments).
"""
from __future__ import annotations

import os
import time
import warnings
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import synth
from .config import PipelineConfig
from .gpt import UnifiedVoice
from .s2mel import S2Mel
from .vocoder import BigVGAN


@dataclass
class InferenceResult:                       # infer_v2.py:58-66
    sampling_rate: int
    audio: object
    duration_sec: float
    saved_path: Optional[str]
    rtf: Optional[float]


@dataclass
class PromptFeatures:
    """What the reference's audio-side prompt encoders leave behind per prompt (infer_v2.py:637-658, 691-694): the inputs of the
    GPT's conditioning encoders plus the s2mel conditioning."""
    spk_cond_emb: torch.Tensor        # [1, T50, 1024]  w2v-bert layer-17 features of the speaker prompt     infer_v2.py:637
    style: torch.Tensor               # [1, 192]        CAMPPlus                                              infer_v2.py:647
    prompt_condition: torch.Tensor    # [1, Tp, 512]    length_regulator(S_ref)                               infer_v2.py:649
    ref_mel: torch.Tensor             # [1, 80, Tp]     mel_fn(prompt audio)                                  infer_v2.py:640
    emo_cond_emb: Optional[torch.Tensor] = None       # [1, T50', 1024] features of the emotion prompt (default: the speaker's, 583-584)

    @staticmethod
    def synthetic(cfg: PipelineConfig, prompt_frames: int = 689, feat_frames: int = 400, tag: str = "prompt", emo_frames: int = 0) -> "PromptFeatures":
        """Seeded stand-in for an 8 s prompt (spk_cond_emb [1,400,1024], Tp = 689)."""
        t = lambda n, shape, s, o=0.0: torch.from_numpy(synth.uniform(f"{tag}/{n}", shape, s, o))
        return PromptFeatures(t("spk_cond_emb", (1, feat_frames, cfg.gpt.cond_module.input_size), 1.0), t("style", (1, cfg.s2mel.style_dim), 1.0),
                              t("prompt_condition", (1, prompt_frames, cfg.s2mel.content_dim), 1.0),
                              t("ref_mel", (1, cfg.s2mel.in_channels, prompt_frames), 2.6, -4.0),
                              t("emo_cond_emb", (1, emo_frames, cfg.gpt.cond_module.input_size), 1.0) if emo_frames else None)


@dataclass
class PromptConditioning:
    """What the reference caches per prompt (infer_v2.py:654-658, 693-694) after its encoders have run."""
    spk_cond_latent: torch.Tensor     # [1, 32, d]   get_conditioning(spk_cond_emb)           model_v2.py:819
    emo_vec: torch.Tensor             # [1, d]       merge_emovec(...)                        model_v2.py:904-910
    style: torch.Tensor               # [1, 192]     CAMPPlus                                 infer_v2.py:647
    prompt_condition: torch.Tensor    # [1, Tp, 512] length_regulator(S_ref)                  infer_v2.py:649
    ref_mel: torch.Tensor             # [1, 80, Tp]  mel_fn(prompt audio)                     infer_v2.py:640

    FIELDS = ("spk_cond_latent", "emo_vec", "style", "prompt_condition", "ref_mel")

    @staticmethod
    def synthetic(cfg: PipelineConfig, prompt_frames: int = 689, tag: str = "prompt") -> "PromptConditioning":
        """Seeded stand-in for an 8 s prompt (Tp = 689)."""
        d = cfg.gpt.model_dim
        t = lambda n, shape, s, o=0.0: torch.from_numpy(synth.uniform(f"{tag}/{n}", shape, s, o))
        return PromptConditioning(
            t("latent", (1, cfg.gpt.cond_latents, d), 0.5), t("emo", (1, d), 0.3), t("style", (1, cfg.s2mel.style_dim), 1.0),
            t("prompt_condition", (1, prompt_frames, cfg.s2mel.content_dim), 1.0),
            t("ref_mel", (1, cfg.s2mel.in_channels, prompt_frames), 2.6, -4.0))

    @staticmethod
    def from_features(gpt: UnifiedVoice, feats: "PromptFeatures", emo_alpha: float = 1.0, emo_mix=None) -> "PromptConditioning":
        """The per-prompt half of the reference's segment loop, once per prompt instead of once per segment:
        `merge_emovec(spk_cond_emb, emo_cond_emb, ..., alpha=emo_alpha)` (infer_v2.py:748-754) and the `get_conditioning` call
        inside `inference_speech` (model_v2.py:819), on the HIP conditioning encoders.  The "lengths" are the reference's:
        `shape[-1]` of the [1,T,1024] tensors (infer_v2.py:751-752), i.e. no padding."""
        dev = gpt.device
        spk = feats.spk_cond_emb.to(dev, torch.float32)
        emo = spk if feats.emo_cond_emb is None else feats.emo_cond_emb.to(dev, torch.float32)
        ln_s, ln_e = torch.tensor([spk.shape[-1]]), torch.tensor([emo.shape[-1]])
        emovec = gpt.merge_emovec(spk, emo, ln_s, ln_e, alpha=emo_alpha)
        if emo_mix is not None:                     # (emovec_mat [1,d], weight_vector): infer_v2.py:756-757
            emovec_mat, weight_vector = emo_mix
            emovec = emovec_mat.to(dev, torch.float32) + (1 - torch.sum(weight_vector.to(dev, torch.float32))) * emovec
        latent = gpt.get_conditioning(spk.transpose(1, 2), ln_s)
        return PromptConditioning(latent, emovec, feats.style, feats.prompt_condition, feats.ref_mel)

    def to(self, device) -> "PromptConditioning":
        return PromptConditioning(*[getattr(self, f).to(device, torch.float32).contiguous() for f in self.FIELDS])

    # one flat buffer for the RCCL broadcast ("one packed buffer")
    def pack(self) -> torch.Tensor:
        return torch.cat([getattr(self, f).reshape(-1) for f in self.FIELDS])

    def shapes(self):
        return [tuple(getattr(self, f).shape) for f in self.FIELDS]

    @staticmethod
    def unpack(flat: torch.Tensor, shapes) -> "PromptConditioning":
        out, off = [], 0
        for shp in shapes:
            n = int(np.prod(shp))
            out.append(flat[off:off + n].reshape(shp))
            off += n
        return PromptConditioning(*out)


class IndexTTS2:
    """Drop-in for the hot path of `indextts.infer_v2.IndexTTS2`.

    Construct from a checkpoint directory with the reference's constructor signature (infer_v2.py:69-72): everything that
    constructor loads (infer_v2.py:138-289) is assembled -- gpt / s2mel / BigVGAN, the semantic model + its statistics, the
    semantic codec, CAMPPlus, the emotion banks, the tokenizer -- so `infer("voice.wav", "text", "out.wav")` works as it does there
    (files are looked up where the reference looks, hub cache included; nothing is downloaded).  Or from state dicts
    (reference key layout) with `from_state_dicts` (+ `attach_prompt_models`).
    """

    def __init__(self, cfg_path="checkpoints/config.yaml", model_dir="checkpoints", use_fp16=False, device=None,
                 use_cuda_kernel=None, use_deepspeed=False, use_accel=False, use_torch_compile=False, gpt_weight_format=None,
                 segment_batch: int = 16, gpt_kv_format=None):
        # use_fp16 (reference: gpt.half() + fp16 autocast, infer_v2.py:109, 145-146) maps to bf16 STORAGE of the GPT weights:
        # the arithmetic of the HIP path stays fp32.  gpt_weight_format ("f32" | "bf16" | "fp8") overrides it.
        if gpt_weight_format is None:
            gpt_weight_format = "bf16" if use_fp16 else "f32"
        need = [os.path.join(model_dir, f) for f in ("gpt.pth", "s2mel.pth")]
        missing = [p for p in need if not os.path.exists(p)]
        if missing:
            raise FileNotFoundError(f"IndexTTS-2 checkpoints not found ({missing}); use IndexTTS2.from_state_dicts(...)")
        from .checkpoint import config_from_yaml, load_reference_checkpoints
        cfg, raw = config_from_yaml(cfg_path, model_dir) if os.path.exists(cfg_path) else (PipelineConfig(), None)
        gpt_sd, s2mel_sd, voc_sd = load_reference_checkpoints(model_dir, raw)
        self._init(cfg, gpt_sd, s2mel_sd, voc_sd, device, gpt_weight_format, segment_batch=segment_batch, gpt_kv_format=gpt_kv_format)
        # the rest of the reference's constructor (infer_v2.py:187-289): semantic model + statistics, semantic codec, CAMPPlus,
        # emotion banks, tokenizer
        from .checkpoint import load_prompt_checkpoints, reference_text_normalizer
        self.model_dir = model_dir
        self.attach_prompt_models(load_prompt_checkpoints(model_dir, raw), normalizer=reference_text_normalizer())

    def attach_prompt_models(self, ck: dict, normalizer=None) -> None:
        """ck: checkpoint.load_prompt_checkpoints()'s dict (state dicts in the reference modules' own key layouts, statistics,
        emotion banks, bpe path) -> self.prompt_encoders, the emotion banks, self.tokenizer."""
        from .prompt import PromptEncoders
        from .tokenizer import TextTokenizer
        self.prompt_encoders = PromptEncoders(ck["w2vbert"], ck["codec"], ck["campplus"], self.s2mel, device=self.device,
                                              w2vbert_cfg=ck["w2vbert_cfg"], codec_cfg=ck["codec_cfg"], campplus_cfg=ck["campplus_cfg"],
                                              semantic_mean=ck["semantic_mean"], semantic_std=ck["semantic_std"], mel_kwargs=ck.get("mel_kwargs"))
        self.set_emotion_matrices(ck["emo_matrix"], ck["spk_matrix"], ck["emo_num"])
        self.normalizer = normalizer
        self.tokenizer = TextTokenizer(ck["bpe_path"], normalizer)
        self.bpe_path = ck["bpe_path"]

    @classmethod
    def from_state_dicts(cls, cfg: PipelineConfig, gpt_sd, s2mel_sd, bigvgan_sd, device=None, gpt_weight_format="f32",
                         keep_effective_gpt=False, segment_batch: int = 16, gpt_kv_format=None) -> "IndexTTS2":
        """gpt_weight_format: "f32" | "bf16" | "fp8" storage of the GPT linear weights, gpt_kv_format: "f32" | "bf16" storage of its KV
        cache (default: fp32 with fp32 weights, bf16 with compact weights) (UnifiedVoice); keep_effective_gpt keeps
        `self.gpt.effective_state_dict` (the rounded model, reference keys) for parity checks; segment_batch: segments of one
        infer() call synthesised together (1 = the reference's loop as written)."""
        self = cls.__new__(cls)
        self._init(cfg, gpt_sd, s2mel_sd, bigvgan_sd, device, gpt_weight_format, keep_effective_gpt, segment_batch, gpt_kv_format)
        return self

    def _init(self, cfg, gpt_sd, s2mel_sd, bigvgan_sd, device, gpt_weight_format="f32", keep_effective_gpt=False, segment_batch=16,
              gpt_kv_format=None):
        if device is None:
            device = f"cuda:{torch.cuda.current_device()}" if torch.cuda.is_available() else "cpu"
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("IndexTTS2 (HIP path) needs an MI355X device; there is no CPU fallback")
        self.cfg = cfg
        self.gpt = UnifiedVoice(gpt_sd, cfg.gpt, device=self.device, weight_format=gpt_weight_format, keep_effective=keep_effective_gpt,
                                kv_format=gpt_kv_format)
        self.s2mel = S2Mel(s2mel_sd, cfg.s2mel, device=self.device)
        self.bigvgan = BigVGAN(bigvgan_sd, cfg.bigvgan)
        self.stop_mel_token = cfg.gpt.stop_mel_token
        self.model_version = 2.0
        self.gr_progress = None
        self._diffusion_steps = int(os.environ.get("TARS_DIFFUSION_STEPS", cfg.diffusion_steps))   # infer_v2.py:125
        self._cfg_rate = float(os.environ.get("TARS_CFG_RATE", cfg.cfg_rate))                      # infer_v2.py:126
        self.last_stage_times = {}
        self.segment_batch = int(segment_batch)      # segments of one infer() call synthesised together

    # ------------------------------------------------------------------------------------------
    def synthesize_batch(self, text_tokens: torch.Tensor, cond: PromptConditioning, max_mel_tokens: int = 1500,
                         repetition_penalty: float = 10.0, noise: Optional[torch.Tensor] = None, sync_timers: bool = False,
                         return_intermediates: bool = False, sampling: Optional[dict] = None, per_row_noise: bool = False):
        """One batch of single-segment utterances sharing a prompt: the body of the reference's segment loop
        (infer_v2.py:732-881) for B rows at once.  text_tokens [B, L] (right-padded with stop_text_token).
        Returns a list of B waveforms, float32 [1, n_b] in int16 range (infer_v2.py:866).
        sampling: None = greedy; else the num_beams=1 sampling kwargs of UnifiedVoice.inference_speech
        (do_sample, temperature, top_k, top_p, sampler, exp_noise / generator).
        per_row_noise: draw the CFM noise row by row, `randn([1, 80, Tp + Tg_b])` in row order -- the draws the reference's
        sequential segment loop makes (flow_matching.py:62 once per segment) -- instead of one [B, 80, T] draw.
        = acoustic_stage(gpt_stage(...)): the two halves are separate entry points so that a serving loop can overlap the
        (latency-bound) decode of one batch with the (MFMA-bound) s2mel + vocoder of the previous one on another stream."""
        st = self.gpt_stage(text_tokens, cond, max_mel_tokens=max_mel_tokens, repetition_penalty=repetition_penalty,
                            sampling=sampling, sync_timers=sync_timers)
        return self.acoustic_stage(st, noise=noise, per_row_noise=per_row_noise, sync_timers=sync_timers,
                                   return_intermediates=return_intermediates)

    def _tick(self, sync: bool) -> float:
        if sync:
            torch.cuda.synchronize(self.device)
        return time.perf_counter()

    def gpt_stage(self, text_tokens: torch.Tensor, cond: PromptConditioning, max_mel_tokens: int = 1500,
                  repetition_penalty: float = 10.0, sampling: Optional[dict] = None, sync_timers: bool = False,
                  codes: Optional[torch.Tensor] = None) -> dict:
        """Decode + stop-token trim + latent pass (infer_v2.py:732-828) on the current stream; returns the state
        acoustic_stage consumes (device tensors + host lengths).  `codes` [B, n] (optional): take these codes instead of decoding --
        the rest of the flow on a given code sequence (parity checks feed the reference's codes through the stages behind the decode)."""
        dev = self.device
        c = cond.to(dev)
        B = text_tokens.shape[0]
        times = {}
        t0 = self._tick(sync_timers)
        lat = c.spk_cond_latent.expand(B, -1, -1) if c.spk_cond_latent.shape[0] == 1 else c.spk_cond_latent
        emo = c.emo_vec.expand(B, -1) if c.emo_vec.shape[0] == 1 else c.emo_vec
        gen = dict(sampling) if sampling else {"do_sample": False}
        gen.setdefault("num_beams", 1)
        if codes is None:
            codes, _ = self.gpt.inference_speech(lat, text_tokens, emo_vec=emo, max_generate_length=max_mel_tokens,
                                                 repetition_penalty=repetition_penalty, **gen)
        else:
            codes = torch.as_tensor(codes).to(dev, torch.long)
        t1 = self._tick(sync_timers)
        times["gpt_gen_time"] = t1 - t0
        # trim at the first stop token (infer_v2.py:795-807)
        hc = codes.cpu().numpy()
        code_lens = []
        for row in hc:
            hits = np.nonzero(row == self.stop_mel_token)[0]
            code_lens.append(int(hits[0]) if len(hits) else len(row))
        if not all(cl > 0 for cl in code_lens):
            raise RuntimeError("a row produced the stop token first: nothing to synthesise")
        if any(cl == len(row) for cl, row in zip(code_lens, hc)) and hc.shape[1] >= max_mel_tokens:
            warnings.warn(f"WARN: generation stopped due to exceeding `max_mel_tokens` ({max_mel_tokens}).", RuntimeWarning)
        max_len = max(code_lens)
        codes = codes[:, :max_len].contiguous()
        code_lens_t = torch.tensor(code_lens, dtype=torch.long)
        # per-row text length (rows are right-padded with stop_text_token); the latent pass left-pads + masks shorter rows
        tt = torch.as_tensor(text_tokens).cpu()
        text_lens = ((tt != self.cfg.gpt.stop_text_token) & (tt != self.cfg.gpt.start_text_token)).sum(1)
        latent = self.gpt.forward(lat, text_tokens, text_lens, codes, code_lens_t, emo_vec=emo)
        t2 = self._tick(sync_timers)
        times["gpt_forward_time"] = t2 - t1
        return {"cond": c, "B": B, "codes": codes, "code_lens": code_lens, "code_lens_t": code_lens_t, "latent": latent, "times": times}

    def acoustic_stage(self, st: dict, noise: Optional[torch.Tensor] = None, per_row_noise: bool = False, sync_timers: bool = False,
                       return_intermediates: bool = False):
        """s2mel (length regulator + CFM) and the vocoder (infer_v2.py:835-866) on the current stream."""
        dev = self.device
        c, B, codes, code_lens, code_lens_t, latent, times = (st["cond"], st["B"], st["codes"], st["code_lens"], st["code_lens_t"],
                                                              st["latent"], dict(st["times"]))
        t2 = self._tick(sync_timers)
        condv, target_lens = self.s2mel.prepare_condition(latent, codes, code_lens_t)
        Tp = c.prompt_condition.shape[1]
        Tg = condv.shape[1]
        cat_condition = torch.cat([c.prompt_condition.expand(B, -1, -1), condv], dim=1)     # infer_v2.py:850
        x_lens = target_lens.cpu() + Tp
        if noise is None and per_row_noise and B > 1:
            noise = torch.zeros([B, self.cfg.s2mel.in_channels, Tp + Tg], device=dev)
            for b in range(B):
                nb = int(x_lens[b])
                noise[b, :, :nb] = torch.randn([1, self.cfg.s2mel.in_channels, nb], device=dev)[0]
        elif noise is None:
            noise = torch.randn([B, self.cfg.s2mel.in_channels, Tp + Tg], device=dev)
        mel = self.s2mel.cfm_inference(cat_condition, x_lens, c.ref_mel.expand(B, -1, -1), c.style.expand(B, -1), None,
                                       self._diffusion_steps, inference_cfg_rate=self._cfg_rate, z=noise)
        vc_target = mel[:, :, Tp:]                                                          # infer_v2.py:856
        t3 = self._tick(sync_timers)
        times["s2mel_time"] = t3 - t2
        # vocoder: one ragged batch; every layer pads at each row's OWN end (zeros for the convolutions, replicate for the
        # anti-alias filters), exactly what the reference's B=1 call per utterance sees (infer_v2.py:860)
        wavs: List[Optional[torch.Tensor]] = [None] * B
        tl = target_lens.cpu().tolist()
        Tmax = max(tl)
        w = self.bigvgan(vc_target[:, :, :Tmax].float().contiguous(), lengths=tl if len(set(tl)) > 1 else None)
        w = torch.clamp(32767 * w, -32767.0, 32767.0)                                       # infer_v2.py:866
        up = self.cfg.bigvgan.total_upsample
        for b in range(B):
            wavs[b] = w[b, :, : tl[b] * up].contiguous()
        t4 = self._tick(sync_timers)
        times["bigvgan_time"] = t4 - t3
        self.last_stage_times = times
        if return_intermediates:
            return wavs, {"codes": codes, "code_lens": code_lens, "latent": latent, "cond": condv, "mel": vc_target,
                          "target_lens": tl}
        return wavs

    # ------------------------------------------------------------------------------------------
    def interval_silence(self, sampling_rate=22050, interval_silence=200):                 # infer_v2.py:484-497
        n = int(sampling_rate * interval_silence / 1000.0)
        return torch.zeros(1, n, device=self.device)

    def infer(self, spk_audio_prompt, text, output_path, emo_audio_prompt=None, emo_alpha=1.0, emo_vector=None,
              use_emo_text=False, emo_text=None, use_random=False, interval_silence=200, verbose=False,
              max_text_tokens_per_segment=120, stream_return=False, more_segment_before=0, return_audio=False,
              return_numpy=False, **generation_kwargs):
        """Reference signature and return contract (infer_v2.py:541-567): the generator itself when `stream_return`, else its
        first (only) item -- the output path, an InferenceResult, or (sampling_rate, int16 [n, 1]) -- or None for empty input.
        `spk_audio_prompt` (and `emo_audio_prompt`): a WAV file path as in the reference, or a PromptAudio / PromptFeatures /
        PromptConditioning; `text`: a string (needs self.tokenizer), a list of token-id segments (List[List[int]]) or one segment
        (List[int] / 1-D tensor)."""
        gen = self.infer_generator(spk_audio_prompt, text, output_path, emo_audio_prompt, emo_alpha, emo_vector, use_emo_text,
                                   emo_text, use_random, interval_silence, verbose, max_text_tokens_per_segment, stream_return,
                                   more_segment_before, return_audio=return_audio, return_numpy=return_numpy, **generation_kwargs)
        if stream_return:
            return gen
        try:
            return list(gen)[0]
        except IndexError:
            return None

    def set_emotion_matrices(self, emo_matrix: torch.Tensor, spk_matrix: torch.Tensor, emo_num):
        """feat2.pt / feat1.pt and cfg.emo_num (infer_v2.py:281-289): the per-emotion banks the emo_vector mode mixes from."""
        emo_num = list(emo_num)
        self.emo_num = emo_num
        self.emo_matrix = torch.split(torch.as_tensor(emo_matrix).to(self.device, torch.float32), emo_num)
        self.spk_matrix = torch.split(torch.as_tensor(spk_matrix).to(self.device, torch.float32), emo_num)

    def emotion_vector_mix(self, style: torch.Tensor, emo_vector, use_random: bool = False):
        """infer_v2.py:668-679: per emotion pick the bank entry whose speaker vector is closest (cosine) to this prompt's style -- or a
        random one --, weight by emo_vector.  Returns (emovec_mat [1,d], weight_vector)."""
        if not hasattr(self, "emo_matrix"):
            raise RuntimeError("emo_vector needs the emotion banks: call set_emotion_matrices(emo_matrix, spk_matrix, emo_num) first")
        if len(emo_vector) != len(self.emo_num):
            raise ValueError(f"emo_vector needs {len(self.emo_num)} weights")
        import random
        weight_vector = torch.tensor(emo_vector, device=self.device, dtype=torch.float32)
        style = style.to(self.device, torch.float32)
        if use_random:
            index = [random.randint(0, x - 1) for x in self.emo_num]
        else:
            index = [int(torch.argmax(torch.nn.functional.cosine_similarity(style, tmp, dim=1))) for tmp in self.spk_matrix]     # find_most_similar_cosine
        emo_matrix = torch.cat([tmp[i].unsqueeze(0) for i, tmp in zip(index, self.emo_matrix)], 0)
        return torch.sum(weight_vector.unsqueeze(1) * emo_matrix, 0).unsqueeze(0), weight_vector

    def infer_generator(self, spk_audio_prompt, text, output_path, emo_audio_prompt=None, emo_alpha=1.0, emo_vector=None,
                        use_emo_text=False, emo_text=None, use_random=False, interval_silence=200, verbose=False,
                        max_text_tokens_per_segment=120, stream_return=False, quick_streaming_tokens=0, return_audio=False,
                        return_numpy=False, **generation_kwargs):
        """infer_v2.py:569-937.  Streaming (`stream_return`): after every segment yields its waveform (`[1, n]` float32 on the
        CPU, already scaled and clamped to +-32767) and then the inter-segment silence, and nothing else (874-879, 885-886)."""
        if stream_return and return_audio:
            raise ValueError("stream_return and return_audio are mutually exclusive")                # infer_v2.py:575-576
        if isinstance(text, str):
            # text front-end (infer_v2.py:697-704): tokenize, split into segments, ids.  The tokenizer (indextts_amd/tokenizer.py) needs
            # the checkpoint's bpe.model and, for real text, the reference's normaliser (WeTextProcessing): both absent offline.
            tok = getattr(self, "tokenizer", None)
            if tok is None:
                raise RuntimeError("a str text needs the tokenizer: tts.tokenizer = TextTokenizer('checkpoints/bpe.model', normalizer); "
                                   "or pass token-id segments")
            tokens = tok.tokenize(text)
            text = [tok.convert_tokens_to_ids(seg) for seg in tok.split_segments(tokens, max_text_tokens_per_segment,
                                                                                 quick_streaming_tokens=quick_streaming_tokens)]
        if use_emo_text:
            raise NotImplementedError("emo_text routing (the Qwen emotion classifier, infer_v2.py:590-598) happens upstream of this path: "
                                      "pass its result as emo_vector")
        if emo_vector is not None:
            emo_audio_prompt = None                                                                  # infer_v2.py:586-589
            scale = max(0.0, min(1.0, emo_alpha))                                                    # 600-608
            if scale != 1.0:
                emo_vector = [int(x * scale * 10000) / 10000 for x in emo_vector]
            emo_alpha = 1.0                                                                          # 610-615: the speaker prompt serves
        from .prompt import PromptAudio
        import os as _os
        if isinstance(spk_audio_prompt, (str, _os.PathLike)):
            # file path, as the reference takes it (infer_v2.py:628-630, 685): read, cut to 15 s, resample on the host
            # (indextts_amd/audioio.py); loaded once per path like cache_spk_audio_prompt / cache_emo_audio_prompt (618, 681)
            from .audioio import load_prompt_audio
            # ONE entry per kind, like the reference's cache_spk_audio_prompt / cache_emo_audio_prompt (infer_v2.py:304-310, 618, 681):
            # another path replaces it, and so does the same path once the file has changed (size / modification time)
            files = getattr(self, "_prompt_files", None)
            if files is None:
                files = self._prompt_files = {}

            def cached(kind, path, **kw):
                path = _os.fspath(path)
                st = _os.stat(path)
                key = (path, st.st_mtime_ns, st.st_size)
                hit = files.get(kind)
                if hit is None or hit[0] != key:
                    hit = files[kind] = (key, load_prompt_audio(path, **kw))
                return hit[1]
            spk_audio_prompt = cached("spk", spk_audio_prompt)
            if isinstance(emo_audio_prompt, (str, _os.PathLike)):
                emo_audio_prompt = cached("emo", emo_audio_prompt, emotion=True)
        if isinstance(spk_audio_prompt, PromptAudio):
            # audio path: w2v-bert / semantic codec / CAMPPlus / mel / length regulator on the GPU (indextts_amd/prompt.py), cached per
            # prompt object like the reference's cache_spk_cond / cache_emo_cond (infer_v2.py:618, 681)
            enc = getattr(self, "prompt_encoders", None)
            if enc is None:
                raise RuntimeError("attach the audio-side encoders first: tts.prompt_encoders = PromptEncoders(w2vbert_sd, codec_sd, campplus_sd, tts.s2mel)")
            if emo_audio_prompt is not None and not isinstance(emo_audio_prompt, PromptAudio):
                raise NotImplementedError("with an audio speaker prompt the emotion prompt must be a PromptAudio too")
            akey = (id(spk_audio_prompt), id(emo_audio_prompt))
            if getattr(self, "_audio_cache_key", None) != akey:
                self._audio_cache = enc.encode(spk_audio_prompt, emo_audio_prompt)
                self._audio_cache_key = akey
                self._audio_cache_refs = (spk_audio_prompt, emo_audio_prompt)
            spk_audio_prompt, emo_audio_prompt = self._audio_cache, None
        if isinstance(spk_audio_prompt, PromptFeatures):
            # prompt-feature path: conformer + perceiver + emotion vector on the GPU, cached per (prompt, emotion prompt, alpha) like
            # the reference caches its prompt features (infer_v2.py:618, 681)
            feats = spk_audio_prompt
            if emo_audio_prompt is not None:
                if not isinstance(emo_audio_prompt, (PromptFeatures, torch.Tensor)):
                    raise NotImplementedError("emo_audio_prompt must be a PromptFeatures or the emotion prompt's [1,T,1024] features")
                emo_emb = emo_audio_prompt.spk_cond_emb if isinstance(emo_audio_prompt, PromptFeatures) else emo_audio_prompt
                feats = PromptFeatures(feats.spk_cond_emb, feats.style, feats.prompt_condition, feats.ref_mel, emo_emb)
            key = (id(spk_audio_prompt), id(emo_audio_prompt), float(emo_alpha), None if emo_vector is None else (tuple(emo_vector), bool(use_random)))
            if getattr(self, "_cond_cache_key", None) != key or (emo_vector is not None and use_random):
                mix = None if emo_vector is None else self.emotion_vector_mix(feats.style, emo_vector, use_random=use_random)
                self._cond_cache = PromptConditioning.from_features(self.gpt, feats, emo_alpha=emo_alpha, emo_mix=mix)
                self._cond_cache_key = key
                self._cond_cache_refs = (spk_audio_prompt, emo_audio_prompt)     # keep the ids alive
            spk_audio_prompt = self._cond_cache
        elif emo_audio_prompt is not None or emo_vector is not None:
            raise NotImplementedError("with a ready PromptConditioning the emotion prompt / vector is already folded into emo_vec")
        if not isinstance(spk_audio_prompt, PromptConditioning):
            raise NotImplementedError("pass a wav file path or a PromptAudio (both need tts.prompt_encoders), PromptFeatures or a PromptConditioning: "
                                      "reading and resampling audio files is left to the caller (librosa / torchaudio are not in this image)")
        segs = text if (len(text) and isinstance(text[0], (list, tuple, np.ndarray, torch.Tensor))) else [text]
        segs = [torch.as_tensor(s, dtype=torch.long).reshape(1, -1) for s in segs]
        if not segs or any(s.numel() == 0 for s in segs):
            return                                                                                   # nothing yielded -> infer() returns None
        # generation kwargs and their defaults: infer_v2.py:714-722.  The reference pops `do_sample` and then hard-codes
        # do_sample=True in the call (line 767): its default mode is beam-sample with 3 beams.  Here the popped value is honoured
        # (do_sample=False, num_beams=1 is the greedy parity mode of BASELINE configs[2]); the defaults are the reference's.
        do_sample = generation_kwargs.pop("do_sample", True)
        top_p = generation_kwargs.pop("top_p", 0.8)
        top_k = generation_kwargs.pop("top_k", 30)
        temperature = generation_kwargs.pop("temperature", 0.8)
        length_penalty = generation_kwargs.pop("length_penalty", 0.0)
        num_beams = generation_kwargs.pop("num_beams", 3)
        repetition_penalty = generation_kwargs.pop("repetition_penalty", 10.0)
        max_mel_tokens = generation_kwargs.pop("max_mel_tokens", 1500)
        sampling = {"do_sample": bool(do_sample), "num_beams": int(num_beams), "generator": generation_kwargs.pop("generator", None)}
        if do_sample:
            sampling.update(top_p=top_p, top_k=top_k, temperature=temperature)
        if num_beams > 1:
            sampling["length_penalty"] = length_penalty
        elif do_sample:
            sampling["sampler"] = generation_kwargs.pop("sampler", "hf")
        if generation_kwargs:
            raise TypeError(f"unsupported generation kwargs: {sorted(generation_kwargs)}")
        start = time.perf_counter()
        wavs = []
        sil = self.interval_silence(interval_silence=interval_silence)
        if stream_return:
            for s in segs:                                                                           # segment loop, infer_v2.py:732
                w = self.synthesize_batch(s, spk_audio_prompt, max_mel_tokens=max_mel_tokens, repetition_penalty=repetition_penalty,
                                          sampling=sampling)[0]
                yield w.cpu()                                                                        # infer_v2.py:874-879
                yield sil.cpu()
            return                                                                                   # infer_v2.py:885-886
        # Not streaming: the segments of one text are independent utterances of the same prompt, so they go through the
        # batched path `segment_batch` at a time (one decode loop, one CFM solve, one ragged vocoder pass) instead of one by
        # one; greedy codes are identical to the one-by-one flow, and the CFM noise is drawn segment by segment in order, so
        # under a seed the waveforms agree with it to fp32 rounding.  segment_batch = 1 is the reference's loop as written.
        nb = max(1, int(self.segment_batch))
        stop_text = self.cfg.gpt.stop_text_token
        for i in range(0, len(segs), nb):
            grp = segs[i:i + nb]
            L = max(s.shape[1] for s in grp)
            toks = torch.full((len(grp), L), stop_text, dtype=torch.long)
            for r, s in enumerate(grp):
                toks[r, : s.shape[1]] = s[0]
            wavs.extend(self.synthesize_batch(toks, spk_audio_prompt, max_mel_tokens=max_mel_tokens, repetition_penalty=repetition_penalty,
                                              sampling=sampling, per_row_noise=True))
        out = []
        for i, w in enumerate(wavs):                                                                 # insert_interval_silence 499-522
            out.append(w)
            if i + 1 < len(wavs):
                out.append(sil)
        wav = torch.cat(out, dim=1).cpu()
        total = time.perf_counter() - start
        mono = wav.squeeze(0).contiguous()
        sr = 22050
        wav_length = mono.shape[-1] / sr
        rtf = (total / wav_length) if wav_length else None
        saved = None
        if output_path and not return_audio:
            from .wavio import write_wav_int16
            if os.path.dirname(output_path):
                os.makedirs(os.path.dirname(output_path), exist_ok=True)
            write_wav_int16(output_path, mono.to(torch.int16).numpy(), sr)
            yield output_path                                                                        # infer_v2.py:917
            return
        if return_audio:
            audio = mono.clone()
            yield InferenceResult(sr, audio.numpy().copy() if return_numpy else audio, float(wav_length), saved, rtf)
            return
        yield (sr, mono.unsqueeze(0).to(torch.int16).numpy().T)                                      # infer_v2.py:935-937
