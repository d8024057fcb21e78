"""ORACLE (test infrastructure, not product): CPU fp32 restatement of the IndexTTS-2 prompt-conditioning encoders.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this.

What it restates (reference file:line, relative to grantjr1842/index-tts):
  * UnifiedVoice.get_conditioning        indextts/gpt/model_v2.py:627-663  (condition_type "conformer_perceiver")
  * UnifiedVoice.get_emo_conditioning    model_v2.py:665-671
  * get_emovec / merge_emovec            model_v2.py:897-910
  * ConformerEncoder / BaseEncoder       indextts/gpt/conformer_encoder.py:284-520 (forward 365-401; layer 219-281;
                                         ConvolutionModule 57-164; PositionwiseFeedForward 20-54)
  * Conv2dSubsampling2                   indextts/gpt/conformer/subsampling.py:131-181
  * RelPositionalEncoding                indextts/gpt/conformer/embedding.py:25-53, 112-141
  * RelPositionMultiHeadedAttention      indextts/gpt/conformer/attention.py:31-117, 164-312 (this fork's forward adds the
                                         position term WITHOUT `rel_shift`: scores = ((q+u)k^T + (q+v)p^T)/sqrt(d_k))
  * PerceiverResampler / Attention / FeedForward / GEGLU / RMSNorm   indextts/gpt/perceiver.py:150-317
PARITY PIN: tests/golden/gpt_ref.npz, produced by the reference's own UnifiedVoice imported from /root/reference
(tests/golden/make_golden.py::make_gpt_ref): tests/test_oracle_gpt_ref.py.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def _t(w, key) -> torch.Tensor:
    v = w[key]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))


def _lin(w, name, x, bias=True):
    y = x @ _t(w, f"{name}.weight").t()
    return y + _t(w, f"{name}.bias") if bias else y


def _ln(w, name, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), _t(w, f"{name}.weight"), _t(w, f"{name}.bias"), eps)


def sinusoid_table(n: int, d: int) -> torch.Tensor:
    """PositionalEncoding.__init__ (embedding.py:45-53): pe[pos, 2i] = sin(pos * exp(-2i ln(1e4)/d)), pe[pos, 2i+1] = cos."""
    pe = torch.zeros(n, d)
    position = torch.arange(0, n).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2) * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def conformer_encoder(w, m, prefix: str, xs: torch.Tensor, xs_lens: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """BaseEncoder.forward (conformer_encoder.py:365-401): xs [B,T,1024], xs_lens [B] -> (out [B,T',D], mask [B,1,T'])."""
    B, T, _ = xs.shape
    D, H = m.output_size, m.attention_heads
    dk = D // H
    valid = (torch.arange(T)[None, :] < xs_lens.reshape(-1, 1))[:, None, :]                       # ~make_pad_mask, [B,1,T]
    # Conv2dSubsampling2 (subsampling.py:165-181): conv2d(1->D, 3x3, stride 2) + ReLU over (time, feature), then
    # [B,D,T',F'] -> [B,T',D*F'] (channel-major) -> Linear; mask[:, :, 2::2]
    x = F.relu(F.conv2d(xs[:, None], _t(w, f"{prefix}.embed.conv.0.weight"), _t(w, f"{prefix}.embed.conv.0.bias"), stride=2))
    b, c, t, f = x.shape
    x = _lin(w, f"{prefix}.embed.out.0", x.transpose(1, 2).contiguous().view(b, t, c * f))
    # RelPositionalEncoding.forward (embedding.py:127-141): x * sqrt(D); pos_emb = pe[:, :T'] (NOT added to x)
    x = x * math.sqrt(D)
    pos_emb = sinusoid_table(t, D)[None]
    mask = valid[:, :, 2::2]
    key_off = ~mask[:, None]                                                                     # [B,1,1,T']
    for i in range(m.num_blocks):
        e = f"{prefix}.encoders.{i}"
        # --- rel-pos self-attention (attention.py:266-312) ---
        h = _ln(w, f"{e}.norm_mha", x)
        q = _lin(w, f"{e}.self_attn.linear_q", h).view(B, t, H, dk)
        k = _lin(w, f"{e}.self_attn.linear_k", h).view(B, t, H, dk).transpose(1, 2)
        v = _lin(w, f"{e}.self_attn.linear_v", h).view(B, t, H, dk).transpose(1, 2)
        p = _lin(w, f"{e}.self_attn.linear_pos", pos_emb, bias=False).view(1, t, H, dk).transpose(1, 2)
        qu = (q + _t(w, f"{e}.self_attn.pos_bias_u")).transpose(1, 2)
        qv = (q + _t(w, f"{e}.self_attn.pos_bias_v")).transpose(1, 2)
        scores = (qu @ k.transpose(-2, -1) + qv @ p.transpose(-2, -1)) / math.sqrt(dk)
        scores = scores.masked_fill(key_off, -float("inf"))
        att = torch.softmax(scores, dim=-1).masked_fill(key_off, 0.0)
        a = (att @ v).transpose(1, 2).contiguous().view(B, t, D)
        x = x + _lin(w, f"{e}.self_attn.linear_out", a)
        # --- convolution module (conformer_encoder.py:114-164): mask, 1x1 -> GLU -> depthwise k -> LN -> SiLU -> 1x1, mask ---
        h = _ln(w, f"{e}.norm_conv", x).transpose(1, 2)
        h = h.masked_fill(~mask, 0.0)
        h = F.conv1d(h, _t(w, f"{e}.conv_module.pointwise_conv1.weight"), _t(w, f"{e}.conv_module.pointwise_conv1.bias"))
        h = F.glu(h, dim=1)
        h = F.conv1d(h, _t(w, f"{e}.conv_module.depthwise_conv.weight"), _t(w, f"{e}.conv_module.depthwise_conv.bias"),
                     padding=(m.cnn_kernel - 1) // 2, groups=D)
        h = F.silu(_ln(w, f"{e}.conv_module.norm", h.transpose(1, 2))).transpose(1, 2)
        h = F.conv1d(h, _t(w, f"{e}.conv_module.pointwise_conv2.weight"), _t(w, f"{e}.conv_module.pointwise_conv2.bias"))
        h = h.masked_fill(~mask, 0.0)
        x = x + h.transpose(1, 2)
        # --- feed forward (ff_scale 1.0: macaron_style is off), then norm_final ---
        h = _ln(w, f"{e}.norm_ff", x)
        x = x + _lin(w, f"{e}.feed_forward.w_2", F.silu(_lin(w, f"{e}.feed_forward.w_1", h)))
        x = _ln(w, f"{e}.norm_final", x)
    return _ln(w, f"{prefix}.after_norm", x), mask


def perceiver_resampler(w, m, prefix: str, x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """PerceiverResampler.forward (perceiver.py:233-245): x [B,T',D_ctx], mask [B, n_latents + T'] bool -> [B, n_latents, dim]."""
    B = x.shape[0]
    H, hd = m.attention_heads, m.perceiver_dim_head
    x = _lin(w, f"{prefix}.proj_context", x)
    lat = _t(w, f"{prefix}.latents")[None].expand(B, -1, -1)
    n = lat.shape[1]
    for l in range(m.perceiver_depth):
        a = f"{prefix}.layers.{l}.0"
        ctx = torch.cat((lat, x), dim=-2)                                                       # cross_attn_include_queries
        q = _lin(w, f"{a}.to_q", lat, bias=False).view(B, n, H, hd).transpose(1, 2)
        kv = _lin(w, f"{a}.to_kv", ctx, bias=False)
        k, v = kv.chunk(2, dim=-1)
        k = k.view(B, -1, H, hd).transpose(1, 2)
        v = v.view(B, -1, H, hd).transpose(1, 2)
        sim = (q @ k.transpose(-2, -1)) * hd ** -0.5
        sim = sim.masked_fill(~mask[:, None, None, :], -torch.finfo(sim.dtype).max)
        o = (sim.softmax(dim=-1) @ v).transpose(1, 2).reshape(B, n, H * hd)
        lat = _lin(w, f"{a}.to_out", o, bias=False) + lat
        f_ = f"{prefix}.layers.{l}.1"
        h, gate = _lin(w, f"{f_}.0", lat).chunk(2, dim=-1)                                      # GEGLU (perceiver.py:174-177)
        lat = _lin(w, f"{f_}.2", F.gelu(gate) * h) + lat
    # RMSNorm (perceiver.py:150-159): F.normalize(x, dim=-1) * sqrt(dim) * gamma
    return F.normalize(lat, dim=-1) * (lat.shape[-1] ** 0.5) * _t(w, f"{prefix}.norm.gamma")


def get_conditioning(w, cfg, speech_conditioning_input: torch.Tensor, cond_lengths: torch.Tensor) -> torch.Tensor:
    """model_v2.py:637-645 on [B,T,1024] features (the method itself takes the transposed tensor and transposes back)."""
    enc, mask = conformer_encoder(w, cfg.cond_module, "conditioning_encoder", speech_conditioning_input, cond_lengths)
    conds_mask = F.pad(mask.squeeze(1), (cfg.cond_latents, 0), value=True)
    return perceiver_resampler(w, cfg.cond_module, "perceiver_encoder", enc, conds_mask)


def get_emo_conditioning(w, cfg, emo_input: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
    """model_v2.py:665-671 -> [B, 1024]."""
    enc, mask = conformer_encoder(w, cfg.emo_cond_module, "emo_conditioning_encoder", emo_input, lengths)
    conds_mask = F.pad(mask.squeeze(1), (1, 0), value=True)
    return perceiver_resampler(w, cfg.emo_cond_module, "emo_perceiver_encoder", enc, conds_mask).squeeze(1)


def get_emovec(w, cfg, emo_input: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
    """model_v2.py:897-902."""
    return _lin(w, "emo_layer", _lin(w, "emovec_layer", get_emo_conditioning(w, cfg, emo_input, lengths)))


def merge_emovec(w, cfg, spk_cond: torch.Tensor, emo_cond: torch.Tensor, cond_lengths: torch.Tensor, emo_cond_lengths: torch.Tensor,
                 alpha: float = 1.0) -> torch.Tensor:
    """model_v2.py:904-910."""
    emo_vec = get_emovec(w, cfg, emo_cond, emo_cond_lengths)
    base_vec = get_emovec(w, cfg, spk_cond, cond_lengths)
    return base_vec + alpha * (emo_vec - base_vec)
