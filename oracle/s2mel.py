"""ORACLE (test infrastructure, not product): CPU fp32 restatement of the IndexTTS-2 semantic-to-mel stage.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this.

Follows (reference file:line, relative to grantjr1842/index-tts):
  * gpt_layer                     indextts/s2mel/modules/commons.py:413 (three Linears, no activation)
  * vq2emb                        utils/maskgct/.../residual_vq.py:144-152, factorized_vector_quantize.py:123-127
  * InterpolateRegulator.forward  s2mel/modules/length_regulator.py:90-141 (continuous input, no f0, no vq)
  * BASECFM.inference/solve_euler s2mel/modules/flow_matching.py:31-115 (noise z passed explicitly)
  * DiT.forward                   s2mel/modules/diffusion_transformer.py:186-257 (+ TimestepEmbedder 19-60,
                                  FinalLayer 84-101)
  * Transformer / blocks          s2mel/modules/gpt_fast/model.py:121-360 (adaLN-RMSNorm, rotary, SwiGLU, U-ViT skips)
  * WN.forward                    s2mel/modules/wavenet.py:138-166 with SConv1d reflect pad (encodec.py:192-228)
                                  and fused_add_tanh_sigmoid_multiply (commons.py:132-141)
Pinned against the imported reference `MyModel` by tests/golden/make_golden.py::make_s2mel -> s2mel.npz.
Weight-norm layers take the FOLDED weight (w = g*v/||v||).
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F


def _t(w, key) -> torch.Tensor:
    v = w[key]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))


def _lin(w, name, x, bias=True):
    return F.linear(x, _t(w, f"{name}.weight"), _t(w, f"{name}.bias") if bias else None)


# ------------------------------------------------------------------------------------------------
def gpt_layer(w, latent: torch.Tensor) -> torch.Tensor:
    x = latent
    for n in range(3):
        x = _lin(w, f"gpt_layer.{n}", x)
    return x


def vq2emb(w, codes: torch.Tensor) -> torch.Tensor:
    """codes [B,M] -> [B,M,codec_hidden] (the caller's transpose(1,2) included, infer_v2.py:841-842)."""
    cb = _t(w, "semantic_codec.quantizer.quantizers.0.codebook.weight")
    ow = _t(w, "semantic_codec.quantizer.quantizers.0.out_project.weight")[:, :, 0]
    ob = _t(w, "semantic_codec.quantizer.quantizers.0.out_project.bias")
    return cb[codes.long()] @ ow.t() + ob


def sequence_mask(length: torch.Tensor, max_length=None) -> torch.Tensor:
    max_length = int(length.max()) if max_length is None else max_length
    return torch.arange(max_length)[None, :] < length[:, None]


def length_regulator(w, cfg, x: torch.Tensor, ylens: torch.Tensor) -> torch.Tensor:
    """x [B,M,in] -> [B,Tg,C] with Tg = ylens.max() (length_regulator.py:117-141)."""
    lr = "length_regulator"
    x = _lin(w, f"{lr}.content_in_proj", x)
    mask = sequence_mask(ylens).unsqueeze(-1)
    x = F.interpolate(x.transpose(1, 2).contiguous(), size=int(ylens.max()), mode="nearest")
    for n in range(cfg.lr_num_convs):
        x = F.conv1d(x, _t(w, f"{lr}.model.{3*n}.weight"), _t(w, f"{lr}.model.{3*n}.bias"), padding=1)
        x = F.group_norm(x, 1, _t(w, f"{lr}.model.{3*n+1}.weight"), _t(w, f"{lr}.model.{3*n+1}.bias"), 1e-5)
        x = F.mish(x)
    x = F.conv1d(x, _t(w, f"{lr}.model.{3*cfg.lr_num_convs}.weight"), _t(w, f"{lr}.model.{3*cfg.lr_num_convs}.bias"))
    return x.transpose(1, 2).contiguous() * mask


# ------------------------------------------------------------------------------------------------
def timestep_embedding(t: torch.Tensor, dim: int = 256) -> torch.Tensor:
    """TimestepEmbedder.timestep_embedding (diffusion_transformer.py:41-55): scale 1000, max_period 10000."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
    args = 1000 * t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def t_embed(w, name: str, t: torch.Tensor) -> torch.Tensor:
    h = _lin(w, f"{name}.mlp.0", timestep_embedding(t))
    return _lin(w, f"{name}.mlp.2", F.silu(h))


def rope_cache(seq_len: int, n_elem: int, base: float = 10000.0) -> torch.Tensor:
    """precompute_freqs_cis (gpt_fast/model.py:336-345) in the weight dtype (fp32) -> [T, n_elem/2, 2]."""
    freqs = 1.0 / (base ** (torch.arange(0, n_elem, 2)[: (n_elem // 2)].float() / n_elem))
    t = torch.arange(seq_len)
    freqs = torch.outer(t, freqs)
    fc = torch.polar(torch.ones_like(freqs), freqs)
    return torch.stack([fc.real, fc.imag], dim=-1).to(torch.float32)


def apply_rotary(x: torch.Tensor, fc: torch.Tensor) -> torch.Tensor:
    """apply_rotary_emb (model.py:348-360): interleaved pairs.  x [B,T,H,hd], fc [T,hd/2,2]."""
    xs = x.float().reshape(*x.shape[:-1], -1, 2)
    fc = fc.view(1, xs.size(1), 1, xs.size(3), 2)
    out = torch.stack([xs[..., 0] * fc[..., 0] - xs[..., 1] * fc[..., 1],
                       xs[..., 1] * fc[..., 0] + xs[..., 0] * fc[..., 1]], -1)
    return out.flatten(3)


def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    return (x * torch.rsqrt(torch.mean(x * x, dim=-1, keepdim=True) + eps)) * weight


def ada_norm(w, name: str, x: torch.Tensor, c: torch.Tensor, eps: float) -> torch.Tensor:
    """AdaptiveLayerNorm (model.py:20-38): weight, bias = split(project_layer(c)); weight*RMSNorm(x)+bias."""
    D = x.shape[-1]
    mod = _lin(w, f"{name}.project_layer", c)
    weight, bias = mod[..., :D], mod[..., D:]
    return weight * rms_norm(x, _t(w, f"{name}.norm.weight"), eps) + bias


def dit_transformer(w, cfg, x: torch.Tensor, c: torch.Tensor, key_mask: torch.Tensor) -> torch.Tensor:
    """Transformer.forward (model.py:160-191).  x [N,T,D], c [N,1,D], key_mask [N,T] bool."""
    N, T, D = x.shape
    H, hd = cfg.num_heads, cfg.head_dim
    fc = rope_cache(cfg.block_size, hd, cfg.rope_base)[:T]
    attn_mask = key_mask[:, None, None, :].expand(N, 1, T, T)
    half = cfg.depth // 2
    skips = []
    for i in range(cfg.depth):
        p = f"cfm.estimator.transformer.layers.{i}"
        if i > half:
            x = _lin(w, f"{p}.skip_in_linear", torch.cat([x, skips.pop(-1)], dim=-1))
        h = ada_norm(w, f"{p}.attention_norm", x, c, cfg.norm_eps)
        q, k, v = _lin(w, f"{p}.attention.wqkv", h, bias=False).split([D, D, D], dim=-1)
        q = apply_rotary(q.view(N, T, H, hd), fc).transpose(1, 2)
        k = apply_rotary(k.view(N, T, H, hd), fc).transpose(1, 2)
        v = v.view(N, T, H, hd).transpose(1, 2)
        y = F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask, dropout_p=0.0)
        y = y.transpose(1, 2).contiguous().view(N, T, D)
        hres = x + _lin(w, f"{p}.attention.wo", y, bias=False)
        f = ada_norm(w, f"{p}.ffn_norm", hres, c, cfg.norm_eps)
        ff = _lin(w, f"{p}.feed_forward.w2",
                  F.silu(_lin(w, f"{p}.feed_forward.w1", f, bias=False)) * _lin(w, f"{p}.feed_forward.w3", f, bias=False), bias=False)
        x = hres + ff
        if i < half:
            skips.append(x)
    return ada_norm(w, "cfm.estimator.transformer.norm", x, c, cfg.norm_eps)


def wavenet(w, cfg, x: torch.Tensor, x_mask: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """WN.forward (wavenet.py:138-166).  x [N,C,T], x_mask [N,1,T], g [N,C,1]."""
    e = "cfm.estimator.wavenet"
    C = cfg.wn_hidden
    out = torch.zeros_like(x)
    g = F.conv1d(g, _t(w, f"{e}.cond_layer.conv.conv.weight"), _t(w, f"{e}.cond_layer.conv.conv.bias"))
    pad = (cfg.wn_kernel - 1) // 2
    for i in range(cfg.wn_layers):
        d = cfg.wn_dilation_rate ** i
        xp = F.pad(x, (pad * d, pad * d), mode="reflect")      # SConv1d: reflect, symmetric for stride 1
        x_in = F.conv1d(xp, _t(w, f"{e}.in_layers.{i}.conv.conv.weight"), _t(w, f"{e}.in_layers.{i}.conv.conv.bias"), dilation=d)
        a = x_in + g[:, i * 2 * C:(i + 1) * 2 * C, :]
        acts = torch.tanh(a[:, :C]) * torch.sigmoid(a[:, C:])
        rs = F.conv1d(acts, _t(w, f"{e}.res_skip_layers.{i}.conv.conv.weight"), _t(w, f"{e}.res_skip_layers.{i}.conv.conv.bias"))
        if i < cfg.wn_layers - 1:
            x = (x + rs[:, :C]) * x_mask
            out = out + rs[:, C:]
        else:
            out = out + rs
    return out * x_mask


def dit_forward(w, cfg, x, prompt_x, x_lens, t, style, cond) -> torch.Tensor:
    """DiT.forward (diffusion_transformer.py:186-257), eval mode, no class dropout.
    x, prompt_x [N,80,T]; x_lens [N]; t [N]; style [N,192]; cond [N,T,512] -> [N,80,T]."""
    e = "cfm.estimator"
    N, _, T = x.shape
    t1 = t_embed(w, f"{e}.t_embedder", t)
    cond = _lin(w, f"{e}.cond_projection", cond)
    xt, pt = x.transpose(1, 2), prompt_x.transpose(1, 2)
    x_in = torch.cat([xt, pt, cond, style[:, None, :].repeat(1, T, 1)], dim=-1)
    x_in = _lin(w, f"{e}.cond_x_merge_linear", x_in)
    key_mask = sequence_mask(x_lens, T)
    x_res = dit_transformer(w, cfg, x_in, t1.unsqueeze(1), key_mask)
    x_res = _lin(w, f"{e}.skip_linear", torch.cat([x_res, xt], dim=-1))
    h = _lin(w, f"{e}.conv1", x_res).transpose(1, 2)
    t2 = t_embed(w, f"{e}.t_embedder2", t)
    h = wavenet(w, cfg, h, key_mask.unsqueeze(1), t2.unsqueeze(2)).transpose(1, 2) + _lin(w, f"{e}.res_projection", x_res)
    # FinalLayer: LN(no affine, 1e-6) modulated by adaLN_modulation(SiLU(t1)), then linear
    mod = _lin(w, f"{e}.final_layer.adaLN_modulation.1", F.silu(t1))
    shift, scale = mod.chunk(2, dim=1)
    h = F.layer_norm(h, (h.shape[-1],), None, None, 1e-6) * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)
    h = _lin(w, f"{e}.final_layer.linear", h).transpose(1, 2)
    return F.conv1d(h, _t(w, f"{e}.conv2.weight"), _t(w, f"{e}.conv2.bias"))


def cfm_inference(w, cfg, mu, x_lens, prompt, style, z, n_timesteps: int, cfg_rate: float = 0.7) -> torch.Tensor:
    """BASECFM.inference + solve_euler (flow_matching.py:31-115) with the noise `z` [B,80,T] given
    (the reference draws it from the global RNG at line 52; temperature 1.0)."""
    B, T = mu.shape[0], mu.shape[1]
    x = z.clone()
    t_span = torch.linspace(0, 1, n_timesteps + 1)
    Tp = prompt.shape[-1]
    prompt_x = torch.zeros_like(x)
    prompt_x[..., :Tp] = prompt[..., :Tp]
    x[..., :Tp] = 0
    t = t_span[0]
    for step in range(1, len(t_span)):
        dt = t_span[step] - t_span[step - 1]
        if cfg_rate > 0:
            sx = torch.cat([x, x], 0)
            sp = torch.cat([prompt_x, torch.zeros_like(prompt_x)], 0)
            ss = torch.cat([style, torch.zeros_like(style)], 0)
            sm = torch.cat([mu, torch.zeros_like(mu)], 0)
            st = torch.stack([t, t])
            if B > 1:
                st = st.repeat_interleave(B)
                lens = torch.cat([x_lens, x_lens])
            else:
                lens = x_lens          # the reference passes x_lens [1] for both stacked rows (broadcast)
            d = dit_forward(w, cfg, sx, sp, lens if B > 1 else x_lens.expand(2), st, ss, sm)
            dphi, cfg_dphi = d.chunk(2, dim=0)
            dphi = (1.0 + cfg_rate) * dphi - cfg_rate * cfg_dphi
        else:
            dphi = dit_forward(w, cfg, x, prompt_x, x_lens, t.expand(B), style, mu)
        x = x + dt * dphi
        t = t + dt
        x[:, :, :Tp] = 0
    return x
