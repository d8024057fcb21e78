"""Synthetic state dicts in the reference's key layout (SURVEY.md §8a "Weight/state-dict layouts").

Every function returns `dict[str, np.ndarray(float32)]` whose keys and shapes are exactly those of
the reference module's `state_dict()` on the inference path, so the same dict can be
 - loaded into the reference classes (`tests/golden/make_golden.py`, in the build container only),
 - fed to the CPU oracle (`oracle/`), and
 - handed tensor-by-tensor to the C-ABI (`idxtts_ctx_load_tensor`), which is also how a real
   checkpoint (`bigvgan_generator.pt['generator']`, `gpt.pth['model']`, `s2mel.pth['net']`) would go in.
Weight-norm parametrised layers are stored folded (w = g*v/||v||), i.e. after
`remove_weight_norm()` (infer_v2.py:263) for BigVGAN and after export-time folding for s2mel.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from . import synth
from .config import BigVGANConfig, CamPPlusConfig, GPTConfig, RepCodecConfig, S2MelConfig, W2VBertConfig

Weights = Dict[str, np.ndarray]


# --------------------------------------------------------------------------------------
# BigVGAN (bigvgan.py:266-386)
# --------------------------------------------------------------------------------------
def synth_bigvgan_weights(cfg: BigVGANConfig, tag: str = "bigvgan") -> Weights:
    w: Weights = {}

    def conv(name, cout, cin, k, gain, bias=True):
        w[f"{name}.weight"] = synth.fan_in_uniform(f"{tag}/{name}.weight", (cout, cin, k), cin * k, gain)
        if bias:
            w[f"{name}.bias"] = synth.uniform(f"{tag}/{name}.bias", (cout,), 0.05)

    c0 = cfg.upsample_initial_channel
    conv("conv_pre", c0, cfg.num_mels, 7, gain=0.35)
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        cin, cout = cfg.channels(i), cfg.channels(i + 1)
        # ConvTranspose1d weight is [Cin, Cout, k]; each output sample sees k/u taps of every input channel
        w[f"ups.{i}.0.weight"] = synth.fan_in_uniform(f"{tag}/ups.{i}.0.weight", (cin, cout, k), cin * k // u, 0.9)
        w[f"ups.{i}.0.bias"] = synth.uniform(f"{tag}/ups.{i}.0.bias", (cout,), 0.05)
        for j, (rk, dils) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
            r = i * cfg.num_kernels + j
            for l in range(len(dils)):
                conv(f"resblocks.{r}.convs1.{l}", cout, cout, rk, gain=0.6)
                conv(f"resblocks.{r}.convs2.{l}", cout, cout, rk, gain=0.45)
            for a in range(2 * len(dils)):
                w[f"resblocks.{r}.activations.{a}.act.alpha"] = synth.uniform(
                    f"{tag}/resblocks.{r}.activations.{a}.act.alpha", (cout,), 0.6)
                w[f"resblocks.{r}.activations.{a}.act.beta"] = synth.uniform(
                    f"{tag}/resblocks.{r}.activations.{a}.act.beta", (cout,), 0.6, offset=0.4)
    cl = cfg.channels(cfg.num_upsamples)
    w["activation_post.act.alpha"] = synth.uniform(f"{tag}/activation_post.act.alpha", (cl,), 0.6)
    w["activation_post.act.beta"] = synth.uniform(f"{tag}/activation_post.act.beta", (cl,), 0.6, offset=0.4)
    conv("conv_post", 1, cl, 7, gain=0.25, bias=False)
    return w


def synth_mel(name: str, batch: int, num_mels: int, frames: int) -> np.ndarray:
    """Log-mel-like input: BASELINE.md config 2 uses randn*1.5-4; here the same range, uniform."""
    return synth.uniform(name, (batch, num_mels, frames), scale=2.6, offset=-4.0)


# --------------------------------------------------------------------------------------
# GPT (UnifiedVoice hot-path tensors; model_v2.py:381-443, HF GPT2Model built at model_v2.py:290-305)
# --------------------------------------------------------------------------------------
def synth_gpt_weights(cfg: GPTConfig, tag: str = "gpt") -> Weights:
    """Keys/shapes of `UnifiedVoice.state_dict()` restricted to the decode + latent-pass path.
    HF `Conv1D` weights are [in, out] (y = x @ W + b)."""
    w: Weights = {}
    d, f = cfg.model_dim, cfg.ffn_dim

    def u(name, shape, scale, offset=0.0):
        w[name] = synth.uniform(f"{tag}/{name}", shape, scale, offset)

    def lin(name, shape, fan_in, gain=1.0):
        w[name] = synth.fan_in_uniform(f"{tag}/{name}", shape, fan_in, gain)

    for i in range(cfg.layers):
        p = f"gpt.h.{i}"
        u(f"{p}.ln_1.weight", (d,), 0.2, 1.0)
        u(f"{p}.ln_1.bias", (d,), 0.1)
        lin(f"{p}.attn.c_attn.weight", (d, 3 * d), d, 1.2)
        u(f"{p}.attn.c_attn.bias", (3 * d,), 0.1)
        lin(f"{p}.attn.c_proj.weight", (d, d), d, 0.7)
        u(f"{p}.attn.c_proj.bias", (d,), 0.05)
        u(f"{p}.ln_2.weight", (d,), 0.2, 1.0)
        u(f"{p}.ln_2.bias", (d,), 0.1)
        lin(f"{p}.mlp.c_fc.weight", (d, f), d, 1.0)
        u(f"{p}.mlp.c_fc.bias", (f,), 0.1)
        lin(f"{p}.mlp.c_proj.weight", (f, d), f, 0.7)
        u(f"{p}.mlp.c_proj.bias", (d,), 0.05)
    u("gpt.ln_f.weight", (d,), 0.2, 1.0)
    u("gpt.ln_f.bias", (d,), 0.1)
    u("final_norm.weight", (d,), 0.2, 1.0)
    u("final_norm.bias", (d,), 0.1)
    lin("mel_head.weight", (cfg.number_mel_codes, d), d, 3.0)
    u("mel_head.bias", (cfg.number_mel_codes,), 0.5)
    u("mel_embedding.weight", (cfg.number_mel_codes, d), 0.6)
    u("text_embedding.weight", (cfg.number_text_tokens + 1, d), 0.6)
    u("mel_pos_embedding.emb.weight", (cfg.mel_pos_len, d), 0.3)
    u("text_pos_embedding.emb.weight", (cfg.text_pos_len, d), 0.3)
    u("speed_emb.weight", (2, d), 0.1)
    return w


def conformer_pe(max_len: int, d_model: int) -> np.ndarray:
    """PositionalEncoding.__init__ (indextts/gpt/conformer/embedding.py:45-53), same torch ops in the same order."""
    import math
    import torch
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0).numpy()


def synth_gpt_cond_weights(cfg: GPTConfig, tag: str = "gpt") -> Weights:
    """Keys/shapes of the prompt-conditioning half of `UnifiedVoice.state_dict()` (model_v2.py:396-423):
    `conditioning_encoder.*` / `emo_conditioning_encoder.*` (ConformerEncoder, conformer_encoder.py:436-520),
    `perceiver_encoder.*` / `emo_perceiver_encoder.*` (PerceiverResampler, perceiver.py:193-245), `emovec_layer`, `emo_layer`.
    The sinusoid buffer `embed.pos_enc.pe` [1, 5000, D] is part of the reference's state_dict too; it is built here with the
    reference's own torch expression (embedding.py:45-53) so that it is bit-identical."""
    w: Weights = {}
    d = cfg.model_dim

    def u(name, shape, scale, offset=0.0):
        w[name] = synth.uniform(f"{tag}/{name}", shape, scale, offset)

    def lin(name, n_out, n_in, gain=1.0, bias=True, bscale=0.05):
        w[f"{name}.weight"] = synth.fan_in_uniform(f"{tag}/{name}.weight", (n_out, n_in), n_in, gain)
        if bias:
            u(f"{name}.bias", (n_out,), bscale)

    def ln(name, n):
        u(f"{name}.weight", (n,), 0.2, 1.0)
        u(f"{name}.bias", (n,), 0.1)

    def conformer(p, m):
        D, hd = m.output_size, m.output_size // m.attention_heads
        w[f"{p}.embed.pos_enc.pe"] = conformer_pe(5000, D)
        w[f"{p}.embed.conv.0.weight"] = synth.fan_in_uniform(f"{tag}/{p}.embed.conv.0.weight", (D, 1, 3, 3), 9, 1.5)
        u(f"{p}.embed.conv.0.bias", (D,), 0.3)
        lin(f"{p}.embed.out.0", D, D * m.sub_freq, 1.6)
        ln(f"{p}.after_norm", D)
        for i in range(m.num_blocks):
            e = f"{p}.encoders.{i}"
            u(f"{e}.self_attn.pos_bias_u", (m.attention_heads, hd), 0.3)
            u(f"{e}.self_attn.pos_bias_v", (m.attention_heads, hd), 0.3)
            for nm in ("linear_q", "linear_k", "linear_v"):
                lin(f"{e}.self_attn.{nm}", D, D, 1.2)
            lin(f"{e}.self_attn.linear_out", D, D, 0.7)
            lin(f"{e}.self_attn.linear_pos", D, D, 1.0, bias=False)
            lin(f"{e}.feed_forward.w_1", m.linear_units, D, 1.0)
            lin(f"{e}.feed_forward.w_2", D, m.linear_units, 0.7)
            w[f"{e}.conv_module.pointwise_conv1.weight"] = synth.fan_in_uniform(f"{tag}/{e}.conv_module.pointwise_conv1.weight", (2 * D, D, 1), D, 1.2)
            u(f"{e}.conv_module.pointwise_conv1.bias", (2 * D,), 0.05)
            w[f"{e}.conv_module.depthwise_conv.weight"] = synth.fan_in_uniform(f"{tag}/{e}.conv_module.depthwise_conv.weight", (D, 1, m.cnn_kernel), m.cnn_kernel, 1.5)
            u(f"{e}.conv_module.depthwise_conv.bias", (D,), 0.05)
            ln(f"{e}.conv_module.norm", D)
            w[f"{e}.conv_module.pointwise_conv2.weight"] = synth.fan_in_uniform(f"{tag}/{e}.conv_module.pointwise_conv2.weight", (D, D, 1), D, 0.8)
            u(f"{e}.conv_module.pointwise_conv2.bias", (D,), 0.05)
            for nm in ("norm_ff", "norm_mha", "norm_conv", "norm_final"):
                ln(f"{e}.{nm}", D)

    def perceiver(p, dim, m, n_latents):
        inner = m.perceiver_dim_head * m.attention_heads
        ff = int(dim * m.perceiver_mult * 2 / 3)             # perceiver.py:181
        u(f"{p}.latents", (n_latents, dim), 0.5)
        lin(f"{p}.proj_context", dim, m.output_size, 1.0)
        for l in range(m.perceiver_depth):
            lin(f"{p}.layers.{l}.0.to_q", inner, dim, 1.2, bias=False)
            lin(f"{p}.layers.{l}.0.to_kv", 2 * inner, dim, 1.2, bias=False)
            lin(f"{p}.layers.{l}.0.to_out", dim, inner, 0.7, bias=False)
            lin(f"{p}.layers.{l}.1.0", 2 * ff, dim, 1.0)
            lin(f"{p}.layers.{l}.1.2", dim, ff, 0.7)
        u(f"{p}.norm.gamma", (dim,), 0.2, 1.0)

    conformer("conditioning_encoder", cfg.cond_module)
    perceiver("perceiver_encoder", d, cfg.cond_module, cfg.cond_latents)
    conformer("emo_conditioning_encoder", cfg.emo_cond_module)
    perceiver("emo_perceiver_encoder", cfg.emo_perceiver_dim, cfg.emo_cond_module, 1)
    lin("emovec_layer", d, cfg.emo_perceiver_dim, 1.0)
    lin("emo_layer", d, d, 1.0)
    return w


# --------------------------------------------------------------------------------------
# s2mel (MyModel: cfm / length_regulator / gpt_layer, commons.py:390-420) + the semantic-codec
# vq2emb tables (residual_vq.py:144-152, factorized_vector_quantize.py:123-127)
# --------------------------------------------------------------------------------------
def synth_s2mel_weights(cfg: S2MelConfig, tag: str = "s2mel") -> Weights:
    """Keys follow `s2mel.pth['net'][{cfm,length_regulator,gpt_layer}]` prefixed with the sub-model name
    (SURVEY.md §8a).  Weight-norm parametrised layers (final_layer.linear, wavenet.*) are stored FOLDED
    under the plain `.weight` key; an exporter for a real checkpoint computes g*v/||v|| once."""
    w: Weights = {}
    D, H, F_, C = cfg.hidden_dim, cfg.num_heads, cfg.ffn_dim, cfg.in_channels

    def lin(name, n_out, n_in, gain=1.0, bias=True, bscale=0.05):
        w[f"{name}.weight"] = synth.fan_in_uniform(f"{tag}/{name}.weight", (n_out, n_in), n_in, gain)
        if bias:
            w[f"{name}.bias"] = synth.uniform(f"{tag}/{name}.bias", (n_out,), bscale)

    def conv(name, cout, cin, k, gain=1.0):
        w[f"{name}.weight"] = synth.fan_in_uniform(f"{tag}/{name}.weight", (cout, cin, k), cin * k, gain)
        w[f"{name}.bias"] = synth.uniform(f"{tag}/{name}.bias", (cout,), 0.05)

    # ---- DiT (diffusion_transformer.py:103-184) ----
    e = "cfm.estimator"
    for i in range(cfg.depth):
        p = f"{e}.transformer.layers.{i}"
        lin(f"{p}.attention.wqkv", 3 * D, D, 1.0, bias=False)
        lin(f"{p}.attention.wo", D, D, 0.7, bias=False)
        lin(f"{p}.feed_forward.w1", F_, D, 1.0, bias=False)
        lin(f"{p}.feed_forward.w3", F_, D, 1.0, bias=False)
        lin(f"{p}.feed_forward.w2", D, F_, 0.7, bias=False)
        for nrm in ("attention_norm", "ffn_norm"):
            lin(f"{p}.{nrm}.project_layer", 2 * D, D, 0.5, bscale=0.5)
            w[f"{p}.{nrm}.project_layer.bias"][:D] += 1.0      # modulation weight ~ 1
            w[f"{p}.{nrm}.norm.weight"] = synth.uniform(f"{tag}/{p}.{nrm}.norm.weight", (D,), 0.2, 1.0)
        lin(f"{p}.skip_in_linear", D, 2 * D, 0.8)       # present in every block, used by blocks > depth//2
    lin(f"{e}.transformer.norm.project_layer", 2 * D, D, 0.5, bscale=0.5)
    w[f"{e}.transformer.norm.project_layer.bias"][:D] += 1.0
    w[f"{e}.transformer.norm.norm.weight"] = synth.uniform(f"{tag}/{e}.transformer.norm.norm.weight", (D,), 0.2, 1.0)
    lin(f"{e}.cond_projection", D, cfg.content_dim)
    lin(f"{e}.cond_x_merge_linear", D, D + 2 * C + cfg.style_dim)
    lin(f"{e}.t_embedder.mlp.0", D, 256)
    lin(f"{e}.t_embedder.mlp.2", D, D)
    lin(f"{e}.t_embedder2.mlp.0", cfg.wn_hidden, 256)
    lin(f"{e}.t_embedder2.mlp.2", cfg.wn_hidden, cfg.wn_hidden)
    lin(f"{e}.skip_linear", D, D + C)
    lin(f"{e}.conv1", cfg.wn_hidden, D)
    lin(f"{e}.res_projection", cfg.wn_hidden, D)
    lin(f"{e}.final_layer.linear", cfg.wn_hidden, cfg.wn_hidden)             # weight-norm folded
    lin(f"{e}.final_layer.adaLN_modulation.1", 2 * cfg.wn_hidden, cfg.wn_hidden, 0.5)
    conv(f"{e}.conv2", C, cfg.wn_hidden, 1)
    # ---- WaveNet (wavenet.py:103-136), weight-norm folded ----
    Wh = cfg.wn_hidden
    conv(f"{e}.wavenet.cond_layer.conv.conv", 2 * Wh * cfg.wn_layers, Wh, 1, 0.5)
    for i in range(cfg.wn_layers):
        conv(f"{e}.wavenet.in_layers.{i}.conv.conv", 2 * Wh, Wh, cfg.wn_kernel, 1.0)
        conv(f"{e}.wavenet.res_skip_layers.{i}.conv.conv", 2 * Wh if i < cfg.wn_layers - 1 else Wh, Wh, 1, 0.6)
    # ---- length regulator (length_regulator.py:28-88) ----
    lr = "length_regulator"
    lin(f"{lr}.content_in_proj", cfg.lr_channels, cfg.lr_in_channels)
    for n in range(cfg.lr_num_convs):
        conv(f"{lr}.model.{3 * n}", cfg.lr_channels, cfg.lr_channels, 3, 1.3)
        w[f"{lr}.model.{3 * n + 1}.weight"] = synth.uniform(f"{tag}/{lr}.model.{3 * n + 1}.weight", (cfg.lr_channels,), 0.2, 1.0)
        w[f"{lr}.model.{3 * n + 1}.bias"] = synth.uniform(f"{tag}/{lr}.model.{3 * n + 1}.bias", (cfg.lr_channels,), 0.1)
    conv(f"{lr}.model.{3 * cfg.lr_num_convs}", cfg.lr_channels, cfg.lr_channels, 1)
    # ---- gpt_layer (commons.py:413) ----
    dims = (cfg.gpt_dim,) + tuple(cfg.gpt_layer_dims)
    for n in range(3):
        lin(f"gpt_layer.{n}", dims[n + 1], dims[n])
    # ---- semantic codec vq2emb: codebook lookup + out_project (weight-norm folded 1x1 conv) ----
    w["semantic_codec.quantizer.quantizers.0.codebook.weight"] = synth.uniform(
        f"{tag}/semantic_codec.codebook", (cfg.codebook_size, cfg.codebook_dim), 1.0)
    conv("semantic_codec.quantizer.quantizers.0.out_project", cfg.codec_hidden, cfg.codebook_dim, 1)
    return w


# --------------------------------------------------------------------------------------
# w2v-bert-2.0 (HF Wav2Vec2BertModel.state_dict(), the layers the reference runs: infer_v2.py:381-408)
# --------------------------------------------------------------------------------------
def synth_w2vbert_weights(cfg: W2VBertConfig, tag: str = "w2vbert", stats: bool = True) -> Weights:
    """Keys / shapes of `Wav2Vec2BertModel.state_dict()` for `feature_projection.*` and `encoder.layers.{i}.*`, i < cfg.num_layers,
    plus `semantic_mean` / `semantic_std` (wav2vec2bert_stats.pt: mean, sqrt(var); maskgct_utils.py:90-92)."""
    w: Weights = {}
    D, F, hd = cfg.hidden_size, cfg.intermediate_size, cfg.hidden_size // cfg.num_heads

    def u(name, shape, scale, offset=0.0):
        w[name] = synth.uniform(f"{tag}/{name}", shape, scale, offset)

    def lin(name, n_out, n_in, gain=1.0, bias=True):
        w[f"{name}.weight"] = synth.fan_in_uniform(f"{tag}/{name}.weight", (n_out, n_in), n_in, gain)
        if bias:
            u(f"{name}.bias", (n_out,), 0.05)

    def ln(name, n):
        u(f"{name}.weight", (n,), 0.2, 1.0)
        u(f"{name}.bias", (n,), 0.1)

    ln("feature_projection.layer_norm", cfg.input_dim)
    lin("feature_projection.projection", D, cfg.input_dim, 1.4)
    for i in range(cfg.num_layers):
        e = f"encoder.layers.{i}"
        for nm in ("ffn1_layer_norm", "self_attn_layer_norm", "conv_module.layer_norm", "conv_module.depthwise_layer_norm", "ffn2_layer_norm",
                   "final_layer_norm"):
            ln(f"{e}.{nm}", D)
        for f in ("ffn1", "ffn2"):
            lin(f"{e}.{f}.intermediate_dense", F, D, 1.2)
            lin(f"{e}.{f}.output_dense", D, F, 0.9)
        for nm in ("linear_q", "linear_k", "linear_v"):
            lin(f"{e}.self_attn.{nm}", D, D, 1.3)
        lin(f"{e}.self_attn.linear_out", D, D, 0.8)
        u(f"{e}.self_attn.distance_embedding.weight", (cfg.left_max + cfg.right_max + 1, hd), 0.4)
        w[f"{e}.conv_module.pointwise_conv1.weight"] = synth.fan_in_uniform(f"{tag}/{e}.conv_module.pointwise_conv1.weight", (2 * D, D, 1), D, 1.3)
        w[f"{e}.conv_module.depthwise_conv.weight"] = synth.fan_in_uniform(f"{tag}/{e}.conv_module.depthwise_conv.weight", (D, 1, cfg.conv_kernel),
                                                                          cfg.conv_kernel, 1.6)
        w[f"{e}.conv_module.pointwise_conv2.weight"] = synth.fan_in_uniform(f"{tag}/{e}.conv_module.pointwise_conv2.weight", (D, D, 1), D, 0.9)
    if stats:
        u("semantic_mean", (D,), 0.3)
        u("semantic_std", (D,), 0.4, 1.0)
    return w


# --------------------------------------------------------------------------------------
# RepCodec, the semantic codec's encoder + quantizer (kmeans/repcodec_model.py:105-146; weight-norm pairs folded)
# --------------------------------------------------------------------------------------
def synth_repcodec_weights(cfg: RepCodecConfig, tag: str = "repcodec") -> Weights:
    """Keys / shapes of `RepCodec.state_dict()` on the `quantize` path: `encoder.0.*` (VocosBackbone), `encoder.1.*` (Linear),
    `quantizer.quantizers.0.{in_project,out_project}.{weight,bias}` (1x1 convs, weight norm folded) and `.codebook.weight`."""
    w: Weights = {}
    Hs, D, F, cd = cfg.hidden_size, cfg.vocos_dim, cfg.vocos_intermediate_dim, cfg.codebook_dim

    def u(name, shape, scale, offset=0.0):
        w[name] = synth.uniform(f"{tag}/{name}", shape, scale, offset)

    def ln(name, n):
        u(f"{name}.weight", (n,), 0.2, 1.0)
        u(f"{name}.bias", (n,), 0.1)

    w["encoder.0.embed.weight"] = synth.fan_in_uniform(f"{tag}/encoder.0.embed.weight", (D, Hs, 7), Hs * 7, 1.5)
    u("encoder.0.embed.bias", (D,), 0.1)
    ln("encoder.0.norm", D)
    for i in range(cfg.vocos_num_layers):
        e = f"encoder.0.convnext.{i}"
        w[f"{e}.dwconv.weight"] = synth.fan_in_uniform(f"{tag}/{e}.dwconv.weight", (D, 1, 7), 7, 1.5)
        u(f"{e}.dwconv.bias", (D,), 0.1)
        ln(f"{e}.norm", D)
        w[f"{e}.pwconv1.weight"] = synth.fan_in_uniform(f"{tag}/{e}.pwconv1.weight", (F, D), D, 1.3)
        u(f"{e}.pwconv1.bias", (F,), 0.1)
        w[f"{e}.pwconv2.weight"] = synth.fan_in_uniform(f"{tag}/{e}.pwconv2.weight", (D, F), F, 1.2)
        u(f"{e}.pwconv2.bias", (D,), 0.1)
        u(f"{e}.gamma", (D,), 0.3, 0.6)
    ln("encoder.0.final_layer_norm", D)
    w["encoder.1.weight"] = synth.fan_in_uniform(f"{tag}/encoder.1.weight", (Hs, D), D, 1.4)
    u("encoder.1.bias", (Hs,), 0.1)
    q = "quantizer.quantizers.0"
    w[f"{q}.in_project.weight"] = synth.fan_in_uniform(f"{tag}/{q}.in_project.weight", (cd, Hs, 1), Hs, 1.5)
    u(f"{q}.in_project.bias", (cd,), 0.1)
    w[f"{q}.out_project.weight"] = synth.fan_in_uniform(f"{tag}/{q}.out_project.weight", (Hs, cd, 1), cd, 1.0)
    u(f"{q}.out_project.bias", (Hs,), 0.1)
    u(f"{q}.codebook.weight", (cfg.codebook_size, cd), 1.0)
    return w


# --------------------------------------------------------------------------------------
# CAMPPlus (s2mel/modules/campplus/DTDNN.py:62-140; campplus_cn_common.bin's key layout)
# --------------------------------------------------------------------------------------
def synth_campplus_weights(cfg: CamPPlusConfig = CamPPlusConfig(), tag: str = "campplus") -> Weights:
    """Keys / shapes of `CAMPPlus(feat_dim, embedding_size).state_dict()` (BatchNorm running statistics included; the
    `num_batches_tracked` counters are left out: load_state_dict(strict=False) on the reference side)."""
    w: Weights = {}
    mc, G, BNC, IC = cfg.m_channels, cfg.growth_rate, cfg.bn_size * cfg.growth_rate, cfg.init_channels

    def u(name, shape, scale, offset=0.0):
        w[name] = synth.uniform(f"{tag}/{name}", shape, scale, offset)

    def bn(name, n, affine=True):
        if affine:
            u(f"{name}.weight", (n,), 0.3, 1.0)
            u(f"{name}.bias", (n,), 0.2)
        u(f"{name}.running_mean", (n,), 0.3)
        u(f"{name}.running_var", (n,), 0.4, 1.0)

    def conv(name, shape, fan_in, gain):
        w[f"{name}.weight"] = synth.fan_in_uniform(f"{tag}/{name}.weight", shape, fan_in, gain)

    conv("head.conv1", (mc, 1, 3, 3), 9, 1.6)
    bn("head.bn1", mc)
    for l in (1, 2):
        for j in (0, 1):
            p = f"head.layer{l}.{j}"
            conv(f"{p}.conv1", (mc, mc, 3, 3), mc * 9, 1.5)
            bn(f"{p}.bn1", mc)
            conv(f"{p}.conv2", (mc, mc, 3, 3), mc * 9, 1.2)
            bn(f"{p}.bn2", mc)
            if j == 0:
                conv(f"{p}.shortcut.0", (mc, mc, 1, 1), mc, 1.0)
                bn(f"{p}.shortcut.1", mc)
    conv("head.conv2", (mc, mc, 3, 3), mc * 9, 1.5)
    bn("head.bn2", mc)
    ch = mc * (cfg.feat_dim // 8)
    conv("xvector.tdnn.linear", (IC, ch, 5), ch * 5, 1.5)
    bn("xvector.tdnn.nonlinear.batchnorm", IC)
    ch = IC
    for bi, n_layers in enumerate(cfg.block_layers):
        for i in range(n_layers):
            p = f"xvector.block{bi + 1}.tdnnd{i + 1}"
            cin = ch + i * G
            bn(f"{p}.nonlinear1.batchnorm", cin)
            conv(f"{p}.linear1", (BNC, cin, 1), cin, 1.6)
            bn(f"{p}.nonlinear2.batchnorm", BNC)
            conv(f"{p}.cam_layer.linear_local", (G, BNC, 3), BNC * 3, 1.6)
            conv(f"{p}.cam_layer.linear1", (BNC // 2, BNC, 1), BNC, 1.4)
            u(f"{p}.cam_layer.linear1.bias", (BNC // 2,), 0.2)
            conv(f"{p}.cam_layer.linear2", (G, BNC // 2, 1), BNC // 2, 1.4)
            u(f"{p}.cam_layer.linear2.bias", (G,), 0.2)
        ch += n_layers * G
        bn(f"xvector.transit{bi + 1}.nonlinear.batchnorm", ch)
        conv(f"xvector.transit{bi + 1}.linear", (ch // 2, ch, 1), ch, 1.5)
        ch //= 2
    bn("xvector.out_nonlinear.batchnorm", ch)
    conv("xvector.dense.linear", (cfg.embedding_size, 2 * ch, 1), 2 * ch, 1.2)
    bn("xvector.dense.nonlinear.batchnorm", cfg.embedding_size, affine=False)
    return w
