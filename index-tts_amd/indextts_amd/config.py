"""Model hyper-parameters of the IndexTTS-2 hot path.

Values mirror the reference's `checkpoints/config.yaml` (gpt: lines 14-44, s2mel: 53-108) and
`indextts/s2mel/modules/bigvgan/config.json` (lines 11-21, 41-48).  `tiny()` variants keep the
same structure at reduced width so that CPU oracles and golden fixtures finish in seconds.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Tuple


@dataclass(frozen=True)
class BigVGANConfig:
    num_mels: int = 80
    upsample_initial_channel: int = 1536
    upsample_rates: Tuple[int, ...] = (4, 4, 2, 2, 2, 2)
    upsample_kernel_sizes: Tuple[int, ...] = (8, 8, 4, 4, 4, 4)
    resblock_kernel_sizes: Tuple[int, ...] = (3, 7, 11)
    resblock_dilation_sizes: Tuple[Tuple[int, ...], ...] = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
    sampling_rate: int = 22050
    hop_size: int = 256

    @property
    def num_upsamples(self) -> int:
        return len(self.upsample_rates)

    @property
    def num_kernels(self) -> int:
        return len(self.resblock_kernel_sizes)

    def channels(self, stage: int) -> int:
        """Channel count after `stage` up-samplings (stage 0 = conv_pre output)."""
        return self.upsample_initial_channel // (2 ** stage)

    @property
    def total_upsample(self) -> int:
        n = 1
        for u in self.upsample_rates:
            n *= u
        return n

    @staticmethod
    def tiny(initial_channel: int = 64) -> "BigVGANConfig":
        return BigVGANConfig(upsample_initial_channel=initial_channel)


@dataclass(frozen=True)
class CondModuleConfig:
    """Conformer + perceiver prompt encoder (config.yaml:29-43 `condition_module` / `emo_condition_module`;
    conformer_encoder.py:436-520, perceiver.py:193-245).  The conformer input is always 1024 wide (w2v-bert features,
    model_v2.py:396, 406) through `Conv2dSubsampling2` (subsampling.py:131-181)."""
    output_size: int = 512
    linear_units: int = 2048
    attention_heads: int = 8
    num_blocks: int = 6
    perceiver_mult: int = 2
    cnn_kernel: int = 15             # ConformerEncoder default cnn_module_kernel
    input_size: int = 1024
    perceiver_depth: int = 2         # PerceiverResampler default
    perceiver_dim_head: int = 64

    @property
    def sub_freq(self) -> int:       # features left by the 3x3 stride-2 conv: (idim - 1) // 2
        return (self.input_size - 1) // 2


@dataclass(frozen=True)
class GPTConfig:
    model_dim: int = 1280
    heads: int = 20
    layers: int = 24
    number_text_tokens: int = 12000   # text vocab = number_text_tokens + 1 (model_v2.py:381)
    number_mel_codes: int = 8194
    start_mel_token: int = 8192
    stop_mel_token: int = 8193
    start_text_token: int = 0
    stop_text_token: int = 1
    max_mel_tokens: int = 1815
    max_text_tokens: int = 600
    cond_latents: int = 32            # perceiver latents; +2 speed embeddings = 34 prefix rows
    cond_module: CondModuleConfig = field(default_factory=CondModuleConfig)
    emo_cond_module: CondModuleConfig = field(default_factory=lambda: CondModuleConfig(linear_units=1024, attention_heads=4, num_blocks=4))
    emo_perceiver_dim: int = 1024     # hard-coded in the reference (model_v2.py:412, 420)

    @property
    def head_dim(self) -> int:
        return self.model_dim // self.heads

    @property
    def ffn_dim(self) -> int:
        return 4 * self.model_dim

    @property
    def mel_pos_len(self) -> int:     # max_mel_tokens + 2 + max_conditioning_inputs (model_v2.py:404)
        return self.max_mel_tokens + 3

    @property
    def text_pos_len(self) -> int:    # max_text_tokens + 2
        return self.max_text_tokens + 2

    @staticmethod
    def tiny() -> "GPTConfig":
        return GPTConfig(model_dim=128, heads=2, layers=3, number_text_tokens=300,
                         number_mel_codes=258, start_mel_token=256, stop_mel_token=257,
                         max_mel_tokens=120, max_text_tokens=60, cond_latents=6,
                         cond_module=CondModuleConfig(output_size=32, linear_units=64, attention_heads=2, num_blocks=2),
                         emo_cond_module=CondModuleConfig(output_size=32, linear_units=48, attention_heads=2, num_blocks=1))


@dataclass(frozen=True)
class S2MelConfig:
    # DiT (config.yaml:80-101)
    hidden_dim: int = 512
    num_heads: int = 8
    depth: int = 13
    in_channels: int = 80
    content_dim: int = 512
    style_dim: int = 192
    block_size: int = 16384          # rotary table length (diffusion_transformer.py:113)
    rope_base: float = 10000.0
    norm_eps: float = 1e-5
    # WaveNet (config.yaml:102-108)
    wn_hidden: int = 512
    wn_layers: int = 8
    wn_kernel: int = 5
    wn_dilation_rate: int = 1
    # length regulator (config.yaml:67-78)
    lr_channels: int = 512
    lr_in_channels: int = 1024
    lr_num_convs: int = 4            # len(sampling_ratios)
    # gpt_layer (commons.py:413)
    gpt_dim: int = 1280
    gpt_layer_dims: Tuple[int, ...] = (256, 128, 1024)
    # semantic codec (config.yaml:46-52)
    codebook_size: int = 8192
    codebook_dim: int = 8
    codec_hidden: int = 1024

    @property
    def head_dim(self) -> int:
        return self.hidden_dim // self.num_heads

    @property
    def ffn_dim(self) -> int:        # gpt_fast ModelArgs.__post_init__ (model.py:58-64)
        n_hidden = int(2 * (4 * self.hidden_dim) / 3)
        return n_hidden if n_hidden % 256 == 0 else n_hidden + 256 - (n_hidden % 256)

    @staticmethod
    def tiny() -> "S2MelConfig":
        return S2MelConfig(hidden_dim=128, num_heads=2, depth=5, in_channels=16, content_dim=64,
                           style_dim=24, block_size=512, wn_hidden=128, wn_layers=3,
                           lr_channels=64, lr_in_channels=96, gpt_dim=128,
                           gpt_layer_dims=(48, 32, 96), codebook_size=256, codebook_dim=8,
                           codec_hidden=96)


@dataclass(frozen=True)
class PipelineConfig:
    gpt: GPTConfig = field(default_factory=GPTConfig)
    s2mel: S2MelConfig = field(default_factory=S2MelConfig)
    bigvgan: BigVGANConfig = field(default_factory=BigVGANConfig)
    code_to_frame: float = 1.72      # infer_v2.py:844
    diffusion_steps: int = 20        # TARS_DIFFUSION_STEPS default (infer_v2.py:125)
    cfg_rate: float = 0.7            # TARS_CFG_RATE default (infer_v2.py:126)

    @staticmethod
    def tiny() -> "PipelineConfig":
        g, s = GPTConfig.tiny(), S2MelConfig.tiny()
        return PipelineConfig(gpt=g, s2mel=s,
                              bigvgan=BigVGANConfig(num_mels=s.in_channels, upsample_initial_channel=64),
                              diffusion_steps=3)


@dataclass(frozen=True)
class W2VBertConfig:
    """facebook/w2v-bert-2.0 as the reference uses it (utils/maskgct_utils.py:87-93; infer_v2.py:381-408 reads
    hidden_states[17]): HF Wav2Vec2BertConfig's defaults, `num_layers` = the layers that have to run."""
    input_dim: int = 160
    hidden_size: int = 1024
    num_heads: int = 16
    intermediate_size: int = 4096
    num_layers: int = 17
    left_max: int = 64
    right_max: int = 8
    conv_kernel: int = 31
    layer_norm_eps: float = 1e-5

    @staticmethod
    def tiny() -> "W2VBertConfig":
        return W2VBertConfig(input_dim=24, hidden_size=64, num_heads=4, intermediate_size=128, num_layers=3, left_max=6, right_max=3,
                             conv_kernel=7)


@dataclass(frozen=True)
class RepCodecConfig:
    """The semantic codec (checkpoints/config.yaml:45-51 `semantic_codec`; RepCodec, kmeans/repcodec_model.py:35-146)."""
    hidden_size: int = 1024
    codebook_size: int = 8192
    codebook_dim: int = 8
    vocos_dim: int = 384
    vocos_intermediate_dim: int = 2048
    vocos_num_layers: int = 12

    @staticmethod
    def tiny() -> "RepCodecConfig":
        return RepCodecConfig(hidden_size=64, codebook_size=57, codebook_dim=4, vocos_dim=32, vocos_intermediate_dim=48, vocos_num_layers=2)


@dataclass(frozen=True)
class CamPPlusConfig:
    """CAMPPlus(feat_dim=80, embedding_size=192) (infer_v2.py:254; DTDNN.py:62-140: growth 32, bn_size 4, 128 initial channels,
    dense blocks of 12 / 24 / 16 layers with kernel 3 and dilation 1 / 2 / 2)."""
    feat_dim: int = 80
    embedding_size: int = 192
    m_channels: int = 32
    growth_rate: int = 32
    bn_size: int = 4
    init_channels: int = 128
    block_layers: tuple = (12, 24, 16)
    block_dilation: tuple = (1, 2, 2)
