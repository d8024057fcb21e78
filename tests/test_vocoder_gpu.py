"""GPU parity: HIP vocoder kernels (through the C ABI) vs the CPU oracle and the reference's golden vectors."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from indextts_amd import synth, weights
from indextts_amd.config import BigVGANConfig

pytestmark = pytest.mark.gpu

# fp32 kernels vs fp32 reference: differences are summation order + sin implementation only.
ACT_ATOL = 1e-5
CONV_RTOL = 2e-5
# Every test runs in both arithmetic modes: "f32" = exact-fp32 MFMA everywhere (tolerances above); "bf16x3" = the wide
# convolutions (> 96 output rows) on bf16 MFMAs with split operands (3 MFMAs per product, ~2^-16 relative per product):
# tolerances x TOL["m"].
TOL = {"m": 1.0}


@pytest.fixture(autouse=True, params=["f32", "bf16x3"])
def arith(request):
    from indextts_amd import _lib
    _lib.set_gemm_mode(_lib.GEMM_F32 if request.param == "f32" else _lib.GEMM_BF16X3)
    TOL["m"] = 1.0 if request.param == "f32" else 8.0
    yield request.param
    _lib.set_gemm_mode(_lib.GEMM_BF16X3)
    TOL["m"] = 1.0


def _golden(golden_dir):
    return np.load(os.path.join(golden_dir, "vocoder.npz"))


def _filt(device):
    from indextts_amd.vocoder import kaiser_sinc_filter12
    return kaiser_sinc_filter12().to(device)


# ------------------------------------------------------------------------------------------
# fused anti-aliased activation
# ------------------------------------------------------------------------------------------
def test_aa_act_matches_reference_golden(device, golden_dir):
    from indextts_amd.vocoder import anti_alias_activation_forward
    g = _golden(golden_dir)
    f = _filt(device)
    assert np.array_equal(f.cpu().numpy(), g["up_filter"])
    for tag in "abcd":
        B, C, T = (int(v) for v in g[f"act_{tag}_shape"])
        la = synth.uniform(f"golden/act/{tag}/alpha", (C,), 0.8)
        lb = synth.uniform(f"golden/act/{tag}/beta", (C,), 0.8, offset=0.2)
        x = synth.uniform(f"golden/act/{tag}/x", (B, C, T), 3.0)
        y = anti_alias_activation_forward(torch.from_numpy(x).to(device), f, f, torch.from_numpy(la).to(device),
                                          torch.from_numpy(lb).to(device))
        np.testing.assert_allclose(y.cpu().numpy(), g[f"act_{tag}_y"], rtol=0, atol=ACT_ATOL)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_aa_act_half_precision_io(device, dtype):
    """fp16 / bf16 tensors in and out (the reference kernel's dtype dispatch, anti_alias_activation_cuda.cu:232-244): 16-bit
    inputs are widened on load, the arithmetic is fp32, the result is rounded ONCE -- i.e. exactly the fp32 result of the same
    (rounded) input, rounded to the dtype; and within a few ulps of the reference's torch path run in that dtype."""
    from indextts_amd.vocoder import anti_alias_activation_forward
    from oracle import vocoder as ov
    f = _filt(device)
    for (B, C, T) in [(2, 5, 300), (1, 3, 1027), (1, 2, 1)]:
        la = torch.from_numpy(synth.uniform(f"t/act16/a/{C}", (C,), 0.8))
        lb = torch.from_numpy(synth.uniform(f"t/act16/b/{C}", (C,), 0.8, offset=0.2))
        x = torch.from_numpy(synth.uniform(f"t/act16/x/{B}{C}{T}", (B, C, T), 3.0)).to(dtype)
        y = anti_alias_activation_forward(x.to(device), f, f, la.to(device), lb.to(device))
        assert y.dtype == dtype and y.shape == x.shape
        y32 = anti_alias_activation_forward(x.float().to(device), f, f, la.to(device), lb.to(device))
        assert torch.equal(y.cpu(), y32.cpu().to(dtype))                              # one rounding, at the store
        ref = ov.activation1d(x.float(), la, lb)                                       # CPU oracle on the rounded input
        ulp = 2.0 ** (-10 if dtype == torch.float16 else -7)
        assert ((y.cpu().float() - ref).abs() <= ulp * ref.abs().clamp_min(1.0)).all()
    with pytest.raises(TypeError):
        anti_alias_activation_forward(torch.zeros(1, 1, 4, dtype=torch.float64, device=device), f, f, la[:1].to(device), lb[:1].to(device))


@pytest.mark.parametrize("shape", [(1, 1, 1), (1, 3, 2), (2, 5, 3), (1, 2, 5), (1, 4, 1023), (1, 3, 1024), (2, 3, 1025),
                                   (1, 2, 1030), (1, 24, 2048 + 7), (3, 7, 4096), (1, 1, 5000)])
def test_aa_act_vs_oracle_ragged(device, shape):
    """tile edges (1024-sample tiles), lengths not multiple of 4, and tiny T where pads dominate."""
    from indextts_amd.vocoder import anti_alias_activation_forward
    from oracle import vocoder as ov
    B, C, T = shape
    x = torch.from_numpy(synth.uniform(f"t/act/x/{shape}", shape, 4.0))
    la = torch.from_numpy(synth.uniform(f"t/act/a/{shape}", (C,), 1.0))
    lb = torch.from_numpy(synth.uniform(f"t/act/b/{shape}", (C,), 1.0))
    f = _filt(device)
    y = anti_alias_activation_forward(x.to(device), f, f, la.to(device), lb.to(device)).cpu()
    ref = ov.activation1d(x, la, lb)
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=0, atol=ACT_ATOL)


def test_aa_act_empty_and_errors(device):
    from indextts_amd.vocoder import anti_alias_activation_forward
    f = _filt(device)
    z = torch.zeros(2, 3, 0, device=device)
    out = anti_alias_activation_forward(z, f, f, torch.zeros(3, device=device), torch.zeros(3, device=device))
    assert out.shape == (2, 3, 0)
    with pytest.raises(ValueError):
        anti_alias_activation_forward(torch.zeros(1, 3, 8, device=device), f, f, torch.zeros(2, device=device),
                                      torch.zeros(3, device=device))


def test_aa_act_properties_full_size(device):
    """BASELINE config-2 stage-6 shape [8,24,204800]: size-independent properties.
    (1) a constant signal passes through both unit-DC-gain filters: y = c + sin^2(c*a)/b
    (2) translation covariance away from the edges: act(shift(x)) == shift(act(x))"""
    from indextts_amd.vocoder import anti_alias_activation_forward
    B, C, T = 8, 24, 204800
    f = _filt(device)
    la = torch.from_numpy(synth.uniform("t/actfull/a", (C,), 0.5)).to(device)
    lb = torch.from_numpy(synth.uniform("t/actfull/b", (C,), 0.5)).to(device)
    cvals = torch.from_numpy(synth.uniform("t/actfull/c", (B, C, 1), 2.0)).to(device)
    x = cvals.expand(B, C, T).contiguous()
    y = anti_alias_activation_forward(x, f, f, la, lb)
    a, b = torch.exp(la)[None, :, None], torch.exp(lb)[None, :, None]
    expect = cvals + torch.sin(cvals * a) ** 2 / (b + 1e-9)
    assert torch.allclose(y, expect.expand_as(y), atol=2e-5)
    g = torch.Generator(device="cpu").manual_seed(1)
    xr = torch.randn(1, C, 40000, generator=g).to(device)
    y0 = anti_alias_activation_forward(xr, f, f, la, lb)
    y1 = anti_alias_activation_forward(torch.roll(xr, 37, dims=2), f, f, la, lb)
    assert torch.allclose(y1[..., 100:-100], torch.roll(y0, 37, dims=2)[..., 100:-100], atol=1e-5)


# ------------------------------------------------------------------------------------------
# implicit-GEMM conv1d on the fp32 matrix core
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [
    # (B, Cin, Cout, K, dil, T)
    (1, 80, 1536, 7, 1, 37),      # conv_pre shape (Cin not a multiple of 16)
    (2, 768, 768, 3, 1, 200),
    (1, 384, 384, 11, 5, 300),    # widest halo (50)
    (1, 192, 192, 7, 3, 513),
    (2, 96, 96, 11, 1, 700),      # 96-row tile config
    (1, 48, 48, 7, 5, 1000),      # 64-row tile config (48 padded)
    (3, 24, 24, 3, 3, 1500),      # 32-row tile config (24 padded)
    (1, 5, 3, 1, 1, 9),           # 1x1, tiny
    (1, 17, 33, 5, 1, 131),
])
def test_conv1d_vs_torch(device, cfg):
    from indextts_amd.vocoder import Conv1d
    B, Cin, Cout, K, dil, T = cfg
    w = torch.from_numpy(synth.fan_in_uniform(f"t/conv/w/{cfg}", (Cout, Cin, K), Cin * K))
    b = torch.from_numpy(synth.uniform(f"t/conv/b/{cfg}", (Cout,), 0.1))
    x = torch.from_numpy(synth.uniform(f"t/conv/x/{cfg}", (B, Cin, T), 1.0))
    ref = F.conv1d(x.double(), w.double(), b.double(), dilation=dil, padding=(K - 1) * dil // 2).float()
    conv = Conv1d(w, b)
    y = conv(x.to(device), dilation=dil).cpu()
    scale = ref.abs().max().item()
    assert (y - ref).abs().max().item() <= TOL["m"] * CONV_RTOL * scale + 1e-6
    # fused epilogue: residual, scale, accumulate
    res = torch.from_numpy(synth.uniform(f"t/conv/r/{cfg}", (B, Cout, T), 1.0))
    out = torch.from_numpy(synth.uniform(f"t/conv/o/{cfg}", (B, Cout, T), 1.0))
    got = conv(x.to(device), dilation=dil, residual=res.to(device), scale=1.0 / 3, out=out.to(device).clone(), accumulate=True).cpu()
    want = out + (ref + res) / 3
    assert (got - want).abs().max().item() <= TOL["m"] * CONV_RTOL * max(scale, 1.0) + 1e-6


@pytest.mark.parametrize("cfg", [(1, 1536, 768, 8, 4, 23), (2, 384, 192, 4, 2, 130), (1, 48, 24, 4, 2, 515), (1, 64, 32, 8, 4, 9)])
def test_conv_transpose1d_vs_torch(device, cfg):
    from indextts_amd.vocoder import Conv1d
    B, Cin, Cout, K, u, T = cfg
    w = torch.from_numpy(synth.fan_in_uniform(f"t/convt/w/{cfg}", (Cin, Cout, K), Cin * K // u))
    b = torch.from_numpy(synth.uniform(f"t/convt/b/{cfg}", (Cout,), 0.1))
    x = torch.from_numpy(synth.uniform(f"t/convt/x/{cfg}", (B, Cin, T), 1.0))
    ref = F.conv_transpose1d(x.double(), w.double(), b.double(), stride=u, padding=(K - u) // 2).float()
    y = Conv1d(w, b, transposed_stride=u)(x.to(device)).cpu()
    assert y.shape == ref.shape
    assert (y - ref).abs().max().item() <= TOL["m"] * CONV_RTOL * ref.abs().max().item() + 1e-6


def test_conv1d_reflect_pad(device):
    """SConv1d (encodec.py:212-228) as used by the s2mel WaveNet: reflect pad 2/2, k=5."""
    from indextts_amd.vocoder import Conv1d
    B, C, T = 2, 64, 333
    w = torch.from_numpy(synth.fan_in_uniform("t/convr/w", (2 * C, C, 5), C * 5))
    b = torch.from_numpy(synth.uniform("t/convr/b", (2 * C,), 0.1))
    x = torch.from_numpy(synth.uniform("t/convr/x", (B, C, T), 1.0))
    ref = F.conv1d(F.pad(x, (2, 2), mode="reflect").double(), w.double(), b.double()).float()
    y = Conv1d(w, b)(x.to(device), pad_left=2, pad_mode=1).cpu()
    assert (y - ref).abs().max().item() <= TOL["m"] * CONV_RTOL * ref.abs().max().item() + 1e-6


# ------------------------------------------------------------------------------------------
# whole vocoder
# ------------------------------------------------------------------------------------------
def test_bigvgan_matches_reference_golden(device, golden_dir):
    from indextts_amd.vocoder import BigVGAN
    g = _golden(golden_dir)
    for tag in ("w64", "w128"):
        c0, B, Tm = (int(v) for v in g[f"bigvgan_{tag}_cfg"])
        cfg = BigVGANConfig.tiny(c0)
        w = weights.synth_bigvgan_weights(cfg, tag=f"golden/bigvgan/{tag}")
        mel = torch.from_numpy(weights.synth_mel(f"golden/bigvgan/{tag}/mel", B, cfg.num_mels, Tm)).to(device)
        voc = BigVGAN(w, cfg)
        pre, st1 = voc(mel, clamp=False, stage=1)
        np.testing.assert_allclose(st1.cpu().numpy(), g[f"bigvgan_{tag}_stage1"], rtol=0, atol=3e-5 * TOL["m"])
        np.testing.assert_allclose(pre.cpu().numpy(), g[f"bigvgan_{tag}_preclamp"], rtol=0, atol=2e-5 * TOL["m"])
        wav = voc(mel)
        np.testing.assert_allclose(wav.cpu().numpy(), g[f"bigvgan_{tag}_wav"], rtol=0, atol=2e-5 * TOL["m"])
        assert wav.abs().max().item() <= 1.0


def test_bigvgan_vs_oracle_mid_width(device):
    """Width 256 (channels 128..4): exercises every conv tile configuration; oracle on CPU in ~seconds."""
    from indextts_amd.vocoder import BigVGAN
    from oracle import vocoder as ov
    cfg = BigVGANConfig.tiny(256)
    w = weights.synth_bigvgan_weights(cfg, tag="t/bigvgan/256")
    mel = torch.from_numpy(weights.synth_mel("t/bigvgan/256/mel", 2, cfg.num_mels, 9))
    ref = ov.bigvgan_forward(w, cfg, mel, clamp=False)
    got = BigVGAN(w, cfg)(mel.to(device), clamp=False).cpu()
    assert ref.abs().max() > 0.05
    assert (got - ref).abs().max().item() <= TOL["m"] * 2e-5 * max(1.0, ref.abs().max().item())


def test_bigvgan_full_size_properties(device):
    """Full IndexTTS-2 vocoder (1536 ch, 112 M params) at BASELINE config-2 shape [8,80,800]:
    batch independence and time-locality (a change in the last 60 mel frames cannot alter audio
    more than the receptive field upstream), plus bounded output."""
    from indextts_amd.vocoder import BigVGAN
    cfg = BigVGANConfig()
    w = weights.synth_bigvgan_weights(cfg, tag="bench/bigvgan")
    voc = BigVGAN(w, cfg)
    mel = torch.from_numpy(weights.synth_mel("t/bigvgan/full/mel", 8, 80, 800)).to(device)
    wav = voc(mel)
    assert wav.shape == (8, 1, 204800)
    assert torch.isfinite(wav).all() and wav.abs().max().item() <= 1.0
    assert wav.std().item() > 1e-3
    # batch rows are independent: row 3 alone gives the same samples
    solo = voc(mel[3:4].contiguous())
    assert torch.allclose(solo[0], wav[3], atol=1e-5 * TOL["m"])
    # locality: perturbing frames >= 740 leaves the first 600 frames' audio unchanged
    mel2 = mel.clone()
    mel2[:, :, 740:] += 0.5
    wav2 = voc(mel2)
    assert torch.equal(wav2[..., : 600 * 256], wav[..., : 600 * 256])
    assert not torch.equal(wav2[..., 760 * 256:], wav[..., 760 * 256:])


def test_bigvgan_ragged_batch_equals_per_row_calls(device):
    """A ragged batch (lengths 9, 23, 1, 16 of 23 mel frames) through idxtts_bigvgan_fwd_ragged: row b must equal the
    vocoder run on that row's own mel alone -- zero padding of the convolutions and replicate padding of the anti-alias
    filters both happen at the row's own end."""
    from indextts_amd.vocoder import BigVGAN
    cfg = BigVGANConfig.tiny(128)
    w = weights.synth_bigvgan_weights(cfg, tag="t/bigvgan/ragged")
    voc = BigVGAN(w, cfg)
    lens = [9, 23, 1, 16]
    mel = torch.from_numpy(weights.synth_mel("t/bigvgan/ragged/mel", len(lens), cfg.num_mels, max(lens))).to(device)
    got = voc(mel, clamp=False, lengths=lens)
    up = cfg.total_upsample
    for b, n in enumerate(lens):
        solo = voc(mel[b:b + 1, :, :n].contiguous(), clamp=False)
        assert solo.shape[-1] == n * up
        assert torch.allclose(got[b:b + 1, :, : n * up], solo, atol=1e-6 * TOL["m"]), (b, (got[b:b + 1, :, : n * up] - solo).abs().max().item())
