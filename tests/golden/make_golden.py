"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own Python modules.

Runs only in the build container (needs /root/reference, which never travels to the GPU box):
    python tests/golden/make_golden.py [vocoder|s2mel|gpt|all]
The fixtures hold ONLY data: the expected outputs of the reference modules (plus small metadata);
weights and inputs are regenerated bit-identically from `indextts_amd.synth` by the tests, so no
reference source, bytecode or checkpoint is stored.

What is imported from the reference (SURVEY.md §8c "What CAN be imported here"):
  * indextts.s2mel.modules.bigvgan.bigvgan.BigVGAN  (+ alias_free_activation.torch.act.Activation1d)
  * indextts.s2mel.modules.commons.MyModel (cfm / length_regulator / gpt_layer)
with inert `sys.modules` placeholders for packages that are absent in this image and unused on
the hot path (munch, librosa, torchaudio, indextts.s2mel.dac).
The GPT-2 block arithmetic lives in the third-party `transformers` package (reference pins 4.52.1,
this image has 5.x; the reference's own model_v2.py cannot import here), so the GPT fixtures are
produced with the container's `transformers.GPT2Model` after the same surgery as
model_v2.py:290-305 -- see `make_gpt()`.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "index-tts_amd"))
REF = "/root/reference"


def _install_placeholders():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class Munch(dict):
        __getattr__ = dict.get

    stub("munch", Munch=Munch)
    lib = stub("librosa")
    lib.util = stub("librosa.util", normalize=None)
    lib.filters = stub("librosa.filters", mel=None)
    stub("torchaudio")
    stub("indextts.s2mel.dac")
    stub("indextts.s2mel.dac.nn")
    stub("indextts.s2mel.dac.nn.quantize", VectorQuantize=object)
    sys.path.insert(0, REF)
    return Munch


def _sd(weights):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()}


# ----------------------------------------------------------------------------------------------
def make_vocoder():
    from indextts_amd import synth, weights
    from indextts_amd.config import BigVGANConfig
    from indextts.s2mel.modules.bigvgan import bigvgan as ref_bv
    from indextts.s2mel.modules.bigvgan import activations as ref_act
    from indextts.s2mel.modules.bigvgan.alias_free_activation.torch.act import Activation1d

    out = {}
    # (1) Activation1d on [2,24,300] and ragged/short lengths (T=1, 7, 13 exercise the replicate pads)
    for tag, (B, C, T) in {"a": (2, 24, 300), "b": (1, 5, 1), "c": (3, 7, 7), "d": (1, 3, 13)}.items():
        act = Activation1d(activation=ref_act.SnakeBeta(C, alpha_logscale=True)).eval()
        la = synth.uniform(f"golden/act/{tag}/alpha", (C,), 0.8)
        lb = synth.uniform(f"golden/act/{tag}/beta", (C,), 0.8, offset=0.2)
        x = synth.uniform(f"golden/act/{tag}/x", (B, C, T), 3.0)
        act.act.alpha.data = torch.from_numpy(la)
        act.act.beta.data = torch.from_numpy(lb)
        with torch.no_grad():
            y = act(torch.from_numpy(x))
        out[f"act_{tag}_shape"] = np.array([B, C, T])
        out[f"act_{tag}_y"] = y.numpy()
        if tag == "a":
            out["up_filter"] = act.upsample.filter.reshape(-1).numpy()
            out["down_filter"] = act.downsample.lowpass.filter.reshape(-1).numpy()

    # (2) full BigVGAN at reduced width, two widths x two lengths
    for tag, (c0, B, Tm) in {"w64": (64, 2, 12), "w128": (128, 1, 5)}.items():
        cfg = BigVGANConfig.tiny(c0)
        h = ref_bv.load_hparams_from_json(os.path.join(REF, "indextts/s2mel/modules/bigvgan/config.json"))
        h["upsample_initial_channel"] = c0
        m = ref_bv.BigVGAN(h, use_cuda_kernel=False)
        m.remove_weight_norm()
        m.eval()
        w = weights.synth_bigvgan_weights(cfg, tag=f"golden/bigvgan/{tag}")
        sd = m.state_dict()
        new = _sd(w)
        for k in sd:  # filters are buffers: keep the reference's own
            if k not in new:
                assert k.endswith("filter"), k
                new[k] = sd[k]
        m.load_state_dict(new, strict=True)
        mel = weights.synth_mel(f"golden/bigvgan/{tag}/mel", B, cfg.num_mels, Tm)
        with torch.no_grad():
            wav = m(torch.from_numpy(mel))
            # pre-clamp waveform too (so a saturated output cannot hide an error)
            m.use_tanh_at_final = False
            x = m.conv_pre(torch.from_numpy(mel))
            for i in range(m.num_upsamples):
                x = m.ups[i][0](x)
                xs = None
                for j in range(m.num_kernels):
                    r = m.resblocks[i * m.num_kernels + j](x)
                    xs = r if xs is None else xs + r
                x = xs / m.num_kernels
                if i == 0:
                    out[f"bigvgan_{tag}_stage1"] = x.numpy().copy()
            x = m.conv_post(m.activation_post(x))
        out[f"bigvgan_{tag}_cfg"] = np.array([c0, B, Tm])
        out[f"bigvgan_{tag}_wav"] = wav.numpy()
        out[f"bigvgan_{tag}_preclamp"] = x.numpy()
    np.savez_compressed(os.path.join(HERE, "vocoder.npz"), **out)
    print("wrote vocoder.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    Munch = _install_placeholders()
    torch.manual_seed(0)
    if which in ("vocoder", "all"):
        make_vocoder()
    if which in ("s2mel", "all") and "make_s2mel" in globals():
        globals()["make_s2mel"](Munch)
    if which in ("gpt", "all") and "make_gpt" in globals():
        globals()["make_gpt"]()
