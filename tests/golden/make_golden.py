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
sys.path.insert(0, ROOT)
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


# ----------------------------------------------------------------------------------------------
def make_s2mel(Munch):
    """s2mel fixtures from the reference's MyModel (cfm + length_regulator + gpt_layer) and the semantic codec's
    FactorizedVectorQuantize.vq2emb, at reduced width (S2MelConfig.tiny())."""
    from indextts_amd import synth, weights
    from indextts_amd.config import S2MelConfig
    from indextts.s2mel.modules.commons import MyModel
    from indextts.utils.maskgct.models.codec.amphion_codec.quantize.factorized_vector_quantize import FactorizedVectorQuantize

    cfg = S2MelConfig.tiny()

    def to_m(d):
        return Munch({k: to_m(v) for k, v in d.items()}) if isinstance(d, dict) else d

    args = to_m({
        "dit_type": "DiT", "reg_loss_type": "l1",
        "style_encoder": {"dim": cfg.style_dim},
        "length_regulator": {"channels": cfg.lr_channels, "is_discrete": False, "in_channels": cfg.lr_in_channels,
                             "content_codebook_size": 2048, "sampling_ratios": [1] * cfg.lr_num_convs, "vector_quantize": False,
                             "n_codebooks": 1, "quantizer_dropout": 0.0, "f0_condition": False, "n_f0_bins": 512},
        "DiT": {"hidden_dim": cfg.hidden_dim, "num_heads": cfg.num_heads, "depth": cfg.depth, "class_dropout_prob": 0.1,
                "block_size": 8192, "in_channels": cfg.in_channels, "style_condition": True, "final_layer_type": "wavenet",
                "target": "mel", "content_dim": cfg.content_dim, "content_codebook_size": 1024, "content_type": "discrete",
                "f0_condition": False, "n_f0_bins": 512, "content_codebooks": 1, "is_causal": False, "long_skip_connection": True,
                "zero_prompt_speech_token": False, "time_as_token": False, "style_as_token": False, "uvit_skip_connection": True,
                "add_resblock_in_transformer": False},
        "wavenet": {"hidden_dim": cfg.wn_hidden, "num_layers": cfg.wn_layers, "kernel_size": cfg.wn_kernel,
                    "dilation_rate": cfg.wn_dilation_rate, "p_dropout": 0.2, "style_condition": True},
    })
    mm = MyModel(args, use_gpt_latent=True)
    # the reference hard-codes gpt_layer = Linear(1280,256) -> (256,128) -> (128,1024): rebuild at the tiny widths
    dims = (cfg.gpt_dim,) + tuple(cfg.gpt_layer_dims)
    mm.models["gpt_layer"] = torch.nn.Sequential(*[torch.nn.Linear(dims[i], dims[i + 1]) for i in range(3)])
    mm.eval()
    w = weights.synth_s2mel_weights(cfg, tag="golden/s2mel")
    ref_sd = mm.state_dict()
    new = {}
    for k, v in ref_sd.items():
        name = k[len("models."):]
        if name.endswith("weight_g") or name.endswith("weight_v"):
            base = name[: -len("_g")]                       # '....weight'
            if base in w:
                wt = torch.from_numpy(w[base])
                if name.endswith("weight_v"):
                    new[k] = wt
                else:                                        # g = ||v|| over all dims but 0 -> g*v/||v|| == v
                    new[k] = wt.reshape(wt.shape[0], -1).norm(dim=1).reshape(v.shape)
            else:
                new[k] = v                                   # x_embedder: unused in forward
        elif name in w:
            new[k] = torch.from_numpy(w[name])
        else:
            assert any(t in name for t in ("input_pos", "freqs", "cond_embedder", "content_mask_embedder", "mask_token",
                                           "x_embedder", "embedding")), name
            new[k] = v
    mm.load_state_dict(new, strict=True)
    used = {k[len("models."):].replace("weight_v", "weight").replace("weight_g", "weight") for k in ref_sd}
    assert all(k in used or k.startswith("semantic_codec") for k in w), [k for k in w if k not in used and not k.startswith("semantic_codec")]
    est = mm.models["cfm"].estimator
    est.setup_caches(max_batch_size=2, max_seq_length=cfg.block_size)
    # the reference builds the rotary table for block_size=16384 regardless of config; rows are position-indexed
    out = {}
    with torch.no_grad():
        # (a) gpt_layer
        lat = torch.from_numpy(synth.uniform("golden/s2mel/latent", (2, 9, cfg.gpt_dim), 1.0))
        out["gpt_layer"] = mm.models["gpt_layer"](lat).numpy()
        # (b) vq2emb
        fvq = FactorizedVectorQuantize(input_dim=cfg.codec_hidden, codebook_size=cfg.codebook_size, codebook_dim=cfg.codebook_dim,
                                       use_l2_normlize=True)
        fvq.eval()
        fvq.codebook.weight.data = torch.from_numpy(w["semantic_codec.quantizer.quantizers.0.codebook.weight"])
        ow = torch.from_numpy(w["semantic_codec.quantizer.quantizers.0.out_project.weight"])
        fvq.out_project.weight_v.data = ow
        fvq.out_project.weight_g.data = ow.reshape(ow.shape[0], -1).norm(dim=1).reshape(fvq.out_project.weight_g.shape)
        fvq.out_project.bias.data = torch.from_numpy(w["semantic_codec.quantizer.quantizers.0.out_project.bias"])
        codes = torch.from_numpy(synth.integers("golden/s2mel/codes", (2, 9), 0, cfg.codebook_size))
        out["vq2emb"] = fvq.vq2emb(codes).transpose(1, 2).numpy()            # [B,M,hidden] as infer_v2.py:841-842
        # (c) length regulator, B=1 at two lengths (M=9 -> 15 frames, M=20 -> 34 frames)
        for tag, M in (("a", 9), ("b", 20)):
            S = torch.from_numpy(synth.uniform(f"golden/s2mel/S_{tag}", (1, M, cfg.lr_in_channels), 1.0))
            ylens = (torch.LongTensor([M]) * 1.72).long()
            out[f"lr_{tag}"] = mm.models["length_regulator"](S, ylens=ylens, n_quantizers=3, f0=None)[0].numpy()
        # (c2) the composed call of infer_v2.py:835-849 per utterance: length_regulator(vq2emb(codes) + gpt_layer(latent))
        for tag, M in (("a", 9), ("b", 20)):
            lat1 = torch.from_numpy(synth.uniform(f"t/s2mel/lat_{tag}", (1, M, cfg.gpt_dim), 1.0))
            codes1 = torch.from_numpy(synth.integers(f"t/s2mel/codes_{tag}", (1, M), 0, cfg.codebook_size))
            S = fvq.vq2emb(codes1).transpose(1, 2) + mm.models["gpt_layer"](lat1)
            ylens = (torch.LongTensor([M]) * 1.72).long()
            out[f"cond_{tag}"] = mm.models["length_regulator"](S, ylens=ylens, n_quantizers=3, f0=None)[0].numpy()
        # (d) one DiT forward, N=2 rows with different x_lens (exercises the key-padding mask and the WaveNet mask)
        T = 37
        x = torch.from_numpy(synth.uniform("golden/s2mel/dit/x", (2, cfg.in_channels, T), 1.0))
        px = torch.from_numpy(synth.uniform("golden/s2mel/dit/prompt", (2, cfg.in_channels, T), 1.0))
        px[..., 12:] = 0
        st = torch.from_numpy(synth.uniform("golden/s2mel/dit/style", (2, cfg.style_dim), 1.0))
        mu = torch.from_numpy(synth.uniform("golden/s2mel/dit/mu", (2, T, cfg.content_dim), 1.0))
        tt = torch.tensor([0.35, 0.35])
        out["dit"] = est(x, px, torch.LongTensor([T, T - 6]), tt, st, mu).numpy()
        # the second row ALONE at its own length (the only way infer_v2 ever calls the estimator: B = 1, x_lens = T): what a
        # ragged batch must reproduce for that row, reflect padding of the WaveNet convolutions at the row's own end included
        out["dit_row1_alone"] = est(x[1:2, :, :T - 6].contiguous(), px[1:2, :, :T - 6].contiguous(), torch.LongTensor([T - 6]), tt[:1], st[1:2],
                                    mu[1:2, :T - 6].contiguous()).numpy()
        # (e) CFM Euler with CFG, B=1 (the reference's only mode), Tp=11 prompt frames + 23 generated, 3 steps
        Tp, Tg = 11, 23
        T = Tp + Tg
        z = torch.from_numpy(synth.uniform("golden/s2mel/cfm/z", (1, cfg.in_channels, T), 1.7))
        mu = torch.from_numpy(synth.uniform("golden/s2mel/cfm/mu", (1, T, cfg.content_dim), 1.0))
        prompt = torch.from_numpy(synth.uniform("golden/s2mel/cfm/prompt", (1, cfg.in_channels, Tp), 1.0))
        st = torch.from_numpy(synth.uniform("golden/s2mel/cfm/style", (1, cfg.style_dim), 1.0))
        t_span = torch.linspace(0, 1, 3 + 1)
        out["cfm"] = mm.models["cfm"].solve_euler(z.clone(), torch.LongTensor([T]), prompt, mu, st, None, t_span, 0.7).numpy()
    np.savez_compressed(os.path.join(HERE, "s2mel.npz"), **out)
    print("wrote s2mel.npz", {k: (v.shape, float(np.abs(v).max())) for k, v in out.items()})


# ----------------------------------------------------------------------------------------------
def make_gpt():
    """GPT-2 stack fixtures from the container's `transformers.GPT2Model` (the reference instantiates the
    same class from transformers==4.52.1 at model_v2.py:290-305) + HF's RepetitionPenaltyLogitsProcessor."""
    import functools
    from transformers import GPT2Config, GPT2Model
    from transformers.cache_utils import DynamicCache
    from transformers.generation.logits_process import RepetitionPenaltyLogitsProcessor
    import torch.nn.functional as F
    from indextts_amd import synth, weights
    from indextts_amd.config import GPTConfig

    def null_position_embeddings(range, dim):          # model_v2.py:32-33
        return torch.zeros((range.shape[0], range.shape[1], dim), device=range.device)

    cfg = GPTConfig.tiny()
    d = cfg.model_dim
    w = weights.synth_gpt_weights(cfg, tag="golden/gpt")
    seq = cfg.max_mel_tokens + cfg.max_text_tokens + 2
    gc = GPT2Config(vocab_size=256, n_positions=seq, n_ctx=seq, n_embd=d, n_layer=cfg.layers, n_head=cfg.heads,
                    use_cache=True, attn_pdrop=0.0, embd_pdrop=0.0, resid_pdrop=0.0)
    gpt = GPT2Model(gc).eval()
    del gpt.wpe
    gpt.wpe = functools.partial(null_position_embeddings, dim=d)
    del gpt.wte
    sd = {k[len("gpt."):]: torch.from_numpy(v) for k, v in w.items() if k.startswith("gpt.")}
    missing, unexpected = gpt.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(("attn.bias" in m or "masked_bias" in m) for m in missing), missing
    tw = {k: torch.from_numpy(v) for k, v in w.items()}

    def head(h):
        h = F.layer_norm(h, (d,), tw["final_norm.weight"], tw["final_norm.bias"], 1e-5)
        return h @ tw["mel_head.weight"].t() + tw["mel_head.bias"]

    out = {}
    # ---- greedy decode: B=3 rows, the second and third left-padded (ragged text lengths) ----
    B, L, NEW = 3, 12, 10
    P = cfg.cond_latents + 2 + L + 2
    conds = torch.from_numpy(synth.uniform("golden/gpt/conds", (B, cfg.cond_latents + 2, d), 0.5))
    text = torch.from_numpy(synth.integers("golden/gpt/text", (B, L), 2, cfg.number_text_tokens))
    text[1, 9:] = cfg.stop_text_token       # row 1 has 9 real tokens, row 2 has 5
    text[2, 5:] = cfg.stop_text_token
    # inputs_embeds / mask built exactly as model_v2.py:749-779 describes (restated in oracle.gpt; checked there)
    from oracle import gpt as og
    fake, inputs_embeds, attention_mask = og.prepare_gpt_inputs(tw, cfg, conds, text)
    me, mp = tw["mel_embedding.weight"], tw["mel_pos_embedding.emb.weight"]
    proc = RepetitionPenaltyLogitsProcessor(penalty=10.0)
    input_ids = fake.clone()
    cache = DynamicCache()
    logits_all = []
    unfinished = torch.ones(B, dtype=torch.long)
    with torch.no_grad():
        for step in range(NEW):
            if step == 0:
                emb = torch.cat([inputs_embeds, (me[cfg.start_mel_token] + mp[0])[None, None].expand(B, 1, d)], 1)
            else:
                emb = (me[input_ids[:, -1]] + mp[attention_mask.shape[1] - P])[:, None]
            o = gpt(inputs_embeds=emb, past_key_values=cache, attention_mask=attention_mask, use_cache=True, return_dict=True)
            cache = o.past_key_values
            logits = head(o.last_hidden_state[:, -1]).float()
            logits_all.append(logits.clone())
            scores = proc(input_ids, logits.clone())
            nxt = torch.argmax(scores, -1)
            nxt = nxt * unfinished + cfg.stop_mel_token * (1 - unfinished)
            input_ids = torch.cat([input_ids, nxt[:, None]], 1)
            attention_mask = torch.cat([attention_mask, torch.ones(B, 1, dtype=torch.long)], 1)
            unfinished = unfinished & (nxt != cfg.stop_mel_token).long()
    out["greedy_codes"] = input_ids[:, P + 1:].numpy()
    out["greedy_logits"] = torch.stack(logits_all, 1).numpy()
    out["greedy_text"] = text.numpy()
    # ---- multinomial sampling, HF warpers: same prompts, do_sample=True / num_beams=1 (transformers_generation_utils.py:
    #      3196-3262, warpers 1036-1044), seeded; the Exp(1) draws torch.multinomial makes internally are re-drawn with the
    #      same seed and stored, so the oracle / GPU path can be fed the identical noise ----
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    TEMP, TOPK, TOPP, SEED = 0.8, 30, 0.8, 20240607
    warpers = [TemperatureLogitsWarper(TEMP), TopKLogitsWarper(top_k=TOPK, min_tokens_to_keep=1), TopPLogitsWarper(top_p=TOPP, min_tokens_to_keep=1)]
    torch.manual_seed(SEED)
    input_ids = fake.clone()
    attention_mask = og.prepare_gpt_inputs(tw, cfg, conds, text)[2]
    cache = DynamicCache()
    unfinished = torch.ones(B, dtype=torch.long)
    with torch.no_grad():
        for step in range(NEW):
            if step == 0:
                emb = torch.cat([inputs_embeds, (me[cfg.start_mel_token] + mp[0])[None, None].expand(B, 1, d)], 1)
            else:
                emb = (me[input_ids[:, -1]] + mp[attention_mask.shape[1] - P])[:, None]
            o = gpt(inputs_embeds=emb, past_key_values=cache, attention_mask=attention_mask, use_cache=True, return_dict=True)
            cache = o.past_key_values
            scores = proc(input_ids, head(o.last_hidden_state[:, -1]).float().clone())
            for wp in warpers:
                scores = wp(input_ids, scores)
            nxt = torch.multinomial(torch.softmax(scores, -1), num_samples=1).squeeze(1)
            nxt = nxt * unfinished + cfg.stop_mel_token * (1 - unfinished)
            input_ids = torch.cat([input_ids, nxt[:, None]], 1)
            attention_mask = torch.cat([attention_mask, torch.ones(B, 1, dtype=torch.long)], 1)
            unfinished = unfinished & (nxt != cfg.stop_mel_token).long()
    out["sample_codes"] = input_ids[:, P + 1:].numpy()
    torch.manual_seed(SEED)
    noise = torch.stack([torch.empty(B, cfg.number_mel_codes).exponential_(1) for _ in range(NEW)])
    out["sample_noise"] = noise.numpy()
    out["sample_params"] = np.array([TEMP, TOPK, TOPP], dtype=np.float64)
    chk = og.generate_sample(tw, cfg, conds, text, NEW, noise, 10.0, TEMP, TOPK, TOPP)
    assert np.array_equal(chk.numpy(), out["sample_codes"][:, :chk.shape[1]]), "explicit-noise restatement != HF multinomial sampling"
    # ---- latent pass (no mask, causal): B=2, L=7, M=9 ----
    B2, L2, M2 = 2, 7, 9
    lat = torch.from_numpy(synth.uniform("golden/gpt/lat", (B2, cfg.cond_latents, d), 0.5))
    emo = torch.from_numpy(synth.uniform("golden/gpt/emo", (B2, d), 0.3))
    text2 = torch.from_numpy(synth.integers("golden/gpt/text2", (B2, L2), 2, cfg.number_text_tokens))
    codes2 = torch.from_numpy(synth.integers("golden/gpt/codes2", (B2, M2), 0, cfg.start_mel_token))
    conds2 = og.conds_latent(tw, cfg, lat, emo)
    tin = F.pad(F.pad(text2, (0, 1), value=cfg.stop_text_token), (1, 0), value=cfg.start_text_token)
    min_ = F.pad(F.pad(codes2, (0, 1), value=cfg.stop_mel_token), (1, 0), value=cfg.start_mel_token)
    emb = torch.cat([conds2, tw["text_embedding.weight"][tin] + tw["text_pos_embedding.emb.weight"][: L2 + 2],
                     me[min_] + mp[: M2 + 2]], 1)
    with torch.no_grad():
        hs = gpt(inputs_embeds=emb, return_dict=True).last_hidden_state
    enc = F.layer_norm(hs[:, conds2.shape[1]:], (d,), tw["final_norm.weight"], tw["final_norm.bias"], 1e-5)
    out["latent"] = enc[:, -(M2 + 2):][:, :-2].numpy()
    out["latent_hidden"] = hs.numpy()
    np.savez_compressed(os.path.join(HERE, "gpt.npz"), **out)
    print("wrote gpt.npz", {k: v.shape for k, v in out.items()}, "codes", out["greedy_codes"].tolist())


def _import_reference_model_v2():
    """`import indextts.gpt.model_v2` from /root/reference.  Its module-level imports name three things this image lacks:
    the vendored `indextts.gpt.transformers_gpt2` (its sibling `transformers_generation_utils.py:28` needs
    `transformers.cache_utils.OffloadedCache`, gone in transformers 5.x), `transformers.utils.model_parallel_utils` (dead
    `parallelize()` code) and `torchaudio` (pulled in by `indextts.utils.common` for an unrelated loader).  The first is
    satisfied by re-exporting the container's `GPT2PreTrainedModel` / `GPT2Model` -- the very classes `model_v2.py:290`
    instantiates from upstream `transformers` anyway -- the other two by inert placeholders."""
    import contextlib
    import importlib.machinery
    import io
    import transformers
    from transformers import GPT2Model, GPT2PreTrainedModel

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        sys.modules[name] = m

    # model_v2.py:149-151, 194-198 converts the KV cache through `DynamicCache.{from,to}_legacy_cache`, two pure container
    # conversions of transformers 4.52.1 (cache_utils.py there: tuple of per-layer (key, value) <-> DynamicCache.update per
    # layer) that 5.x dropped.  Restored here, in the fixture generator's process only, so the reference's forward runs unmodified.
    from transformers.cache_utils import DynamicCache
    if not hasattr(DynamicCache, "to_legacy_cache"):
        def to_legacy_cache(self):
            return tuple((l.keys, l.values) for l in self.layers)

        def from_legacy_cache(cls, past_key_values=None):
            c = cls()
            for i, (k, v) in enumerate(past_key_values or ()):
                c.update(k, v, i)
            return c
        DynamicCache.to_legacy_cache = to_legacy_cache
        DynamicCache.from_legacy_cache = classmethod(from_legacy_cache)
    stub("indextts.gpt.transformers_gpt2", GPT2PreTrainedModel=GPT2PreTrainedModel, GPT2Model=GPT2Model)
    stub("transformers.utils.model_parallel_utils", assert_device_map=None, get_device_map=None)
    stub("torchaudio")
    if REF not in sys.path:
        sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        import indextts.gpt.model_v2 as mv
    return mv


def build_reference_unified_voice(mv, cfg, w):
    """The reference's own `UnifiedVoice` (model_v2.py:338-560) at `cfg`'s sizes carrying the synthetic weights `w`."""
    import contextlib
    import io
    cm, em = cfg.cond_module, cfg.emo_cond_module
    as_dict = lambda m: dict(output_size=m.output_size, linear_units=m.linear_units, attention_heads=m.attention_heads,
                             num_blocks=m.num_blocks, input_layer="conv2d2", perceiver_mult=m.perceiver_mult)
    uv = mv.UnifiedVoice(layers=cfg.layers, model_dim=cfg.model_dim, heads=cfg.heads, max_text_tokens=cfg.max_text_tokens,
                         max_mel_tokens=cfg.max_mel_tokens, number_text_tokens=cfg.number_text_tokens,
                         number_mel_codes=cfg.number_mel_codes, start_mel_token=cfg.start_mel_token,
                         stop_mel_token=cfg.stop_mel_token, start_text_token=cfg.start_text_token,
                         stop_text_token=cfg.stop_text_token, condition_num_latent=cfg.cond_latents,
                         condition_type="conformer_perceiver", condition_module=as_dict(cm), emo_condition_module=as_dict(em))
    missing, unexpected = uv.load_state_dict(_sd(w), strict=False)
    assert not unexpected, unexpected
    for k in missing:   # buffers and the unused text head only
        assert k.endswith("pos_enc.pe") or k.startswith("text_head.") or k.endswith("attn.bias") or k.endswith("masked_bias"), k
    with contextlib.redirect_stdout(io.StringIO()):
        uv.post_init_gpt2_config(use_deepspeed=False, kv_cache=True, half=False)
    return uv.eval()


def _import_reference_beam_scorer():
    """The reference's vendored `BeamSearchScorer` / `BeamHypotheses` (indextts/gpt/transformers_beam_search.py:123-420,
    930-1013).  Its only missing import, `transformers.generation.beam_constraints`, serves the constrained scorer, which is
    never used: inert placeholder."""
    import importlib.util
    bc = types.ModuleType("transformers.generation.beam_constraints")
    bc.Constraint = object
    bc.ConstraintListState = object
    sys.modules.setdefault("transformers.generation.beam_constraints", bc)
    spec = importlib.util.spec_from_file_location("ref_transformers_beam_search", os.path.join(REF, "indextts/gpt/transformers_beam_search.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def drive_vendored_beam_search(uv, scorer_mod, cfg, conds, text, max_new, num_beams, do_sample, temperature, top_k, top_p,
                               repetition_penalty, length_penalty):
    """The vendored `GenerationMixin._beam_search` loop (transformers_generation_utils.py:3325-3516; that module cannot be
    imported under transformers 5.x) driven by hand, statement for statement, around code that DOES run: the reference's
    `GPT2InferenceModel.prepare_inputs_for_generation` / `forward`, this image's HF logits processors (the classes lines
    900-901, 1034-1043 instantiate) and the reference's vendored `BeamSearchScorer.process` / `finalize`.  Sampling goes
    through `torch.multinomial` on the global RNG exactly as line 3510 does."""
    from transformers.generation.logits_process import (LogitsProcessorList, RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper,
                                                        TopKLogitsWarper, TopPLogitsWarper)
    im = uv.inference_model
    fake, inputs_embeds, attention_mask = uv.prepare_gpt_inputs(conds, text)
    im.store_mel_emb(inputs_embeds)
    B = fake.shape[0]
    procs = LogitsProcessorList([RepetitionPenaltyLogitsProcessor(penalty=repetition_penalty)])
    if do_sample:
        procs += [TemperatureLogitsWarper(temperature), TopKLogitsWarper(top_k=top_k, min_tokens_to_keep=2),
                  TopPLogitsWarper(top_p=top_p, min_tokens_to_keep=2)]
    eos = torch.tensor([cfg.stop_mel_token])
    max_length = fake.shape[1] + max_new
    scorer = scorer_mod.BeamSearchScorer(batch_size=B, num_beams=num_beams, device=fake.device, length_penalty=length_penalty,
                                         do_early_stopping=False, num_beam_hyps_to_keep=1, max_length=max_length)
    input_ids = fake.repeat_interleave(num_beams, 0)
    mask = attention_mask.repeat_interleave(num_beams, 0)
    beam_scores = torch.zeros((B, num_beams), dtype=torch.float)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view((B * num_beams,))
    prompt_len = input_ids.shape[-1]
    past = None
    with torch.no_grad():
        while True:
            mi = im.prepare_inputs_for_generation(input_ids, past_key_values=past, attention_mask=mask, use_cache=True)
            o = im(**mi, return_dict=True)
            past = o.past_key_values
            mask = torch.cat([mask, mask.new_ones((mask.shape[0], 1))], dim=-1)
            next_token_logits = o.logits[:, -1, :].clone().float()
            next_token_scores = torch.nn.functional.log_softmax(next_token_logits, dim=-1)
            processed = procs(input_ids, next_token_scores)
            next_token_scores = processed + beam_scores[:, None].expand_as(processed)
            vocab = next_token_scores.shape[-1]
            next_token_scores = next_token_scores.view(B, num_beams * vocab)
            n_keep = max(2, 1 + 1) * num_beams
            if do_sample:
                probs = torch.nn.functional.softmax(next_token_scores, dim=-1)
                next_tokens = torch.multinomial(probs, num_samples=n_keep)
                next_token_scores = torch.gather(next_token_scores, -1, next_tokens)
                next_token_scores, _indices = torch.sort(next_token_scores, descending=True, dim=1)
                next_tokens = torch.gather(next_tokens, -1, _indices)
            else:
                next_token_scores, next_tokens = torch.topk(next_token_scores, n_keep, dim=1, largest=True, sorted=True)
            next_indices = torch.div(next_tokens, vocab, rounding_mode="floor")
            next_tokens = next_tokens % vocab
            bo = scorer.process(input_ids, next_token_scores, next_tokens, next_indices, pad_token_id=cfg.stop_mel_token,
                                eos_token_id=eos, beam_indices=None, decoder_prompt_len=prompt_len)
            beam_scores, beam_next, beam_idx = bo["next_beam_scores"], bo["next_beam_tokens"], bo["next_beam_indices"]
            input_ids = torch.cat([input_ids[beam_idx, :], beam_next.unsqueeze(-1)], dim=-1)
            past = im._reorder_cache(past, beam_idx)                      # model_v2.py:227-240
            if scorer.is_done or input_ids.shape[-1] >= max_length:
                break
        seq = scorer.finalize(input_ids, beam_scores, next_tokens, next_indices, pad_token_id=cfg.stop_mel_token, eos_token_id=eos,
                              max_length=max_length, beam_indices=None, decoder_prompt_len=prompt_len)["sequences"]
    return seq[:, prompt_len:]


def make_gpt_ref():
    """GPT fixtures produced by the REFERENCE's own `UnifiedVoice` / `GPT2InferenceModel` code (model_v2.py), imported from
    /root/reference: conditioning encoders, emotion vector, prompt layout, cached decode forward, latent pass and -- through
    the container's HF `generate` driving the reference's `prepare_inputs_for_generation` / `forward` -- greedy and
    beam-sample token sequences."""
    import contextlib
    import io
    from indextts_amd import synth, weights
    from indextts_amd.config import GPTConfig
    mv = _import_reference_model_v2()
    cfg = GPTConfig.tiny()
    d = cfg.model_dim
    w = weights.synth_gpt_weights(cfg, tag="golden/gptref")
    w.update(weights.synth_gpt_cond_weights(cfg, tag="golden/gptref"))
    uv = build_reference_unified_voice(mv, cfg, w)
    out = {}
    quiet = lambda: contextlib.redirect_stdout(io.StringIO())

    # ---- a8: conditioning (model_v2.py:627-671, 897-910).  One prompt as infer_v2.py:748-765 passes it ([1,T,1024], the
    #      "length" is shape[-1] = 1024, i.e. no padding), plus a ragged pair with true lengths for the mask path ----
    spk = torch.from_numpy(synth.uniform("golden/gptref/spk", (1, 23, 1024), 1.0))
    emo = torch.from_numpy(synth.uniform("golden/gptref/emo", (1, 19, 1024), 1.0))
    with torch.no_grad():
        ln_spk = torch.tensor([spk.shape[-1]])
        ln_emo = torch.tensor([emo.shape[-1]])
        lat = uv.get_conditioning(spk.transpose(1, 2), ln_spk)                 # [1, 6, d]
        out["cond_latent"] = lat.numpy()
        out["emovec_spk"] = uv.get_emovec(spk, ln_spk).numpy()
        out["emovec_emo"] = uv.get_emovec(emo, ln_emo).numpy()
        out["emovec_merged"] = uv.merge_emovec(spk, emo, ln_spk, ln_emo, alpha=0.6).numpy()
        enc, mask = uv.conditioning_encoder(spk, ln_spk)
        out["conformer_out"] = enc.numpy()
        pair = torch.from_numpy(synth.uniform("golden/gptref/pair", (2, 21, 1024), 1.0))
        plen = torch.tensor([21, 14])
        pair[1, 14:] = 0.0
        out["cond_latent_ragged"] = uv.get_conditioning(pair.transpose(1, 2), plen).numpy()
        out["emovec_ragged"] = uv.get_emovec(pair, plen).numpy()
    emovec = torch.from_numpy(out["emovec_merged"])

    # ---- a2: prompt layout, the reference's own prepare_gpt_inputs on ragged rows (model_v2.py:725-794) ----
    B, L, NEW = 3, 12, 10
    text = torch.from_numpy(synth.integers("golden/gptref/text", (B, L), 2, cfg.number_text_tokens))
    text[1, 9:] = cfg.stop_text_token
    text[2, 5:] = cfg.stop_text_token
    with torch.no_grad():
        zero = torch.zeros(B).long()
        conds = torch.cat((lat.expand(B, -1, -1) + emovec.expand(B, -1).unsqueeze(1),
                           uv.speed_emb(torch.ones_like(zero)).unsqueeze(1), uv.speed_emb(zero).unsqueeze(1)), 1)   # model_v2.py:830-834
        fake, inputs_embeds, attention_mask = uv.prepare_gpt_inputs(conds, text)
    out["text"] = text.numpy()
    out["conds"] = conds.numpy()
    out["prep_fake"] = fake.numpy()
    out["prep_embeds"] = inputs_embeds.numpy()
    out["prep_mask"] = attention_mask.numpy()

    # ---- a3: GPT2InferenceModel.forward (model_v2.py:131-225) driven the way generate drives it: prefill, then cached
    #      single-token steps through its own prepare_inputs_for_generation (101-129); greedy choice by hand (argmax of
    #      the HF repetition-penalty processor) so every step's logits are recorded ----
    from transformers.generation.logits_process import RepetitionPenaltyLogitsProcessor
    proc = RepetitionPenaltyLogitsProcessor(penalty=10.0)
    im = uv.inference_model
    im.store_mel_emb(inputs_embeds)
    ids, mask, past = fake.clone(), attention_mask.clone(), None
    logits_all = []
    unfinished = torch.ones(B, dtype=torch.long)
    with torch.no_grad():
        for step in range(NEW):
            mi = im.prepare_inputs_for_generation(ids, past_key_values=past, attention_mask=mask, use_cache=True)
            o = im(**mi, return_dict=True)
            past = o.past_key_values
            logits = o.logits[:, -1].float()
            logits_all.append(logits.clone())
            nxt = torch.argmax(proc(ids, logits.clone()), -1)
            nxt = nxt * unfinished + cfg.stop_mel_token * (1 - unfinished)
            ids = torch.cat([ids, nxt[:, None]], 1)
            mask = torch.cat([mask, torch.ones(B, 1, dtype=torch.long)], 1)
            unfinished = unfinished & (nxt != cfg.stop_mel_token).long()
    out["step_logits"] = torch.stack(logits_all, 1).numpy()
    out["step_codes"] = ids[:, fake.shape[1]:].numpy()

    # ---- a5/a6: the reference's inference_speech (model_v2.py:796-895) end to end: greedy, HF generate of this image ----
    def speech(**kw):
        with torch.no_grad(), quiet():
            codes, lat_ = uv.inference_speech(spk, text, emo, cond_lengths=ln_spk, emo_cond_lengths=ln_emo, emo_vec=emovec.expand(B, -1),
                                              num_return_sequences=1, max_generate_length=NEW, **kw)
        return codes, lat_
    try:
        # transformers 5.x `generate` pre-populates its default cache with one (empty) layer object per block, which is truthy;
        # the reference's `if past_key_values:` (model_v2.py:106), written against 4.52.1's empty-and-falsy DynamicCache(),
        # would then treat the prefill as a cached step.  Handing generate an empty DynamicCache() restores the 4.52.1 flow.
        from transformers.cache_utils import DynamicCache
        codes, lat_ = speech(do_sample=False, num_beams=1, repetition_penalty=10.0, past_key_values=DynamicCache())
        out["speech_greedy_codes"] = codes.numpy()
        out["speech_latent"] = lat_.numpy()
        print("inference_speech greedy:", codes.tolist())
    except Exception as e:      # recorded, not hidden: DESIGN section 2 quotes this line
        print("inference_speech(greedy) through container HF generate failed:", type(e).__name__, e)

    # ---- a7: latent pass, the reference's UnifiedVoice.forward (model_v2.py:673-723) ----
    B2, L2, M2 = 2, 7, 9
    text2 = torch.from_numpy(synth.integers("golden/gptref/text2", (B2, L2), 2, cfg.number_text_tokens))
    codes2 = torch.from_numpy(synth.integers("golden/gptref/codes2", (B2, M2), 0, cfg.start_mel_token))
    with torch.no_grad():
        latent = uv(lat.expand(B2, -1, -1), text2.clone(), torch.tensor([L2, L2]), codes2.clone(), torch.tensor([M2, M2]), emo,
                    cond_mel_lengths=ln_spk, emo_cond_mel_lengths=ln_emo, emo_vec=emovec.expand(B2, -1), use_speed=torch.zeros(B2).long())
    out["latent_text"] = text2.numpy()
    out["latent_codes"] = codes2.numpy()
    out["latent"] = latent.numpy()
    # ---- f2: beam search, the mode infer() really runs by default (infer_v2.py:714-722, 767): num_beams=3, do_sample=True,
    #      top_p .8, top_k 30, temperature .8, repetition_penalty 10, length_penalty 0 -- the reference's inference_speech
    #      driven by this image's HF generate, seeded; the Exp(1) draws torch.multinomial consumed are re-drawn with the same
    #      seed and stored.  mel_head's stop bias is raised so that some hypotheses finish (exercises BeamHypotheses) ----
    from oracle import gpt as og
    from transformers.cache_utils import DynamicCache
    NB, NEWB, SEED = 3, 24, 1234
    scorer_mod = _import_reference_beam_scorer()
    # cases a / b: the reference's default warpers (temperature .8, top_k 30, top_p .8) -- on this tiny model they leave so few
    # candidates that the sampled beams of b equal its deterministic ones.  Case c widens the nucleus (temperature 1.6, top_p .97)
    # so that beam-sample really samples (asserted: c's sampled codes differ from its deterministic codes) while hypotheses still
    # finish before the last step (asserted: a stop token before the final position).
    cases = (("a", 3.0, 0.8, 30, 0.8), ("b", 5.5, 0.8, 30, 0.8), ("c", 4.0, 1.6, 30, 0.97))
    for tag, stop_bias, temperature, top_k, top_p in cases:
        wb = dict(w)
        wb["mel_head.bias"] = w["mel_head.bias"].copy()
        wb["mel_head.bias"][cfg.stop_mel_token] += stop_bias
        uvb = build_reference_unified_voice(mv, cfg, wb)
        twb = {k: torch.from_numpy(v) for k, v in wb.items()}
        for mode, do_sample in (("det", False), ("sample", True)):
            torch.manual_seed(SEED)
            with torch.no_grad(), quiet():
                codes, _ = uvb.inference_speech(spk, text, emo, cond_lengths=ln_spk, emo_cond_lengths=ln_emo, emo_vec=emovec.expand(B, -1),
                                                num_return_sequences=1, max_generate_length=NEWB, do_sample=do_sample, top_p=top_p, top_k=top_k,
                                                temperature=temperature, length_penalty=0.0, num_beams=NB, repetition_penalty=10.0,
                                                past_key_values=DynamicCache())
            # the same decode through the vendored loop + the reference's own BeamSearchScorer: THE expected value.  Where the
            # hypotheses finish early, transformers 5.x's rewritten beam search (no BeamScorer any more) stops by a different
            # rule than the vendored 4.x scorer; everywhere else the two agree, which checks the hand-driven loop.
            torch.manual_seed(SEED)
            vend = drive_vendored_beam_search(uvb, scorer_mod, cfg, conds, text, NEWB, NB, do_sample, temperature, top_k, top_p, 10.0, 0.0)
            same = vend.shape == codes.shape and bool((vend == codes).all())
            print(f"beam {tag} {mode}: vendored-scorer drive {'==' if same else '!='} transformers-5.x generate;", vend.tolist())
            out[f"beam_{tag}_{mode}_codes"] = vend.numpy()
            out[f"beam_{tag}_{mode}_hf5_agrees"] = np.array(same)
            torch.manual_seed(SEED)
            noise = torch.stack([torch.empty(B, NB * cfg.number_mel_codes).exponential_(1) for _ in range(NEWB)])
            if do_sample:
                out[f"beam_{tag}_noise"] = noise.numpy()
            chk = og.generate_beam(twb, cfg, conds, text, NEWB, noise, num_beams=NB, do_sample=do_sample, temperature=temperature, top_k=top_k,
                                   top_p=top_p)
            assert chk.shape == vend.shape and bool((chk == vend).all()), "beam-search restatement != vendored beam search on the reference model"
        out[f"beam_{tag}_stop_bias"] = np.array(stop_bias, dtype=np.float32)
        out[f"beam_{tag}_warpers"] = np.array([temperature, top_k, top_p], dtype=np.float64)
    cs, cd = out["beam_c_sample_codes"], out["beam_c_det_codes"]
    assert cs.shape != cd.shape or not np.array_equal(cs, cd), "case c must really sample"
    early = [(row == cfg.stop_mel_token).any() and int(np.argmax(row == cfg.stop_mel_token)) < row.shape[0] - 1 for row in cs]
    assert any(early), "case c must have a hypothesis that finishes early under sampling"
    print("beam c: sampled != deterministic, rows finishing early under sampling:", early)
    np.savez_compressed(os.path.join(HERE, "gpt_ref.npz"), **out)
    print("wrote gpt_ref.npz", {k: v.shape for k, v in out.items()})
    return uv, cfg, w, out


def make_segments():
    """Segment-splitter fixtures from the reference's TextTokenizer.split_segments_by_token (front.py:345-422): random token
    streams over a small SentencePiece-like vocabulary (words, commas, dashes, apostrophes, sentence punctuation)."""
    import json
    import random
    import warnings
    from indextts.utils.front import TextTokenizer
    vocab = ["▁the", "▁a", "ing", "▁in", "dex", "▁tts", "s", "▁to", "▁of", "▁mi", "▁wave", "front", "▁x", "y", "z", "▁go", "od",
             ",", "▁,", "-", "'", "▁'", ".", "!", "?", "▁.", "▁?", "▁..."]
    weights_ = [6] * 17 + [3, 2, 2, 2, 1, 4, 2, 2, 2, 1, 1]
    rng = random.Random(20240611)
    cases = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for n in [0, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 200] * 6:
            toks = rng.choices(vocab, weights=weights_, k=n)
            for limit, quick in ((120, 0), (20, 0), (8, 0), (30, 25), (12, 40)):
                out = TextTokenizer.split_segments_by_token(list(toks), TextTokenizer.punctuation_marks_tokens, limit, quick)
                cases.append({"tokens": toks, "limit": limit, "quick": quick, "segments": out})
    path = os.path.join(HERE, "segments.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump({"punctuation": TextTokenizer.punctuation_marks_tokens, "cases": cases}, f, ensure_ascii=False)
    print("wrote segments.json", len(cases), "cases")


def make_w2vbert():
    """Semantic-feature fixtures from the container's own `transformers.Wav2Vec2BertModel` (the class the reference instantiates,
    utils/maskgct_utils.py:88) carrying the synthetic weights: `hidden_states[n]` exactly as `IndexTTS2.get_emb` reads it
    (infer_v2.py:399-406), a ragged batch with an attention mask and a single unpadded row, at the tiny sizes."""
    from transformers import Wav2Vec2BertConfig, Wav2Vec2BertModel
    from indextts_amd import synth, weights
    from indextts_amd.config import W2VBertConfig
    cfg = W2VBertConfig.tiny()
    w = weights.synth_w2vbert_weights(cfg, tag="golden/w2vbert")
    hc = Wav2Vec2BertConfig(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_layers + 1, num_attention_heads=cfg.num_heads,
                            intermediate_size=cfg.intermediate_size, feature_projection_input_dim=cfg.input_dim,
                            left_max_position_embeddings=cfg.left_max, right_max_position_embeddings=cfg.right_max,
                            conv_depthwise_kernel_size=cfg.conv_kernel, layer_norm_eps=cfg.layer_norm_eps, apply_spec_augment=False,
                            hidden_dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, feat_proj_dropout=0.0, layerdrop=0.0,
                            conformer_conv_dropout=0.0)
    model = Wav2Vec2BertModel(hc).eval()
    sd = {k: torch.from_numpy(v) for k, v in w.items() if k not in ("semantic_mean", "semantic_std")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(m.startswith(f"encoder.layers.{cfg.num_layers}.") or m == "masked_spec_embed" for m in missing), missing
    B, T = 3, 37
    feats = torch.from_numpy(synth.uniform("golden/w2vbert/feats", (B, T, cfg.input_dim), 1.5))
    lens = torch.tensor([37, 22, 30])
    mask = (torch.arange(T)[None, :] < lens[:, None]).long()
    mean, std = torch.from_numpy(w["semantic_mean"]), torch.from_numpy(w["semantic_std"])
    with torch.no_grad():
        hs = model(input_features=feats, attention_mask=mask, output_hidden_states=True).hidden_states
        assert len(hs) == cfg.num_layers + 2
        ragged = (hs[cfg.num_layers] - mean) / std
        solo = (model(input_features=feats[1:2, :22], output_hidden_states=True).hidden_states[cfg.num_layers] - mean) / std
        proj = hs[0]
        layer0 = hs[1]
    out = {"feats": feats.numpy(), "lens": lens.numpy().astype(np.int32), "mask": mask.numpy().astype(np.int32), "emb_ragged": ragged.numpy(),
           "emb_row1_alone": solo.numpy(), "hidden0": proj.numpy(), "hidden1": layer0.numpy()}
    # the ragged batch's valid frames must equal the unpadded run (what "padding never reaches a valid frame" means)
    assert np.abs(out["emb_ragged"][1, :22] - out["emb_row1_alone"][0]).max() < 1e-4
    np.savez_compressed(os.path.join(HERE, "w2vbert.npz"), **out)
    print("w2vbert.npz", {k: v.shape for k, v in out.items()})


def make_repcodec():
    """Semantic-codec fixtures from the reference's own RepCodec class (kmeans/repcodec_model.py) carrying the synthetic weights:
    `quantize(x)` as the prompt block calls it (infer_v2.py:637), plus the encoder output and the projected latents so a test can
    tell a near-tie of the nearest-code search from an error."""
    _install_placeholders()
    ff = types.ModuleType("torchaudio.functional.functional")
    ff._hz_to_mel = ff._mel_to_hz = None
    sys.modules["torchaudio.functional"] = types.ModuleType("torchaudio.functional")
    sys.modules["torchaudio.functional.functional"] = ff
    from indextts.utils.maskgct.models.codec.kmeans.repcodec_model import RepCodec
    from indextts_amd import synth, weights
    from indextts_amd.config import RepCodecConfig
    cfg = RepCodecConfig.tiny()
    w = weights.synth_repcodec_weights(cfg, tag="golden/repcodec")
    m = RepCodec(codebook_size=cfg.codebook_size, hidden_size=cfg.hidden_size, codebook_dim=cfg.codebook_dim, vocos_dim=cfg.vocos_dim,
                 vocos_intermediate_dim=cfg.vocos_intermediate_dim, vocos_num_layers=cfg.vocos_num_layers).eval()
    sd = m.state_dict()
    loaded = {}
    for k, v in w.items():
        if k.endswith("_project.weight"):          # weight-norm pair: g = ||v|| makes the effective weight v itself
            t = torch.from_numpy(v)
            loaded[k[:-len("weight")] + "weight_v"] = t
            loaded[k[:-len("weight")] + "weight_g"] = t.reshape(t.shape[0], -1).norm(dim=1).reshape(-1, 1, 1)
        else:
            loaded[k] = torch.from_numpy(v)
    assert set(loaded) <= set(sd), sorted(set(loaded) - set(sd))
    assert all(k.startswith("decoder.") for k in set(sd) - set(loaded)), sorted(k for k in set(sd) - set(loaded) if not k.startswith("decoder."))
    m.load_state_dict(loaded, strict=False)
    B, T = 2, 23
    x = torch.from_numpy(synth.uniform("golden/repcodec/x", (B, T, cfg.hidden_size), 1.0))
    with torch.no_grad():
        idx, q = m.quantize(x)
        enc = m.encoder(x.transpose(1, 2))                                     # [B, T, hidden]
        ze = m.quantizer.quantizers[0].in_project(enc.transpose(1, 2))         # [B, d, T]
        x1 = x[:1]
        idx1, q1 = m.quantize(x1)
    assert idx.shape == (B, T) and q.shape == (B, T, cfg.hidden_size) and idx1.shape == (1, T)
    out = {"x": x.numpy(), "indices": idx.numpy(), "quantized": q.numpy(), "encoded": enc.numpy(), "z_e": ze.transpose(1, 2).numpy(),
           "indices_b1": idx1.numpy(), "quantized_b1": q1.numpy()}
    np.savez_compressed(os.path.join(HERE, "repcodec.npz"), **out)
    print("repcodec.npz", {k: v.shape for k, v in out.items()})


def make_melspec():
    """Prompt log-mel fixtures from the reference's own `mel_spectrogram` (s2mel/modules/audio.py:45-83) with infer_v2.py's arguments.
    `librosa.filters.mel`, which that module imports at load time, is absent in this image: it is supplied by
    transformers.audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney") (a third-party port of the librosa function)."""
    from transformers.audio_utils import mel_filter_bank
    _install_placeholders()

    def librosa_mel(sr, n_fft, n_mels, fmin, fmax):
        fmax = sr / 2 if fmax is None else fmax
        return mel_filter_bank(num_frequency_bins=n_fft // 2 + 1, num_mel_filters=n_mels, min_frequency=fmin, max_frequency=fmax,
                               sampling_rate=sr, norm="slaney", mel_scale="slaney").T.astype(np.float32)

    sys.modules["librosa.filters"].mel = librosa_mel
    sys.modules["librosa"].filters = sys.modules["librosa.filters"]
    from indextts.s2mel.modules.audio import mel_spectrogram
    from indextts_amd import synth
    n = 22050 + 333
    t = np.arange(n) / 22050.0
    y = np.stack([0.4 * np.sin(2 * np.pi * 310 * t) + 0.2 * np.sin(2 * np.pi * 2400 * t + 1.0) + 0.05 * synth.uniform("golden/mel/a", (n,), 1.0),
                  0.3 * np.sin(2 * np.pi * (150 + 800 * t) * t) + 0.02 * synth.uniform("golden/mel/b", (n,), 1.0)]).astype(np.float32)
    with torch.no_grad():
        mel = mel_spectrogram(torch.from_numpy(y), n_fft=1024, num_mels=80, sampling_rate=22050, hop_size=256, win_size=1024, fmin=0, fmax=None,
                              center=False)
    out = {"audio": y, "mel": mel.numpy(), "mel_basis": librosa_mel(22050, 1024, 80, 0, None)}
    np.savez_compressed(os.path.join(HERE, "melspec.npz"), **out)
    print("melspec.npz", {k: v.shape for k, v in out.items()})


def make_campplus():
    """Style-vector fixtures from the reference's own CAMPPlus class (s2mel/modules/campplus/DTDNN.py) in eval mode carrying the
    synthetic weights, at the real configuration (feat_dim 80, embedding 192): a 2.3 s and a 0.6 s feature sequence (three 100-frame
    context segments, the last one partial; and a single partial one), plus the FCM head's output for the shorter one."""
    _install_placeholders()
    from indextts.s2mel.modules.campplus.DTDNN import CAMPPlus
    from indextts_amd import synth, weights
    from indextts_amd.config import CamPPlusConfig
    cfg = CamPPlusConfig()
    w = weights.synth_campplus_weights(cfg, tag="golden/campplus")
    m = CAMPPlus(feat_dim=cfg.feat_dim, embedding_size=cfg.embedding_size).eval()
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    out = {}
    with torch.no_grad():
        for tag, T in (("a", 461), ("b", 61)):
            feat = torch.from_numpy(synth.uniform(f"golden/campplus/feat_{tag}", (1, T, cfg.feat_dim), 2.0))
            feat = feat - feat.mean(dim=1, keepdim=True)                       # infer_v2.py:646
            out[f"feat_{tag}"] = feat.numpy()
            out[f"style_{tag}"] = m(feat).numpy()
        out["fcm_b"] = m.head(torch.from_numpy(out["feat_b"]).permute(0, 2, 1)).numpy()
    np.savez_compressed(os.path.join(HERE, "campplus.npz"), **out)
    print("campplus.npz", {k: v.shape for k, v in out.items()})


def make_tokenizer():
    """Tokenizer fixtures from the reference's own TextTokenizer (indextts/utils/front.py:231-343, normalizer=None: WeTextProcessing is
    absent) over a small SentencePiece BPE model trained here on a synthetic corpus and committed next to the fixture
    (tests/golden/tiny_bpe.model): tokens, ids, decoded strings and the segments of `split_segments`."""
    import io
    import json
    import sentencepiece as spm
    _install_placeholders()
    from indextts.utils.front import TextTokenizer
    corpus = ["the quick brown fox jumps over the lazy dog.", "hello world, this is a test of the tokenizer!", "你好世界, 这是一个测试。",
              "今天天气很好, we are going to the park.", "what's the matter? it's nothing - really.", "一二三四五六七八九十, 百千万。",
              "speech synthesis with large language models is fun.", "语音合成 is what we do here, 每天都是。"] * 40
    model = io.BytesIO()
    spm.SentencePieceTrainer.train(sentence_iterator=iter([tokenize_up(c) for c in corpus]), model_writer=model, vocab_size=180, model_type="bpe",
                                   character_coverage=1.0, bos_id=0, eos_id=1, unk_id=2, pad_id=-1, user_defined_symbols=[])
    path = os.path.join(HERE, "tiny_bpe.model")
    with open(path, "wb") as f:
        f.write(model.getvalue())
    tok = TextTokenizer(path, normalizer=None)
    texts = ["Hello world, this is a test!", "你好世界是 hello world 的中文", "What's the matter? It's nothing - really. 今天天气很好。", "a", " ", "",
             "the quick brown fox, the lazy dog. speech synthesis is fun! 一二三, 四五六。 what's next? nothing."]
    cases = []
    for t in texts:
        tokens = tok.tokenize(t)
        ids = tok.convert_tokens_to_ids(tokens)
        cases.append({"text": t, "tokens": tokens, "ids": ids, "encode": tok.encode(t), "decoded": tok.decode(ids) if ids else "",
                      "decoded_lower": tok.decode(ids, do_lower_case=True) if ids else "",
                      "segments_8": tok.split_segments(tokens, 8), "segments_20_q4": tok.split_segments(tokens, 20, quick_streaming_tokens=4)})
    meta = {"vocab_size": tok.vocab_size, "unk_token_id": tok.unk_token_id, "batch_encode": tok.batch_encode(texts[:3]),
            "vocab_head": [tok.convert_ids_to_tokens(i) for i in range(12)]}
    with open(os.path.join(HERE, "tokenizer.json"), "w", encoding="utf-8") as f:
        json.dump({"cases": cases, "meta": meta}, f, ensure_ascii=False, indent=1)
    print("tokenizer.json", len(cases), "cases; vocab", tok.vocab_size)


def tokenize_up(line):
    from indextts.utils.common import tokenize_by_CJK_char
    return tokenize_by_CJK_char(line)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    torch.manual_seed(0)
    if which in ("gpt", "all"):          # before the placeholders: transformers probes for torchaudio
        make_gpt()
    if which in ("gpt_ref", "all"):
        make_gpt_ref()
    if which in ("w2vbert", "all"):
        make_w2vbert()
    if which in ("vocoder", "s2mel", "all"):
        Munch = _install_placeholders()
        if which in ("vocoder", "all"):
            make_vocoder()
        if which in ("s2mel", "all") and "make_s2mel" in globals():
            globals()["make_s2mel"](Munch)
    if which in ("segments", "all"):
        _install_placeholders()
        make_segments()
    if which in ("repcodec", "all"):
        make_repcodec()
    if which in ("melspec", "all"):
        make_melspec()
    if which in ("campplus", "all"):
        make_campplus()
    if which in ("tokenizer", "all"):
        _install_placeholders()
        make_tokenizer()
