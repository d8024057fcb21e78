"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain torch functions over a dict of tensors) of the semantic feature model of
the reference's prompt block.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Follows  IndexTTS2.get_emb                       /root/reference/indextts/infer_v2.py:381-408
         build_semantic_model                    /root/reference/indextts/utils/maskgct_utils.py:87-93
and, for the model itself -- a third-party dependency absent from /root/reference (transformers, pinned 4.52.1 in the reference's
pyproject; the container has 5.x, whose modeling_wav2vec2_bert.py has the same forward) -- the published algorithm of
Wav2Vec2BertModel: feature_projection, Wav2Vec2BertEncoder / EncoderLayer / SelfAttention (relative_key) / ConvolutionModule /
FeedForward.  Pinned by tests/golden/w2vbert.npz, produced by the container's own Wav2Vec2BertModel on the synthetic weights
(tests/golden/make_golden.py::make_w2vbert)."""
import math

import torch
import torch.nn.functional as F


def _ln(x, w, p, eps):
    return F.layer_norm(x, (x.shape[-1],), w[p + ".weight"], w[p + ".bias"], eps)


def _ffn(x, w, p):
    # Wav2Vec2BertFeedForward: intermediate_dense -> swish -> output_dense
    h = F.silu(F.linear(x, w[p + ".intermediate_dense.weight"], w[p + ".intermediate_dense.bias"]))
    return F.linear(h, w[p + ".output_dense.weight"], w[p + ".output_dense.bias"])


def _self_attn(x, w, p, cfg, key_mask):
    # Wav2Vec2BertSelfAttention.forward, position_embeddings_type == "relative_key"
    B, T, D = x.shape
    H, dh = cfg.num_heads, D // cfg.num_heads
    q = F.linear(x, w[p + ".linear_q.weight"], w[p + ".linear_q.bias"]).view(B, T, H, dh).transpose(1, 2)
    k = F.linear(x, w[p + ".linear_k.weight"], w[p + ".linear_k.bias"]).view(B, T, H, dh).transpose(1, 2)
    v = F.linear(x, w[p + ".linear_v.weight"], w[p + ".linear_v.bias"]).view(B, T, H, dh).transpose(1, 2)
    scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dh)
    pos = torch.arange(T)
    dist = torch.clamp(pos[None, :] - pos[:, None], -cfg.left_max, cfg.right_max) + cfg.left_max
    pe = w[p + ".distance_embedding.weight"][dist]                         # [T, T, dh]
    scores = scores + torch.einsum("bhld,lrd->bhlr", q, pe) / math.sqrt(dh)
    if key_mask is not None:                                               # additive finfo.min at padded keys
        scores = scores + (~key_mask)[:, None, None, :].to(scores.dtype) * torch.finfo(scores.dtype).min
    o = torch.matmul(torch.softmax(scores, dim=-1), v).transpose(1, 2).reshape(B, T, D)
    return F.linear(o, w[p + ".linear_out.weight"], w[p + ".linear_out.bias"])


def _conv_module(x, w, p, cfg, key_mask):
    # Wav2Vec2BertConvolutionModule.forward
    h = _ln(x, w, p + ".layer_norm", cfg.layer_norm_eps)
    if key_mask is not None:
        h = h.masked_fill(~key_mask.unsqueeze(-1), 0.0)
    h = h.transpose(1, 2)
    h = F.glu(F.conv1d(h, w[p + ".pointwise_conv1.weight"]), dim=1)
    h = F.pad(h, (cfg.conv_kernel - 1, 0))                                 # causal: all padding on the left
    h = F.conv1d(h, w[p + ".depthwise_conv.weight"], groups=h.shape[1])
    h = _ln(h.transpose(1, 2), w, p + ".depthwise_layer_norm", cfg.layer_norm_eps).transpose(1, 2)
    h = F.conv1d(F.silu(h), w[p + ".pointwise_conv2.weight"])
    return h.transpose(1, 2)


def encoder_layer(x, w, p, cfg, key_mask):
    # Wav2Vec2BertEncoderLayer.forward
    eps = cfg.layer_norm_eps
    x = _ffn(_ln(x, w, p + ".ffn1_layer_norm", eps), w, p + ".ffn1") * 0.5 + x
    x = _self_attn(_ln(x, w, p + ".self_attn_layer_norm", eps), w, p + ".self_attn", cfg, key_mask) + x
    x = x + _conv_module(x, w, p + ".conv_module", cfg, key_mask)
    x = _ffn(_ln(x, w, p + ".ffn2_layer_norm", eps), w, p + ".ffn2") * 0.5 + x
    return _ln(x, w, p + ".final_layer_norm", eps)


def hidden_state(w, cfg, input_features, attention_mask=None):
    """Wav2Vec2BertModel(...).hidden_states[cfg.num_layers]: feature_projection, then the first cfg.num_layers layers."""
    key_mask = None if attention_mask is None else attention_mask.bool()
    x = _ln(input_features, w, "feature_projection.layer_norm", cfg.layer_norm_eps)
    x = F.linear(x, w["feature_projection.projection.weight"], w["feature_projection.projection.bias"])
    if key_mask is not None:
        x = x.masked_fill(~key_mask.unsqueeze(-1), 0.0)                    # Wav2Vec2BertEncoder.forward
    for i in range(cfg.num_layers):
        x = encoder_layer(x, w, f"encoder.layers.{i}", cfg, key_mask)
    return x


def get_emb(w, cfg, input_features, attention_mask=None):
    """infer_v2.py:402-408: (hidden_states[17] - semantic_mean) / semantic_std."""
    feat = hidden_state(w, cfg, input_features, attention_mask)
    if "semantic_mean" in w:
        feat = (feat - w["semantic_mean"]) / w["semantic_std"]
    return feat
