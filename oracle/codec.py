"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain torch functions over a dict of tensors) of the semantic codec's `quantize`.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Follows  RepCodec.quantize             /root/reference/indextts/utils/maskgct/models/codec/kmeans/repcodec_model.py:179-199
         VocosBackbone, ConvNeXtBlock  /root/reference/indextts/utils/maskgct/models/codec/kmeans/vocos.py:719-782, 468-526
         ResidualVQ.forward            /root/reference/indextts/utils/maskgct/models/codec/amphion_codec/quantize/residual_vq.py:68-140
         FactorizedVectorQuantize      .../amphion_codec/quantize/factorized_vector_quantize.py:66-119
Pinned by tests/golden/repcodec.npz, produced by the reference's own RepCodec class on the synthetic weights
(tests/golden/make_golden.py::make_repcodec)."""
import torch
import torch.nn.functional as F


def _ln(x, w, p):
    return F.layer_norm(x, (x.shape[-1],), w[p + ".weight"], w[p + ".bias"], 1e-6)


def encoder(w, x):
    """x [B,T,hidden] -> [B,T,hidden]: `self.encoder(x.transpose(1, 2)).transpose(1, 2)` BEFORE the final transpose, i.e. token-major."""
    h = F.conv1d(x.transpose(1, 2), w["encoder.0.embed.weight"], w["encoder.0.embed.bias"], padding=3)          # vocos.py:772
    h = _ln(h.transpose(1, 2), w, "encoder.0.norm").transpose(1, 2)                                             # 777-778
    i = 0
    while f"encoder.0.convnext.{i}.gamma" in w:                                                                 # ConvNeXtBlock.forward 507-526
        p = f"encoder.0.convnext.{i}"
        r = h
        y = F.conv1d(h, w[p + ".dwconv.weight"], w[p + ".dwconv.bias"], padding=3, groups=h.shape[1]).transpose(1, 2)
        y = _ln(y, w, p + ".norm")
        y = F.linear(F.gelu(F.linear(y, w[p + ".pwconv1.weight"], w[p + ".pwconv1.bias"])), w[p + ".pwconv2.weight"], w[p + ".pwconv2.bias"])
        h = r + (w[p + ".gamma"] * y).transpose(1, 2)
        i += 1
    h = _ln(h.transpose(1, 2), w, "encoder.0.final_layer_norm")                                                 # 781
    return F.linear(h, w["encoder.1.weight"], w["encoder.1.bias"])


def quantize(w, x):
    """-> (indices [B,T] int64, quantized [B,T,hidden])"""
    q = "quantizer.quantizers.0"
    z = encoder(w, x).transpose(1, 2)                                       # [B, hidden, T]
    z_e = F.conv1d(z, w[q + ".in_project.weight"], w[q + ".in_project.bias"])
    B, d, T = z_e.shape
    enc = F.normalize(z_e.transpose(1, 2).reshape(B * T, d))                # decode_latents
    cb = F.normalize(w[q + ".codebook.weight"])
    dist = enc.pow(2).sum(1, keepdim=True) - 2 * enc @ cb.t() + cb.pow(2).sum(1, keepdim=True).t()
    idx = (-dist).max(1)[1].reshape(B, T)
    z_q = F.embedding(idx, w[q + ".codebook.weight"]).transpose(1, 2)
    z_q = z_e + (z_q - z_e)
    out = F.conv1d(z_q, w[q + ".out_project.weight"], w[q + ".out_project.bias"])
    return idx, out.transpose(1, 2)
