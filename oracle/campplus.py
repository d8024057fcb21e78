"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain torch functions over a dict of tensors) of the CAMPPlus speaker encoder in
inference mode.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Follows  CAMPPlus, FCM            /root/reference/indextts/s2mel/modules/campplus/DTDNN.py:25-140
         layers                   /root/reference/indextts/s2mel/modules/campplus/layers.py:23-259
Pinned by tests/golden/campplus.npz, produced by the reference's own CAMPPlus class on the synthetic weights
(tests/golden/make_golden.py::make_campplus)."""
import torch
import torch.nn.functional as F


def _bn(x, w, p, affine=True):
    return F.batch_norm(x, w[p + ".running_mean"], w[p + ".running_var"], w.get(p + ".weight") if affine else None,
                        w.get(p + ".bias") if affine else None, False, 0.0, 1e-5)


def _resblock(x, w, p, stride):                                    # BasicResBlock.forward, layers.py:252-259
    out = F.relu(_bn(F.conv2d(x, w[p + ".conv1.weight"], stride=(stride, 1), padding=1), w, p + ".bn1"))
    out = _bn(F.conv2d(out, w[p + ".conv2.weight"], padding=1), w, p + ".bn2")
    sc = x
    if p + ".shortcut.0.weight" in w:
        sc = _bn(F.conv2d(x, w[p + ".shortcut.0.weight"], stride=(stride, 1)), w, p + ".shortcut.1")
    return F.relu(out + sc)


def fcm(x, w):                                                     # FCM.forward, DTDNN.py:49-59; x [B, F, T]
    out = F.relu(_bn(F.conv2d(x.unsqueeze(1), w["head.conv1.weight"], padding=1), w, "head.bn1"))
    for l in (1, 2):
        out = _resblock(out, w, f"head.layer{l}.0", 2)
        out = _resblock(out, w, f"head.layer{l}.1", 1)
    out = F.relu(_bn(F.conv2d(out, w["head.conv2.weight"], stride=(2, 1), padding=1), w, "head.bn2"))
    return out.reshape(out.shape[0], out.shape[1] * out.shape[2], out.shape[3])


def _seg_pooling(x, seg_len=100):                                  # CAMLayer.seg_pooling, layers.py:102-113
    seg = F.avg_pool1d(x, kernel_size=seg_len, stride=seg_len, ceil_mode=True)
    shape = seg.shape
    seg = seg.unsqueeze(-1).expand(*shape, seg_len).reshape(*shape[:-1], -1)
    return seg[..., : x.shape[-1]]


def _cam_dense_layer(x, w, p, dil):                                # CAMDenseTDNNLayer.forward + CAMLayer.forward
    h = F.conv1d(F.relu(_bn(x, w, p + ".nonlinear1.batchnorm")), w[p + ".linear1.weight"])
    h = F.relu(_bn(h, w, p + ".nonlinear2.batchnorm"))
    y = F.conv1d(h, w[p + ".cam_layer.linear_local.weight"], padding=dil, dilation=dil)
    ctx = h.mean(-1, keepdim=True) + _seg_pooling(h)
    ctx = F.relu(F.conv1d(ctx, w[p + ".cam_layer.linear1.weight"], w[p + ".cam_layer.linear1.bias"]))
    return y * torch.sigmoid(F.conv1d(ctx, w[p + ".cam_layer.linear2.weight"], w[p + ".cam_layer.linear2.bias"]))


def forward(w, cfg, feat):
    """feat [B, T, feat_dim] -> [B, embedding_size]"""
    x = fcm(feat.permute(0, 2, 1), w)
    x = F.relu(_bn(F.conv1d(x, w["xvector.tdnn.linear.weight"], stride=2, padding=2), w, "xvector.tdnn.nonlinear.batchnorm"))
    for bi, (n_layers, dil) in enumerate(zip(cfg.block_layers, cfg.block_dilation)):
        for i in range(n_layers):
            x = torch.cat([x, _cam_dense_layer(x, w, f"xvector.block{bi + 1}.tdnnd{i + 1}", dil)], dim=1)
        x = F.conv1d(F.relu(_bn(x, w, f"xvector.transit{bi + 1}.nonlinear.batchnorm")), w[f"xvector.transit{bi + 1}.linear.weight"])
    x = F.relu(_bn(x, w, "xvector.out_nonlinear.batchnorm"))
    stats = torch.cat([x.mean(dim=-1), x.std(dim=-1, unbiased=True)], dim=-1)                  # StatsPool
    out = F.conv1d(stats.unsqueeze(-1), w["xvector.dense.linear.weight"]).squeeze(-1)
    return _bn(out, w, "xvector.dense.nonlinear.batchnorm", affine=False)
