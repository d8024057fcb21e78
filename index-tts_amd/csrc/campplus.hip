// CAMPPlus speaker encoder on MI355X: the global style vector of the prompt block
// (`style = self.campplus_model(feat.unsqueeze(0))`, infer_v2.py:251-257, 641-647).
//
// Reference: CAMPPlus, FCM                 indextts/s2mel/modules/campplus/DTDNN.py:25-140
//            BasicResBlock, TDNNLayer, CAMDenseTDNNBlock / Layer, CAMLayer (context mask = global mean + 100-frame segment
//            mean), TransitLayer, StatsPool, DenseLayer     indextts/s2mel/modules/campplus/layers.py:23-259
//
// Inference-mode BatchNorms are folded at load: into the convolution that precedes them where there is one, else kept as a per-channel
// scale / shift applied with the ReLU that follows.  The 2-D head works channels-first on [C][80 / s][T] planes with a direct 3x3 kernel
// (32 channels: nothing for the MFMA to do); from the TDNN layer on everything is token-major [T][channels] in ONE growing buffer -- a
// dense layer reads the first c_in columns and appends its 32 output columns in place, which is the reference's torch.cat -- with the
// 1x1 / k3 / k5 convolutions on the exact-fp32 MFMA GEMM.  Runs once per prompt.
#include <cmath>

#include "campplus.h"
#include "model_util.h"

namespace idxtts {

namespace {

constexpr float BN_EPS = 1e-5f;

struct BN { std::vector<float> s, t; };

int bn_from(std::map<std::string, HostTensor>& t, const std::string& p, int n, bool affine, BN* out) {
  HostTensor *g = nullptr, *b = nullptr, *rm = nullptr, *rv = nullptr;
  if (need(t, p + ".running_mean", {n}, &rm) || need(t, p + ".running_var", {n}, &rv)) return 1;
  if (affine && (need(t, p + ".weight", {n}, &g) || need(t, p + ".bias", {n}, &b))) return 1;
  out->s.resize(n); out->t.resize(n);
  for (int i = 0; i < n; ++i) {
    const float s = (affine ? g->data[i] : 1.0f) / std::sqrt(rv->data[i] + BN_EPS);
    out->s[i] = s;
    out->t[i] = (affine ? b->data[i] : 0.0f) - rm->data[i] * s;
  }
  return 0;
}

// Conv2d (no bias) followed by BatchNorm2d -> weights scaled per output channel + bias
int conv2d_bn(std::map<std::string, HostTensor>& t, DeviceArena& arena, const std::string& conv, const std::string& bn, int cout, int cin, int k,
              int stride_h, Conv2dW* out) {
  HostTensor* w = nullptr;
  BN b;
  if (need(t, conv + ".weight", {cout, cin, k, k}, &w) || bn_from(t, bn, cout, true, &b)) return 1;
  std::vector<float> ws(w->data.size());
  const size_t per = (size_t)cin * k * k;
  for (int co = 0; co < cout; ++co)
    for (size_t i = 0; i < per; ++i) ws[co * per + i] = w->data[co * per + i] * b.s[co];
  out->cin = cin; out->cout = cout; out->k = k; out->stride_h = stride_h;
  return up(arena, ws, &out->w) || up(arena, b.t, &out->b);
}

// Conv1d weight [N][Cin][k] (+ optional per-output scale / shift of a following BatchNorm) -> rows [N][tap * Cin + ci]
int conv1d_linear(DeviceArena& arena, const HostTensor& w, int N, int Cin, int k, const BN* bn, const float* bias, LinearWeights* out) {
  std::vector<float> r((size_t)N * k * Cin), b(N, 0.0f);
  for (int n = 0; n < N; ++n) {
    const float s = bn ? bn->s[n] : 1.0f;
    for (int ci = 0; ci < Cin; ++ci)
      for (int kk = 0; kk < k; ++kk) r[((size_t)n * k + kk) * Cin + ci] = w.data[((size_t)n * Cin + ci) * k + kk] * s;
    b[n] = (bias ? bias[n] * s : 0.0f) + (bn ? bn->t[n] : 0.0f);
  }
  return make_linear(arena, r.data(), (bn || bias) ? b.data() : nullptr, N, k * Cin, k * Cin, out);
}

}  // namespace

int CamPPlusModel::finalize(std::map<std::string, HostTensor>& t, DeviceArena& arena) {
  const int mc = cfg.m_channels, F = cfg.feat_dim, G = cfg.growth_rate, BNC = cfg.bn_size * cfg.growth_rate, IC = cfg.init_channels;
  IDX_CHECK(mc > 0 && F > 0 && F % 8 == 0 && G > 0 && (G & 3) == 0 && (BNC & 31) == 0 && (IC & 31) == 0 && cfg.embedding_size > 0, "CAMPPlus shape");
  IDX_CHECK(cfg.num_blocks >= 1 && cfg.num_blocks <= 4, "1..4 dense blocks");
  // ---- FCM head ----
  if (conv2d_bn(t, arena, "head.conv1", "head.bn1", mc, 1, 3, 1, &conv1)) return 1;
  for (int l = 1; l <= 2; ++l)
    for (int j = 0; j < 2; ++j) {
      FcmResBlock rb;
      const std::string p = "head.layer" + std::to_string(l) + "." + std::to_string(j);
      const int stride = j == 0 ? 2 : 1;
      if (conv2d_bn(t, arena, p + ".conv1", p + ".bn1", mc, mc, 3, stride, &rb.c1) || conv2d_bn(t, arena, p + ".conv2", p + ".bn2", mc, mc, 3, 1, &rb.c2)) return 1;
      rb.has_sc = stride != 1;
      if (rb.has_sc && conv2d_bn(t, arena, p + ".shortcut.0", p + ".shortcut.1", mc, mc, 1, stride, &rb.sc)) return 1;
      res.push_back(rb);
    }
  if (conv2d_bn(t, arena, "head.conv2", "head.bn2", mc, mc, 3, 2, &conv2)) return 1;
  fcm_out = mc * (F / 8);
  IDX_CHECK((fcm_out & 31) == 0, "FCM output channels must be a multiple of 32");
  // ---- TDNN layer ----
  {
    HostTensor* w = nullptr;
    BN b;
    if (need(t, "xvector.tdnn.linear.weight", {IC, fcm_out, 5}, &w) || bn_from(t, "xvector.tdnn.nonlinear.batchnorm", IC, true, &b)) return 1;
    if (conv1d_linear(arena, *w, IC, fcm_out, 5, &b, nullptr, &tdnn)) return 1;
  }
  // ---- dense blocks ----
  int channels = IC;
  max_c = IC;
  blocks.resize(cfg.num_blocks);
  for (int bi = 0; bi < cfg.num_blocks; ++bi) {
    CamBlock& B = blocks[bi];
    B.dil = cfg.block_dilation[bi];
    B.cin = channels;
    const std::string bp = "xvector.block" + std::to_string(bi + 1);
    IDX_CHECK(cfg.block_layers[bi] > 0 && B.dil > 0, "dense block shape");
    B.layers.resize(cfg.block_layers[bi]);
    for (int i = 0; i < cfg.block_layers[bi]; ++i) {
      CamLayer& L = B.layers[i];
      L.cin = channels + i * G;
      const std::string p = bp + ".tdnnd" + std::to_string(i + 1);
      BN b1, b2;
      HostTensor *w1 = nullptr, *wl = nullptr, *c1 = nullptr, *c1b = nullptr, *c2 = nullptr, *c2b = nullptr;
      if (bn_from(t, p + ".nonlinear1.batchnorm", L.cin, true, &b1) || up(arena, b1.s, &L.bn1_s) || up(arena, b1.t, &L.bn1_t)) return 1;
      if (need(t, p + ".linear1.weight", {BNC, L.cin, 1}, &w1) || bn_from(t, p + ".nonlinear2.batchnorm", BNC, true, &b2)) return 1;
      if (conv1d_linear(arena, *w1, BNC, L.cin, 1, &b2, nullptr, &L.lin1)) return 1;
      if (need(t, p + ".cam_layer.linear_local.weight", {G, BNC, 3}, &wl) || conv1d_linear(arena, *wl, G, BNC, 3, nullptr, nullptr, &L.local)) return 1;
      if (need(t, p + ".cam_layer.linear1.weight", {BNC / 2, BNC, 1}, &c1) || need(t, p + ".cam_layer.linear1.bias", {BNC / 2}, &c1b) ||
          need(t, p + ".cam_layer.linear2.weight", {G, BNC / 2, 1}, &c2) || need(t, p + ".cam_layer.linear2.bias", {G}, &c2b)) return 1;
      if (up(arena, c1->data, &L.w1) || up(arena, c1b->data, &L.b1) || up(arena, c2->data, &L.w2) || up(arena, c2b->data, &L.b2)) return 1;
    }
    channels += cfg.block_layers[bi] * G;
    max_c = std::max(max_c, channels);
    BN tb;
    HostTensor* tw = nullptr;
    const std::string tp = "xvector.transit" + std::to_string(bi + 1);
    if (bn_from(t, tp + ".nonlinear.batchnorm", channels, true, &tb) || up(arena, tb.s, &B.tbn_s) || up(arena, tb.t, &B.tbn_t)) return 1;
    if (need(t, tp + ".linear.weight", {channels / 2, channels, 1}, &tw) || conv1d_linear(arena, *tw, channels / 2, channels, 1, nullptr, nullptr, &B.transit)) return 1;
    channels /= 2;
    B.cout = channels;
    IDX_CHECK((channels & 31) == 0, "transit output channels must be a multiple of 32");
  }
  final_c = channels;
  BN ob, db;
  if (bn_from(t, "xvector.out_nonlinear.batchnorm", channels, true, &ob) || up(arena, ob.s, &out_s) || up(arena, ob.t, &out_t)) return 1;
  HostTensor* dw = nullptr;
  if (need(t, "xvector.dense.linear.weight", {cfg.embedding_size, 2 * channels, 1}, &dw) ||
      bn_from(t, "xvector.dense.nonlinear.batchnorm", cfg.embedding_size, false, &db)) return 1;
  return conv1d_linear(arena, *dw, cfg.embedding_size, 2 * channels, 1, &db, nullptr, &dense);
}

// -------------------------------------------------------------------------------------------------------------------------
namespace {

// Direct Conv2d over planes [Cin][H][W] -> [Cout][Ho][W] (stride (stride_h, 1), padding k/2), + bias (+ residual) (+ ReLU).
// tok_major: the output is written as rows [W][Cout * Ho] (channel index co * Ho + ho): FCM's final reshape + the transpose to
// token-major in one go.
__global__ __launch_bounds__(256) void conv2d_kernel(float* out, const float* in, const float* w, const float* bias, const float* res, int Cin,
                                                     int Cout, int H, int W, int Ho, int k, int stride_h, int relu, int tok_major) {
  const int x = blockIdx.x * 256 + threadIdx.x, ho = blockIdx.y, co = blockIdx.z;
  if (x >= W) return;
  const int pad = k / 2;
  float acc = bias[co];
  for (int ci = 0; ci < Cin; ++ci)
    for (int i = 0; i < k; ++i) {
      const int h = ho * stride_h + i - pad;
      if (h < 0 || h >= H) continue;
      const float* row = in + ((size_t)ci * H + h) * W;
      const float* wr = w + (((size_t)co * Cin + ci) * k + i) * k;
      for (int j = 0; j < k; ++j) {
        const int xx = x + j - pad;
        if (xx >= 0 && xx < W) acc = fmaf(wr[j], row[xx], acc);
      }
    }
  if (res) acc += res[((size_t)co * Ho + ho) * W + x];
  if (relu) acc = fmaxf(acc, 0.0f);
  if (tok_major) out[(size_t)x * ((size_t)Cout * Ho) + (size_t)co * Ho + ho] = acc;
  else out[((size_t)co * Ho + ho) * W + x] = acc;
}

int conv2d(const Conv2dW& c, const float* in, float* out, const float* res, int H, int W, bool relu, bool tok_major, hipStream_t st) {
  const int Ho = (H + 2 * (c.k / 2) - c.k) / c.stride_h + 1;
  hipLaunchKernelGGL(conv2d_kernel, dim3(cdiv(W, 256), Ho, c.cout), dim3(256), 0, st, out, in, c.w, c.b, res, c.cin, c.cout, H, W, Ho, c.k,
                     c.stride_h, relu ? 1 : 0, tok_major ? 1 : 0);
  IDX_LAUNCH_CHECK();
  return 0;
}

// feat [T][F] -> plane [1][F][T]
__global__ __launch_bounds__(256) void transpose_feat_kernel(float* out, const float* feat, int T, int F) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= T * F) return;
  const int f = i / T, t = i - f * T;
  out[i] = feat[(size_t)t * F + f];
}

// y[t][c] = relu(x[t][c] * s[c] + b[c]), c < C   (x row stride ldx, y dense)
__global__ __launch_bounds__(256) void bn_relu_rows_kernel(float* y, const float* x, int ldx, const float* s, const float* b, int C) {
  const int t = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) y[(size_t)t * C + c] = fmaxf(fmaf(x[(size_t)t * ldx + c], s[c], b[c]), 0.0f);
}

// y[t2][:] = x[2 * t2][:]
__global__ __launch_bounds__(256) void take_even_rows_kernel(float* y, int ldy, const float* x, int C) {
  const int t2 = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) y[(size_t)t2 * ldy + c] = x[(size_t)2 * t2 * C + c];
}

// sums[seg][c] = sum of x[t][c] over the frames of segment seg (seg_len frames each, the last one shorter)
__global__ __launch_bounds__(256) void seg_sums_kernel(float* sums, const float* x, int T, int C, int seg_len) {
  const int sg = blockIdx.x, t0 = sg * seg_len, t1 = min(T, t0 + seg_len);
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.0f;
    for (int t = t0; t < t1; ++t) a += x[(size_t)t * C + c];
    sums[(size_t)sg * C + c] = a;
  }
}

// CAMLayer's mask for the frames of one segment (layers.py:92-113): context = mean over all frames + mean over the segment
// (avg_pool1d, ceil_mode: the last window averages the frames it has); m = sigmoid(W2 relu(W1 context + b1) + b2)
__global__ __launch_bounds__(256) void cam_gate_kernel(float* gate, const float* sums, int nseg, int T, int C, int Ch, int G, int seg_len,
                                                       const float* w1, const float* b1, const float* w2, const float* b2) {
  extern __shared__ float sm[];
  float* ctx = sm;          // [C]
  float* hid = sm + C;      // [Ch]
  const int sg = blockIdx.x, tid = threadIdx.x;
  const int n_in_seg = min(T, (sg + 1) * seg_len) - sg * seg_len;
  for (int c = tid; c < C; c += 256) {
    float tot = 0.0f;
    for (int s = 0; s < nseg; ++s) tot += sums[(size_t)s * C + c];
    ctx[c] = tot / T + sums[(size_t)sg * C + c] / n_in_seg;
  }
  __syncthreads();
  for (int h = tid; h < Ch; h += 256) {
    float a = b1[h];
    for (int c = 0; c < C; ++c) a = fmaf(w1[(size_t)h * C + c], ctx[c], a);
    hid[h] = fmaxf(a, 0.0f);
  }
  __syncthreads();
  for (int g = tid; g < G; g += 256) {
    float a = b2[g];
    for (int h = 0; h < Ch; ++h) a = fmaf(w2[(size_t)g * Ch + h], hid[h], a);
    gate[(size_t)sg * G + g] = 1.0f / (1.0f + expf(-a));
  }
}

// X[t][col0 + g] = y[t][g] * gate[t / seg_len][g]     (the append of torch.cat([x, layer(x)], dim=1))
__global__ __launch_bounds__(64) void gate_append_kernel(float* X, int ldX, int col0, const float* y, const float* gate, int G, int seg_len) {
  const int t = blockIdx.x;
  for (int g = threadIdx.x; g < G; g += 64) X[(size_t)t * ldX + col0 + g] = y[(size_t)t * G + g] * gate[(size_t)(t / seg_len) * G + g];
}

// StatsPool over relu(bn(x)): out[c] = mean_t, out[C + c] = unbiased std_t   (layers.py:23-36; out_nonlinear folded in)
__global__ __launch_bounds__(256) void stats_pool_kernel(float* out, const float* x, int ldx, const float* s, const float* b, int T, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float sum = 0.0f;
  for (int t = 0; t < T; ++t) sum += fmaxf(fmaf(x[(size_t)t * ldx + c], s[c], b[c]), 0.0f);
  const float mean = sum / T;
  float ss = 0.0f;
  for (int t = 0; t < T; ++t) { const float d = fmaxf(fmaf(x[(size_t)t * ldx + c], s[c], b[c]), 0.0f) - mean; ss = fmaf(d, d, ss); }
  out[c] = mean;
  out[C + c] = sqrtf(ss / (T - 1));
}

constexpr int SEG_LEN = 100;

struct CamBuf {
  float *pa, *pb, *pc, *tok, *full, *X, *X2, *h1, *h2, *y, *sums, *gate, *stats;
  size_t bytes;
};

CamBuf carve_cam(const CamPPlusModel& m, void* ws, int T) {
  const auto& c = m.cfg;
  const int T2 = (T - 1) / 2 + 1, nseg = cdiv(T2, SEG_LEN), BNC = c.bn_size * c.growth_rate;
  const size_t plane = (size_t)c.m_channels * c.feat_dim * T;
  CamBuf b;
  Carver k(ws);
  b.pa = k.take<float>(plane); b.pb = k.take<float>(plane); b.pc = k.take<float>(plane);
  b.tok = k.take<float>((size_t)T * m.fcm_out);
  b.full = k.take<float>((size_t)T * c.init_channels);
  b.X = k.take<float>((size_t)T2 * m.max_c);
  b.X2 = k.take<float>((size_t)T2 * m.max_c);
  b.h1 = k.take<float>((size_t)T2 * m.max_c);
  b.h2 = k.take<float>((size_t)T2 * BNC);
  b.y = k.take<float>((size_t)T2 * c.growth_rate);
  b.sums = k.take<float>((size_t)nseg * BNC);
  b.gate = k.take<float>((size_t)nseg * c.growth_rate);
  b.stats = k.take<float>((size_t)2 * m.max_c);
  b.bytes = (k.off + 255) & ~(size_t)255;
  return b;
}

}  // namespace

size_t CamPPlusModel::workspace_bytes(int T) const { return carve_cam(*this, nullptr, T).bytes; }

int CamPPlusModel::forward(const float* feat, int B, int T, float* out, void* ws, size_t ws_bytes, hipStream_t st) {
  IDX_CHECK(feat && out, "null pointer");
  IDX_CHECK(B > 0 && T >= 8, "CAMPPlus needs a few frames");
  IDX_CHECK(ws && ws_bytes >= workspace_bytes(T), "workspace too small");
  const int F = cfg.feat_dim, G = cfg.growth_rate, BNC = cfg.bn_size * cfg.growth_rate, IC = cfg.init_channels;
  CamBuf w = carve_cam(*this, ws, T);
  const int T2 = (T - 1) / 2 + 1, nseg = cdiv(T2, SEG_LEN);
  IDX_CHECK(T2 >= 2, "the unbiased standard deviation needs two frames");
  for (int b = 0; b < B; ++b) {
    // ---- FCM head on planes [C][H][T] ----
    hipLaunchKernelGGL(transpose_feat_kernel, dim3(cdiv(T * F, 256)), dim3(256), 0, st, w.pa, feat + (size_t)b * T * F, T, F);
    IDX_LAUNCH_CHECK();
    if (conv2d(conv1, w.pa, w.pb, nullptr, F, T, true, false, st)) return 1;
    // three planes: x (the block's input AND output: conv2 reads t1 and adds the shortcut element by element, so it may overwrite x),
    // t1 (conv1's output), t2 (the strided shortcut)
    float *x = w.pb, *t1 = w.pa, *t2 = w.pc;
    int H = F;
    for (const FcmResBlock& rb : res) {
      const int Ho = (H + 2 - 3) / rb.c1.stride_h + 1;
      if (conv2d(rb.c1, x, t1, nullptr, H, T, true, false, st)) return 1;
      const float* sc = x;
      if (rb.has_sc) {
        if (conv2d(rb.sc, x, t2, nullptr, H, T, false, false, st)) return 1;
        sc = t2;
      }
      if (conv2d(rb.c2, t1, x, sc, Ho, T, true, false, st)) return 1;
      H = Ho;
    }
    if (conv2d(conv2, x, w.tok, nullptr, H, T, true, true, st)) return 1;       // -> token-major [T][mc * H / 2]
    // ---- TDNN layer: k5, stride 2 (the stride-1 result on the even frames), BatchNorm folded, ReLU ----
    {
      GemmArgs g;
      g.x = w.tok; g.ldx = fcm_out; g.y = w.full; g.ldy = IC; g.M = T; g.taps = 5; g.seq_len = T; g.dil = 1; g.pad_left = 2; g.pad_mode = 0; g.act = ACT_RELU;
      if (gemm_tn_forward(tdnn, g, st)) return 1;
    }
    float* X = w.X;
    float* Xn = w.X2;
    int ld = blocks.empty() ? IC : blocks[0].cin + (int)blocks[0].layers.size() * G;
    hipLaunchKernelGGL(take_even_rows_kernel, dim3(T2), dim3(256), 0, st, X, ld, w.full, IC);
    IDX_LAUNCH_CHECK();
    for (size_t bi = 0; bi < blocks.size(); ++bi) {
      const CamBlock& Bk = blocks[bi];
      for (const CamLayer& L : Bk.layers) {
        hipLaunchKernelGGL(bn_relu_rows_kernel, dim3(T2), dim3(256), 0, st, w.h1, X, ld, L.bn1_s, L.bn1_t, L.cin);
        IDX_LAUNCH_CHECK();
        if (lin(L.lin1, w.h1, L.cin, w.h2, BNC, T2, st, ACT_RELU)) return 1;
        GemmArgs g;
        g.x = w.h2; g.ldx = BNC; g.y = w.y; g.ldy = G; g.M = T2; g.taps = 3; g.seq_len = T2; g.dil = Bk.dil; g.pad_left = Bk.dil; g.pad_mode = 0;
        if (gemm_tn_forward(L.local, g, st)) return 1;
        hipLaunchKernelGGL(seg_sums_kernel, dim3(nseg), dim3(256), 0, st, w.sums, w.h2, T2, BNC, SEG_LEN);
        IDX_LAUNCH_CHECK();
        hipLaunchKernelGGL(cam_gate_kernel, dim3(nseg), dim3(256), (size_t)(BNC + BNC / 2) * sizeof(float), st, w.gate, w.sums, nseg, T2, BNC, BNC / 2, G,
                           SEG_LEN, L.w1, L.b1, L.w2, L.b2);
        IDX_LAUNCH_CHECK();
        hipLaunchKernelGGL(gate_append_kernel, dim3(T2), dim3(64), 0, st, X, ld, L.cin, w.y, w.gate, G, SEG_LEN);
        IDX_LAUNCH_CHECK();
      }
      const int ctot = Bk.cin + (int)Bk.layers.size() * G;
      hipLaunchKernelGGL(bn_relu_rows_kernel, dim3(T2), dim3(256), 0, st, w.h1, X, ld, Bk.tbn_s, Bk.tbn_t, ctot);
      IDX_LAUNCH_CHECK();
      const int ldn = bi + 1 < blocks.size() ? blocks[bi + 1].cin + (int)blocks[bi + 1].layers.size() * G : Bk.cout;
      if (lin(Bk.transit, w.h1, ctot, Xn, ldn, T2, st)) return 1;
      std::swap(X, Xn);
      ld = ldn;
    }
    hipLaunchKernelGGL(stats_pool_kernel, dim3(cdiv(final_c, 256)), dim3(256), 0, st, w.stats, X, ld, out_s, out_t, T2, final_c);
    IDX_LAUNCH_CHECK();
    if (lin(dense, w.stats, 2 * final_c, out + (size_t)b * cfg.embedding_size, cfg.embedding_size, 1, st)) return 1;
  }
  return 0;
}

}  // namespace idxtts
