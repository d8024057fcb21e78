// BigVGAN-v2 forward on MI355X: host-side sequencing of the HIP kernels.
// Reference: BigVGAN.forward bigvgan.py:360-386, AMPBlock1.forward bigvgan.py:132-141.
//
// Data layout in HBM: activations stay channels-first [B][C][T] fp32 (time contiguous), the layout
// the reference uses, so the time axis is the coalesced/GEMM-N axis of every kernel.  Six scratch
// buffers of max_i(B*C_i*T_i) floats live in the caller's workspace:
//   P  = stage input            X0 = transposed-conv output (shared input of the 3 AMP blocks)
//   ACT= activation output      T1 = convs1 output
//   R  = running AMP residual   XS = (1/3) * sum of the three AMP block outputs -> next stage's P
// Fusions relative to the reference graph: bias, residual add (x = xt + x), the xs += / xs /= 3
// averaging and the transposed-conv phase interleave all happen in the conv epilogue; the three
// Activation1d sub-ops are one kernel.
#include <cmath>

#include "bigvgan.h"

namespace idxtts {

int aa_act_forward(void* y, const void* x, const float* up_f, const float* down_f, const float* log_alpha,
                   const float* log_beta, int B, int C, int T, hipStream_t stream, const int* lens = nullptr, int len_mul = 1, int dtype = 0);
int conv_post_forward(float* y, const float* x, const float* w, int B, int C, int T, int clamp, hipStream_t stream);

// kaiser-sinc low-pass, cutoff 0.25, half-width 0.3, 12 taps (filter.py:30-62), in double.
static void kaiser_sinc12(float* out) {
  const int K = 12, half = 6;
  const double cutoff = 0.25, half_width = 0.3;
  const double A = 2.285 * (half - 1) * M_PI * (4 * half_width) + 7.95;
  double beta = A > 50.0 ? 0.1102 * (A - 8.7) : (A >= 21.0 ? 0.5842 * std::pow(A - 21.0, 0.4) + 0.07886 * (A - 21.0) : 0.0);
  auto i0 = [](double x) {
    double s = 1.0, term = 1.0;
    for (int k = 1; k < 64; ++k) { term *= (x / (2.0 * k)) * (x / (2.0 * k)); s += term; }
    return s;
  };
  double f[12], sum = 0.0;
  for (int n = 0; n < K; ++n) {
    const double r = 2.0 * n / (K - 1) - 1.0;
    const double win = i0(beta * std::sqrt(std::max(0.0, 1.0 - r * r))) / i0(beta);
    const double t = (n - half) + 0.5;
    const double arg = 2 * cutoff * t;
    const double sinc = std::sin(M_PI * arg) / (M_PI * arg);
    f[n] = 2 * cutoff * win * sinc;
    sum += f[n];
  }
  for (int n = 0; n < K; ++n) out[n] = (float)(f[n] / sum);
}

BigVGANModel::BigVGANModel(const idxtts_bigvgan_config& c) : cfg(c) {}

int BigVGANModel::stage_channels(int i) const { return cfg.upsample_initial_channel >> i; }

bool BigVGANModel::accepts(const std::string& name) const {
  auto ends = [&](const char* s) { const std::string e(s); return name.size() >= e.size() && name.compare(name.size() - e.size(), e.size(), e) == 0; };
  if (ends(".filter")) return true;
  if (name.rfind("conv_pre.", 0) == 0 || name.rfind("conv_post.", 0) == 0) return true;
  if (name.rfind("ups.", 0) == 0 || name.rfind("resblocks.", 0) == 0 || name.rfind("activation_post.", 0) == 0) return true;
  return false;
}

static int need(std::map<std::string, HostTensor>& t, const std::string& key, std::vector<int64_t> shape, HostTensor** out) {
  auto it = t.find(key);
  if (it == t.end()) IDX_FAIL("missing tensor '" + key + "'");
  if (it->second.shape != shape) {
    std::string s = "tensor '" + key + "' has shape [";
    for (auto d : it->second.shape) s += std::to_string(d) + ",";
    s += "] expected [";
    for (auto d : shape) s += std::to_string(d) + ",";
    IDX_FAIL(s + "]");
  }
  *out = &it->second;
  return 0;
}

static int make_conv(std::map<std::string, HostTensor>& t, DeviceArena& arena, const std::string& prefix, int Cout,
                     int Cin, int K, bool bias, int ups, ConvWeights* out) {
  HostTensor* w = nullptr;
  HostTensor* b = nullptr;
  if (ups > 1) {
    if (need(t, prefix + ".weight", {Cin, Cout, K}, &w)) return 1;
  } else {
    if (need(t, prefix + ".weight", {Cout, Cin, K}, &w)) return 1;
  }
  if (bias && need(t, prefix + ".bias", {Cout}, &b)) return 1;
  ConvWeights cw;
  cw.Cin = Cin;
  cw.nchunk = cdiv(Cin, CONV_KC);
  cw.ups = ups;
  if (ups > 1) {
    IDX_CHECK(K == 2 * ups, "ConvTranspose1d kernel must be 2*stride");
    cw.M = Cout * ups;
    cw.K = 3;
  } else {
    cw.M = Cout;
    cw.K = K;
  }
  std::vector<float> packed(conv_packed_floats(cw.M, Cin, cw.K));
  if (ups > 1) pack_conv_transpose1d(packed.data(), w->data.data(), Cin, Cout, K, ups);
  else pack_conv1d(packed.data(), w->data.data(), Cout, Cin, K);
  float* d = nullptr;
  if (arena.upload(packed.data(), packed.size(), &d)) return 1;
  cw.wp = d;
  {                     // split-bf16 images (conv1d_bf16x3.hip); same size as the fp32 pack
    std::vector<float> p16(packed.size());      // hi + lo bf16 = 4 bytes per weight, as the fp32 pack
    pack_conv_bf16x3(p16.data(), packed.data(), packed.size() / CONV_SUB);
    if (arena.upload(p16.data(), p16.size(), &d)) return 1;
    cw.wp16 = d;
  }
  if (b) {
    if (arena.upload(b->data.data(), b->data.size(), &d)) return 1;
    cw.bias = d;
  }
  *out = cw;
  return 0;
}

static int make_vec(std::map<std::string, HostTensor>& t, DeviceArena& arena, const std::string& key, int n, const float** out) {
  HostTensor* v = nullptr;
  if (need(t, key, {n}, &v)) return 1;
  float* d = nullptr;
  if (arena.upload(v->data.data(), v->data.size(), &d)) return 1;
  *out = d;
  return 0;
}

int BigVGANModel::finalize(std::map<std::string, HostTensor>& t, DeviceArena& arena) {
  IDX_CHECK(cfg.num_upsamples > 0 && cfg.num_upsamples <= 8 && cfg.num_kernels > 0 && cfg.num_kernels <= 4, "config");
  IDX_CHECK((cfg.upsample_initial_channel >> cfg.num_upsamples) >= 1, "upsample_initial_channel too small");
  if (make_conv(t, arena, "conv_pre", cfg.upsample_initial_channel, cfg.num_mels, 7, true, 1, &conv_pre)) return 1;
  ups.resize(cfg.num_upsamples);
  blocks.resize(cfg.num_upsamples * cfg.num_kernels);
  for (int i = 0; i < cfg.num_upsamples; ++i) {
    const int cin = stage_channels(i), cout = stage_channels(i + 1);
    if (make_conv(t, arena, "ups." + std::to_string(i) + ".0", cout, cin, cfg.upsample_kernel_sizes[i], true,
                  cfg.upsample_rates[i], &ups[i])) return 1;
    for (int j = 0; j < cfg.num_kernels; ++j) {
      AmpBlock& blk = blocks[i * cfg.num_kernels + j];
      const std::string pre = "resblocks." + std::to_string(i * cfg.num_kernels + j);
      blk.kernel = cfg.resblock_kernel_sizes[j];
      for (int l = 0; l < 3; ++l) {
        blk.dil[l] = cfg.resblock_dilations[j][l];
        if (make_conv(t, arena, pre + ".convs1." + std::to_string(l), cout, cout, blk.kernel, true, 1, &blk.convs1[l])) return 1;
        if (make_conv(t, arena, pre + ".convs2." + std::to_string(l), cout, cout, blk.kernel, true, 1, &blk.convs2[l])) return 1;
      }
      for (int a = 0; a < 6; ++a) {
        if (make_vec(t, arena, pre + ".activations." + std::to_string(a) + ".act.alpha", cout, &blk.alpha[a])) return 1;
        if (make_vec(t, arena, pre + ".activations." + std::to_string(a) + ".act.beta", cout, &blk.beta[a])) return 1;
      }
    }
  }
  const int cl = stage_channels(cfg.num_upsamples);
  if (make_vec(t, arena, "activation_post.act.alpha", cl, &post_alpha)) return 1;
  if (make_vec(t, arena, "activation_post.act.beta", cl, &post_beta)) return 1;
  HostTensor* wpost = nullptr;
  if (need(t, "conv_post.weight", {1, cl, 7}, &wpost)) return 1;
  float* d = nullptr;
  if (arena.upload(wpost->data.data(), wpost->data.size(), &d)) return 1;
  conv_post_w = d;

  // anti-alias filters: take the reference's registered buffers when they were handed over, else derive.
  float upf[12], downf[12];
  kaiser_sinc12(upf);
  kaiser_sinc12(downf);
  auto pick = [&](const char* key, float* dst) {
    auto it = t.find(key);
    if (it != t.end() && it->second.numel() == 12)
      for (int k = 0; k < 12; ++k) dst[k] = it->second.data[k];
  };
  pick("activation_post.upsample.filter", upf);
  pick("activation_post.downsample.lowpass.filter", downf);
  if (arena.upload(upf, 12, &d)) return 1;
  up_filter = d;
  if (arena.upload(downf, 12, &d)) return 1;
  down_filter = d;
  return 0;
}

size_t BigVGANModel::max_elems(int B, int Tm) const {
  size_t best = (size_t)cfg.upsample_initial_channel * Tm;
  size_t T = Tm;
  for (int i = 0; i < cfg.num_upsamples; ++i) {
    T *= cfg.upsample_rates[i];
    best = std::max(best, (size_t)stage_channels(i + 1) * T);
  }
  return best * B;
}

size_t BigVGANModel::workspace_bytes(int B, int Tm) const {
  const size_t per = (max_elems(B, Tm) * sizeof(float) + 255) & ~(size_t)255;
  return 6 * per;
}

// lens: optional device int32 [B], valid mel frames per row (<= Tm; the mel must be zero beyond them).  Every layer then keeps
// a shorter row's tail at zero and pads / replicates at that row's own end, so row b equals the B = 1 call on mel[b, :, :lens[b]].
int BigVGANModel::forward(const float* mel, float* wav, int B, int Tm, void* workspace, size_t workspace_bytes_in,
                          int clamp, int stage_idx, float* stage_out, hipStream_t stream, const int* lens) {
  IDX_CHECK(mel && wav, "null pointer");
  if (B == 0 || Tm == 0) return 0;
  IDX_CHECK(workspace && workspace_bytes_in >= workspace_bytes(B, Tm), "workspace too small");
  const size_t per = (max_elems(B, Tm) * sizeof(float) + 255) & ~(size_t)255;
  char* base = static_cast<char*>(workspace);
  float* P = reinterpret_cast<float*>(base + 0 * per);
  float* X0 = reinterpret_cast<float*>(base + 1 * per);
  float* ACT = reinterpret_cast<float*>(base + 2 * per);
  float* T1 = reinterpret_cast<float*>(base + 3 * per);
  float* R = reinterpret_cast<float*>(base + 4 * per);
  float* XS = reinterpret_cast<float*>(base + 5 * per);

  ConvArgs a;
  a.B = B;
  // conv_pre: Conv1d(num_mels -> C0, k7, pad 3)   bigvgan.py:362
  a.x = mel; a.y = P; a.T = Tm; a.dil = 1; a.pad_left = 3; a.lens = lens; a.len_mul_out = 1;
  if (conv1d_forward(conv_pre, a, stream)) return 1;

  int T = Tm;
  int mul = 1;          // samples per mel frame at the current stage
  for (int i = 0; i < cfg.num_upsamples; ++i) {
    const int u = cfg.upsample_rates[i];
    const int C = stage_channels(i + 1);
    // ConvTranspose1d as a 3-tap conv over x[s-1..s+1] with interleaved store   bigvgan.py:367
    ConvArgs up;
    up.B = B; up.x = P; up.y = X0; up.T = T; up.dil = 1; up.pad_left = 1; up.lens = lens; up.len_mul_out = mul * u;
    if (conv1d_forward(ups[i], up, stream)) return 1;
    T *= u;
    mul *= u;
    for (int j = 0; j < cfg.num_kernels; ++j) {
      AmpBlock& blk = blocks[i * cfg.num_kernels + j];
      const int k = blk.kernel;
      for (int l = 0; l < 3; ++l) {
        const float* src = (l == 0) ? X0 : R;
        const int d = blk.dil[l];
        if (aa_act_forward(ACT, src, up_filter, down_filter, blk.alpha[2 * l], blk.beta[2 * l], B, C, T, stream, lens, mul)) return 1;
        ConvArgs c1;
        c1.B = B; c1.x = ACT; c1.y = T1; c1.T = T; c1.dil = d; c1.pad_left = (k * d - d) / 2; c1.lens = lens; c1.len_mul_out = mul;
        if (conv1d_forward(blk.convs1[l], c1, stream)) return 1;
        if (aa_act_forward(ACT, T1, up_filter, down_filter, blk.alpha[2 * l + 1], blk.beta[2 * l + 1], B, C, T, stream, lens, mul)) return 1;
        ConvArgs c2;
        c2.B = B; c2.x = ACT; c2.T = T; c2.dil = 1; c2.pad_left = (k - 1) / 2; c2.res = src; c2.lens = lens; c2.len_mul_out = mul;
        if (l < 2) {
          c2.y = R;                       // x = xt + x            bigvgan.py:139
        } else {
          c2.y = XS;                      // xs (+)= block output ; x = xs / num_kernels   bigvgan.py:369-375
          c2.scale = 1.0f / cfg.num_kernels;
          c2.accum = j > 0;
        }
        if (conv1d_forward(blk.convs2[l], c2, stream)) return 1;
      }
    }
    if (stage_out && stage_idx == i + 1)
      IDX_HIP(hipMemcpyAsync(stage_out, XS, (size_t)B * C * T * sizeof(float), hipMemcpyDeviceToDevice, stream));
    std::swap(P, XS);
  }
  const int cl = stage_channels(cfg.num_upsamples);
  if (aa_act_forward(ACT, P, up_filter, down_filter, post_alpha, post_beta, B, cl, T, stream, lens, mul)) return 1;   // bigvgan.py:378
  if (conv_post_forward(wav, ACT, conv_post_w, B, cl, T, clamp, stream)) return 1;                        // bigvgan.py:379-384
  return 0;
}

}  // namespace idxtts
