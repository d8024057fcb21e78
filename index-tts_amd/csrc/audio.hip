// Log-mel spectrogram of the prompt audio on MI355X (`ref_mel = self.mel_fn(audio_22k)`, infer_v2.py:640).
//
// Reference: mel_spectrogram   indextts/s2mel/modules/audio.py:45-83  (reflect pad (n_fft - hop) / 2, torch.stft(center=False, hann),
//                              sqrt(re^2 + im^2 + 1e-9), mel_basis @ spec, log(clamp(., 1e-5)))
// The mel basis (librosa.filters.mel in the reference) and the window are tensors of the context: the host mirror builds them.
//
// The STFT runs as a GEMM: frames [B*T][n_fft] (gathered with the reflect padding) x the windowed DFT matrix [2 * bins][n_fft]
// on the exact-fp32 MFMA; 1292 frames for a 15 s prompt.
#include <cmath>

#include "audio.h"
#include "model_util.h"

namespace idxtts {

int MelSpecModel::finalize(std::map<std::string, HostTensor>& t, DeviceArena& arena) {
  const int nf = cfg.n_fft, win = cfg.win_size;
  IDX_CHECK(nf >= 16 && (nf & 3) == 0 && win > 0 && win <= nf && cfg.hop_size > 0 && cfg.hop_size <= nf && cfg.num_mels > 0, "mel spectrogram shape");
  nbins = nf / 2 + 1;
  nbins4 = (nbins + 3) & ~3;
  HostTensor *mb = nullptr, *wd = nullptr;
  if (need(t, "mel_basis", {cfg.num_mels, nbins}, &mb) || need(t, "window", {win}, &wd)) return 1;
  // torch.stft centres a window shorter than n_fft inside the frame
  std::vector<double> wfull(nf, 0.0);
  const int off = (nf - win) / 2;
  for (int i = 0; i < win; ++i) wfull[off + i] = wd->data[i];
  std::vector<float> m((size_t)2 * nbins * nf);
  const double w0 = 2.0 * M_PI / nf;
  for (int n = 0; n < nbins; ++n)
    for (int k = 0; k < nf; ++k) {
      const double ang = w0 * (double)(((long long)n * k) % nf);
      m[(size_t)n * nf + k] = (float)(wfull[k] * std::cos(ang));
      m[(size_t)(nbins + n) * nf + k] = (float)(-wfull[k] * std::sin(ang));
    }
  if (make_linear(arena, m.data(), nullptr, 2 * nbins, nf, nf, &dft)) return 1;
  return make_linear(arena, mb->data.data(), nullptr, cfg.num_mels, nbins, nbins4, &mel);
}

int MelSpecModel::frames(int N) const {
  const int pad = (cfg.n_fft - cfg.hop_size) / 2;
  const long long L = (long long)N + 2 * pad;
  return L < cfg.n_fft ? 0 : (int)((L - cfg.n_fft) / cfg.hop_size) + 1;
}

namespace {

struct MelBuf { float *fr, *spec, *mag, *mel; size_t bytes; };

MelBuf carve_mel(const MelSpecModel& m, void* ws, int B, int N) {
  const size_t M = (size_t)B * std::max(0, m.frames(N));
  MelBuf b;
  Carver k(ws);
  b.fr = k.take<float>(M * m.cfg.n_fft);
  b.spec = k.take<float>(M * 2 * m.nbins);
  b.mag = k.take<float>(M * m.nbins4);
  b.mel = k.take<float>(M * m.cfg.num_mels);
  b.bytes = (k.off + 255) & ~(size_t)255;
  return b;
}

// frame (b, t) = padded[b][t * hop .. t * hop + n_fft), padded = reflect-padded audio (pad samples each side, edge not repeated)
__global__ __launch_bounds__(256) void gather_frames_kernel(float* fr, const float* audio, int N, int T, int nf, int hop, int pad) {
  const int t = blockIdx.x, b = blockIdx.y;
  const float* a = audio + (size_t)b * N;
  float* o = fr + ((size_t)b * T + t) * nf;
  for (int i = threadIdx.x; i < nf; i += 256) {
    int s = t * hop + i - pad;
    if (s < 0) s = -s;
    if (s >= N) s = 2 * (N - 1) - s;
    o[i] = a[s];
  }
}

__global__ __launch_bounds__(256) void magnitude_kernel(float* mag, const float* spec, int nbins, int nbins4) {
  const size_t m = blockIdx.x;
  const float* re = spec + m * 2 * nbins;
  const float* im = re + nbins;
  for (int n = threadIdx.x; n < nbins4; n += 256) mag[m * nbins4 + n] = n < nbins ? sqrtf(re[n] * re[n] + im[n] * im[n] + 1e-9f) : 0.0f;
}

// out[b][c][t] = log(max(mel[(b, t)][c], 1e-5))
__global__ __launch_bounds__(256) void log_transpose_kernel(float* out, const float* mel, int T, int C) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= T * C) return;
  const int c = i / T, t = i - c * T;
  out[((size_t)b * C + c) * T + t] = logf(fmaxf(mel[((size_t)b * T + t) * C + c], 1e-5f));
}

}  // namespace

size_t MelSpecModel::workspace_bytes(int B, int N) const { return carve_mel(*this, nullptr, B, N).bytes; }

int MelSpecModel::forward(const float* audio, int B, int N, float* out, void* ws, size_t ws_bytes, hipStream_t st) {
  IDX_CHECK(audio && out, "null pointer");
  const int pad = (cfg.n_fft - cfg.hop_size) / 2;
  IDX_CHECK(B > 0 && N > pad, "reflect padding needs more samples than the pad width");
  const int T = frames(N);
  IDX_CHECK(T > 0, "audio shorter than one frame");
  IDX_CHECK(ws && ws_bytes >= workspace_bytes(B, N), "workspace too small");
  MelBuf w = carve_mel(*this, ws, B, N);
  const int M = B * T;
  hipLaunchKernelGGL(gather_frames_kernel, dim3(T, B), dim3(256), 0, st, w.fr, audio, N, T, cfg.n_fft, cfg.hop_size, pad);
  IDX_LAUNCH_CHECK();
  if (lin(dft, w.fr, cfg.n_fft, w.spec, 2 * nbins, M, st)) return 1;
  hipLaunchKernelGGL(magnitude_kernel, dim3(M), dim3(256), 0, st, w.mag, w.spec, nbins, nbins4);
  IDX_LAUNCH_CHECK();
  if (lin(mel, w.mag, nbins4, w.mel, cfg.num_mels, M, st)) return 1;
  hipLaunchKernelGGL(log_transpose_kernel, dim3(cdiv(T * cfg.num_mels, 256), B), dim3(256), 0, st, out, w.mel, T, cfg.num_mels);
  IDX_LAUNCH_CHECK();
  return 0;
}

}  // namespace idxtts
