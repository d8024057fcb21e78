#pragma once
#include "../../include/idxtts.h"
#include "ctx.h"
#include "gemm.h"
#include "prof.h"

namespace idxtts {

// mel_spectrogram of the reference (s2mel/modules/audio.py:45-83) as two exact-fp32 GEMMs: windowed real DFT of the reflect-padded
// frames, magnitude, mel basis, log.
struct MelSpecModel : ModelBase {
  idxtts_melspec_config cfg;
  int nbins = 0, nbins4 = 0;
  LinearWeights dft;      // [2 * nbins][n_fft]: rows n < nbins = window * cos, rows nbins.. = -window * sin
  LinearWeights mel;      // [n_mels][nbins4] (zero columns beyond nbins)

  explicit MelSpecModel(const idxtts_melspec_config& c) : cfg(c) {}
  bool accepts(const std::string& name) const override { return name == "mel_basis" || name == "window"; }
  int finalize(std::map<std::string, HostTensor>& t, DeviceArena& arena) override;
  int frames(int N) const;
  size_t workspace_bytes(int B, int N) const;
  int forward(const float* audio, int B, int N, float* out, void* ws, size_t ws_bytes, hipStream_t st);
};

}  // namespace idxtts
