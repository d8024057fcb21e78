/* libidxtts_hip -- C ABI of the MI355X-native IndexTTS-2 hot path (gfx950 only).
 *
 * Drop-in boundary.  Every entry point takes plain device pointers, explicit
 * shapes and a hipStream_t (passed as void*), returns 0 on success / non-zero on error (message via
 * idxtts_last_error(), thread-local) and never throws.  No ownership transfer: the caller's allocator
 * (torch) owns inputs, outputs and workspaces; the library owns only an opaque idxtts_ctx holding the
 * kernel-layout (packed) weights.  One ctx per (process, device); calls on one ctx are serialised by
 * the caller (same rule as the reference object, infer_v2.py:304-310 / serve_tars.py:139-140).
 * All tensors are float32, contiguous, unless a parameter says otherwise.
 *
 * Reference interfaces replaced (file:line relative to grantjr1842/index-tts):
 *   idxtts_aa_act_fwd      <- pybind `anti_alias_activation_cuda.forward(input, up_ftr, down_ftr, alpha, beta)`
 *                             indextts/s2mel/modules/bigvgan/alias_free_activation/cuda/anti_alias_activation.cpp:19-23,
 *                             `extern "C" fwd_cuda` anti_alias_activation_cuda.cu:212-246, used by
 *                             `FusedAntiAliasActivation.forward` activation1d.py:21-26
 *   idxtts_bigvgan_*       <- `BigVGAN.forward` indextts/s2mel/modules/bigvgan/bigvgan.py:360-386 as called at
 *                             indextts/infer_v2.py:860 (`self.bigvgan(vc_target.float())`)
 *   idxtts_conv1d_*        <- torch.nn.Conv1d / ConvTranspose1d call sites on the hot path
 *                             (bigvgan.py:136-139, 362, 367; wavenet.py:149, 161; length_regulator.py:51, 61)
 */
#ifndef IDXTTS_H
#define IDXTTS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct idxtts_ctx idxtts_ctx;

#define IDXTTS_DTYPE_F32 0

int idxtts_version(void);
/* Last error message of the calling thread ("" if none). */
const char* idxtts_last_error(void);

/* ---- fused anti-aliased SnakeBeta activation (reference: fwd_cuda, .cu:212-246) -------------------
 * out/in: [B][C][T]; up_filter/down_filter: 12 taps; log_alpha/log_beta: [C] (log-scale, exp'd in-kernel,
 * .cu:86-89).  T == 0 is a no-op (.cu:193).  out must not alias in.  dtype must be IDXTTS_DTYPE_F32. */
int idxtts_aa_act_fwd(float* out, const float* in, const float* up_filter, const float* down_filter,
                      const float* log_alpha, const float* log_beta, int B, int C, int T, int dtype,
                      void* stream);

/* ---- stand-alone Conv1d on the fp32 matrix core (used by the tests and by s2mel) -------------------
 * weight: torch layout [Cout][Cin][K] (or [Cin][Cout][Kt] when transposed_stride > 1), host or device.
 * The returned handle owns the packed copy; free with idxtts_conv1d_destroy. */
typedef struct idxtts_conv1d idxtts_conv1d;
int idxtts_conv1d_create(const float* weight, const float* bias /* may be NULL */, int Cout, int Cin, int K,
                         int transposed_stride /* 1 = Conv1d, u>1 = ConvTranspose1d(k=2u, stride u, pad u/2) */,
                         idxtts_conv1d** out);
/* y[b][co][t] = scale*(bias + sum x[..t + k*dil - pad_left..] + residual) (+ y if accumulate).
 * pad_mode: 0 zero, 1 reflect.  x: [B][Cin][T]; y/residual: [B][Cout][T*transposed_stride]. */
int idxtts_conv1d_fwd(const idxtts_conv1d* conv, const float* x, float* y, const float* residual /* may be NULL */,
                      int B, int T, int dilation, int pad_left, int pad_mode, float scale, int accumulate,
                      void* stream);
int idxtts_conv1d_destroy(idxtts_conv1d* conv);

/* ---- generic weight hand-over for model contexts -------------------------------------------------
 * `name` is the reference module's state_dict key (e.g. "resblocks.3.convs1.0.weight"); data may be a
 * host or a device pointer (copied during the call).  Unknown keys are rejected; buffers the kernels
 * derive themselves ("*.filter") are accepted and cross-checked. */
int idxtts_ctx_load_tensor(idxtts_ctx* ctx, const char* name, const float* data, const int64_t* shape, int ndim);
/* Pack everything loaded so far into kernel layouts on the current device; fails if a tensor is missing. */
int idxtts_ctx_finalize(idxtts_ctx* ctx);
int idxtts_ctx_destroy(idxtts_ctx* ctx);

/* ---- BigVGAN-v2 vocoder (reference: BigVGAN.forward, bigvgan.py:360-386) ------------------------- */
typedef struct idxtts_bigvgan_config {
  int num_mels;                   /* 80 */
  int upsample_initial_channel;   /* 1536 */
  int num_upsamples;              /* 6 */
  int upsample_rates[8];          /* 4,4,2,2,2,2 */
  int upsample_kernel_sizes[8];   /* 8,8,4,4,4,4 */
  int num_kernels;                /* 3 */
  int resblock_kernel_sizes[4];   /* 3,7,11 */
  int resblock_dilations[4][3];   /* 1,3,5 each */
} idxtts_bigvgan_config;

int idxtts_bigvgan_create(const idxtts_bigvgan_config* cfg, idxtts_ctx** out);
/* Bytes of scratch idxtts_bigvgan_fwd needs for a [B][num_mels][Tm] input. */
size_t idxtts_bigvgan_workspace_bytes(const idxtts_ctx* ctx, int B, int Tm);
/* mel: [B][num_mels][Tm] -> wav: [B][1][Tm*prod(upsample_rates)], clamped to [-1,1] (bigvgan.py:384).
 * If stage_out != NULL the activation tensor after up-sampling stage `stage_idx` (1-based, after the
 * resblock average, [B][C_i][T_i]) is also copied there (parity tests).  If clamp == 0 the final clamp
 * is skipped (tests compare the pre-clamp waveform so saturation cannot hide an error). */
int idxtts_bigvgan_fwd(idxtts_ctx* ctx, const float* mel, float* wav, int B, int Tm, void* workspace,
                       size_t workspace_bytes, int clamp, int stage_idx, float* stage_out, void* stream);

/* ---- per-kernel timing for the benchmark's roofline report ------------------------------------------
 * When enabled, every kernel launch is bracketed by HIP events on its own stream and the library
 * accumulates, per kernel family, the launch count, elapsed milliseconds and the ALGORITHMIC flops /
 * bytes of the launches (DESIGN.md "work units").  Enabling resets the accumulators.  Not for use
 * inside a timed region (event records serialise nothing but add host work per launch). */
int idxtts_profile_enable(int on);
int idxtts_profile_num_kernels(void);
const char* idxtts_profile_kernel_name(int index);
int idxtts_profile_read(int index, double* total_ms, double* flops, double* bytes, long* launches);

#ifdef __cplusplus
}
#endif
#endif /* IDXTTS_H */
