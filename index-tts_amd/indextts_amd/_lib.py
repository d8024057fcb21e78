"""ctypes binding of libidxtts_hip.so (the C ABI declared in include/idxtts.h).

The product path has NO fallback: if the HIP library is missing or fails to load, importing an
op raises -- loudly -- instead of routing through PyTorch or the CPU oracle.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libidxtts_hip.so")

# every symbol include/idxtts.h declares (tests/test_abi.py checks the list against the header)
SYMBOLS = [
    "idxtts_version", "idxtts_last_error", "idxtts_aa_act_fwd",
    "idxtts_conv1d_create", "idxtts_conv1d_fwd", "idxtts_conv1d_destroy",
    "idxtts_ctx_load_tensor", "idxtts_ctx_finalize", "idxtts_ctx_get_tensor", "idxtts_ctx_destroy",
    "idxtts_fp8_e4m3_decode", "idxtts_fp8_e4m3_encode",
    "idxtts_bigvgan_create", "idxtts_bigvgan_workspace_bytes", "idxtts_bigvgan_fwd", "idxtts_bigvgan_fwd_ragged",
    "idxtts_profile_enable", "idxtts_profile_num_kernels", "idxtts_profile_kernel_name", "idxtts_profile_read", "idxtts_profile_event_overhead",
    "idxtts_linear_create", "idxtts_linear_fwd", "idxtts_linear_destroy", "idxtts_release_stream", "idxtts_attention_fwd", "idxtts_attention_bf16x3_fwd", "idxtts_attention_relkey_fwd", "idxtts_layernorm_fwd",
    "idxtts_gpt_create", "idxtts_gpt_quantize_weights", "idxtts_gpt_set_kv_format", "idxtts_gpt_get_kv_format", "idxtts_gpt_workspace_bytes", "idxtts_gpt_embed", "idxtts_gpt_generate", "idxtts_gpt_generate_forced", "idxtts_gpt_generate_sampled", "idxtts_gpt_latent",
    "idxtts_gpt_beam_workspace_bytes", "idxtts_gpt_generate_beam",
    "idxtts_s2mel_create", "idxtts_s2mel_cond_workspace_bytes", "idxtts_s2mel_prepare_cond",
    "idxtts_s2mel_cfm_workspace_bytes", "idxtts_s2mel_cfm", "idxtts_set_gemm_mode", "idxtts_get_gemm_mode", "idxtts_set_decode_geometry", "idxtts_get_decode_geometry", "idxtts_set_decode_plane_rows", "idxtts_get_decode_plane_rows", "idxtts_s2mel_set_overlap", "idxtts_s2mel_get_overlap",
    "idxtts_s2mel_estimator", "idxtts_s2mel_regulate", "idxtts_cond_create", "idxtts_cond_workspace_bytes", "idxtts_cond_forward", "idxtts_emovec_merge",
    "idxtts_gpt_graph_cache_entries", "idxtts_w2vbert_create", "idxtts_w2vbert_workspace_bytes", "idxtts_w2vbert_forward",
    "idxtts_repcodec_create", "idxtts_repcodec_workspace_bytes", "idxtts_repcodec_quantize",
    "idxtts_melspec_create", "idxtts_melspec_frames", "idxtts_melspec_workspace_bytes", "idxtts_melspec_forward",
    "idxtts_campplus_create", "idxtts_campplus_workspace_bytes", "idxtts_campplus_forward",
]


class BigVGANConfigC(ctypes.Structure):
    _fields_ = [
        ("num_mels", c_int), ("upsample_initial_channel", c_int), ("num_upsamples", c_int),
        ("upsample_rates", c_int * 8), ("upsample_kernel_sizes", c_int * 8),
        ("num_kernels", c_int), ("resblock_kernel_sizes", c_int * 4),
        ("resblock_dilations", (c_int * 3) * 4),
    ]


class GPTConfigC(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("model_dim", "heads", "layers", "number_mel_codes", "number_text_tokens",
                                     "start_mel_token", "stop_mel_token", "mel_pos_len", "text_pos_len")]


class SamplingC(ctypes.Structure):          # idxtts_sampling (include/idxtts.h)
    _fields_ = [("mode", c_int), ("temperature", c_float), ("top_k", c_int), ("top_p", c_float), ("exp_noise", c_void_p),
                ("seed", ctypes.c_ulonglong)]


class CondConfigC(ctypes.Structure):         # idxtts_cond_config (include/idxtts.h)
    _fields_ = [(n, c_int) for n in ("input_size", "output_size", "linear_units", "attention_heads", "num_blocks", "cnn_kernel",
                                     "perceiver_dim", "num_latents", "perceiver_depth", "perceiver_dim_head", "perceiver_mult",
                                     "emotion", "model_dim")]


class W2VBertConfigC(ctypes.Structure):      # idxtts_w2vbert_config (include/idxtts.h)
    _fields_ = [(n, c_int) for n in ("input_dim", "hidden_size", "num_heads", "intermediate_size", "num_layers", "left_max", "right_max",
                                     "conv_kernel")] + [("layer_norm_eps", c_float)]


class RepCodecConfigC(ctypes.Structure):     # idxtts_repcodec_config (include/idxtts.h)
    _fields_ = [(n, c_int) for n in ("hidden_size", "codebook_size", "codebook_dim", "vocos_dim", "vocos_intermediate_dim", "vocos_num_layers")]


class MelSpecConfigC(ctypes.Structure):      # idxtts_melspec_config (include/idxtts.h)
    _fields_ = [(n, c_int) for n in ("n_fft", "hop_size", "win_size", "num_mels")]


class CamPPlusConfigC(ctypes.Structure):     # idxtts_campplus_config (include/idxtts.h)
    _fields_ = [(n, c_int) for n in ("feat_dim", "embedding_size", "m_channels", "growth_rate", "bn_size", "init_channels", "num_blocks")] + [
        ("block_layers", c_int * 4), ("block_dilation", c_int * 4)]


class BeamC(ctypes.Structure):               # idxtts_beam (include/idxtts.h)
    _fields_ = [("num_beams", c_int), ("do_sample", c_int), ("temperature", c_float), ("top_k", c_int), ("top_p", c_float),
                ("length_penalty", c_float), ("early_stopping", c_int), ("exp_noise", c_void_p), ("seed", ctypes.c_ulonglong)]


class S2MelConfigC(ctypes.Structure):
    _fields_ = ([(n, c_int) for n in ("hidden_dim", "num_heads", "depth", "in_channels", "content_dim", "style_dim", "wn_hidden",
                                      "wn_layers", "wn_kernel", "wn_dilation_rate", "lr_channels", "lr_in_channels",
                                      "lr_num_convs", "gpt_dim")]
                + [("gpt_layer_dims", c_int * 3)]
                + [(n, c_int) for n in ("codebook_size", "codebook_dim", "codec_hidden")] + [("norm_eps", c_float)])


_lib = None


def load() -> ctypes.CDLL:
    """Load (once) and return the library; raise if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libidxtts_hip.so not found at {LIB_PATH}: build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C index-tts_amd/csrc`). "
            "There is no CPU/PyTorch fallback for the HIP path.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.idxtts_version.restype = c_int
    lib.idxtts_last_error.restype = c_char_p
    lib.idxtts_aa_act_fwd.argtypes = [c_void_p] * 6 + [c_int] * 4 + [c_void_p]
    lib.idxtts_conv1d_create.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, POINTER(c_void_p)]
    lib.idxtts_conv1d_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                      c_float, c_int, c_void_p]
    lib.idxtts_conv1d_destroy.argtypes = [c_void_p]
    lib.idxtts_ctx_load_tensor.argtypes = [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int]
    lib.idxtts_ctx_finalize.argtypes = [c_void_p]
    lib.idxtts_ctx_get_tensor.argtypes = [c_void_p, c_char_p, c_void_p, c_size_t]
    lib.idxtts_fp8_e4m3_decode.argtypes = [ctypes.c_ubyte]
    lib.idxtts_fp8_e4m3_decode.restype = ctypes.c_float
    lib.idxtts_fp8_e4m3_encode.argtypes = [ctypes.c_float]
    lib.idxtts_fp8_e4m3_encode.restype = ctypes.c_ubyte
    lib.idxtts_gpt_quantize_weights.argtypes = [c_void_p, c_int]
    lib.idxtts_gpt_set_kv_format.argtypes = [c_void_p, c_int]
    lib.idxtts_gpt_get_kv_format.argtypes = [c_void_p]
    lib.idxtts_ctx_destroy.argtypes = [c_void_p]
    lib.idxtts_bigvgan_create.argtypes = [POINTER(BigVGANConfigC), POINTER(c_void_p)]
    lib.idxtts_bigvgan_workspace_bytes.argtypes = [c_void_p, c_int, c_int]
    lib.idxtts_bigvgan_workspace_bytes.restype = c_size_t
    lib.idxtts_bigvgan_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_int, c_int,
                                       c_void_p, c_void_p]
    lib.idxtts_bigvgan_fwd_ragged.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_int, c_void_p]
    c_long = ctypes.c_long
    lib.idxtts_linear_create.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, POINTER(c_void_p)]
    lib.idxtts_linear_fwd.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    lib.idxtts_set_gemm_mode.argtypes = [c_int]
    lib.idxtts_linear_destroy.argtypes = [c_void_p]
    lib.idxtts_attention_fwd.argtypes = [c_void_p] * 4 + [c_long, c_int, c_long, c_int, c_long, c_int, c_int, c_int, c_int, c_int,
                                                          c_int, c_void_p, c_void_p, c_float, c_void_p]
    lib.idxtts_attention_bf16x3_fwd.argtypes = lib.idxtts_attention_fwd.argtypes
    lib.idxtts_attention_relkey_fwd.argtypes = [c_void_p] * 4 + [c_long, c_int, c_long, c_int, c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_int, c_int,
                                                                 c_int, c_void_p]
    lib.idxtts_layernorm_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]
    lib.idxtts_gpt_create.argtypes = [POINTER(GPTConfigC), POINTER(c_void_p)]
    lib.idxtts_gpt_workspace_bytes.argtypes = [c_void_p, c_int, c_int, c_int]
    lib.idxtts_gpt_workspace_bytes.restype = c_size_t
    lib.idxtts_gpt_embed.argtypes = [c_void_p, c_void_p, c_int] + [c_void_p] * 7
    lib.idxtts_gpt_generate.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, POINTER(c_int),
                                        c_void_p, c_void_p, c_size_t, c_int, c_void_p]
    lib.idxtts_gpt_generate_forced.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, POINTER(c_int),
                                               c_void_p, c_void_p, c_size_t, c_void_p]
    lib.idxtts_gpt_generate_sampled.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, POINTER(SamplingC), c_void_p,
                                                POINTER(c_int), c_void_p, c_void_p, c_size_t, c_int, c_void_p]
    lib.idxtts_gpt_beam_workspace_bytes.argtypes = [c_void_p, c_int, c_int, c_int, c_int]
    lib.idxtts_gpt_beam_workspace_bytes.restype = c_size_t
    lib.idxtts_gpt_generate_beam.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, POINTER(BeamC), c_void_p,
                                             POINTER(c_int), c_void_p, c_size_t, c_int, c_void_p]
    lib.idxtts_gpt_latent.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.idxtts_s2mel_create.argtypes = [POINTER(S2MelConfigC), POINTER(c_void_p)]
    lib.idxtts_s2mel_cond_workspace_bytes.argtypes = [c_void_p, c_int, c_int, c_int]
    lib.idxtts_s2mel_cond_workspace_bytes.restype = c_size_t
    lib.idxtts_s2mel_prepare_cond.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                              c_void_p, c_size_t, c_void_p]
    lib.idxtts_s2mel_cfm_workspace_bytes.argtypes = [c_void_p, c_int, c_int, c_int]
    lib.idxtts_s2mel_cfm_workspace_bytes.restype = c_size_t
    lib.idxtts_s2mel_cfm.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int, c_float, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]
    lib.idxtts_s2mel_regulate.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.idxtts_s2mel_estimator.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int, c_int, c_void_p, c_size_t, c_void_p]
    lib.idxtts_cond_create.argtypes = [POINTER(CondConfigC), POINTER(c_void_p)]
    lib.idxtts_cond_workspace_bytes.argtypes = [c_void_p, c_int, c_int]
    lib.idxtts_cond_workspace_bytes.restype = c_size_t
    lib.idxtts_cond_forward.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.idxtts_emovec_merge.argtypes = [c_void_p, c_void_p, c_void_p, c_float, c_size_t, c_void_p]
    lib.idxtts_campplus_create.argtypes = [POINTER(CamPPlusConfigC), POINTER(c_void_p)]
    lib.idxtts_campplus_workspace_bytes.argtypes = [c_void_p, c_int]
    lib.idxtts_campplus_workspace_bytes.restype = c_size_t
    lib.idxtts_campplus_forward.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.idxtts_melspec_create.argtypes = [POINTER(MelSpecConfigC), POINTER(c_void_p)]
    lib.idxtts_melspec_frames.argtypes = [c_void_p, c_int]
    lib.idxtts_melspec_workspace_bytes.argtypes = [c_void_p, c_int, c_int]
    lib.idxtts_melspec_workspace_bytes.restype = c_size_t
    lib.idxtts_melspec_forward.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.idxtts_repcodec_create.argtypes = [POINTER(RepCodecConfigC), POINTER(c_void_p)]
    lib.idxtts_repcodec_workspace_bytes.argtypes = [c_void_p, c_int, c_int]
    lib.idxtts_repcodec_workspace_bytes.restype = c_size_t
    lib.idxtts_repcodec_quantize.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.idxtts_gpt_graph_cache_entries.argtypes = [c_void_p]
    lib.idxtts_w2vbert_create.argtypes = [POINTER(W2VBertConfigC), POINTER(c_void_p)]
    lib.idxtts_w2vbert_workspace_bytes.argtypes = [c_void_p, c_int, c_int]
    lib.idxtts_w2vbert_workspace_bytes.restype = c_size_t
    lib.idxtts_w2vbert_forward.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]
    lib.idxtts_profile_enable.argtypes = [c_int]
    lib.idxtts_profile_kernel_name.argtypes = [c_int]
    lib.idxtts_profile_kernel_name.restype = c_char_p
    lib.idxtts_profile_read.argtypes = [c_int, POINTER(ctypes.c_double), POINTER(ctypes.c_double),
                                        POINTER(ctypes.c_double), POINTER(ctypes.c_long)]
    for name in SYMBOLS:
        getattr(lib, name)   # AttributeError if the .so lacks a declared symbol
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().idxtts_last_error()
        raise RuntimeError("libidxtts_hip: " + (msg.decode("utf-8", "replace") if msg else f"error {rc}"))


def ptr(t) -> c_void_p:
    """Device (or host) pointer of a contiguous float32 torch tensor, or NULL for None."""
    if t is None:
        return c_void_p(0)
    return c_void_p(t.data_ptr())


def current_stream() -> c_void_p:
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def load_state_dict(ctx: c_void_p, state_dict, before_finalize=None) -> None:
    """Hand every tensor of a (reference-layout) state dict to the context, then finalize it.
    before_finalize(ctx): optional hook between the last tensor and finalize (weight quantisation, read-back)."""
    import numpy as np
    import torch
    lib = load()
    for name, value in state_dict.items():
        if isinstance(value, torch.Tensor):
            arr = value.detach().to(dtype=torch.float32).contiguous()
            shape = tuple(arr.shape)
            dptr = c_void_p(arr.data_ptr())
        else:
            arr = np.ascontiguousarray(value, dtype=np.float32)
            shape = arr.shape
            dptr = c_void_p(arr.ctypes.data)
        cshape = (c_int64 * max(1, len(shape)))(*shape)
        check(lib.idxtts_ctx_load_tensor(ctx, name.encode(), dptr, cshape, len(shape)))
    if before_finalize is not None:
        before_finalize(ctx)
    check(lib.idxtts_ctx_finalize(ctx))


def get_tensor(ctx: c_void_p, name: str, shape):
    """Staged (not yet finalized) tensor `name` as a numpy fp32 array of `shape`."""
    import numpy as np
    out = np.empty(shape, dtype=np.float32)
    check(load().idxtts_ctx_get_tensor(ctx, name.encode(), c_void_p(out.ctypes.data), out.size))
    return out


def profile_enable(on: bool) -> None:
    check(load().idxtts_profile_enable(int(on)))


def profile_event_overhead(launches: int = 200) -> float:
    """ms an event pair reads around one launch of an empty kernel on the current stream (fixed cost of the per-launch timing)."""
    out = ctypes.c_double()
    lib = load()
    lib.idxtts_profile_event_overhead.argtypes = [c_void_p, c_int, POINTER(ctypes.c_double)]
    check(lib.idxtts_profile_event_overhead(current_stream(), int(launches), ctypes.byref(out)))
    return out.value


def profile_read() -> dict:
    """{kernel family: {"ms", "flops", "bytes", "launches"}} accumulated since profile_enable(True)."""
    lib = load()
    out = {}
    for i in range(lib.idxtts_profile_num_kernels()):
        ms, fl, by, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_long()
        check(lib.idxtts_profile_read(i, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by), ctypes.byref(n)))
        if n.value:
            out[lib.idxtts_profile_kernel_name(i).decode()] = {
                "ms": ms.value, "flops": fl.value, "bytes": by.value, "launches": n.value}
    return out


GEMM_F32, GEMM_BF16X3 = 0, 1


class StreamWorkspaces:
    """Grow-only device workspaces, one per HIP stream: calls on one stream are ordered and may share a buffer, calls on different
    streams (several host threads, serving.BatchPipeline) may not.  Least-recently-used streams beyond `max_entries` are dropped."""

    def __init__(self, max_entries: int = 8):
        import collections
        import threading
        self.max_entries = max_entries
        self._ws = collections.OrderedDict()
        self._lock = threading.Lock()

    def get(self, need: int, device):
        import torch
        if need <= 0:
            raise RuntimeError("workspace query returned 0")
        device = torch.device(device)
        key = (device.index, int(torch.cuda.current_stream(device).cuda_stream))
        with self._lock:
            ws = self._ws.pop(key, None)
            if ws is None or ws.numel() < need:
                ws = None
                ws = torch.empty(need, dtype=torch.uint8, device=device)
            self._ws[key] = ws
            while len(self._ws) > self.max_entries:
                self._ws.popitem(last=False)
        return ws


def set_gemm_mode(mode: int) -> None:
    """0 = exact fp32 MFMA everywhere; 1 (library default) = split-bf16 for the GEMM-shaped passes with M >= 256."""
    check(load().idxtts_set_gemm_mode(int(mode)))


def get_gemm_mode() -> int:
    return int(load().idxtts_get_gemm_mode())


def set_decode_geometry(narrow: bool) -> None:
    """Decode GEMVs as 512-thread workgroups (narrow: for decodes that run beside acoustic stages, serving.py) or 1024-thread ones
    (default: fastest alone).  Process-wide, set before generating; results differ in the last bits between the two."""
    check(load().idxtts_set_decode_geometry(int(bool(narrow))))


def get_decode_geometry() -> bool:
    return bool(load().idxtts_get_decode_geometry())


def release_stream(stream) -> None:
    """Hand a torch.cuda.Stream that is being retired back to the library (per-stream scratch and side streams; include/idxtts.h)."""
    check(load().idxtts_release_stream(c_void_p(int(stream.cuda_stream))))


def set_decode_plane_rows(min_rows: int) -> None:
    """From how many decode rows on (compact weight streams) the decode step runs on the plane GEMV (include/idxtts.h): 0 = default (17),
    5..64, 65 = off.  Process-wide; set before generating."""
    check(load().idxtts_set_decode_plane_rows(int(min_rows)))


def get_decode_plane_rows() -> int:
    return int(load().idxtts_get_decode_plane_rows())


def set_s2mel_overlap(on: bool) -> None:
    """The CFM solver's two CFG halves on two streams (on) or as one stacked 2B batch on the caller's stream (off, the default: faster
    beside concurrent decode chains, profiles/README.md "Round 3").  Same result bit for bit."""
    check(load().idxtts_s2mel_set_overlap(int(bool(on))))


def get_s2mel_overlap() -> bool:
    return bool(load().idxtts_s2mel_get_overlap())
