"""Prompt audio from a file (indextts_amd/audioio.py): WAV decoding, the torchaudio-style windowed-sinc resampler (restated from the
published algorithm -- torchaudio is not in this image, so these are property checks, not a pin) and the reference's cut rule."""
import math
import os
import struct
import sys
import wave

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "index-tts_amd"))
from indextts_amd import audioio  # noqa: E402


def _write_pcm16(path, x, sr, nch=1):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(nch)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())


def test_read_wav_pcm16_stereo_and_float32(tmp_path):
    rng = np.random.default_rng(0)
    x = (rng.random((1000, 2)) * 2 - 1).astype(np.float32) * 0.9
    _write_pcm16(tmp_path / "s.wav", x.reshape(-1), 22050, nch=2)
    y, sr = audioio.read_wav(str(tmp_path / "s.wav"))
    assert sr == 22050 and y.shape == (2, 1000)
    assert np.abs(y.T - x).max() <= 2.0 / 32768             # truncation to int16 by the writer + the 32767 / 32768 scale
    # IEEE float file written by hand (the wave module cannot): format tag 3
    f = (rng.random(777) * 2 - 1).astype("<f4")
    body = f.tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 3, 1, 16000, 64000, 4, 32)
    (tmp_path / "f.wav").write_bytes(hdr + b"data" + struct.pack("<I", len(body)) + body)
    y, sr = audioio.read_wav(str(tmp_path / "f.wav"))
    assert sr == 16000 and y.shape == (1, 777) and np.array_equal(y[0], f)


def test_read_wav_pcm24_and_8(tmp_path):
    v = np.array([0, 1, -1, (1 << 23) - 1, -(1 << 23), 12345, -54321], np.int32)
    raw = b"".join(int(a & 0xFFFFFF).to_bytes(3, "little") for a in v)
    with wave.open(str(tmp_path / "p24.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(3); w.setframerate(8000); w.writeframes(raw)
    y, sr = audioio.read_wav(str(tmp_path / "p24.wav"))
    assert sr == 8000 and np.allclose(y[0], v / float(1 << 23), atol=0)
    with wave.open(str(tmp_path / "p8.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(1); w.setframerate(8000); w.writeframes(bytes([0, 128, 255]))
    y, _ = audioio.read_wav(str(tmp_path / "p8.wav"))
    assert np.allclose(y[0], [-1.0, 0.0, 127 / 128])


@pytest.mark.parametrize("orig,new", [(22050, 16000), (16000, 22050), (44100, 22050), (48000, 16000), (24000, 22050)])
def test_sinc_resample_length_dc_and_tone(orig, new):
    n = 3 * orig // 10 + 17
    g = math.gcd(orig, new)
    # output length: ceil(new * length / orig) on the reduced rates (torchaudio _apply_sinc_resample_kernel)
    y = audioio.sinc_resample(np.ones(n, np.float32), orig, new)
    assert y.shape == (math.ceil((new // g) * n / (orig // g)),)
    mid = y[len(y) // 4: 3 * len(y) // 4]
    assert np.abs(mid - 1.0).max() < 2e-3                     # unit DC gain away from the zero-padded ends
    # a tone well inside both pass bands keeps its frequency and amplitude
    f0 = 1000.0
    t = np.arange(n) / orig
    y = audioio.sinc_resample(np.sin(2 * np.pi * f0 * t).astype(np.float32), orig, new)
    tt = np.arange(len(y)) / new
    ref = np.sin(2 * np.pi * f0 * tt)
    sl = slice(len(y) // 4, 3 * len(y) // 4)
    assert np.abs(y[sl] - ref[sl]).max() < 5e-3
    # a tone above the new Nyquist (when downsampling) is removed
    if new < orig:
        fh = 0.5 * new * 1.25
        if fh < 0.5 * orig:
            yh = audioio.sinc_resample(np.sin(2 * np.pi * fh * t).astype(np.float32), orig, new)
            assert np.abs(yh[sl]).max() < 0.05


def test_sinc_resample_kernel_shape_and_identity():
    k, width = audioio.sinc_resample_kernel(441, 320)          # 22 050 -> 16 000 reduced by gcd 50
    assert width == math.ceil(6 * 441 / (320 * 0.99)) and k.shape == (320, 2 * width + 441) and k.dtype == np.float32
    # phase 0 is centred on the input sample `width`: its largest tap sits there and equals base / orig
    assert int(np.argmax(k[0])) == width and abs(k[0, width] - 320 * 0.99 / 441) < 1e-6
    x = np.random.default_rng(1).standard_normal((2, 500)).astype(np.float32)
    assert audioio.sinc_resample(x, 16000, 16000) is x or np.array_equal(audioio.sinc_resample(x, 16000, 16000), x)
    y = audioio.sinc_resample(x, 22050, 16000)                  # batch axis kept, rows independent
    assert y.shape[0] == 2 and np.array_equal(y[1], audioio.sinc_resample(x[1], 22050, 16000))


def test_load_prompt_audio_cut_and_rates(tmp_path):
    sr = 22050
    n = 16 * sr + 123                                          # longer than the 15 s cut
    x = 0.5 * np.sin(2 * np.pi * 440 * np.arange(n) / sr).astype(np.float32)
    _write_pcm16(tmp_path / "p.wav", x, sr)
    a, r = audioio.load_and_cut_audio(str(tmp_path / "p.wav"), 15)
    assert r == 22050 and a.shape == (1, 15 * 22050)           # int(max_seconds * sr) samples (infer_v2.py:516-522)
    pa = audioio.load_prompt_audio(str(tmp_path / "p.wav"))
    assert pa.audio_22k.shape == (15 * 22050,) and pa.audio_16k.shape == (math.ceil(320 * 15 * 22050 / 441),)
    assert np.abs(pa.audio_22k - a[0]).max() == 0.0            # 22 050 -> 22 050 is the identity, like torchaudio's transform
    e = audioio.load_prompt_audio(str(tmp_path / "p.wav"), emotion=True)
    assert e.audio_22k is None and e.audio_16k.shape == (15 * 16000,)
    # stereo file: channel mean (librosa to_mono)
    st = np.stack([x[:1000], -x[:1000]], 1).reshape(-1)
    _write_pcm16(tmp_path / "st.wav", st, sr, nch=2)
    m, _ = audioio.load_and_cut_audio(str(tmp_path / "st.wav"), 15)
    assert np.abs(m).max() < 1e-4
