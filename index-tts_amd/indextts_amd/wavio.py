"""Minimal PCM-16 wav writer (the reference uses torchaudio.save, infer_v2.py:912; torchaudio is not on the path here)."""
import wave

import numpy as np


def write_wav_int16(path: str, samples: np.ndarray, sampling_rate: int = 22050) -> None:
    data = np.ascontiguousarray(samples, dtype="<i2").reshape(-1)
    with wave.open(path, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sampling_rate)
        f.writeframes(data.tobytes())
