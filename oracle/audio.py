"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's log-mel spectrogram of the prompt audio.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Follows  mel_spectrogram   /root/reference/indextts/s2mel/modules/audio.py:45-83  (arguments: infer_v2.py:291-301)
with the mel basis passed in (the reference builds it with librosa.filters.mel, absent here).  Pinned by tests/golden/melspec.npz: the
reference's own function run with `librosa.filters.mel` supplied by transformers.audio_utils.mel_filter_bank(norm="slaney",
mel_scale="slaney") -- a third-party port of that librosa function, not this repo's restatement of it."""
import torch


def mel_spectrogram(y, mel_basis, n_fft=1024, hop_size=256, win_size=1024):
    pad = int((n_fft - hop_size) / 2)
    y = torch.nn.functional.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    spec = torch.view_as_real(torch.stft(y, n_fft, hop_length=hop_size, win_length=win_size, window=torch.hann_window(win_size), center=False,
                                         pad_mode="reflect", normalized=False, onesided=True, return_complex=True))
    spec = torch.sqrt(spec.pow(2).sum(-1) + 1e-9)
    return torch.log(torch.clamp(torch.matmul(mel_basis, spec), min=1e-5))
