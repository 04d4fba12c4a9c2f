"""Mel filterbank, Hann window and FFT twiddle tables for the log-mel frontend.

The reference builds its (n_freqs, n_mels) matrix with the third-party call
``torchaudio.functional.melscale_fbanks(n_freqs=257, f_min=0, f_max=8000,
n_mels=80, sample_rate=16000, norm='slaney', mel_scale='slaney')``
(reference: speechcatcher/model/frontend/stft_frontend.py:73-81).  torchaudio
is an unpinned dependency that is absent from /root/reference and from this
image, so the published closed-form (Slaney mel scale + Slaney area
normalisation, the librosa / torchaudio definition) is restated here in
float32 torch ops.  "Parity unpinned" at this one boundary: see DESIGN.md.
"""
import math

import numpy as np
import torch


def _hz_to_mel_slaney(f: float) -> float:
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    if f >= min_log_hz:
        mels = min_log_mel + math.log(f / min_log_hz) / logstep
    return mels


def _mel_to_hz_slaney(mels: torch.Tensor) -> torch.Tensor:
    f_sp = 200.0 / 3
    freqs = f_sp * mels
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    log_t = mels >= min_log_mel
    freqs[log_t] = min_log_hz * torch.exp(logstep * (mels[log_t] - min_log_mel))
    return freqs


def melscale_fbanks_slaney(n_freqs: int, f_min: float, f_max: float,
                           n_mels: int, sample_rate: int) -> torch.Tensor:
    """(n_freqs, n_mels) float32 triangular filters, slaney scale + norm."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = _hz_to_mel_slaney(f_min)
    m_max = _hz_to_mel_slaney(f_max)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = _mel_to_hz_slaney(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.max(torch.zeros(1), torch.min(down, up))
    enorm = 2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])
    return (fb * enorm.unsqueeze(0)).contiguous()


def hann_window_periodic(win_length: int) -> torch.Tensor:
    """torch.hann_window(win_length) (periodic=True default).

    reference: speechcatcher/model/frontend/stft_frontend.py:68
    """
    return torch.hann_window(win_length)


def fft_twiddles(n_fft: int) -> np.ndarray:
    """(n_fft/2, 2) float32 table of (cos, -sin)(2*pi*k/n_fft), k < n_fft/2,
    computed in float64 then rounded once."""
    k = np.arange(n_fft // 2, dtype=np.float64)
    ang = 2.0 * np.pi * k / n_fft
    return np.stack([np.cos(ang), -np.sin(ang)], axis=1).astype(np.float32)


def positional_encoding_table(max_len: int, d_model: int) -> torch.Tensor:
    """Sinusoidal table (max_len, d_model), float32, same op order as the
    reference so the values are bit-identical.

    reference: speechcatcher/model/layers/positional_encoding.py:38-48
    """
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(
        torch.arange(0, d_model, 2, dtype=torch.float32)
        * -(math.log(10000.0) / d_model)
    )
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe
