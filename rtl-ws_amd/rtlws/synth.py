"""Seeded synthetic u8 IQ: the stand-in for dongle data.

tone_noise_iq: per frame a complex tone (amplitude 0.6, random normalised
frequency in [-0.5, 0.5)) plus Gaussian noise (sigma 0.05 per component),
quantised like an RTL2832U sample: clip(round(x*128 + 128), 0, 255).
uniform_iq: uniformly random bytes (worst case for nothing in particular,
but every bit pattern appears).
"""
import numpy as np


def tone_noise_iq(nframes, N, seed=1234, amp=0.6, sigma=0.05):
    rng = np.random.default_rng(seed)
    f = rng.uniform(-0.5, 0.5, size=(nframes, 1))
    ph = rng.uniform(0, 2 * np.pi, size=(nframes, 1))
    n = np.arange(N)[None, :]
    x = amp * np.exp(1j * (2 * np.pi * f * n + ph))
    x = x + sigma * (rng.standard_normal((nframes, N)) + 1j * rng.standard_normal((nframes, N)))
    iq = np.empty((nframes, N, 2), dtype=np.uint8)
    iq[..., 0] = np.clip(np.round(x.real * 128 + 128), 0, 255).astype(np.uint8)
    iq[..., 1] = np.clip(np.round(x.imag * 128 + 128), 0, 255).astype(np.uint8)
    return iq


def pure_tone_iq(nframes, N, seed=99, amp=0.9):
    """Full-scale tone with quantisation noise only (widest dynamic range)."""
    return tone_noise_iq(nframes, N, seed=seed, amp=amp, sigma=0.0)


def uniform_iq(nframes, N, seed=4321):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(nframes, N, 2), dtype=np.uint8)


def hann(N):
    """Periodic Hann, w[n] = 0.5 - 0.5 cos(2 pi n / N) (build extension; the
    reference applies no window, src/spectrum.c:54-60)."""
    n = np.arange(N, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / N)
