"""Synthetic inputs of SURVEY.md 8(d), shared by bench.py and the tests (host numpy, not part of the path):
Gaussian initial half-spectrum with P(k) = A k^-2 exp(-(k/k_s)^2), k_s = pi/2 in cell units, A set so that the
rms 1LPT displacement at a = 1 is `rms_disp` cells."""
import numpy as np
import scipy.fft


def init_mesh(n, seed=0, rms_disp=2.0, dtype=np.complex64):
    shape = (n, n, n) if np.ndim(n) == 0 else tuple(n)
    g = np.random.default_rng(seed).standard_normal(shape)
    spec = scipy.fft.rfftn(g, workers=-1)
    del g
    kx = 2 * np.pi * np.fft.fftfreq(shape[0])[:, None, None]
    ky = 2 * np.pi * np.fft.fftfreq(shape[1])[None, :, None]
    kz = 2 * np.pi * np.fft.rfftfreq(shape[2])[None, None, :]
    kk = kx ** 2 + ky ** 2 + kz ** 2
    kk[0, 0, 0] = 1.0
    pk = np.exp(-kk / (np.pi / 2) ** 2) / kk
    pk[0, 0, 0] = 0.0
    spec *= np.sqrt(pk)
    # mean |psi|^2 = sum_k |delta_k|^2 / k^2 / M^2 over the full spectrum (Parseval), psi_k = i k / k^2 delta_k
    w = np.full(shape[2] // 2 + 1, 2.0)
    w[0] = 1.0
    if shape[2] % 2 == 0:
        w[-1] = 1.0
    M = float(np.prod(shape))
    psi2 = float(np.sum((spec.real ** 2 + spec.imag ** 2) / kk * w)) / M ** 2
    spec *= rms_disp / np.sqrt(psi2)
    return spec.astype(dtype)
