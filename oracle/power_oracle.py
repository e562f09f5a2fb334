"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Linear power spectrum behind `lin_power(cosmo, kpow=None)`
(montecosmo/bricks.py:69-79): jax_cosmo's `linear_matter_power` with the Eisenstein-Hu transfer function.  jax_cosmo is a
third-party dependency absent from the reference repository and from this container, so this is a restatement of the
PUBLISHED fit (Eisenstein & Hu 1998, ApJ 496, 605; "parity unpinned" against jax_cosmo's own arithmetic, in particular
its sigma8 quadrature).  Pinned by the closed-form limits in tests/test_oracle_known_answers.py.  Written as one scalar
function of k (a loop over the table), deliberately unlike the vectorised product code."""
import math

import numpy as np

TCMB = 2.7255


def _eh_scalar(k_hmpc, Om, Ob, h):
    if k_hmpc <= 0.0:
        return 1.0
    k = k_hmpc * h
    wm, wb = Om * h * h, Ob * h * h
    fb = Ob / Om
    fc = 1.0 - fb
    t27 = TCMB / 2.7
    zeq = 2.5e4 * wm * t27 ** -4
    keq = 0.0746 * wm * t27 ** -2
    zd = 1291.0 * wm ** 0.251 / (1.0 + 0.659 * wm ** 0.828) * (1.0 + 0.313 * wm ** -0.419 * (1.0 + 0.607 * wm ** 0.674) * wb ** (0.238 * wm ** 0.223))
    Rd = 31.5 * wb * t27 ** -4 * 1000.0 / zd
    Req = 31.5 * wb * t27 ** -4 * 1000.0 / zeq
    s = 2.0 / (3.0 * keq) * math.sqrt(6.0 / Req) * math.log((math.sqrt(1.0 + Rd) + math.sqrt(Rd + Req)) / (1.0 + math.sqrt(Req)))
    ksilk = 1.6 * wb ** 0.52 * wm ** 0.73 * (1.0 + (10.4 * wm) ** -0.95)
    q = k / (13.41 * keq)

    def t0(alpha, beta):
        c = 14.2 / alpha + 386.0 / (1.0 + 69.9 * q ** 1.08)
        ln = math.log(math.e + 1.8 * beta * q)
        return ln / (ln + c * q * q)

    ac = ((46.9 * wm) ** 0.670 * (1.0 + (32.1 * wm) ** -0.532)) ** -fb * ((12.0 * wm) ** 0.424 * (1.0 + (45.0 * wm) ** -0.582)) ** -(fb ** 3)
    bc = 1.0 / (1.0 + 0.944 / (1.0 + (458.0 * wm) ** -0.708) * (fc ** ((0.395 * wm) ** -0.0266) - 1.0))
    f = 1.0 / (1.0 + (k * s / 5.4) ** 4)
    tc = f * t0(1.0, bc) + (1.0 - f) * t0(ac, bc)
    y = (1.0 + zeq) / (1.0 + zd)
    sq = math.sqrt(1.0 + y)
    ab = 2.07 * keq * s * (1.0 + Rd) ** -0.75 * y * (-6.0 * sq + (2.0 + 3.0 * y) * math.log((sq + 1.0) / (sq - 1.0)))
    bb = 0.5 + fb + (3.0 - 2.0 * fb) * math.sqrt((17.2 * wm) ** 2 + 1.0)
    bnode = 8.41 * wm ** 0.435
    st = s / (1.0 + (bnode / (k * s)) ** 3) ** (1.0 / 3.0)
    x = k * st
    j0 = math.sin(x) / x if x != 0 else 1.0
    tb = (t0(1.0, 1.0) / (1.0 + (k * s / 5.2) ** 2) + ab / (1.0 + (bb / (k * s)) ** 3) * math.exp(-(k / ksilk) ** 1.4)) * j0
    return fb * tb + fc * tc


def eisenstein_hu(cosmo, ks):
    return np.array([_eh_scalar(float(k), cosmo.Omega_m, cosmo.Omega_b, cosmo.h) for k in np.atleast_1d(ks)])


def sound_horizon(cosmo):
    """EH98 eq. 6, Mpc."""
    h = cosmo.h
    wm, wb, t27 = cosmo.Omega_m * h * h, cosmo.Omega_b * h * h, TCMB / 2.7
    zeq, keq = 2.5e4 * wm * t27 ** -4, 0.0746 * wm * t27 ** -2
    zd = 1291.0 * wm ** 0.251 / (1.0 + 0.659 * wm ** 0.828) * (1.0 + 0.313 * wm ** -0.419 * (1.0 + 0.607 * wm ** 0.674) * wb ** (0.238 * wm ** 0.223))
    Rd, Req = 31.5 * wb * t27 ** -4 * 1000.0 / zd, 31.5 * wb * t27 ** -4 * 1000.0 / zeq
    return 2.0 / (3.0 * keq) * math.sqrt(6.0 / Req) * math.log((math.sqrt(1.0 + Rd) + math.sqrt(Rd + Req)) / (1.0 + math.sqrt(Req)))


def sigma8_of(ks, pows, r=8.0):
    """Top-hat rms at r Mpc/h by Simpson's rule in ln k (ks log-spaced, odd count)."""
    from scipy.integrate import simpson
    x = ks * r
    w = 3.0 * (np.sin(x) - x * np.cos(x)) / x ** 3
    return float(np.sqrt(simpson(ks ** 3 * pows * w * w, x=np.log(ks)) / (2.0 * np.pi ** 2)))


def lin_power_table(cosmo, n_interp=256):
    """(ks, pows normalised to sigma8 = 1) on logspace(-4, 1, n_interp) h/Mpc (bricks.py:73)."""
    ks = np.logspace(-4, 1, n_interp)
    kf = np.logspace(-5, 2, 4097)
    pf = kf ** cosmo.n_s * eisenstein_hu(cosmo, kf) ** 2
    amp = 1.0 / sigma8_of(kf, pf) ** 2
    return ks, amp * ks ** cosmo.n_s * eisenstein_hu(cosmo, ks) ** 2
