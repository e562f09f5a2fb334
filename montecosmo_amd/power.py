"""Linear matter power spectrum without a tabulation: what `lin_power(cosmo, kpow=None)` (montecosmo/bricks.py:69-79)
obtains from jax_cosmo (`power.linear_matter_power` with the Eisenstein & Hu transfer function, a third-party dependency
that is not part of the reference repository).  Restated from the published fit -- Eisenstein & Hu 1998, ApJ 496, 605,
eqs. 2-7, 10-12, 14-24 (cold-dark-matter + baryon transfer function with acoustic oscillations and Silk damping) -- with
P(k) = A k^n_s T(k)^2 normalised so that the top-hat variance at 8 Mpc/h is sigma8^2.

Host float64 on a 256-point k table (bricks.py:73: logspace(-4, 1, 256) h/Mpc): set-up sized, the mesh-sized work is the
device interpolation `mcpm_power_mult_f32`.
"""
import numpy as np

T_CMB = 2.7255   # K (jax_cosmo.constants.tcmb)


def eisenstein_hu_transfer(cosmo, k):
    """T(k), k in h/Mpc, for the cosmology's Omega_m, Omega_b, h (EH98 full fit, no neutrinos)."""
    k = np.asarray(k, dtype=np.float64) * cosmo.h                      # 1/Mpc
    om, ob = cosmo.Omega_m * cosmo.h ** 2, cosmo.Omega_b * cosmo.h ** 2
    fb = cosmo.Omega_b / cosmo.Omega_m
    fc = 1.0 - fb
    th = T_CMB / 2.7
    # eqs. 2-6: equality, drag epoch, sound horizon
    z_eq = 2.50e4 * om / th ** 4
    k_eq = 7.46e-2 * om / th ** 2
    b1 = 0.313 * om ** -0.419 * (1.0 + 0.607 * om ** 0.674)
    b2 = 0.238 * om ** 0.223
    z_d = 1291.0 * om ** 0.251 / (1.0 + 0.659 * om ** 0.828) * (1.0 + b1 * ob ** b2)
    R = lambda z: 31.5 * ob / th ** 4 * (1000.0 / z)
    R_d, R_eq = R(z_d), R(z_eq)
    s = 2.0 / (3.0 * k_eq) * np.sqrt(6.0 / R_eq) * np.log((np.sqrt(1.0 + R_d) + np.sqrt(R_d + R_eq)) / (1.0 + np.sqrt(R_eq)))
    k_silk = 1.6 * ob ** 0.52 * om ** 0.73 * (1.0 + (10.4 * om) ** -0.95)                     # eq. 7
    q = k / (13.41 * k_eq)                                                                    # eq. 10

    def T0(alpha, beta):                                                                      # eqs. 19-20
        C = 14.2 / alpha + 386.0 / (1.0 + 69.9 * q ** 1.08)
        L = np.log(np.e + 1.8 * beta * q)
        return L / (L + C * q * q)

    # cold dark matter, eqs. 11-12, 17-18
    a1 = (46.9 * om) ** 0.670 * (1.0 + (32.1 * om) ** -0.532)
    a2 = (12.0 * om) ** 0.424 * (1.0 + (45.0 * om) ** -0.582)
    alpha_c = a1 ** -fb * a2 ** -(fb ** 3)
    bb1 = 0.944 / (1.0 + (458.0 * om) ** -0.708)
    bb2 = (0.395 * om) ** -0.0266
    beta_c = 1.0 / (1.0 + bb1 * (fc ** bb2 - 1.0))
    f = 1.0 / (1.0 + (k * s / 5.4) ** 4)
    Tc = f * T0(1.0, beta_c) + (1.0 - f) * T0(alpha_c, beta_c)
    # baryons, eqs. 14-15, 21-24
    y = (1.0 + z_eq) / (1.0 + z_d)
    G = y * (-6.0 * np.sqrt(1.0 + y) + (2.0 + 3.0 * y) * np.log((np.sqrt(1.0 + y) + 1.0) / (np.sqrt(1.0 + y) - 1.0)))
    alpha_b = 2.07 * k_eq * s * (1.0 + R_d) ** -0.75 * G
    beta_b = 0.5 + fb + (3.0 - 2.0 * fb) * np.sqrt((17.2 * om) ** 2 + 1.0)
    beta_node = 8.41 * om ** 0.435
    ks = k * s
    with np.errstate(divide="ignore", invalid="ignore"):
        s_tilde = s / (1.0 + (beta_node / ks) ** 3) ** (1.0 / 3.0)
        Tb = (T0(1.0, 1.0) / (1.0 + (ks / 5.2) ** 2) + alpha_b / (1.0 + (beta_b / ks) ** 3) * np.exp(-(k / k_silk) ** 1.4)) \
            * np.sinc(k * s_tilde / np.pi)
    Tb = np.where(k > 0, Tb, 1.0)
    return fb * Tb + fc * Tc


def sigma_r(ks, pows, r=8.0):
    """Top-hat rms fluctuation at radius r (Mpc/h) of a tabulated P(k): sigma^2 = 1/(2 pi^2) int k^3 P W(kr)^2 dln k."""
    x = ks * r
    w = 3.0 * (np.sin(x) - x * np.cos(x)) / x ** 3
    y = ks ** 3 * pows * w * w
    lk = np.log(ks)
    return float(np.sqrt(np.sum(0.5 * (y[1:] + y[:-1]) * np.diff(lk)) / (2.0 * np.pi ** 2)))


def lin_power_table(cosmo, n_interp=256, unit_sigma8=True):
    """(ks, pows): ks = logspace(-4, 1, n_interp) h/Mpc (bricks.py:73) and the a = 1 linear power there, normalised to
    sigma8 = 1 (the `lin_kpow` convention, model.py:539: the model multiplies by cosmo.sigma8^2) or to cosmo.sigma8."""
    ks = np.logspace(-4, 1, n_interp)
    # the normalisation integral runs on a finer, wider grid than the interpolation table
    kf = np.logspace(-5, 2, 4096)
    pf = kf ** cosmo.n_s * eisenstein_hu_transfer(cosmo, kf) ** 2
    amp = 1.0 / sigma_r(kf, pf) ** 2
    pows = amp * ks ** cosmo.n_s * eisenstein_hu_transfer(cosmo, ks) ** 2
    return ks, (pows if unit_sigma8 else pows * cosmo.sigma8 ** 2)
