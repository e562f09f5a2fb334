"""
TEST INFRASTRUCTURE (see oracle/__init__.py).

Restatement of the parts of jax_cosmo==0.1.0 (pinned in the reference's montenv.yml:360, NOT
vendored under /root/reference) that the hot path calls:

  reference call sites: montecosmo/nbody.py:6-7 (imports), :705-707 (Omega_m_a, w, Omega_de_a),
  :716 and :850 (odeint), :848 (dchioverda), :929-930 (Esqr).

Written from the published algorithm of jax_cosmo 0.1.0 (`jax_cosmo/background.py`,
`jax_cosmo/scipy/ode.py`, `jax_cosmo/constants.py`): matter + curvature + w0-wa dark energy,
NO radiation term.  PARITY UNPINNED: no reference fixture pins these numbers beyond the
3-digit growth check of tests_old/valid_fastpm.ipynb:747-749.
"""
import numpy as np

rh = 2997.92458  # h^-1 Mpc, jax_cosmo.constants.rh


class Cosmology:
    """Duck-type of jax_cosmo.Cosmology: the path reads Omega_m, Omega_de, Omega_k, w0, wa and
    uses `_workspace` as a dict cache (montecosmo/nbody.py:699, :745)."""

    def __init__(self, Omega_c=0.2607, Omega_b=0.0490, h=0.6766, n_s=0.9665, sigma8=0.8102,
                 Omega_k=0.0, w0=-1.0, wa=0.0):
        self.Omega_c, self.Omega_b, self.h, self.n_s, self.sigma8 = Omega_c, Omega_b, h, n_s, sigma8
        self.Omega_k, self.w0, self.wa = Omega_k, w0, wa
        self._workspace = {}

    @property
    def Omega_m(self):
        return self.Omega_b + self.Omega_c

    @property
    def Omega_de(self):
        return 1.0 - self.Omega_k - self.Omega_m


def Planck18(**kw):
    """montecosmo/bricks.py:28-37 (Omega_m = 0.3111 + ... = 0.3097; values copied as data)."""
    args = dict(Omega_c=0.2607, Omega_b=0.0490, sigma8=0.8102, Omega_k=0.0, h=0.6766, n_s=0.9665,
                w0=-1.0, wa=0.0)
    args.update(kw)
    return Cosmology(**args)


def w(cosmo, a):
    return cosmo.w0 + (1.0 - a) * cosmo.wa


def f_de(cosmo, a):
    # jax_cosmo guards log(a) at a=1 with a float32 epsilon
    epsilon = np.finfo(np.float32).eps
    a = np.asarray(a, dtype=np.float64)
    return -3.0 * (1.0 + cosmo.w0) + 3.0 * cosmo.wa * ((a - 1.0) / np.log(a - epsilon) - 1.0)


def Esqr(cosmo, a):
    a = np.asarray(a, dtype=np.float64)
    return (cosmo.Omega_m * np.power(a, -3) + cosmo.Omega_k * np.power(a, -2)
            + cosmo.Omega_de * np.power(a, f_de(cosmo, a)))


def Omega_m_a(cosmo, a):
    a = np.asarray(a, dtype=np.float64)
    return cosmo.Omega_m * np.power(a, -3) / Esqr(cosmo, a)


def Omega_de_a(cosmo, a):
    a = np.asarray(a, dtype=np.float64)
    return cosmo.Omega_de * np.power(a, f_de(cosmo, a)) / Esqr(cosmo, a)


def dchioverda(cosmo, a):
    a = np.asarray(a, dtype=np.float64)
    return rh / (a ** 2 * np.sqrt(Esqr(cosmo, a)))


def odeint(fn, y0, t):
    """jax_cosmo.scipy.ode.odeint: classical fixed-step RK4 on the supplied grid, fn(y, t).
    The scan starts with t_prev = t[0], so the first output is y0 itself."""
    t = np.asarray(t, dtype=np.float64)
    y = np.asarray(y0, dtype=np.float64)
    out = []
    t_prev = t[0]
    for ti in t:
        h = ti - t_prev
        k1 = fn(y, t_prev)
        k2 = fn(y + h * k1 / 2, t_prev + h / 2)
        k3 = fn(y + h * k2 / 2, t_prev + h / 2)
        k4 = fn(y + k3 * h, ti)
        y = y + 1.0 / 6.0 * h * (k1 + 2 * k2 + 2 * k3 + k4)
        t_prev = ti
        out.append(y)
    return np.stack(out)
