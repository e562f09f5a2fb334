"""
MI355X-native operator surface with the names, argument names, defaults and return structure of
`montecosmo/nbody.py` (reference file; line numbers below refer to it), so that `bricks.py` /
`model.py` can `from montecosmo_amd.nbody import ...` unchanged.  Arrays are torch tensors on the
GPU (numpy inputs are uploaded); every mesh/particle operation runs in hand-written HIP kernels of
libmcpm.so through the C ABI of include/mcpm.h.  There is no autodiff: every differentiable entry
has an explicit `*_vjp` twin.

Host-side pieces (wavevector helpers, growth tables, BullFrog/FastPM coefficients) are float64 numpy,
as in the reference where they are numpy / 128-entry tables.
"""
from __future__ import annotations

import ctypes as C
import os
from itertools import product

import numpy as np
import torch

from . import _lib
from ._lib import lib, check, POS_ABSOLUTE, POS_LATTICE
from .utils import safe_div, ch2rshape, r2chshape, scale_shape, chreshape, chreshape_vjp  # noqa: F401 (re-exported like the reference)

__all__ = [
    "rfftk", "fftk", "invlaplace_hat", "gradient_hat", "gaussian_hat", "top_hat", "rectangular", "rectangular_hat", "k2ell", "ell2k",
    "paint", "read", "paint_vjp", "read_vjp", "pm_forces", "pm_forces_vjp", "pm_forces2", "lpt", "lpt_vjp",
    "a2g", "a2g2", "a2f", "a2f2", "a2dg2dg", "g2a", "g2g2", "g2f", "g2f2", "g2dg2dg", "a2chi", "chi2a",
    "bullfrog_vf", "nbody_bf", "nbody_bf_vjp", "cosmo_vjp", "alpha_bf", "alpha_fpm", "LatticePos", "get_plan",
    "deconv_paint", "interlace", "nufft", "nufft_vjp",
    "safe_div", "ch2rshape", "r2chshape", "scale_shape",
]


# ------------------------------------------------------------------------------------------------
# device plumbing
def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("montecosmo_amd needs a ROCm GPU (MI355X): there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _fd(order):
    if order == np.inf:
        return _lib.FD_INF
    if order in (2, 4):
        return int(order)
    raise ValueError("Only orders 2, 4, and inf are supported.")


class Plan:
    """Owns one `mcpm_plan` (rocFFT plans + scratch) for a (mesh shape, particle lattice) pair."""

    def __init__(self, mesh_shape, ptcl_shape=None):
        mesh_shape = tuple(int(s) for s in mesh_shape)
        ptcl_shape = mesh_shape if ptcl_shape is None else tuple(int(s) for s in ptcl_shape)
        if len(mesh_shape) != 3 or len(ptcl_shape) != 3:
            raise ValueError("only 3D meshes are supported")
        self.device = _device()
        self.mesh_shape, self.ptcl_shape = mesh_shape, ptcl_shape
        self.M = int(np.prod(mesh_shape))
        self.Mh = mesh_shape[0] * mesh_shape[1] * (mesh_shape[2] // 2 + 1)
        self.N = int(np.prod(ptcl_shape))
        self.stream = torch.cuda.current_stream(self.device)
        h = C.c_void_p()
        rc = lib.mcpm_plan_create(*mesh_shape, *ptcl_shape, C.c_void_p(self.stream.cuda_stream), C.byref(h))
        check(rc, None, "mcpm_plan_create")
        self.h = h

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        try:
            if h:
                lib.mcpm_plan_destroy(h)
        except Exception:  # interpreter shutdown: the library handle may already be gone
            pass

    def call(self, name, *args):
        check(getattr(lib, name)(self.h, *args), self.h, name)

    def last_bucketed(self):
        n = C.c_int64()
        self.call("mcpm_plan_last_bucketed", C.byref(n))
        return n.value

    def last_outliers(self):
        n = C.c_int64()
        self.call("mcpm_plan_last_outliers", C.byref(n))
        return n.value


_PLANS = {}


def get_plan(mesh_shape, ptcl_shape=None) -> Plan:
    mesh_shape = tuple(int(s) for s in mesh_shape)
    ptcl_shape = mesh_shape if ptcl_shape is None else tuple(int(s) for s in ptcl_shape)
    key = (mesh_shape, ptcl_shape, torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    if key not in _PLANS:
        _PLANS[key] = Plan(mesh_shape, ptcl_shape)
    return _PLANS[key]


def clear_plans():
    _PLANS.clear()


def _f32(x, shape=None):
    """Contiguous float32 CUDA tensor from numpy / torch input."""
    if isinstance(x, LatticePos):
        raise TypeError("LatticePos given where a plain array is expected")
    t = torch.as_tensor(x)
    t = t.to(device=_device(), dtype=torch.float32).contiguous()
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


def _c64(x, shape=None):
    t = torch.as_tensor(x).to(device=_device(), dtype=torch.complex64).contiguous()
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


_DEV_CONST = {}


def _dev_const(values, device, dtype=torch.float32):
    """A short constant vector on the device, uploaded once per (values, device)."""
    key = (tuple(float(v) for v in values), str(device), dtype)
    t = _DEV_CONST.get(key)
    if t is None:
        if len(_DEV_CONST) > 64:
            _DEV_CONST.clear()
        t = _DEV_CONST[key] = torch.tensor(key[0], dtype=dtype, device=device)
    return t


class LatticePos:
    """Particle positions stored as float32 displacements from the regular lattice
    `regular_pos(mesh_shape, ptcl_shape)` (bricks.py:593-603).  This is how the kernels keep cell
    indices exact and fractions at full fp32 precision on large meshes (absolute fp32 coordinates lose
    ~3e-5 cells at n = 512).  `to_absolute()` gives the reference's (N,3) array."""

    def __init__(self, disp, mesh_shape, ptcl_shape=None, is_regular=False):
        self.is_regular = bool(is_regular)        # zero displacements: the particles ARE the lattice (set by `regular`)
        self.mesh_shape = tuple(int(s) for s in mesh_shape)
        self.ptcl_shape = self.mesh_shape if ptcl_shape is None else tuple(int(s) for s in ptcl_shape)
        n = int(np.prod(self.ptcl_shape))
        self.disp = _f32(disp, (n, 3))

    @classmethod
    def regular(cls, mesh_shape, ptcl_shape=None):
        ptcl = mesh_shape if ptcl_shape is None else ptcl_shape
        return cls(torch.zeros((int(np.prod(ptcl)), 3), dtype=torch.float32, device=_device()), mesh_shape, ptcl_shape,
                   is_regular=True)

    def lattice(self, dtype=torch.float64):
        axes = [torch.arange(p, device=self.disp.device, dtype=dtype) * (m / p)
                for m, p in zip(self.mesh_shape, self.ptcl_shape)]
        return torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1).reshape(-1, 3)

    def to_absolute(self, dtype=torch.float64):
        return self.lattice(dtype) + self.disp.to(dtype)

    def __add__(self, dpos):
        return LatticePos(self.disp + _f32(dpos, self.disp.shape), self.mesh_shape, self.ptcl_shape)

    __radd__ = __add__

    def __len__(self):
        return self.disp.shape[0]


def _pos_args(pos, mesh_shape):
    """-> (plan, float32 tensor, n, pos_mode)"""
    if isinstance(pos, LatticePos):
        if tuple(pos.mesh_shape) != tuple(int(s) for s in mesh_shape):
            raise ValueError("LatticePos mesh shape does not match the mesh")
        return get_plan(mesh_shape, pos.ptcl_shape), pos.disp, pos.disp.shape[0], POS_LATTICE
    t = _f32(pos)
    if t.ndim != 2 or t.shape[1] != 3:
        raise ValueError("pos must have shape (N, 3)")
    return get_plan(mesh_shape), t, t.shape[0], POS_ABSOLUTE


# ------------------------------------------------------------------------------------------------
# wavevectors and k-space kernels: host numpy, exactly as the reference builds them (nbody.py:50-163)
def _kaxes(shape, box_size, real_last):
    dim = len(shape)
    scales = dim * (2 * np.pi,) if box_size is None else tuple(2 * np.pi * s / b for s, b in zip(shape, box_size))
    out = []
    for ax, (s, sc) in enumerate(zip(shape, scales)):
        freq = np.fft.rfftfreq(s) if (real_last and ax == dim - 1) else np.fft.fftfreq(s)
        bshape = [1] * dim
        bshape[ax] = -1
        out.append((freq * sc).reshape(bshape))
    return tuple(out)


def rfftk(shape, box_size=None):
    """Wavevectors for rfftn, broadcastable; cell units (k in [-pi, pi[) unless `box_size` (nbody.py:50-77)."""
    return _kaxes(shape, box_size, True)


def fftk(shape, box_size=None):
    """Wavevectors for fftn (nbody.py:80-103)."""
    return _kaxes(shape, box_size, False)


def invlaplace_hat(kvec, fd_order=np.inf):
    """Fourier transform of the inverse Laplace kernel (nbody.py:109-133)."""
    if fd_order == 2:
        kk = sum(2 * (np.cos(k) - 1) for k in kvec)
    elif fd_order == 4:
        kk = sum((np.cos(2 * k) - 16 * np.cos(k) + 15) / 6 for k in kvec)
    elif fd_order == np.inf:
        kk = sum(k ** 2 for k in kvec)
    else:
        raise ValueError("Only orders 2, 4, and inf are supported.")
    return -safe_div(1, kk)


def gradient_hat(kvec, direction: int, fd_order=np.inf):
    """Fourier transform of the gradient kernel along `direction` (nbody.py:136-163)."""
    k = kvec[direction]
    if fd_order == 2:
        k = np.sin(k)
    elif fd_order == 4:
        k = (8 * np.sin(k) - np.sin(2 * k)) / 6
    elif fd_order != np.inf:
        raise ValueError("Only orders 2, 4, and inf are supported.")
    return 1j * k


def gaussian_hat(kvec, kcut=np.inf):
    """nbody.py:166-188"""
    if kcut == np.inf:
        return 1.
    rcut = 2 * np.pi / kcut
    return np.exp(-sum(k ** 2 for k in kvec) * rcut ** 2 / 2)


def top_hat(kvec, kcut=np.inf):
    """Top-hat kernel in the Fourier domain, isotropic (nbody.py:191-217): 1. for kcut = inf, else the boolean mask
    k^2 < kcut^2."""
    if kcut == np.inf:
        return 1.
    kk = sum(ki ** 2 for ki in kvec)
    return np.where(kk < kcut ** 2, True, False)


def rectangular(s, order):
    """1-D assignment weights of |s| (nbody.py:220-246); numpy or torch input."""
    xp = torch if isinstance(s, torch.Tensor) else np
    u = xp.abs(s)
    if order == 1:
        return xp.ones_like(u)
    if order == 2:
        return 1 - u
    if order == 3:
        return (u <= 0.5) * (0.75 - u ** 2) + (u > 0.5) * 0.5 * (1.5 - u) ** 2
    if order == 4:
        return (u <= 1) * (4 - 6 * u ** 2 + 3 * u ** 3) / 6 + (u > 1) * (2 - u) ** 3 / 6
    raise ValueError("order must be 1 (NGP), 2 (CIC), 3 (TSC) or 4 (PCS)")


def rectangular_hat(kvec, order: int = 2):
    """nbody.py:249-277"""
    out = 1.
    for k in kvec:
        out = out * np.sinc(k / (2 * np.pi)) ** order
    return out


def kaiser_bessel(s, order, kcut):
    """Kaiser-Bessel kernel (nbody.py:280-290), host numpy."""
    from scipy.special import i0
    s = np.asarray(s, dtype=np.float64) * 2 / order
    kc = kcut * order / 2
    return i0(kc * (1 - s ** 2) ** .5) / (order * np.sinh(kc) / kc)


def kaiser_bessel_hat(kvec, order, kcut):
    """Fourier transform of the Kaiser-Bessel kernel (nbody.py:293-312), host numpy."""
    def kernel(k, kc):
        k = np.asarray(k, dtype=np.float64) * order / 2
        kc = kc * order / 2
        dist = np.abs(kc ** 2 - k ** 2) ** .5
        safe = np.where(dist == 0, 1., dist)
        bulk = np.where(dist == 0, 1., np.sinh(safe) / safe)
        tail = np.where(dist == 0, 1., np.sin(safe) / safe)
        return np.where(np.abs(k) <= kc, bulk, tail) / (np.sinh(kc) / kc)
    out = 1.
    for ki in kvec:
        out = out * kernel(ki, kcut)
    return out


def optim_kcut(oversamp, safety=0.98):
    """Optimal wavenumber cutoff of the Kaiser-Bessel kernel (nbody.py:357-363)."""
    return safety * np.pi * (2 - 1 / oversamp)


# ------------------------------------------------------------------------------------------------
# mass assignment
def _check_kernel(kernel_type, order):
    if kernel_type not in ("rectangular", "kaiser_bessel"):
        raise ValueError(f"Unknown kernel type: {kernel_type}")
    if order not in (1, 2, 3, 4):
        raise ValueError("order must be 1 (NGP), 2 (CIC), 3 (TSC) or 4 (PCS)")


def _weights_args(weights, n):
    if np.ndim(weights) == 0 and not isinstance(weights, torch.Tensor):
        return None, 1, float(weights)
    w = _f32(weights)
    if w.ndim == 0:
        return None, 1, float(w)
    if w.shape != (n,):
        raise ValueError("weights must be a scalar or have shape (N,)")
    return w, 1, 0.0


def paint(pos, shape: tuple, weights=1., order: int = 2, kernel_type='rectangular', oversamp=1.):
    """Paint the positions onto a mesh of given shape (nbody.py:365-396)."""
    _check_kernel(kernel_type, order)
    shape = tuple(int(s) for s in shape)
    plan, p, n, mode = _pos_args(pos, shape)
    w, ws, wsc = _weights_args(weights, n)
    mesh = torch.empty(shape, dtype=torch.float32, device=p.device)
    if kernel_type == "kaiser_bessel":
        plan.call("mcpm_paint_kb_f32", _ptr(p), n, mode, _ptr(w), ws, wsc, order, float(optim_kcut(oversamp)), _ptr(mesh), 0)
    else:
        plan.call("mcpm_paint_f32", _ptr(p), n, mode, _ptr(w), ws, wsc, order, _ptr(mesh), 0)
    return mesh


def read(pos, mesh, order: int = 2, kernel_type='rectangular', oversamp=1.):
    """Read the value at the positions from the mesh (nbody.py:398-427)."""
    _check_kernel(kernel_type, order)
    mesh = _f32(mesh)
    plan, p, n, mode = _pos_args(pos, mesh.shape)
    out = torch.empty((n,), dtype=torch.float32, device=p.device)
    if kernel_type == "kaiser_bessel":
        plan.call("mcpm_read_kb_f32", _ptr(p), n, mode, _ptr(mesh), order, float(optim_kcut(oversamp)), _ptr(out), None, 1, 0.0, None)
    else:
        plan.call("mcpm_read_f32", _ptr(p), n, mode, _ptr(mesh), 1, order, _ptr(out))
    return out


def paint_vjp(pos, shape, weights, mesh_bar, order: int = 2, kernel_type='rectangular', oversamp=1.):
    """VJP of paint: -> (pos_bar (N,3), weights_bar (N,) or its sum for scalar weights)."""
    _check_kernel(kernel_type, order)
    shape = tuple(int(s) for s in shape)
    plan, p, n, mode = _pos_args(pos, shape)
    w, ws, wsc = _weights_args(weights, n)
    mb = _f32(mesh_bar, shape)
    pos_bar = torch.empty((n, 3), dtype=torch.float32, device=p.device)
    w_bar = torch.empty((n,), dtype=torch.float32, device=p.device)
    if kernel_type == "kaiser_bessel":
        plan.call("mcpm_read_kb_f32", _ptr(p), n, mode, _ptr(mb), order, float(optim_kcut(oversamp)), _ptr(w_bar), _ptr(w), ws, wsc,
                  _ptr(pos_bar))
    else:
        plan.call("mcpm_paint_vjp_f32", _ptr(p), n, mode, _ptr(w), ws, wsc, order, _ptr(mb), _ptr(pos_bar), _ptr(w_bar))
    return pos_bar, (w_bar if w is not None else w_bar.double().sum())


def read_vjp(pos, mesh, out_bar, order: int = 2, kernel_type='rectangular', oversamp=1.):
    """VJP of read: -> (pos_bar (N,3), mesh_bar)."""
    _check_kernel(kernel_type, order)
    mesh = _f32(mesh)
    plan, p, n, mode = _pos_args(pos, mesh.shape)
    ob = _f32(out_bar, (n,))
    pos_bar = torch.empty((n, 3), dtype=torch.float32, device=p.device)
    mesh_bar = torch.empty(tuple(mesh.shape), dtype=torch.float32, device=p.device)
    if kernel_type == "kaiser_bessel":
        kc = float(optim_kcut(oversamp))
        plan.call("mcpm_read_kb_f32", _ptr(p), n, mode, _ptr(mesh), order, kc, None, _ptr(ob), 1, 0.0, _ptr(pos_bar))
        plan.call("mcpm_paint_kb_f32", _ptr(p), n, mode, _ptr(ob), 1, 0.0, order, kc, _ptr(mesh_bar), 0)
    else:
        plan.call("mcpm_read_vjp_pos_f32", _ptr(p), n, mode, _ptr(mesh), 1, order, _ptr(ob), _ptr(pos_bar))
        plan.call("mcpm_paint_f32", _ptr(p), n, mode, _ptr(ob), 1, 0.0, order, _ptr(mesh_bar), 0)
    return pos_bar, mesh_bar


def cell_index(pos, shape, order: int = 2):
    """wrap(id0) of nbody.py:372-375 as an (N,3) int16 tensor."""
    shape = tuple(int(s) for s in shape)
    plan, p, n, mode = _pos_args(pos, shape)
    idx = torch.empty((n, 3), dtype=torch.int16, device=p.device)
    plan.call("mcpm_cell_index", _ptr(p), n, mode, order, _ptr(idx))
    return idx


# ------------------------------------------------------------------------------------------------
# FFT helpers (numpy/jax conventions on top of the unnormalised C ABI)
def rfftn(mesh):
    mesh = _f32(mesh)
    plan = get_plan(mesh.shape)
    out = torch.empty(r2chshape(mesh.shape), dtype=torch.complex64, device=mesh.device)
    plan.call("mcpm_fft_r2c", _ptr(mesh), _ptr(out), 1)
    return out


def irfftn(spec):
    spec = _c64(spec).clone()  # C2R destroys its input
    shape = ch2rshape(spec.shape)
    plan = get_plan(shape)
    out = torch.empty(shape, dtype=torch.float32, device=spec.device)
    plan.call("mcpm_fft_c2r", _ptr(spec), _ptr(out), 1)
    return out / plan.M


# ------------------------------------------------------------------------------------------------
# observation-side painting (nbody.py:315-334, :513-577): runs once per log-prob, outside the step loop
def _shift_pos(pos, shift):
    if isinstance(pos, LatticePos):
        return LatticePos(pos.disp + shift, pos.mesh_shape, pos.ptcl_shape)
    return _f32(pos) + shift


def deconv_paint(mesh, order: int = 2, kernel_type='rectangular', oversamp=1.):
    """Deconvolve the mesh by the paint kernel of given order and type (nbody.py:315-334); real or half-spectrum input."""
    _check_kernel(kernel_type, order)
    t = torch.as_tensor(mesh)
    if kernel_type == 'kaiser_bessel':      # separable: three 1-D factors (host float64, nbody.py:293-312) divided out on the device
        real = not t.is_complex()
        spec = rfftn(t) if real else _c64(t)
        shape = ch2rshape(spec.shape)
        kc = optim_kcut(oversamp)
        hx, hy, hz = (torch.from_numpy(kaiser_bessel_hat((k.reshape(-1),), order, kc).astype(np.float32)).to(spec.device) for k in rfftk(shape))
        out = spec / (hx[:, None, None] * hy[None, :, None] * hz[None, None, :])
        return irfftn(out) if real else out
    if not t.is_complex():
        spec = rfftn(t)
        plan = get_plan(tuple(t.shape))
        plan.call("mcpm_kspace_phase_f32", _ptr(spec), _ptr(spec), 1.0, 0.0, int(order), 0, 0, 0)
        return irfftn(spec)
    spec = _c64(t)
    out = torch.empty_like(spec)
    get_plan(ch2rshape(spec.shape)).call("mcpm_kspace_phase_f32", _ptr(spec), _ptr(out), 1.0, 0.0, int(order), 0, 0, 0)
    return out


def interlace(pos, shape: tuple, weights=1., paint_order: int = 2, interlace_order: int = 2, kernel_type='rectangular',
              paint_oversamp: float = 1.):
    """Equal-spacing interlacing (nbody.py:513-529): mean over shifts s = j/interlace_order of
    rfftn(paint(pos + s)) * exp(i s (kx+ky+kz)).  Returns a half-spectrum."""
    _check_kernel(kernel_type, paint_order)
    shape = tuple(int(s) for s in shape)
    plan = get_plan(shape, pos.ptcl_shape if isinstance(pos, LatticePos) else None)
    out = torch.zeros(r2chshape(shape), dtype=torch.complex64, device=_device())
    tmp = torch.empty_like(out)
    for j in range(int(interlace_order)):
        s = j / interlace_order
        mesh = paint(_shift_pos(pos, s), shape, weights, paint_order, kernel_type=kernel_type, oversamp=paint_oversamp)
        plan.call("mcpm_fft_r2c", _ptr(mesh), _ptr(tmp), 1)
        plan.call("mcpm_kspace_phase_f32", _ptr(tmp), _ptr(out), 1.0 / interlace_order, float(s), 0, 0, 0, 1)
    return out


def _nufft_shapes(final_shape, paint_shape):
    final_shape = tuple(int(s) for s in final_shape)
    if paint_shape is None:
        return final_shape, final_shape
    if isinstance(paint_shape, float):
        return final_shape, scale_shape(final_shape, paint_shape)
    return final_shape, tuple(int(s) for s in paint_shape)


def _scale_pos(pos, ratio, shape, final_shape):
    """Positions in cell units of final_shape -> cell units of `shape` (nbody.py:568).  A LatticePos keeps its particle
    lattice: displacements scale with the cell ratio."""
    if all(r == 1.0 for r in ratio):
        return pos
    if isinstance(pos, LatticePos):
        return LatticePos(pos.disp * _dev_const(ratio, pos.disp.device), shape, pos.ptcl_shape)
    p = torch.as_tensor(pos)
    return p * torch.as_tensor(np.asarray(ratio), dtype=p.dtype, device=p.device)


def nufft(pos, final_shape: tuple, paint_shape=None, weights=1., paint_order: int = 2, interlace_order: int = 2,
          kernel_type='rectangular', paint_deconv=True):
    """Non-uniform FFT with oversampling, interlacing and kernel deconvolution (nbody.py:532-577).  `pos` in cell
    units of `final_shape`; the particles are painted on `paint_shape` (tuple, or float = oversampling factor on
    final_shape), deconvolved there and reshaped to the half-spectrum of final_shape with `chreshape`."""
    paint_oversamp = _nufft_oversamp(final_shape, paint_shape)
    final_shape, paint_shape = _nufft_shapes(final_shape, paint_shape)
    ratio = tuple(p / f for p, f in zip(paint_shape, final_shape))
    mesh = interlace(_scale_pos(pos, ratio, paint_shape, final_shape), paint_shape, weights, paint_order, interlace_order,
                     kernel_type=kernel_type, paint_oversamp=paint_oversamp)
    jac = float(np.prod(ratio))
    if jac != 1.0:
        mesh *= jac                       # jacobian of final units to paint units (nbody.py:570)
    if paint_deconv:
        mesh = deconv_paint(mesh, paint_order, kernel_type=kernel_type, oversamp=paint_oversamp)
    if final_shape != paint_shape:
        mesh = chreshape(mesh, r2chshape(final_shape))
    return mesh


def _nufft_oversamp(final_shape, paint_shape):
    """paint_oversamp as nufft derives it (nbody.py:557-566): 1, the float given, or -- for a shape -- the geometric mean of
    final / paint (sic)."""
    if paint_shape is None:
        return 1.
    if isinstance(paint_shape, float):
        return paint_shape
    return float(np.exp(np.log(np.divide(tuple(final_shape), tuple(paint_shape))).mean()))


def nufft_vjp(pos, final_shape: tuple, weights, mesh_bar, paint_order: int = 2, interlace_order: int = 2, paint_deconv=True,
              paint_shape=None, kernel_type='rectangular'):
    """VJP of nufft w.r.t. (pos, weights); mesh_bar is the cotangent of the returned half-spectrum (shape
    r2chshape(final_shape)) in the real-pair convention.  Returns (pos_bar (N,3) in final_shape cell units, weights_bar)."""
    paint_oversamp = _nufft_oversamp(final_shape, paint_shape)
    final_shape, shape = _nufft_shapes(final_shape, paint_shape)
    ratio = tuple(p / f for p, f in zip(shape, final_shape))
    mb = _c64(mesh_bar, r2chshape(final_shape))
    if shape != final_shape:
        mb = chreshape_vjp(mb, r2chshape(shape))
    jac = float(np.prod(ratio))
    ppos = _scale_pos(pos, ratio, shape, final_shape)
    plan = get_plan(shape, ppos.ptcl_shape if isinstance(ppos, LatticePos) else None)
    kb = kernel_type == "kaiser_bessel"
    if kb and paint_deconv:      # the deconvolution is a real multiplier: its adjoint is the same division
        mb = deconv_paint(mb.clone(), paint_order, kernel_type=kernel_type, oversamp=paint_oversamp)
    tmp = torch.empty_like(mb)
    real = torch.empty(shape, dtype=torch.float32, device=mb.device)
    pos_bar, w_bar = None, None
    for j in range(int(interlace_order)):
        s = j / interlace_order
        # adjoint of (x phase / deconv / interlace_order) then of rfftn: C2R(conj(mult) * bar / multiplicity)
        plan.call("mcpm_kspace_phase_f32", _ptr(mb), _ptr(tmp), jac / interlace_order, float(s), int(paint_order) if (paint_deconv and not kb) else 0, 1, 1, 0)
        plan.call("mcpm_fft_c2r", _ptr(tmp), _ptr(real), 1)
        pb, wb = paint_vjp(_shift_pos(ppos, s), shape, weights, real, paint_order, kernel_type=kernel_type, oversamp=paint_oversamp)
        pos_bar = pb if pos_bar is None else pos_bar + pb
        w_bar = wb if w_bar is None else w_bar + wb
    if jac != 1.0 or any(r != 1.0 for r in ratio):
        pos_bar = pos_bar * _dev_const(ratio, pos_bar.device)
    return pos_bar, w_bar


# ------------------------------------------------------------------------------------------------
# forces
def pm_forces(pos, mesh, read_order: int = 2, paint_deconv: bool = False, grad_fd=np.inf, lap_fd=np.inf, kcut=np.inf):
    """Gravitational forces on particles with a PM scheme (nbody.py:583-604).  `mesh` is a shape tuple
    (the particles are painted first) or a half-spectrum."""
    kc = 0.0 if kcut == np.inf else float(kcut)
    if isinstance(mesh, tuple):
        shape = tuple(int(s) for s in mesh)
        plan, p, n, mode = _pos_args(pos, shape)
        out = torch.empty((n, 3), dtype=torch.float32, device=p.device)
        plan.call("mcpm_pm_forces_f32", _ptr(p), n, mode, read_order, int(bool(paint_deconv)), _fd(lap_fd), _fd(grad_fd), kc, _ptr(out))
        return out
    spec = _c64(mesh)
    shape = ch2rshape(spec.shape)
    plan, p, n, mode = _pos_args(pos, shape)
    out = torch.empty((n, 3), dtype=torch.float32, device=p.device)
    plan.call("mcpm_pm_forces_spec_f32", _ptr(spec), _ptr(p), n, mode, read_order, _fd(lap_fd), _fd(grad_fd), kc, _ptr(out))
    return out


def pm_forces_vjp(pos, mesh, forces_bar, read_order: int = 2):
    """VJP of pm_forces (spectral kernels): -> (pos_bar (N,3), mesh_bar).  `mesh` a shape tuple (painted case,
    mesh_bar is None and pos_bar includes the dependence through the painted density) or a half-spectrum (mesh_bar
    in the real-pair convention)."""
    if isinstance(mesh, tuple):
        shape, spec = tuple(int(s) for s in mesh), None
    else:
        spec = _c64(mesh)
        shape = ch2rshape(spec.shape)
    plan, p, n, mode = _pos_args(pos, shape)
    fb = _f32(forces_bar, (n, 3))
    pos_bar = torch.empty((n, 3), dtype=torch.float32, device=p.device)
    spec_bar = torch.empty(tuple(spec.shape), dtype=torch.complex64, device=p.device) if spec is not None else None
    plan.call("mcpm_pm_forces_vjp_f32", _ptr(spec), _ptr(p), n, mode, int(read_order), _ptr(fb), _ptr(pos_bar), _ptr(spec_bar))
    return pos_bar, spec_bar


def pm_forces2(pos, mesh, read_order: int = 2, grad_fd=np.inf, lap_fd=np.inf):
    """2LPT source term forces (nbody.py:607-631)."""
    spec = _c64(mesh)
    shape = ch2rshape(spec.shape)
    plan, p, n, mode = _pos_args(pos, shape)
    out = torch.empty((n, 3), dtype=torch.float32, device=p.device)
    plan.call("mcpm_pm_forces2_f32", _ptr(spec), _ptr(p), n, mode, read_order, _fd(lap_fd), _fd(grad_fd), _ptr(out))
    return out


def interp_dev(x, xp, fp, scale=1.0):
    """np.interp(x, xp, fp) * scale for a float32 device tensor x (mcpm_interp_f32); xp, fp host float64 tables."""
    x = _f32(x).reshape(-1)
    tab = torch.from_numpy(np.concatenate([np.asarray(xp, dtype=np.float64), np.asarray(fp, dtype=np.float64)])).to(x.device)
    out = torch.empty_like(x)
    nt = len(xp)
    get_plan((8, 8, 8)).call("mcpm_interp_f32", _ptr(x), x.numel(), _ptr(tab), C.c_void_p(tab.data_ptr() + 8 * nt), nt, float(scale), _ptr(out))
    return out


def growth_dev(cosmo, a, which):
    """Growth quantity `which` in {'g', 'g2', 'dg2dg', 'f'} at per-particle scale factors `a` given as a device tensor
    (same tables and linear interpolation as the host functions a2g, a2g2, a2dg2dg, a2f)."""
    c = _growth_cache(cosmo)
    if which == "g":
        return interp_dev(a, c["a"], c["g"])
    if which == "g2":
        return interp_dev(a, c["a"], c["g2"], -3 / 7)
    if which == "f":
        return interp_dev(a, c["a"], c["f"])
    if which == "f2":
        return interp_dev(a, c["a"], c["f2"])
    if which == "dg2dg":      # a2dg2dg = safe_div(g2 f2, g f) of the four separately interpolated values (nbody.py:775-777)
        num = growth_dev(cosmo, a, "g2") * growth_dev(cosmo, a, "f2")
        den = growth_dev(cosmo, a, "g") * growth_dev(cosmo, a, "f")
        return torch.where(den != 0, num / torch.where(den != 0, den, torch.ones_like(den)), torch.zeros_like(den))
    raise ValueError(which)


def _growth_tab3(cosmo, a, n):
    """(n,3) float32 device table of (a2g, a2g2, a2dg2dg) at the per-particle scale factors a (N,1)."""
    if isinstance(a, torch.Tensor) and a.is_cuda:
        if a.numel() != n:
            raise ValueError(f"a must have one entry per particle ({n}), got {a.numel()}")
        return torch.stack([growth_dev(cosmo, a, "g"), growth_dev(cosmo, a, "g2"), growth_dev(cosmo, a, "dg2dg")], dim=-1).contiguous()
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    if a.size != n:
        raise ValueError(f"a must be a scalar or have one entry per particle ({n}), got {a.size}")
    return _f32(np.stack([a2g(cosmo, a), a2g2(cosmo, a), a2dg2dg(cosmo, a)], axis=-1))


class LptCtx:
    """What `lpt(..., return_ctx=True)` keeps for `lpt_vjp(..., ctx=...)`: the force / Hessian meshes of the forward pass (`save`) and, on the
    light cone, the per-particle force arrays F1, F2 -- the adjoint then recomputes neither (2.9 of 22.5 ms per gradient at 256^3)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def lpt(cosmo, init_mesh, pos, a, lpt_order: int = 2, read_order: int = 2, grad_fd=np.inf, lap_fd=np.inf, return_ctx=False):
    """First or second order LPT displacement and growth-time velocity at scale factor(s) `a`
    (nbody.py:634-667).  `a` may be a scalar or an (N,1) array (light-cone: per-particle growth applied by
    mcpm_lpt_combine_f32).  `return_ctx=True` (fused path: regular lattice, NGP, infinite-order kernels) also returns an LptCtx."""
    init_mesh = torch.as_tensor(init_mesh)
    if not init_mesh.is_complex():
        init_mesh = rfftn(init_mesh)
    scalar_a = not isinstance(a, torch.Tensor) and (np.ndim(a) == 0 or np.size(a) == 1)
    if isinstance(pos, LatticePos) and pos.is_regular and int(read_order) == 1:
        # the model's call (model.py:763-764: regular lattice, NGP): the fused library path -- force meshes of both
        # orders accumulated straight onto the lattice (mcpm_lpt_f32), no separate read passes
        spec = _c64(init_mesh)
        plan = get_plan(ch2rshape(spec.shape), pos.ptcl_shape)
        n = plan.N
        dpos = torch.empty((n, 3), dtype=torch.float32, device=spec.device)
        vel = torch.empty((n, 3), dtype=torch.float32, device=spec.device)
        keep = bool(return_ctx) and grad_fd == np.inf and lap_fd == np.inf
        save = torch.empty(((12 if int(lpt_order) == 2 else 3) * plan.M,), dtype=torch.float32, device=spec.device) if keep else None

        def run(g, g2, dg2dg, d_out, v_out):
            if keep:
                plan.call("mcpm_lpt_save_f32", _ptr(spec), int(lpt_order), g, g2, dg2dg, _ptr(d_out), _ptr(v_out), _ptr(save))
            else:
                plan.call("mcpm_lpt_f32", _ptr(spec), int(lpt_order), g, g2, dg2dg, _fd(lap_fd), _fd(grad_fd), _ptr(d_out), _ptr(v_out))

        if scalar_a:
            run(float(a2g(cosmo, a)), float(a2g2(cosmo, a)), float(a2dg2dg(cosmo, a)), dpos, vel)
            return ((dpos, vel), LptCtx(save=save, F1=None, F2=None)) if return_ctx else (dpos, vel)
        F2, F1 = dpos, vel                        # (g, g2, dg2dg) = (0, -1, 0): dpos = F2, vel = F1
        run(0.0, -1.0, 0.0, F2, F1)
        gt = _growth_tab3(cosmo, a, n)
        dpos, vel = torch.empty_like(F1), torch.empty_like(F1)
        plan.call("mcpm_lpt_combine_f32", _ptr(F1), _ptr(F2) if lpt_order == 2 else None, _ptr(gt), n, _ptr(dpos), _ptr(vel))
        return ((dpos, vel), LptCtx(save=save, F1=F1, F2=F2)) if return_ctx else (dpos, vel)
    force1 = pm_forces(pos, init_mesh, read_order, grad_fd=grad_fd, lap_fd=lap_fd)
    force2 = pm_forces2(pos, init_mesh, read_order, grad_fd=grad_fd, lap_fd=lap_fd) if lpt_order == 2 else None
    n = force1.shape[0]
    if scalar_a:
        dpos, vel = float(a2g(cosmo, a)) * force1, force1
        if force2 is not None:
            dpos = dpos - float(a2g2(cosmo, a)) * force2
            vel = vel - float(a2dg2dg(cosmo, a)) * force2
        return dpos, vel
    gt = _growth_tab3(cosmo, a, n)
    dpos, vel = torch.empty_like(force1), torch.empty_like(force1)
    plan = get_plan(ch2rshape(init_mesh.shape), pos.ptcl_shape if isinstance(pos, LatticePos) else None)
    plan.call("mcpm_lpt_combine_f32", _ptr(force1), _ptr(force2), _ptr(gt), n, _ptr(dpos), _ptr(vel))
    return dpos, vel


# ------------------------------------------------------------------------------------------------
# growth and distance tables: host float64 (nbody.py:675-896); the RK4 integration runs in libmcpm.so
growth_log10_amin: float = -3.
growth_steps: int = 128
dist_log10_amin: float = -3.
dist_steps: int = 256


def _cosmo_params(cosmo):
    return [float(cosmo.Omega_m), float(cosmo.Omega_de), float(cosmo.Omega_k), float(cosmo.w0), float(cosmo.wa)]


def _growth_cache(cosmo, log10_amin=growth_log10_amin, steps=growth_steps):
    key = "background.growth_factor"
    if key not in cosmo._workspace:
        names = ["a", "g", "f", "h", "g2", "f2", "h2"]
        arrs = [np.zeros(steps) for _ in names]
        rc = lib.mcpm_growth_table(*_cosmo_params(cosmo), float(log10_amin), int(steps),
                                   *[x.ctypes.data_as(C.POINTER(C.c_double)) for x in arrs])
        check(rc, None, "mcpm_growth_table")
        cosmo._workspace[key] = dict(zip(names, arrs))
    return cosmo._workspace[key]


def _interp(x, xp, fp):
    return np.interp(np.asarray(x, dtype=np.float64), xp, fp)


def a2g(cosmo, a):
    c = _growth_cache(cosmo)
    return _interp(a, c["a"], c["g"])


def a2g2(cosmo, a):
    c = _growth_cache(cosmo)
    return _interp(a, c["a"], c["g2"]) * -3 / 7


def a2f(cosmo, a):
    c = _growth_cache(cosmo)
    return _interp(a, c["a"], c["f"])


def a2f2(cosmo, a):
    c = _growth_cache(cosmo)
    return _interp(a, c["a"], c["f2"])


def a2dg2dg(cosmo, a):
    g, g2, f, f2 = a2g(cosmo, a), a2g2(cosmo, a), a2f(cosmo, a), a2f2(cosmo, a)
    return safe_div(g2 * f2, g * f)


def g2a(cosmo, g):
    c = _growth_cache(cosmo)
    return _interp(g, c["g"], c["a"])


def g2g2(cosmo, g):
    c = _growth_cache(cosmo)
    return _interp(g, c["g"], c["g2"]) * -3 / 7


def g2f(cosmo, g):
    c = _growth_cache(cosmo)
    return _interp(g, c["g"], c["f"])


def g2f2(cosmo, g):
    c = _growth_cache(cosmo)
    return _interp(g, c["g"], c["f2"])


def g2dg2dg(cosmo, g):
    g2, f, f2 = g2g2(cosmo, g), g2f(cosmo, g), g2f2(cosmo, g)
    return safe_div(g2 * f2, g * f)


def _dist_cache(cosmo, log10_amin=dist_log10_amin, steps=dist_steps):
    key = "background.radial_comoving_distance"
    if key not in cosmo._workspace:
        a, chi = np.zeros(steps), np.zeros(steps)
        rc = lib.mcpm_distance_table(*_cosmo_params(cosmo), float(log10_amin), int(steps),
                                     a.ctypes.data_as(C.POINTER(C.c_double)), chi.ctypes.data_as(C.POINTER(C.c_double)))
        check(rc, None, "mcpm_distance_table")
        cosmo._workspace[key] = {"a": a, "chi": chi}
    return cosmo._workspace[key]


def a2chi(cosmo, a, log10_amin=dist_log10_amin, steps=dist_steps):
    """Radial comoving distance in Mpc/h (nbody.py:817-859)."""
    c = _dist_cache(cosmo, log10_amin, steps)
    return np.clip(_interp(a, c["a"], c["chi"]), 0.0, None)


def chi2a(cosmo, chi, log10_amin=dist_log10_amin, steps=dist_steps):
    """nbody.py:862-884"""
    c = _dist_cache(cosmo, log10_amin, steps)
    return _interp(chi, c["chi"][::-1], c["a"][::-1])


def k2ell(cosmo, a, k):
    """Comoving wavenumber k -> multipole ell in the Limber approximation (nbody.py:886-890)."""
    return a2chi(cosmo, a) * k - 0.5


def ell2k(cosmo, a, ell):
    """Multipole ell -> comoving wavenumber k in the Limber approximation (nbody.py:892-896)."""
    return (ell + 0.5) / a2chi(cosmo, a)


# ------------------------------------------------------------------------------------------------
# BullFrog / FastPM
def _Esqr(cosmo, a):
    # jax_cosmo.background.Esqr (matter + curvature + w0-wa dark energy), the only background term alpha_fpm needs
    a = np.asarray(a, dtype=np.float64)
    eps = np.finfo(np.float32).eps
    f_de = -3.0 * (1.0 + cosmo.w0) + 3.0 * cosmo.wa * ((a - 1.0) / np.log(a - eps) - 1.0)
    return cosmo.Omega_m * a ** -3 + cosmo.Omega_k * a ** -2 + cosmo.Omega_de * a ** f_de


def alpha_bf(cosmo, g0, dg):
    """BullFrog growth-time integrator coefficient (closure at nbody.py:907-919)."""
    g1, g2 = g0 + dg / 2, g0 + dg
    d0, d2 = g2dg2dg(cosmo, g0), g2dg2dg(cosmo, g2)
    lin = (g2g2(cosmo, g0) + d0 * dg / 2) / g1 - g1
    return (d2 - lin) / (d0 - lin)


def alpha_fpm(cosmo, g0, dg):
    """FastPM growth-time integrator coefficient (closure at nbody.py:921-931)."""
    g2 = g0 + dg
    a0, a2 = g2a(cosmo, g0), g2a(cosmo, g2)
    c0 = _Esqr(cosmo, a0) ** .5 * g0 * g2f(cosmo, g0) * a0 ** 2
    c2 = _Esqr(cosmo, a2) ** .5 * g2 * g2f(cosmo, g2) * a2 ** 2
    return c0 / c2


_ALPHAS = {"bullfrog": alpha_bf, "fastpm": alpha_fpm}


def bullfrog_vf(cosmo, dg, mesh_shape: tuple, paint_order: int = 2, paint_deconv=False, grad_fd=np.inf, lap_fd=np.inf,
                integrator="bullfrog"):
    """BullFrog vector field (nbody.py:902-959): state (pos, vel) -> ((new - old)/dg) after drift(dg/2),
    kick, drift(dg/2).  `pos` may be an (N,3) array or a LatticePos."""
    mesh_shape = tuple(int(s) for s in mesh_shape)
    alpha_fn = _ALPHAS[integrator]

    def vector_field(g0, state, args=None):
        pos, vel = state
        lat = isinstance(pos, LatticePos)
        x = pos.disp if lat else _f32(pos)
        v = _f32(vel)
        x1 = x + v * (dg / 2)
        p1 = LatticePos(x1, pos.mesh_shape, pos.ptcl_shape) if lat else x1
        forces = pm_forces(p1, mesh_shape, paint_order, paint_deconv=paint_deconv, grad_fd=grad_fd, lap_fd=lap_fd)
        alpha = float(alpha_fn(cosmo, g0, dg))
        v1 = alpha * v + (1 - alpha) / (g0 + dg / 2) * forces
        x2 = x1 + v1 * (dg / 2)
        return (x2 - x) / dg, (v1 - v) / dg

    return vector_field


class NbodyCtx:
    """What `nbody_bf_vjp` needs: the plan, host scalars and the device checkpoint buffer."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def state(self, i):
        """Checkpoint i of the composite path as (x'_i, v_i) views: x'_i = x_i + v_i dg / 2, the position offsets the step's paint
        and read use (arrays 2 i and 2 i + 1 of `ckpt`, `pitch` floats apart: include/mcpm.h mcpm_plan_particle_pitch)."""
        N, pt = self.plan.N, self.pitch
        return (self.ckpt[2 * i * pt: 2 * i * pt + 3 * N].view(N, 3), self.ckpt[(2 * i + 1) * pt: (2 * i + 1) * pt + 3 * N].view(N, 3))


def _step_scalars(cosmo, a0, a1, n_steps, integrator):
    """Host float64 scalars of the Euler/BullFrog loop, with diffrax's accumulated time (nbody.py:974-976, :999)."""
    g0, g1 = float(a2g(cosmo, a0)), float(a2g(cosmo, a1))
    dg = (g1 - g0) / n_steps
    alpha_fn = _ALPHAS[integrator]
    t, alphas, betas = g0, [], []
    for _ in range(n_steps):
        al = float(alpha_fn(cosmo, t, dg))
        alphas.append(al)
        betas.append((1 - al) / (t + dg / 2))
        t = g1 if t + dg > g1 - 1e-10 else t + dg   # diffrax snaps the last step onto t1
    lpt_s = [float(a2g(cosmo, a0)), float(a2g2(cosmo, a0)), float(a2dg2dg(cosmo, a0))]
    return dg, np.array(alphas), np.array(betas), np.array(lpt_s)


def _dptr(x):
    return x.ctypes.data_as(C.POINTER(C.c_double))


def save_y(t, y, args=None):
    """diffrax's default SaveAt function (nbody.py:964, `fn=save_y`): the state itself."""
    return y


def nbody_bf(cosmo, init_mesh, pos, a0=0., a1=1., n_steps=5, paint_order: int = 2, lpt_order: int = 2,
             paint_deconv=False, grad_fd=np.inf, lap_fd=np.inf, snapshots=None, fn=save_y,
             integrator="bullfrog", return_ctx=False, lattice_out=False):    # signature and defaults of nbody.py:967-969 (+ three keyword extras)
    """N-body simulation with the BullFrog solver (nbody.py:967-1002).

    `pos` must be the regular lattice (`bricks.regular_pos(mesh_shape, ptcl_shape)`, as at model.py:738) given
    as a LatticePos or as its (N,3) array.  Returns (pos, vel) with a leading snapshot axis of 1 like
    diffrax's SaveAt(t1=True); with `lattice_out=True`, pos is a LatticePos (no leading axis).
    `integrator='fastpm'` selects the alpha_fpm coefficient (nbody.py:921-931).
    `return_ctx=True` also returns the context for `nbody_bf_vjp` (checkpoints are then kept).
    """
    n_steps = int(n_steps)
    spec = _c64(init_mesh)
    mesh_shape = ch2rshape(spec.shape)
    if isinstance(pos, LatticePos):
        ptcl_shape = pos.ptcl_shape
        # (a lattice made by LatticePos.regular carries the flag: scanning 12 N bytes and waiting for the answer stalled the device
        # for 0.4 ms per call at 256^3 -- the host computes the step scalars only after the wait)
        if not pos.is_regular and float(pos.disp.abs().max()) != 0.0:
            raise ValueError("nbody_bf starts from the undisplaced lattice")
    else:
        ptcl_shape = _infer_lattice(pos, mesh_shape)
    plan = get_plan(mesh_shape, ptcl_shape)
    dg, alphas, betas, lpt_s = _step_scalars(cosmo, a0, a1, n_steps, integrator)
    N = plan.N
    want_snaps = not (snapshots is None or (isinstance(snapshots, int) and snapshots <= 1))
    if paint_deconv or grad_fd != np.inf or lap_fd != np.inf:
        # The options the model never selects on this branch (model.py:771-773; nbody.py:590-593 deconvolved paint, :125-163
        # finite-difference kernels): one pm_forces call per step (paint -> R2C -> FD / deconvolved k-space kernel -> 3 C2R ->
        # read) instead of the fused loop; the step states are kept for the reverse sweep (`_nbody_bf_opts_vjp`) and snapshots.
        lp0 = LatticePos.regular(mesh_shape, ptcl_shape)
        dpos, v = lpt(cosmo, spec, lp0, a0, lpt_order=lpt_order, read_order=1, grad_fd=grad_fd, lap_fd=lap_fd)
        keep = return_ctx or want_snaps
        states = torch.empty((n_steps, 2, N, 3), dtype=torch.float32, device=spec.device) if keep else None
        x = dpos + v * (dg / 2)
        for i in range(n_steps):
            if keep:
                states[i, 0], states[i, 1] = x, v
            F = pm_forces(LatticePos(x, mesh_shape, ptcl_shape), mesh_shape, paint_order, paint_deconv=paint_deconv, grad_fd=grad_fd,
                          lap_fd=lap_fd)
            v = float(alphas[i]) * v + float(betas[i]) * F
            x = x + v * (dg / 2 if i == n_steps - 1 else dg)
        lp = LatticePos(x, mesh_shape, ptcl_shape)
        if want_snaps:
            if lattice_out:
                raise NotImplementedError("snapshots are returned as absolute positions")
            out = _snapshots(cosmo, snapshots, a0, a1, n_steps, dg, states.reshape(-1), N, lp, v)
        else:
            out = (lp, v) if lattice_out else (lp.to_absolute()[None], v[None])
        if fn is not None:
            out = _apply_fn(fn, out, lattice_out)
        if return_ctx:
            return out, NbodyCtx(plan=plan, init_mesh=spec, n_steps=n_steps, dg=dg, alphas=alphas, betas=betas, lpt_s=lpt_s,
                                 lpt_order=int(lpt_order), paint_order=int(paint_order), states=states, cosmo=cosmo, a0=a0, a1=a1,
                                 integrator=integrator, mesh_shape=mesh_shape, ptcl_shape=ptcl_shape,
                                 opts=dict(paint_deconv=bool(paint_deconv), grad_fd=grad_fd, lap_fd=lap_fd))
        return out
    x = torch.empty((N, 3), dtype=torch.float32, device=spec.device)
    v = torch.empty((N, 3), dtype=torch.float32, device=spec.device)
    ckpt = None
    if return_ctx or want_snaps:
        nck = lib.mcpm_nbody_ckpt_floats(plan.h, n_steps, lpt_order)
        ckpt = torch.empty((nck,), dtype=torch.float32, device=spec.device)
        if not getattr(plan, "_pitch_probed", False):
            # once per plan: the library times its adjoint particle kernel on THIS buffer for three layouts of the checkpoint
            # arrays (back to back, or shifted against each other by a few KB) and keeps the fastest (include/mcpm.h
            # mcpm_plan_probe_particle_pitch; meshes below 2^23 particles keep the plain layout, nothing is timed)
            plan._pitch_probed = True
            if os.environ.get("MCPM_PARTICLE_PITCH") is not None:
                plan.call("mcpm_plan_set_particle_pitch", int(os.environ["MCPM_PARTICLE_PITCH"]))
            else:
                plan.call("mcpm_plan_probe_particle_pitch", _ptr(ckpt), nck, None)
    pitch = C.c_int64()
    plan.call("mcpm_plan_particle_pitch", C.byref(pitch))
    pitch = int(pitch.value)
    plan.call("mcpm_nbody_bf_f32", _ptr(spec), n_steps, _dptr(alphas), _dptr(betas), float(dg), _dptr(lpt_s),
              int(lpt_order), int(paint_order), _ptr(x), _ptr(v), _ptr(ckpt))
    lp = LatticePos(x, mesh_shape, ptcl_shape)
    if want_snaps:
        if lattice_out:
            raise NotImplementedError("snapshots are returned as absolute positions")
        out = _snapshots(cosmo, snapshots, a0, a1, n_steps, dg, ckpt, N, lp, v, pitch=pitch)
    else:
        out = (lp, v) if lattice_out else (lp.to_absolute()[None], v[None])
    if fn is not None:
        out = _apply_fn(fn, out, lattice_out)
    if return_ctx:
        ctx = NbodyCtx(plan=plan, init_mesh=spec, n_steps=n_steps, dg=dg, alphas=alphas, betas=betas, lpt_s=lpt_s,
                       lpt_order=int(lpt_order), paint_order=int(paint_order), ckpt=ckpt, cosmo=cosmo, a0=a0, a1=a1,
                       integrator=integrator, pitch=pitch)
        return out, ctx
    return out


def _apply_fn(fn, out, lattice_out):
    """`fn(t, y, args)` of diffrax's SubSaveAt (nbody.py:968, :979-986) applied to every saved state y = (pos, vel): the
    solver stacks what fn returns along the leading (snapshot) axis.  t is not available per snapshot here (None)."""
    if fn is save_y:
        return out
    if lattice_out:
        return fn(None, out, None)
    pos, vel = out
    res = [fn(None, (pos[i], vel[i]), None) for i in range(pos.shape[0])]
    if isinstance(res[0], (tuple, list)):
        return tuple(torch.stack([r[k] for r in res]) for k in range(len(res[0])))
    return torch.stack(res)


def _nbody_bf_opts_vjp(ctx, xb, vb):
    """Reverse sweep of the options branch of nbody_bf (deconvolved paint / finite-difference kernels): the adjoint of
    each kick through mcpm_pm_forces_vjp_opts_f32, of the LPT start through mcpm_lpt_vjp_opts_f32."""
    plan, K, dg, o = ctx.plan, ctx.n_steps, ctx.dg, ctx.opts
    N = plan.N
    xb, vb = xb.clone(), vb.clone()
    abar, bbar, dgbar = np.zeros(K), np.zeros(K), 0.0
    for i in reversed(range(K)):
        tau = dg / 2 if i == K - 1 else dg
        x, v = ctx.states[i, 0], ctx.states[i, 1]
        lp = LatticePos(x, ctx.mesh_shape, ctx.ptcl_shape)
        F = pm_forces(lp, ctx.mesh_shape, ctx.paint_order, paint_deconv=o["paint_deconv"], grad_fd=o["grad_fd"], lap_fd=o["lap_fd"])
        vnew = float(ctx.alphas[i]) * v + float(ctx.betas[i]) * F
        dgbar += (0.5 if i == K - 1 else 1.0) * float((xb.double() * vnew.double()).sum())      # x' += v_new tau(dg)
        vt = vb + tau * xb
        abar[i] = float((vt.double() * v.double()).sum())
        bbar[i] = float((vt.double() * F.double()).sum())
        Fb = (float(ctx.betas[i]) * vt).contiguous()
        pb = torch.empty((N, 3), dtype=torch.float32, device=x.device)
        plan.call("mcpm_pm_forces_vjp_opts_f32", _ptr(x), N, POS_LATTICE, ctx.paint_order, int(o["paint_deconv"]), _fd(o["lap_fd"]),
                  _fd(o["grad_fd"]), _ptr(Fb), _ptr(pb))
        xb = xb + pb
        vb = float(ctx.alphas[i]) * vt
    v0 = ctx.states[0, 1]
    dgbar += 0.5 * float((xb.double() * v0.double()).sum())          # initial half drift x'_0 = x_0 + v_0 dg/2
    vb = vb + xb * (dg / 2)
    out = torch.empty(tuple(ctx.init_mesh.shape), dtype=torch.complex64, device=xb.device)
    sb = np.zeros(3)
    plan.call("mcpm_lpt_vjp_opts_f32", _ptr(ctx.init_mesh), ctx.lpt_order, _dptr(np.asarray(ctx.lpt_s, dtype=np.float64)), _fd(o["lap_fd"]),
              _fd(o["grad_fd"]), _ptr(xb.contiguous()), _ptr(vb.contiguous()), _ptr(out), _dptr(sb))
    return out, {"alpha": abar, "beta": bbar, "g": sb[0], "g2": sb[1], "dg2dg": sb[2], "dg": dgbar}


def _snapshots(cosmo, snapshots, a0, a1, n_steps, dg, ckpt, N, lp_final, v_final, pitch=None):
    """diffrax SaveAt(ts=...) on the Euler solution (nbody.py:990-997): linear interpolation between the step states.
    Step state i is rebuilt from the checkpoint (x'_i - v_i dg/2, v_i); the last one is the returned state."""
    g0, g1 = float(a2g(cosmo, a0)), float(a2g(cosmo, a1))
    if isinstance(snapshots, int):
        ts = np.linspace(g0, g1, snapshots)
    else:
        ts = np.atleast_1d(a2g(cosmo, np.asarray(snapshots, dtype=np.float64)))
    tgrid = [g0]
    for _ in range(n_steps):
        tgrid.append(g1 if tgrid[-1] + dg > g1 - 1e-10 else tgrid[-1] + dg)
    lat = lp_final.lattice(torch.float64)

    def state(i):
        if i == n_steps:
            return lp_final.disp.double(), v_final.double()
        pt = 3 * N if pitch is None else pitch      # floats between consecutive (N, 3) arrays (mcpm_plan_particle_pitch)
        xs = ckpt[2 * i * pt: 2 * i * pt + 3 * N].view(N, 3).double()
        vs = ckpt[(2 * i + 1) * pt: (2 * i + 1) * pt + 3 * N].view(N, 3).double()
        return xs - vs * (dg / 2), vs

    pos_out, vel_out = [], []
    for t in ts:
        i = int(np.clip(np.searchsorted(tgrid, t, side="right") - 1, 0, n_steps - 1))
        th = (t - tgrid[i]) / (tgrid[i + 1] - tgrid[i])
        (x0, v0), (x1, v1) = state(i), state(i + 1)
        pos_out.append(lat + x0 + (x1 - x0) * th)
        vel_out.append((v0 + (v1 - v0) * th).float())
    return torch.stack(pos_out), torch.stack(vel_out)


def nbody_bf_vjp(ctx, pos_bar, vel_bar):
    """Reverse sweep of nbody_bf.  `pos_bar`, `vel_bar`: cotangents of the final (pos, vel), shape (N,3) or
    (1,N,3).  Returns (init_mesh_bar, scalar_bars): init_mesh_bar is a complex64 half-spectrum in the
    real-pair convention dL = Re(sum(conj(bar) * d init_mesh)) (the conjugate of jax.grad's); scalar_bars is a
    dict of float64 cotangents of the host scalars (alpha_i, beta_i per step, and the LPT growth scalars)."""
    plan, n = ctx.plan, ctx.n_steps
    xb = _f32(pos_bar).reshape(-1, 3)
    vb = _f32(vel_bar).reshape(-1, 3)
    if xb.shape[0] != plan.N or vb.shape[0] != plan.N:
        raise ValueError("cotangent shape does not match the particle count")
    if getattr(ctx, "opts", None) is not None:
        return _nbody_bf_opts_vjp(ctx, xb, vb)
    out = torch.empty(tuple(ctx.init_mesh.shape), dtype=torch.complex64, device=xb.device)
    sb = np.zeros(2 * n + 4)
    plan.call("mcpm_plan_set_particle_pitch", 0 if ctx.pitch == 3 * plan.N else ctx.pitch)      # the layout this checkpoint was written in
    plan.call("mcpm_nbody_bf_vjp_f32", _ptr(ctx.init_mesh), n, _dptr(ctx.alphas), _dptr(ctx.betas), float(ctx.dg),
              _dptr(ctx.lpt_s), ctx.lpt_order, ctx.paint_order, _ptr(ctx.ckpt), _ptr(xb), _ptr(vb), _ptr(out), _dptr(sb))
    bars = {"alpha": sb[:n].copy(), "beta": sb[n:2 * n].copy(), "g": sb[2 * n], "g2": sb[2 * n + 1], "dg2dg": sb[2 * n + 2],
            "dg": sb[2 * n + 3]}
    return out, bars


def cosmo_vjp(ctx, scalar_bars, params=("Omega_c",), rel_eps=1e-5):
    """Chains the scalar cotangents of `nbody_bf_vjp` through the host float64 growth tables to cosmological
    parameters: dL/dtheta = sum_i alpha_bar_i dalpha_i/dtheta + beta_bar_i dbeta_i/dtheta + dg_bar ddg/dtheta
    + g_bar dg(a0)/dtheta + g2_bar dg2(a0)/dtheta + dg2dg_bar d(dg2dg)(a0)/dtheta, with the Jacobian of the
    (128-point RK4) tables taken by central finite differences (nbody.py:679-808, :907-931 are host scalars).
    `params` are attribute names of the cosmology object (e.g. 'Omega_c', 'Omega_b', 'w0')."""
    import copy
    out = {}
    for name in params:
        base = float(getattr(ctx.cosmo, name))
        h = rel_eps * max(abs(base), 1e-2)
        vals = []
        for sgn in (+1, -1):
            c = copy.copy(ctx.cosmo)
            c._workspace = {}
            setattr(c, name, base + sgn * h)
            vals.append(_step_scalars(c, ctx.a0, ctx.a1, ctx.n_steps, ctx.integrator))
        (dgp, ap, bp, lp), (dgm, am, bm, lm) = vals
        d = lambda p_, m_: (np.asarray(p_) - np.asarray(m_)) / (2 * h)
        out[name] = float(np.dot(scalar_bars["alpha"], d(ap, am)) + np.dot(scalar_bars["beta"], d(bp, bm))
                          + scalar_bars["dg"] * d(dgp, dgm)
                          + np.dot([scalar_bars["g"], scalar_bars["g2"], scalar_bars["dg2dg"]], d(lp, lm)))
    return out


def lpt_vjp(cosmo, init_mesh, pos, a, dpos_bar, vel_bar, lpt_order: int = 2, ctx=None):
    """VJP of `lpt(..., read_order=1)` on the regular lattice w.r.t. init_mesh (half-spectrum).  Scalar `a`: returns
    (init_mesh_bar, {'g','g2','dg2dg'} scalar cotangents).  `a` of shape (N,1) (light cone): the growth cotangents are
    per-particle float32 device tensors (N,).  `ctx`: the LptCtx of the forward call (nothing is recomputed then)."""
    spec = _c64(init_mesh)
    mesh_shape = ch2rshape(spec.shape)
    ptcl_shape = pos.ptcl_shape if isinstance(pos, LatticePos) else _infer_lattice(pos, mesh_shape)
    plan = get_plan(mesh_shape, ptcl_shape)
    xb, vb = _f32(dpos_bar, (plan.N, 3)), _f32(vel_bar, (plan.N, 3))
    out = torch.empty(tuple(spec.shape), dtype=torch.complex64, device=spec.device)
    sb = np.zeros(3)
    saved = ctx.save if ctx is not None else None
    if not isinstance(a, torch.Tensor) and (np.ndim(a) == 0 or np.size(a) == 1):
        sc = np.array([float(a2g(cosmo, a)), float(a2g2(cosmo, a)), float(a2dg2dg(cosmo, a))])
        if saved is not None:
            plan.call("mcpm_lpt_vjp_saved_f32", _ptr(spec), int(lpt_order), _dptr(sc), _ptr(saved), _ptr(xb), _ptr(vb), _ptr(out), _dptr(sb))
        else:
            plan.call("mcpm_lpt_vjp_f32", _ptr(spec), int(lpt_order), _dptr(sc), _ptr(xb), _ptr(vb), _ptr(out), _dptr(sb))
        return out, {"g": sb[0], "g2": sb[1], "dg2dg": sb[2]}
    # light cone: (F2, F1) = mcpm_lpt_f32 with (g, g2, dg2dg) = (0, -1, 0); the per-particle combination is adjointed by
    # mcpm_lpt_combine_vjp_f32, the rest by the scalar LPT adjoint with the same three scalars
    gt = _growth_tab3(cosmo, a, plan.N)
    if ctx is not None and ctx.F1 is not None:
        F1, F2 = ctx.F1, ctx.F2
    else:
        F2 = torch.empty((plan.N, 3), dtype=torch.float32, device=spec.device)
        F1 = torch.empty((plan.N, 3), dtype=torch.float32, device=spec.device)
        plan.call("mcpm_lpt_f32", _ptr(spec), int(lpt_order), 0.0, -1.0, 0.0, 0, 0, _ptr(F2), _ptr(F1))
    xb, vb = xb.clone(), vb.clone()
    gtb = torch.empty((plan.N, 3), dtype=torch.float32, device=spec.device)
    plan.call("mcpm_lpt_combine_vjp_f32", _ptr(F1), _ptr(F2) if lpt_order == 2 else None, _ptr(gt), plan.N, _ptr(xb), _ptr(vb), _ptr(gtb))
    sc = np.array([0.0, -1.0, 0.0])
    if saved is not None:
        plan.call("mcpm_lpt_vjp_saved_f32", _ptr(spec), int(lpt_order), _dptr(sc), _ptr(saved), _ptr(xb), _ptr(vb), _ptr(out), _dptr(sb))
    else:
        plan.call("mcpm_lpt_vjp_f32", _ptr(spec), int(lpt_order), _dptr(sc), _ptr(xb), _ptr(vb), _ptr(out), _dptr(sb))
    return out, {"g": gtb[:, 0], "g2": gtb[:, 1], "dg2dg": gtb[:, 2]}      # per-particle cotangents stay on the device


def _infer_lattice(pos, mesh_shape):
    """Particle-lattice shape of an (N,3) regular_pos array (bricks.py:593-603): checks it IS that lattice."""
    t = torch.as_tensor(pos)
    n = t.shape[0]
    if n == int(np.prod(mesh_shape)):
        ptcl = tuple(mesh_shape)
    else:
        # number of distinct values per axis from the strides of the meshgrid ordering
        t64 = t.to(torch.float64)
        pz = int((t64[:, 2] == t64[0, 2]).nonzero()[1]) if n > 1 else 1
        py = int((t64[::pz, 1] == t64[0, 1]).nonzero()[1]) if n > pz else 1
        ptcl = (n // (py * pz), py, pz)
    lat = LatticePos.regular(mesh_shape, ptcl).lattice(torch.float64)
    if lat.shape[0] != n or not torch.allclose(lat, t.to(lat.device, torch.float64), atol=1e-6):
        raise ValueError("nbody_bf needs pos = regular_pos(mesh_shape, ptcl_shape) (the reference's call, model.py:738)")
    return ptcl
