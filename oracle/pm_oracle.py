"""
TEST INFRASTRUCTURE (see oracle/__init__.py) -- numpy float64 restatement of the PM hot path of
hsimonfroy/montecosmo, function by function, each citing the reference lines it follows
(paths relative to /root/reference), plus hand-derived VJPs (validated against central finite
differences of this same oracle in tests/test_oracle_vjp.py).

Third-party semantics restated here (none vendored in the reference; "parity unpinned"):
  * jax.numpy (jax==0.4.25): rfftn unnormalised / irfftn x 1/M and = ifft over leading axes then
    c2r over the last; `.at[].add` scatter-add; gather; jnp.interp clamps; jnp.round half-to-even;
    d|s|/ds = sign(s) (0 at s = 0).  numpy has the same semantics.
  * diffrax==0.5.0 `diffeqsolve(ODETerm(vf), Euler(), t0, t1, dt0, y0, max_steps)`: constant
    step, t_{i+1} = min(t_i + dt0, t1) (last step clipped to t1), y <- y + vf(t_i, y)*(t_{i+1}-t_i);
    SaveAt(t1=True) returns ys with a leading axis of length 1.
Cotangent convention for complex arrays: "real pair", bar = dL/dRe + i dL/dIm, so that
dL = Re(sum(conj(bar) * dz)) (this is the conjugate of what jax.grad returns).
"""
from itertools import product
import numpy as np

from . import background

# --------------------------------------------------------------------------- optional multi-threaded back end
# `set_threads(n)` with n > 1 switches (a) every FFT to scipy.fft (pocketfft, float64) with `workers = n` and (b) paint / read
# and their VJPs to the OpenMP float64 kernels of oracle/csrc/pm_kernels.c (oracle/_build/libpmo.so, built by
# `make -C oracle`).  The arithmetic restated is the same (same index semantics, same kernels, float64); sums are
# accumulated in a different order, so results agree with the single-threaded numpy path to ~1e-14 (checked in
# tests/test_oracle_threads.py).  Used for the parity cases at 128^3 / 256^3 and for bench.py's cpu_baseline.
_THREADS = 1
_CLIB = None


def _load_clib():
    global _CLIB
    if _CLIB is None:
        import ctypes as C
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libpmo.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `make -C oracle` (or __graft_entry__.build())")
        lib = C.CDLL(path)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int16)
        lib.pmo_paint.argtypes = [dp, C.c_int64, dp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, dp]
        lib.pmo_read.argtypes = [dp, C.c_int64, dp, C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, C.c_double, dp]
        lib.pmo_cell_index.argtypes = [dp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, ip]
        for f in (lib.pmo_paint, lib.pmo_read, lib.pmo_cell_index):
            f.restype = None
        _CLIB = lib
    return _CLIB


def set_threads(n):
    """n > 1: threaded FFTs (scipy.fft workers) and OpenMP particle kernels; 1: plain numpy (the default, which
    generated the golden fixtures).  Returns the previous setting."""
    global _THREADS
    import os
    prev, n = _THREADS, max(1, int(n))
    if n > 1:
        _load_clib()
        os.environ["OMP_NUM_THREADS"] = str(n)
        try:
            import ctypes as C
            C.CDLL("libgomp.so.1").omp_set_num_threads(n)
        except OSError:
            pass
    _THREADS = n
    return prev


def _dptr(a):
    import ctypes as C
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _rfftn(x):
    if _THREADS > 1:
        import scipy.fft
        return scipy.fft.rfftn(x, workers=_THREADS)
    return np.fft.rfftn(x)


def _irfftn(X, s=None, axes=None):
    if _THREADS > 1:
        import scipy.fft
        return scipy.fft.irfftn(X, s=s, axes=axes, workers=_THREADS)
    return np.fft.irfftn(X, s=s, axes=axes) if s is not None else np.fft.irfftn(X)


# --------------------------------------------------------------------------- helpers
def safe_div(x, y):
    """montecosmo/utils.py:21-29"""
    y = np.asarray(y)
    y_nozeros = np.where(y == 0, 1, y)
    return np.where(y == 0, 0, x / y_nozeros)


def ch2rshape(shape):
    """montecosmo/utils.py:769-776"""
    return (*shape[:-1], 2 * (shape[-1] - 1))


def r2chshape(shape):
    """montecosmo/utils.py:778-782"""
    return (*shape[:-1], shape[-1] // 2 + 1)


def scale_shape(shape, scale=1.):
    """montecosmo/utils.py:1163-1168"""
    out = 2 * np.rint(np.multiply(shape, scale) / 2).astype(int)
    return tuple(map(int, out))


def regular_pos(mesh_shape, ptcl_shape=None):
    """montecosmo/bricks.py:593-603: x slowest, z fastest."""
    if ptcl_shape is None:
        ptcl_shape = mesh_shape
    pos = [np.linspace(0, m, p, endpoint=False) for m, p in zip(mesh_shape, ptcl_shape)]
    return np.stack(np.meshgrid(*pos, indexing='ij'), axis=-1).reshape(-1, 3)


# --------------------------------------------------------------------------- k-space kernels
def rfftk(shape, box_size=None):
    """montecosmo/nbody.py:50-77"""
    dim = len(shape)
    if box_size is None:
        scales = dim * (2 * np.pi,)
    else:
        scales = tuple(2 * np.pi * s / b for s, b in zip(shape, box_size))
    kvec = ()
    shapes = np.eye(dim, dtype=int) * -2 + 1
    for ax, (s, sc, ss) in enumerate(zip(shape, scales, shapes)):
        if ax < dim - 1:
            kvec += ((np.fft.fftfreq(s) * sc).reshape(ss),)
        else:
            kvec += ((np.fft.rfftfreq(s) * sc).reshape(ss),)
    return kvec


def fftk(shape, box_size=None):
    """montecosmo/nbody.py:80-103"""
    dim = len(shape)
    if box_size is None:
        scales = dim * (2 * np.pi,)
    else:
        scales = tuple(2 * np.pi * s / b for s, b in zip(shape, box_size))
    shapes = np.eye(dim, dtype=int) * -2 + 1
    return tuple((np.fft.fftfreq(s) * sc).reshape(ss) for s, sc, ss in zip(shape, scales, shapes))


def invlaplace_hat(kvec, fd_order=np.inf):
    """montecosmo/nbody.py:109-133"""
    if fd_order == 2:
        kk = sum((np.cos(ki) - 1) * 2 for ki in kvec)
    elif fd_order == 4:
        kk = sum((np.cos(2 * ki) - 16 * np.cos(ki) + 15) / 6 for ki in kvec)
    elif fd_order == np.inf:
        kk = sum(ki ** 2 for ki in kvec)
    else:
        raise ValueError("Only orders 2, 4, and inf are supported.")
    return -safe_div(1, kk)


def gradient_hat(kvec, direction, fd_order=np.inf):
    """montecosmo/nbody.py:136-163.  NB: no Nyquist zeroing."""
    ki = kvec[direction]
    if fd_order == 2:
        ki = np.sin(ki)
    elif fd_order == 4:
        ki = (8 * np.sin(ki) - np.sin(2 * ki)) / 6
    elif fd_order == np.inf:
        pass
    else:
        raise ValueError("Only orders 2, 4, and inf are supported.")
    return 1j * ki


def gaussian_hat(kvec, kcut=np.inf):
    """montecosmo/nbody.py:166-188"""
    if kcut == np.inf:
        return 1.
    kk = sum(ki ** 2 for ki in kvec)
    rcut = 2 * np.pi / kcut
    return np.exp(-kk * rcut ** 2 / 2)


def rectangular(s, order):
    """montecosmo/nbody.py:220-246, applied to |s|.  Order 1 (NGP) is the constant 1 (the
    reference returns a length-3 vector of ones whose product over axes is 1)."""
    s = np.abs(s)
    if order == 0:
        return np.full(np.shape(s), np.inf)
    if order == 1:
        return np.ones(np.shape(s))
    if order == 2:
        return 1 - s
    if order == 3:
        return (s <= 1 / 2) * (3 / 4 - s ** 2) + (1 / 2 < s) / 2 * (3 / 2 - s) ** 2
    if order == 4:
        return (s <= 1) / 6 * (4 - 6 * s ** 2 + 3 * s ** 3) + (1 < s) / 6 * (2 - s) ** 3
    raise ValueError(order)


def rectangular_grad(s, order):
    """d/ds rectangular(s, order) = k'(|s|) sign(s); sign(0) = 0 as in jax's abs JVP."""
    u = np.abs(s)
    sg = np.sign(s)
    if order == 1:
        return np.zeros(np.shape(s))
    if order == 2:
        return -sg
    if order == 3:
        return ((u <= 1 / 2) * (-2 * u) + (1 / 2 < u) * (-(3 / 2 - u))) * sg
    if order == 4:
        return ((u <= 1) / 6 * (-12 * u + 9 * u ** 2) + (1 < u) / 6 * (-3) * (2 - u) ** 2) * sg
    raise ValueError(order)


def rectangular_hat(kvec, order=2):
    """montecosmo/nbody.py:249-277"""
    out = 1.
    for ki in kvec:
        out = out * np.sinc(ki / (2 * np.pi)) ** order
    return out


def kaiser_bessel(s, order, kcut):
    """montecosmo/nbody.py:280-290"""
    from scipy.special import i0
    s = np.asarray(s, dtype=np.float64) * 2 / order
    kcut = kcut * order / 2
    return i0(kcut * np.sqrt(np.maximum(1 - s ** 2, 0.))) / (order * np.sinh(kcut) / kcut)


def kaiser_bessel_grad(s, order, kcut):
    """d/ds kaiser_bessel(s): -kc^2 s' (2/order) I1(z)/z / norm with s' = 2 s / order, z = kc sqrt(1 - s'^2)."""
    from scipy.special import i1
    sp = np.asarray(s, dtype=np.float64) * 2 / order
    kc = kcut * order / 2
    z = kc * np.sqrt(np.maximum(1 - sp ** 2, 0.))
    r = np.where(z > 1e-12, i1(np.maximum(z, 1e-12)) / np.maximum(z, 1e-12), 0.5)
    return -kc ** 2 * sp * (2 / order) * r / (order * np.sinh(kc) / kc)


def kaiser_bessel_hat(kvec, order, kcut):
    """montecosmo/nbody.py:293-312"""
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
    """montecosmo/nbody.py:357-363"""
    return safety * np.pi * (2 - 1 / oversamp)


# --------------------------------------------------------------------------- paint / read
def _id0_shifts(pos, ndim, order):
    """montecosmo/nbody.py:375-377: id0 = floor (even order) or round-half-even (odd order), cast to
    int16; shifts in itertools.product (lexicographic) order."""
    id0 = (np.round if order % 2 else np.floor)(pos).astype(np.int16)
    ishifts = np.arange(order) - (order - 1) // 2
    ishifts = np.array(list(product(*ndim * (ishifts,))), dtype=np.int16)
    return id0, ishifts


def cell_index(pos, shape, order=2):
    """Wrapped base-cell index of every particle, (N,3) int16: `wrap(id0)` of nbody.py:372-375."""
    if _THREADS > 1 and len(shape) == 3:
        import ctypes as C
        p = np.ascontiguousarray(pos, dtype=np.float64)
        idx = np.empty((len(p), 3), np.int16)
        _load_clib().pmo_cell_index(_dptr(p), len(p), order, *(int(v) for v in shape), idx.ctypes.data_as(C.POINTER(C.c_int16)))
        return idx
    shape16 = np.asarray(shape, dtype=np.int16)
    id0, _ = _id0_shifts(np.asarray(pos, dtype=np.float64), len(shape), order)
    return id0 % shape16


def stencil(pos, shape, order, kernel_type="rectangular", oversamp=1.):
    """Yield (flat wrapped index (N,), per-axis kernel values (N,3), per-axis kernel derivatives
    w.r.t. pos (N,3)) for each stencil point, following nbody.py:379-389."""
    pos = np.asarray(pos, dtype=np.float64)
    shape16 = np.asarray(shape, dtype=np.int16)
    id0, ishifts = _id0_shifts(pos, len(shape), order)
    for ishift in ishifts:
        idx = id0 + ishift                      # int16, unwrapped
        s = idx - pos                           # kernel argument on the UNWRAPPED index
        if kernel_type == "kaiser_bessel":
            ker, dker = kaiser_bessel(s, order, optim_kcut(oversamp)), -kaiser_bessel_grad(s, order, optim_kcut(oversamp))
        elif kernel_type == "rectangular":
            ker, dker = rectangular(s, order), -rectangular_grad(s, order)      # d/dpos K(idx - pos)
        else:
            raise ValueError(f"Unknown kernel type: {kernel_type}")
        flat = np.ravel_multi_index(tuple((idx % shape16).T.astype(np.int64)), tuple(int(v) for v in shape))
        yield flat, ker, dker


def paint(pos, shape, weights=1., order=2, kernel_type="rectangular", oversamp=1.):
    """montecosmo/nbody.py:365-396."""
    shape = tuple(int(s) for s in shape)
    size = int(np.prod(shape))
    mesh = np.zeros(size)
    n = len(pos)
    if _THREADS > 1 and len(shape) == 3 and 1 <= order <= 4 and kernel_type == "rectangular":
        p = np.ascontiguousarray(pos, dtype=np.float64)
        w = None if np.ndim(weights) == 0 else np.ascontiguousarray(np.broadcast_to(weights, (n,)), dtype=np.float64)
        _load_clib().pmo_paint(_dptr(p), n, _dptr(w), float(weights) if w is None else 0.0, order, *shape, _dptr(mesh))
        return mesh.reshape(shape)
    weights = np.broadcast_to(np.asarray(weights, dtype=np.float64), (n,))
    for flat, ker, _ in stencil(pos, shape, order, kernel_type, oversamp):
        mesh += np.bincount(flat, weights=weights * ker.prod(-1), minlength=size)
    return mesh.reshape(shape)


def read(pos, mesh, order=2, kernel_type="rectangular", oversamp=1.):
    """montecosmo/nbody.py:398-427"""
    mesh = np.asarray(mesh)
    if _THREADS > 1 and mesh.ndim == 3 and mesh.dtype == np.float64 and 1 <= order <= 4 and kernel_type == "rectangular":
        p, m = np.ascontiguousarray(pos, dtype=np.float64), np.ascontiguousarray(mesh)
        out = np.empty(len(p))
        _load_clib().pmo_read(_dptr(p), len(p), _dptr(m), order, *m.shape, _dptr(out), None, 0.0, None)
        return out
    out = np.zeros(len(pos), dtype=mesh.dtype)
    flatmesh = mesh.reshape(-1)
    for flat, ker, _ in stencil(pos, mesh.shape, order, kernel_type, oversamp):
        out = out + flatmesh[flat] * ker.prod(-1)
    return out


def _dprod(ker, dker):
    """d/dpos_a of prod_b ker_b, (N,3)."""
    out = np.empty_like(ker)
    for a in range(ker.shape[-1]):
        t = dker[:, a].copy()
        for b in range(ker.shape[-1]):
            if b != a:
                t = t * ker[:, b]
        out[:, a] = t
    return out


def paint_vjp(pos, shape, weights, mesh_bar, order=2, kernel_type="rectangular", oversamp=1.):
    """VJP of paint w.r.t. (pos, weights).  weights may be a scalar (its bar is then summed)."""
    n = len(pos)
    w = np.broadcast_to(np.asarray(weights, dtype=np.float64), (n,))
    if _THREADS > 1 and len(shape) == 3 and 1 <= order <= 4 and kernel_type == "rectangular":
        p, mb = np.ascontiguousarray(pos, dtype=np.float64), np.ascontiguousarray(mesh_bar, dtype=np.float64)
        wc = np.ascontiguousarray(w)
        pos_bar, w_bar = np.zeros((n, 3)), np.empty(n)
        _load_clib().pmo_read(_dptr(p), n, _dptr(mb), order, *(int(v) for v in shape), _dptr(w_bar), _dptr(wc), 0.0, _dptr(pos_bar))
        return pos_bar, (w_bar.sum() if np.ndim(weights) == 0 else w_bar)
    flatbar = np.asarray(mesh_bar).reshape(-1)
    pos_bar = np.zeros((n, len(shape)))
    w_bar = np.zeros(n)
    for flat, ker, dker in stencil(pos, shape, order, kernel_type, oversamp):
        mb = flatbar[flat]
        w_bar += mb * ker.prod(-1)
        pos_bar += (w * mb)[:, None] * _dprod(ker, dker)
    if np.ndim(weights) == 0:
        w_bar = w_bar.sum()
    return pos_bar, w_bar


def read_vjp(pos, mesh, out_bar, order=2, kernel_type="rectangular", oversamp=1.):
    """VJP of read w.r.t. (pos, mesh)."""
    mesh = np.asarray(mesh)
    if _THREADS > 1 and mesh.ndim == 3 and mesh.dtype == np.float64 and 1 <= order <= 4 and kernel_type == "rectangular":
        p, m = np.ascontiguousarray(pos, dtype=np.float64), np.ascontiguousarray(mesh)
        ob = np.ascontiguousarray(np.broadcast_to(out_bar, (len(p),)), dtype=np.float64)
        pos_bar = np.zeros((len(p), 3))
        _load_clib().pmo_read(_dptr(p), len(p), _dptr(m), order, *m.shape, None, _dptr(ob), 0.0, _dptr(pos_bar))
        return pos_bar, paint(pos, mesh.shape, weights=out_bar, order=order)
    flatmesh = mesh.reshape(-1)
    pos_bar = np.zeros((len(pos), mesh.ndim))
    for flat, ker, dker in stencil(pos, mesh.shape, order, kernel_type, oversamp):
        pos_bar += (out_bar * flatmesh[flat])[:, None] * _dprod(ker, dker)
    mesh_bar = paint(pos, mesh.shape, weights=out_bar, order=order, kernel_type=kernel_type, oversamp=oversamp)
    return pos_bar, mesh_bar


# --------------------------------------------------------------------------- observation-side painting
def deconv_paint(mesh, order=2, kernel_type="rectangular", oversamp=1.):
    """montecosmo/nbody.py:315-334"""
    hat = (lambda kv: rectangular_hat(kv, order)) if kernel_type == "rectangular" else \
        (lambda kv: kaiser_bessel_hat(kv, order, optim_kcut(oversamp)))
    if np.isrealobj(mesh):
        kvec = rfftk(mesh.shape)
        return _irfftn(_rfftn(mesh) / hat(kvec), s=mesh.shape, axes=(0, 1, 2))
    return mesh / hat(rfftk(ch2rshape(mesh.shape)))


def interlace(pos, shape, weights=1., paint_order=2, interlace_order=2, kernel_type="rectangular", paint_oversamp=1.):
    """montecosmo/nbody.py:513-529"""
    kvec = rfftk(shape)
    mesh = np.zeros(r2chshape(shape), dtype=complex)
    for shift in np.arange(interlace_order) / interlace_order:
        m = paint(pos + shift, shape, weights, paint_order, kernel_type, paint_oversamp)
        mesh = mesh + _rfftn(m) * np.exp(1j * shift * sum(kvec)) / interlace_order
    return mesh


# --------------------------------------------------------------------------- real Gaussian <-> complex Hermitian
def _rg2cgh(mesh, part="real", norm="backward"):
    """montecosmo/utils.py:785-836, statement by statement (numpy slices instead of .at[].set)."""
    shape = np.array(mesh.shape)
    assert np.all(shape % 2 == 0), "dimension lengths must be even."
    hx, hy, hz = shape // 2
    meshk = np.zeros(r2chshape(tuple(int(v) for v in shape)))
    if part == "imag":
        slix, sliy, sliz = slice(hx + 1, None), slice(hy + 1, None), slice(hz + 1, None)
    else:
        assert part == "real"
        slix, sliy, sliz = slice(1, hx), slice(1, hy), slice(1, hz)
    meshk[:, :, 1:-1] = mesh[:, :, sliz]
    for k in [0, hz]:
        meshk[:, 1:hy, k] = mesh[:, sliy, k]
        meshk[1:, hy + 1:, k] = mesh[1:, sliy, k][::-1, ::-1]
        meshk[0, hy + 1:, k] = mesh[0, sliy, k][::-1]
        if part == "imag" and norm != "amp":
            meshk[:, hy + 1:, k] *= -1.
        for j in [0, hy]:
            meshk[1:hx, j, k] = mesh[slix, j, k]
            meshk[hx + 1:, j, k] = mesh[slix, j, k][::-1]
            if part == "imag" and norm != "amp":
                meshk[hx + 1:, j, k] *= -1.
            for i in [0, hx]:
                if part == "real":
                    meshk[i, j, k] = mesh[i, j, k]
                    if norm != "amp":
                        meshk[i, j, k] *= 2 ** .5
    shape = shape.astype(float)
    if norm == "backward":
        meshk /= (2 / shape.prod()) ** .5
    elif norm == "ortho":
        meshk /= 2 ** .5
    elif norm == "forward":
        meshk /= (2 * shape.prod()) ** .5
    else:
        assert norm == "amp"
    return meshk


def rg2cgh(mesh, norm="backward"):
    """montecosmo/utils.py:892-906: permute and reweight a real Gaussian tensor into a complex Gaussian Hermitian one,
    distributed as rfftn of a real Gaussian tensor."""
    real = _rg2cgh(mesh, "real", norm)
    if norm == "amp":
        return real
    return real + 1j * _rg2cgh(mesh, "imag", norm)


def _cgh2rg(meshk, part="real", norm="backward"):
    """montecosmo/utils.py:839-889."""
    shape = np.array(ch2rshape(meshk.shape))
    assert np.all(shape % 2 == 0)
    hx, hy, hz = shape // 2
    mesh = np.zeros(tuple(int(v) for v in shape))
    if part == "imag":
        slix, sliy, sliz = slice(hx + 1, None), slice(hy + 1, None), slice(hz + 1, None)
    else:
        slix, sliy, sliz = slice(1, hx), slice(1, hy), slice(1, hz)
    mesh[:, :, sliz] = meshk[:, :, 1:-1]
    for k in [0, hz]:
        mesh[:, sliy, k] = meshk[:, 1:hy, k]
        mesh[1:, sliy, k] = meshk[1:, hy + 1:, k][::-1, ::-1]
        mesh[0, sliy, k] = meshk[0, hy + 1:, k][::-1]
        if part == "imag" and norm != "amp":
            mesh[:, sliy, k] *= -1.
        for j in [0, hy]:
            mesh[slix, j, k] = meshk[1:hx, j, k]
            mesh[slix, j, k] = meshk[hx + 1:, j, k][::-1]
            if part == "imag" and norm != "amp":
                mesh[slix, j, k] *= -1.
            for i in [0, hx]:
                if part == "real":
                    mesh[i, j, k] = meshk[i, j, k]
                    if norm != "amp":
                        mesh[i, j, k] /= 2 ** .5
    shape = shape.astype(float)
    if norm == "backward":
        mesh *= (2 / shape.prod()) ** .5
    elif norm == "ortho":
        mesh *= 2 ** .5
    elif norm == "forward":
        mesh *= (2 * shape.prod()) ** .5
    return mesh


def cgh2rg(meshk, norm="backward"):
    """montecosmo/utils.py:909-921."""
    meshk = np.asarray(meshk)
    real = _cgh2rg(meshk.real, "real", norm)
    imag = _cgh2rg(meshk.real if norm == "amp" else meshk.imag, "imag", norm)
    return real + imag


# --------------------------------------------------------------------------- spectrum reshape
def hermitian_symmetric(arr):
    """montecosmo/utils.py:968-978: conj of the index-reversed array, rolled by one along every axis
    (element i -> conj(arr[-i mod n]))."""
    arr = np.asarray(arr)
    out = arr[(slice(None, None, -1),) * arr.ndim].conj()
    for ax in range(arr.ndim):
        out = np.roll(out, 1, axis=ax)
    return out


def _chreshape_naive(mesh, shape):
    """montecosmo/utils.py:924-964: crop / zero-pad the centred wavevectors, scale by the real cell-count ratio."""
    scale = np.divide(ch2rshape(shape), ch2rshape(mesh.shape)).prod()
    for ax, s in enumerate(mesh.shape[:-1]):
        mesh = np.roll(mesh, s // 2, ax)
    slices = ()
    for ax, (ms, s) in enumerate(zip(mesh.shape, shape)):
        trunc = max(ms - s, 0)
        if ax < len(shape) - 1:
            trunc //= 2
            slices += (slice(trunc, None if trunc == 0 else -trunc),)
        else:
            slices += (slice(0, None if trunc == 0 else -trunc),)
    mesh = mesh[slices]
    pad_width = ()
    for ax, (ms, s) in enumerate(zip(mesh.shape, shape)):
        pad = max(s - ms, 0)
        if ax < len(shape) - 1:
            pad //= 2
            pad_width += ((pad, pad),)
        else:
            pad_width += ((0, pad),)
    mesh = np.pad(mesh, pad_width=pad_width)
    for ax, s in enumerate(mesh.shape[:-1]):
        mesh = np.roll(mesh, -s // 2, ax)
    return mesh * scale


def chreshape(mesh, shape):
    """montecosmo/utils.py:981-1013: reshape a half-spectrum to the half-spectrum shape `shape`, truncating or
    padding so that Hermitian symmetry and the mean power are preserved (Nyquist planes aggregated / split with
    1/sqrt(2) weights).  Axes are visited last-to-first for the truncations, first-to-last for the pads."""
    mesh = np.array(mesh, dtype=np.complex128)
    shape = tuple(int(v) for v in shape)
    nd = len(shape)
    for ax, (ms, s) in reversed(list(enumerate(zip(mesh.shape, shape)))):
        if s < ms:
            if ax < nd - 1:
                neg_ids = (slice(None),) * ax + (-s // 2,)
                pos_ids = (slice(None),) * ax + (s // 2,)
                mesh[neg_ids] = (mesh[pos_ids] + mesh[neg_ids]) / 2 ** .5
            else:
                pos_ids = (slice(None),) * ax + (s - 1,)
                nyq = mesh[pos_ids].copy()
                mesh[pos_ids] = (nyq + hermitian_symmetric(nyq)) / 2 ** .5
    in_shape = mesh.shape
    out = _chreshape_naive(mesh, shape)
    for ax, (ms, s) in enumerate(zip(in_shape, shape)):
        if s > ms:
            if ax < nd - 1:
                neg_ids = (slice(None),) * ax + (-ms // 2,)
                pos_ids = (slice(None),) * ax + (ms // 2,)
                out[neg_ids] /= 2 ** .5
                out[pos_ids] = out[neg_ids]
            else:
                pos_ids = (slice(None),) * ax + (ms - 1,)
                out[pos_ids] /= 2 ** .5
    return out


def chreshape_vjp(out_bar, in_shape):
    """VJP of chreshape (a real-linear map: it conjugates on the truncated last-axis Nyquist plane) under
    dL = Re sum conj(bar) dz.  Built by transposing the forward statement step by step."""
    ob = np.array(out_bar, dtype=np.complex128)
    shape = ob.shape
    in_shape = tuple(int(v) for v in in_shape)
    nd = len(shape)
    # transpose of the pad fix-ups, last applied first
    for ax, (ms, s) in reversed(list(enumerate(zip(in_shape, shape)))):
        if s > ms:
            if ax < nd - 1:
                neg_ids = (slice(None),) * ax + (-ms // 2,)
                pos_ids = (slice(None),) * ax + (ms // 2,)
                ob[neg_ids] = (ob[neg_ids] + ob[pos_ids]) / 2 ** .5   # out[neg] /= r2; out[pos] = out[neg]
                ob[pos_ids] = 0.
            else:
                pos_ids = (slice(None),) * ax + (ms - 1,)
                ob[pos_ids] /= 2 ** .5
    # transpose of the naive crop/pad (a selection matrix times a scale) = the naive map the other way round,
    # rescaled: crop <-> zero-pad
    scale = np.divide(ch2rshape(shape), ch2rshape(in_shape)).prod()
    back = _chreshape_naive(ob, in_shape) * scale / np.divide(ch2rshape(in_shape), ch2rshape(shape)).prod()
    # transpose of the truncation aggregations, first axis first (they were applied last axis first)
    for ax, (ms, s) in enumerate(zip(in_shape, shape)):
        if s < ms:
            if ax < nd - 1:
                neg_ids = (slice(None),) * ax + (-s // 2,)
                pos_ids = (slice(None),) * ax + (s // 2,)
                t = back[neg_ids] / 2 ** .5       # new_neg = (pos + neg)/r2, pos kept
                back[pos_ids] = back[pos_ids] + t
                back[neg_ids] = t
            else:
                pos_ids = (slice(None),) * ax + (s - 1,)
                t = back[pos_ids] / 2 ** .5       # new = (z + Hsym(z))/r2 ; adjoint of z -> Hsym(z) is w -> Hsym(w)
                back[pos_ids] = t + hermitian_symmetric(t)
    return back


def _nufft_oversamp(final_shape, paint_shape):
    """paint_oversamp as montecosmo/nbody.py:557-566 derives it."""
    if paint_shape is None:
        return 1.
    if isinstance(paint_shape, float):
        return paint_shape
    return float(np.exp(np.log(np.divide(tuple(final_shape), tuple(paint_shape))).mean()))


def nufft(pos, final_shape, paint_shape=None, weights=1., paint_order=2, interlace_order=2, paint_deconv=True,
          kernel_type="rectangular"):
    """montecosmo/nbody.py:532-577: `pos` in cell units of final_shape; the particles are painted on paint_shape
    (tuple, or float oversampling factor), deconvolved there and reshaped to final_shape."""
    final_shape = tuple(int(v) for v in final_shape)
    paint_oversamp = _nufft_oversamp(final_shape, paint_shape)
    if paint_shape is None:
        paint_shape = final_shape
    elif isinstance(paint_shape, float):
        paint_shape = scale_shape(final_shape, paint_shape)
    paint_shape = tuple(int(v) for v in paint_shape)
    ratio = np.divide(paint_shape, final_shape)
    mesh = interlace(np.asarray(pos) * ratio, paint_shape, weights, paint_order, interlace_order, kernel_type, paint_oversamp)
    mesh = mesh * ratio.prod()
    if paint_deconv:
        mesh = deconv_paint(mesh, paint_order, kernel_type, paint_oversamp)
    if final_shape != paint_shape:
        mesh = chreshape(mesh, r2chshape(final_shape))
    return mesh


def nufft_vjp(pos, final_shape, weights, mesh_bar, paint_order=2, interlace_order=2, paint_deconv=True, paint_shape=None,
              kernel_type="rectangular"):
    """VJP of nufft w.r.t. (pos, weights); mesh_bar in the real-pair convention."""
    final_shape = tuple(int(v) for v in final_shape)
    paint_oversamp = _nufft_oversamp(final_shape, paint_shape)
    if paint_shape is None:
        paint_shape = final_shape
    elif isinstance(paint_shape, float):
        paint_shape = scale_shape(final_shape, paint_shape)
    shape = tuple(int(v) for v in paint_shape)
    ratio = np.divide(shape, final_shape)
    if shape != final_shape:
        mesh_bar = chreshape_vjp(mesh_bar, r2chshape(shape))
    mesh_bar = mesh_bar * ratio.prod()
    kvec = rfftk(shape)
    hat = rectangular_hat(kvec, paint_order) if kernel_type == "rectangular" else kaiser_bessel_hat(kvec, paint_order, optim_kcut(paint_oversamp))
    mult = 1.0 / hat if paint_deconv else 1.0
    pos_bar, w_bar = 0., 0.
    for shift in np.arange(interlace_order) / interlace_order:
        sb = mesh_bar * np.conj(mult * np.exp(1j * shift * sum(kvec)) / interlace_order)
        pb, wb = paint_vjp(np.asarray(pos) * ratio + shift, shape, weights, rfftn_vjp(sb, shape), paint_order, kernel_type, paint_oversamp)
        pos_bar, w_bar = pos_bar + pb * ratio, w_bar + wb
    return pos_bar, w_bar


# --------------------------------------------------------------------------- FFT adjoints
def _zweights(shape_r):
    """w = (1,2,...,2,1) along the half axis: multiplicity of each stored mode in irfftn."""
    nzh = shape_r[-1] // 2 + 1
    w = np.full(nzh, 2.0)
    w[0] = 1.0
    if shape_r[-1] % 2 == 0:
        w[-1] = 1.0
    return w


def irfftn_vjp(real_bar):
    """y = irfftn(X) (numpy: ifft over leading axes, c2r over last)  ->  X_bar = (w/M) rfftn(y_bar)."""
    M = real_bar.size
    return _rfftn(real_bar) * (_zweights(real_bar.shape) / M)


def rfftn_vjp(spec_bar, shape_r):
    """X = rfftn(rho), rho real  ->  rho_bar = M * irfftn(X_bar / w)."""
    M = int(np.prod(shape_r))
    return _irfftn(spec_bar / _zweights(shape_r), s=shape_r, axes=tuple(range(len(shape_r)))) * M


# --------------------------------------------------------------------------- forces
def pm_forces(pos, mesh, read_order=2, paint_deconv=False, grad_fd=np.inf, lap_fd=np.inf, kcut=np.inf):
    """montecosmo/nbody.py:583-604.  `mesh` a shape tuple (paint first) or a half-spectrum."""
    if isinstance(mesh, tuple):
        mesh = _rfftn(paint(pos, mesh, order=read_order))
        if paint_deconv:
            kvec = rfftk(ch2rshape(mesh.shape))
            mesh = mesh / rectangular_hat(kvec, order=read_order) ** 2
    kvec = rfftk(ch2rshape(mesh.shape))
    pot = mesh * invlaplace_hat(kvec, lap_fd)
    if kcut != np.inf:
        pot = pot * gaussian_hat(kvec, kcut)
    return np.stack([read(pos, _irfftn(-gradient_hat(kvec, i, grad_fd) * pot), read_order)
                     for i in range(len(kvec))], axis=-1)


def force_meshes(mesh_k, grad_fd=np.inf, lap_fd=np.inf):
    """The three real force meshes irfftn(-(i k_c) * (-1/k^2) * mesh_k) of nbody.py:597-603."""
    kvec = rfftk(ch2rshape(mesh_k.shape))
    pot = mesh_k * invlaplace_hat(kvec, lap_fd)
    return [_irfftn(-gradient_hat(kvec, i, grad_fd) * pot) for i in range(3)]


def pm_forces_vjp(pos, mesh, forces_bar, read_order=2, paint_deconv=False, grad_fd=np.inf, lap_fd=np.inf,
                  kcut=np.inf):
    """VJP of pm_forces.  Returns (pos_bar, mesh_bar) with mesh_bar None when `mesh` is a shape tuple
    (then pos_bar also carries the dependence through the painted density)."""
    painted = isinstance(mesh, tuple)
    if painted:
        shape = mesh
        spec = _rfftn(paint(pos, shape, order=read_order))
    else:
        spec = mesh
        shape = ch2rshape(mesh.shape)
    kvec = rfftk(shape)
    kern = invlaplace_hat(kvec, lap_fd)
    if painted and paint_deconv:
        kern = kern / rectangular_hat(kvec, order=read_order) ** 2
    if kcut != np.inf:
        kern = kern * gaussian_hat(kvec, kcut)
    pot = spec * kern
    pos_bar = np.zeros_like(pos, dtype=np.float64)
    pot_bar = np.zeros_like(pot)
    for c in range(3):
        gk = -gradient_hat(kvec, c, grad_fd)
        fmesh = _irfftn(gk * pot)
        pb, mb = read_vjp(pos, fmesh, forces_bar[:, c], read_order)
        pos_bar += pb
        pot_bar += np.conj(gk) * irfftn_vjp(mb)
    spec_bar = pot_bar * np.conj(kern)
    if painted:
        rho_bar = rfftn_vjp(spec_bar, shape)
        pos_bar += paint_vjp(pos, shape, 1., rho_bar, order=read_order)[0]
        return pos_bar, None
    return pos_bar, spec_bar


def delta2_mesh(mesh_k, grad_fd=np.inf, lap_fd=np.inf, return_hess=False):
    """2LPT source of nbody.py:611-627 (running-sum form kept)."""
    kvec = rfftk(ch2rshape(mesh_k.shape))
    pot = mesh_k * invlaplace_hat(kvec, lap_fd)
    delta2 = 0.
    hesses = 0.
    hess = {}
    for i in range(3):
        hess_ii = _irfftn(gradient_hat(kvec, i, grad_fd) ** 2 * pot)
        hess[(i, i)] = hess_ii
        delta2 = delta2 + hess_ii * hesses
        hesses = hesses + hess_ii
        for j in range(i + 1, 3):
            hess_ij = _irfftn(gradient_hat(kvec, i, grad_fd) * gradient_hat(kvec, j, grad_fd) * pot)
            hess[(i, j)] = hess_ij
            delta2 = delta2 - hess_ij ** 2
    return (delta2, hess) if return_hess else delta2


def pm_forces2(pos, mesh_k, read_order=2, grad_fd=np.inf, lap_fd=np.inf):
    """montecosmo/nbody.py:607-631"""
    delta2 = delta2_mesh(mesh_k, grad_fd, lap_fd)
    return pm_forces(pos, _rfftn(delta2), read_order, grad_fd=grad_fd, lap_fd=lap_fd)


def pm_forces2_vjp(pos, mesh_k, forces_bar, read_order=2, grad_fd=np.inf, lap_fd=np.inf):
    """VJP of pm_forces2 w.r.t. (pos, mesh_k)."""
    shape = ch2rshape(mesh_k.shape)
    kvec = rfftk(shape)
    delta2, hess = delta2_mesh(mesh_k, grad_fd, lap_fd, return_hess=True)
    pos_bar, d2k_bar = pm_forces_vjp(pos, _rfftn(delta2), forces_bar, read_order, grad_fd=grad_fd, lap_fd=lap_fd)
    d2_bar = rfftn_vjp(d2k_bar, shape)
    pot_bar = 0.
    for i in range(3):
        others = sum(hess[(j, j)] for j in range(3) if j != i)
        pot_bar = pot_bar + np.conj(gradient_hat(kvec, i, grad_fd) ** 2) * irfftn_vjp(d2_bar * others)
        for j in range(i + 1, 3):
            gij = gradient_hat(kvec, i, grad_fd) * gradient_hat(kvec, j, grad_fd)
            pot_bar = pot_bar + np.conj(gij) * irfftn_vjp(-2 * d2_bar * hess[(i, j)])
    return pos_bar, pot_bar * invlaplace_hat(kvec, lap_fd)


def lpt(cosmo, init_mesh, pos, a, lpt_order=2, read_order=2, grad_fd=np.inf, lap_fd=np.inf):
    """montecosmo/nbody.py:634-667"""
    if np.isrealobj(init_mesh):
        init_mesh = _rfftn(init_mesh)
    force1 = pm_forces(pos, init_mesh, read_order, grad_fd=grad_fd, lap_fd=lap_fd)
    dpos = a2g(cosmo, a) * force1
    vel = force1
    if lpt_order == 2:
        force2 = pm_forces2(pos, init_mesh, read_order, grad_fd=grad_fd, lap_fd=lap_fd)
        dpos = dpos - a2g2(cosmo, a) * force2
        vel = vel - a2dg2dg(cosmo, a) * force2
    return dpos, vel


def lpt_vjp(cosmo, init_mesh, pos, a, dpos_bar, vel_bar, lpt_order=2, read_order=2, grad_fd=np.inf, lap_fd=np.inf):
    """VJP of lpt w.r.t. init_mesh (complex half-spectrum), pos, and the three growth scalars.
    Returns (init_mesh_bar, pos_bar, {'g':..., 'g2':..., 'dg2dg':...}); the growth cotangents are scalars for a scalar
    `a` and per-particle arrays (N,) for `a` of shape (N,1) (light cone)."""
    g, g2, c = a2g(cosmo, a), a2g2(cosmo, a), a2dg2dg(cosmo, a)
    red = (lambda x: float(np.sum(x))) if np.ndim(a) == 0 else (lambda x: np.sum(x, axis=-1))
    force1 = pm_forces(pos, init_mesh, read_order, grad_fd=grad_fd, lap_fd=lap_fd)
    f1_bar = g * dpos_bar + vel_bar
    sbar = {'g': red(dpos_bar * force1), 'g2': 0., 'dg2dg': 0.}
    pos_bar, mesh_bar = pm_forces_vjp(pos, init_mesh, f1_bar, read_order, grad_fd=grad_fd, lap_fd=lap_fd)
    if lpt_order == 2:
        force2 = pm_forces2(pos, init_mesh, read_order, grad_fd=grad_fd, lap_fd=lap_fd)
        sbar['g2'] = -red(dpos_bar * force2)
        sbar['dg2dg'] = -red(vel_bar * force2)
        f2_bar = -g2 * dpos_bar - c * vel_bar
        pb, mb = pm_forces2_vjp(pos, init_mesh, f2_bar, read_order, grad_fd=grad_fd, lap_fd=lap_fd)
        pos_bar = pos_bar + pb
        mesh_bar = mesh_bar + mb
    return mesh_bar, pos_bar, sbar


# --------------------------------------------------------------------------- growth tables
growth_log10_amin = -3.
growth_steps = 128


def growth_table(cosmo, log10_amin=growth_log10_amin, steps=growth_steps):
    """montecosmo/nbody.py:679-745: RK4 on atab = logspace(-3, 0, 128) for (D, D2, D', D2')."""
    if "background.growth_factor" in cosmo._workspace:
        return cosmo._workspace["background.growth_factor"]
    atab = np.logspace(log10_amin, 0.0, steps)

    def D_derivs(y, x):
        q = 2.0
        q = q - (background.Omega_m_a(cosmo, x) + (1.0 + 3.0 * background.w(cosmo, x)) * background.Omega_de_a(cosmo, x)) / 2
        q = q / x
        r = 1.5 * background.Omega_m_a(cosmo, x) / x ** 2
        g1, g2 = y[0]
        f1, f2 = y[1]
        dy1da = [f1, -q * f1 + r * g1]
        dy2da = [f2, -q * f2 + r * g2 - r * g1 ** 2]
        return np.array([[dy1da[0], dy2da[0]], [dy1da[1], dy2da[1]]])

    y0 = np.array([[atab[0], -3.0 / 7 * atab[0] ** 2], [1.0, -6.0 / 7 * atab[0]]])
    y = background.odeint(D_derivs, y0, atab)
    dyda2 = D_derivs(np.transpose(y, (1, 2, 0)), atab)
    dyda2 = np.transpose(dyda2, (2, 0, 1))
    y1 = y[:, 0, 0]
    gtab = y1 / y1[-1]
    y2 = y[:, 0, 1]
    g2tab = y2 / y2[-1]
    ftab = y[:, 1, 0] / y1[-1] * atab / gtab
    f2tab = y[:, 1, 1] / y2[-1] * atab / g2tab
    htab = dyda2[:, 1, 0] / y1[-1] * atab / gtab
    h2tab = dyda2[:, 1, 1] / y2[-1] * atab / g2tab
    cache = {"a": atab, "g": gtab, "f": ftab, "h": htab, "g2": g2tab, "f2": f2tab, "h2": h2tab}
    cosmo._workspace["background.growth_factor"] = cache
    return cache


def a2g(cosmo, a):
    """nbody.py:750-754 (np.interp clamps like jnp.interp)."""
    c = growth_table(cosmo)
    return np.interp(a, c["a"], c["g"])


def a2g2(cosmo, a):
    """nbody.py:756-761"""
    c = growth_table(cosmo)
    return np.interp(a, c["a"], c["g2"]) * -3 / 7


def a2f(cosmo, a):
    """nbody.py:763-767"""
    c = growth_table(cosmo)
    return np.interp(a, c["a"], c["f"])


def a2f2(cosmo, a):
    """nbody.py:769-773"""
    c = growth_table(cosmo)
    return np.interp(a, c["a"], c["f2"])


def a2dg2dg(cosmo, a):
    """nbody.py:775-777"""
    g, g2, f, f2 = a2g(cosmo, a), a2g2(cosmo, a), a2f(cosmo, a), a2f2(cosmo, a)
    return safe_div(g2 * f2, g * f)


def g2a(cosmo, g):
    """nbody.py:781-785"""
    c = growth_table(cosmo)
    return np.interp(g, c["g"], c["a"])


def g2g2(cosmo, g):
    """nbody.py:787-792"""
    c = growth_table(cosmo)
    return np.interp(g, c["g"], c["g2"]) * -3 / 7


def g2f(cosmo, g):
    """nbody.py:794-798"""
    c = growth_table(cosmo)
    return np.interp(g, c["g"], c["f"])


def g2f2(cosmo, g):
    """nbody.py:800-804"""
    c = growth_table(cosmo)
    return np.interp(g, c["g"], c["f2"])


def g2dg2dg(cosmo, g):
    """nbody.py:806-808"""
    g2, f, f2 = g2g2(cosmo, g), g2f(cosmo, g), g2f2(cosmo, g)
    return safe_div(g2 * f2, g * f)


dist_log10_amin = -3.
dist_steps = 256


def _dist_table(cosmo, log10_amin=dist_log10_amin, steps=dist_steps):
    """nbody.py:842-856"""
    key = "background.radial_comoving_distance"
    if key not in cosmo._workspace:
        atab = np.logspace(log10_amin, 0.0, steps)

        def dchioverdlna(y, x):
            xa = np.exp(x)
            return background.dchioverda(cosmo, xa) * xa

        chitab = background.odeint(dchioverdlna, 0.0, np.log(atab))
        chitab = chitab[-1] - chitab
        cosmo._workspace[key] = {"a": atab, "chi": chitab}
    return cosmo._workspace[key]


def a2chi(cosmo, a):
    """nbody.py:817-859"""
    c = _dist_table(cosmo)
    return np.clip(np.interp(a, c["a"], c["chi"]), 0.0, None)


def chi2a(cosmo, chi):
    """nbody.py:862-884"""
    c = _dist_table(cosmo)
    return np.interp(chi, c["chi"][::-1], c["a"][::-1])


# --------------------------------------------------------------------------- BullFrog / FastPM
def alpha_bf(cosmo, g0, dg):
    """nbody.py:907-919"""
    g1 = g0 + dg / 2
    g2 = g0 + dg
    dg2dg0, dg2dg2 = g2dg2dg(cosmo, g0), g2dg2dg(cosmo, g2)
    lin_ratio = (g2g2(cosmo, g0) + dg2dg0 * dg / 2) / g1 - g1
    return (dg2dg2 - lin_ratio) / (dg2dg0 - lin_ratio)


def alpha_fpm(cosmo, g0, dg):
    """nbody.py:921-931"""
    g2 = g0 + dg
    a0, a2 = g2a(cosmo, g0), g2a(cosmo, g2)
    coeff0 = background.Esqr(cosmo, a0) ** .5 * g0 * g2f(cosmo, g0) * a0 ** 2
    coeff2 = background.Esqr(cosmo, a2) ** .5 * g2 * g2f(cosmo, g2) * a2 ** 2
    return coeff0 / coeff2


def bullfrog_vf(cosmo, dg, mesh_shape, paint_order=2, paint_deconv=False, grad_fd=np.inf, lap_fd=np.inf,
                alpha_fn=alpha_bf):
    """nbody.py:902-959.  The reference's kick always calls alpha_bf (nbody.py:937); `alpha_fn` lets the
    unused closure alpha_fpm (nbody.py:921-931) be selected for the "FastPM" configs."""
    mesh_shape = tuple(int(s) for s in mesh_shape)

    def kick(state, g0):
        pos, vel = state
        g1 = g0 + dg / 2
        forces = pm_forces(pos, mesh_shape, paint_order, paint_deconv=paint_deconv, grad_fd=grad_fd, lap_fd=lap_fd)
        alpha = alpha_fn(cosmo, g0, dg)
        return pos, alpha * vel + (1 - alpha) * forces / g1

    def drift(state, d):
        pos, vel = state
        return pos + vel * d, vel

    def vector_field(g0, state, args=None):
        old = state
        state = drift(state, dg / 2)
        state = kick(state, g0)
        state = drift(state, dg / 2)
        return tuple((new - o) / dg for new, o in zip(state, old))

    return vector_field


def euler_times(g0, g1, dg, n_steps):
    """diffrax==0.5.0 ConstantStepSize time grid: t accumulates by +dg and a step that lands within 1e-10
    (float64 tolerance of diffrax's `_clip_to_end`) of g1, or beyond it, is snapped to g1."""
    ts = [g0]
    for _ in range(n_steps):
        tn = ts[-1] + dg
        ts.append(g1 if tn > g1 - 1e-10 else tn)
    return ts


def nbody_bf(cosmo, init_mesh, pos, a0=0., a1=1., n_steps=5, paint_order=2, lpt_order=2, paint_deconv=False,
             grad_fd=np.inf, lap_fd=np.inf, snapshots=None, alpha_fn=alpha_bf, return_traj=False):
    """nbody.py:967-1002.  snapshots=None -> SaveAt(t1=True) (leading axis of 1); an int > 1 -> SaveAt(ts=linspace(g0,
    g1, n)); a list of scale factors -> SaveAt(ts=a2g(list)): diffrax interpolates the Euler solution linearly
    between step states."""
    n_steps = int(n_steps)
    g0 = float(a2g(cosmo, a0))
    g1 = float(a2g(cosmo, a1))
    dg = (g1 - g0) / n_steps
    mesh_shape = ch2rshape(init_mesh.shape)
    vf = bullfrog_vf(cosmo, dg, mesh_shape, paint_order, paint_deconv, grad_fd, lap_fd, alpha_fn)
    dpos, vel = lpt(cosmo, init_mesh, pos=pos, a=a0, lpt_order=lpt_order, read_order=1, grad_fd=grad_fd, lap_fd=lap_fd)
    state = (pos + dpos, vel)
    ts = euler_times(g0, g1, dg, n_steps)
    traj = [state]
    for i in range(n_steps):
        d = vf(ts[i], state)
        dt = ts[i + 1] - ts[i]
        state = tuple(y + v * dt for y, v in zip(state, d))
        traj.append(state)
    if snapshots is None or (isinstance(snapshots, int) and snapshots <= 1):
        out = (state[0][None], state[1][None])
    else:
        tq = np.linspace(g0, g1, snapshots) if isinstance(snapshots, int) else np.atleast_1d(a2g(cosmo, np.asarray(snapshots, dtype=float)))
        ps, vs = [], []
        for t in tq:
            i = int(np.clip(np.searchsorted(ts, t, side="right") - 1, 0, n_steps - 1))
            th = (t - ts[i]) / (ts[i + 1] - ts[i])
            ps.append(traj[i][0] + (traj[i + 1][0] - traj[i][0]) * th)
            vs.append(traj[i][1] + (traj[i + 1][1] - traj[i][1]) * th)
        out = (np.stack(ps), np.stack(vs))
    return (out, traj, ts, dg) if return_traj else out


def dkd_vjp(pos, vel, pos2_bar, vel1_bar, dg, alpha, g1mid, mesh_shape, paint_order=2, paint_deconv=False, grad_fd=np.inf,
            lap_fd=np.inf):
    """VJP of one drift-kick-drift map (nbody.py:946-950) at fixed scalars.
    Returns (pos_bar, vel_bar, alpha_bar, beta_bar, dg_bar) with beta = (1-alpha)/g1mid."""
    beta = (1 - alpha) / g1mid
    x1 = pos + vel * (dg / 2)
    F = pm_forces(x1, tuple(mesh_shape), paint_order, paint_deconv=paint_deconv, grad_fd=grad_fd, lap_fd=lap_fd)
    v1 = alpha * vel + beta * F
    dg_bar = 0.5 * float(np.sum(pos2_bar * v1))
    v1_bar = vel1_bar + pos2_bar * (dg / 2)
    x1_bar = pos2_bar.copy()
    alpha_bar = float(np.sum(v1_bar * vel))
    beta_bar = float(np.sum(v1_bar * F))
    x1_bar += pm_forces_vjp(x1, tuple(mesh_shape), beta * v1_bar, paint_order, paint_deconv=paint_deconv, grad_fd=grad_fd, lap_fd=lap_fd)[0]
    vel_bar = alpha * v1_bar + x1_bar * (dg / 2)
    dg_bar += 0.5 * float(np.sum(x1_bar * vel))
    return x1_bar, vel_bar, alpha_bar, beta_bar, dg_bar


def nbody_bf_vjp(cosmo, init_mesh, pos, pos_bar, vel_bar, a0=0., a1=1., n_steps=5, paint_order=2, lpt_order=2,
                 alpha_fn=alpha_bf, paint_deconv=False, grad_fd=np.inf, lap_fd=np.inf):
    """Reverse sweep of nbody_bf at fixed growth scalars (options as nbody.py:967-973).
    Returns (init_mesh_bar, scalar_bars) where scalar_bars holds per-step alpha_bar/beta_bar, dg_bar and the
    LPT growth-scalar bars; `pos_bar`/`vel_bar` are cotangents of the final (pos, vel), shape (N,3)."""
    (_, traj, ts, dg) = nbody_bf(cosmo, init_mesh, pos, a0, a1, n_steps, paint_order, lpt_order, paint_deconv, grad_fd, lap_fd,
                                 alpha_fn=alpha_fn, return_traj=True)
    mesh_shape = ch2rshape(init_mesh.shape)
    xb, vb = np.array(pos_bar, dtype=np.float64), np.array(vel_bar, dtype=np.float64)
    abar, bbar, dgbar = np.zeros(n_steps), np.zeros(n_steps), 0.
    for i in reversed(range(n_steps)):
        x, v = traj[i]
        r = (ts[i + 1] - ts[i]) / dg
        alpha = float(alpha_fn(cosmo, ts[i], dg))
        g1mid = ts[i] + dg / 2
        # y_{i+1} = y_i + (DKD(y_i) - y_i) * r
        xb2, vb2, ab, bb, db = dkd_vjp(x, v, r * xb, r * vb, dg, alpha, g1mid, mesh_shape, paint_order, paint_deconv, grad_fd, lap_fd)
        xb = (1 - r) * xb + xb2
        vb = (1 - r) * vb + vb2
        abar[i], bbar[i] = ab, bb
        dgbar += db
    mesh_bar, _, sbar = lpt_vjp(cosmo, init_mesh, pos, a0, xb, vb, lpt_order=lpt_order, read_order=1, grad_fd=grad_fd, lap_fd=lap_fd)
    sbar.update(alpha=abar, beta=bbar, dg=dgbar)
    return mesh_bar, sbar
