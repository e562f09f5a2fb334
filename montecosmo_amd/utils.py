"""Host helpers with the names of montecosmo/utils.py that the PM path uses (utils.py:21-29, :769-782,
:1163-1168)."""
import numpy as np


def safe_div(x, y):
    """Division where x / 0 := 0 (utils.py:21-29)."""
    y = np.asarray(y)
    nz = y != 0
    return np.where(nz, x / np.where(nz, y, 1), 0)


def ch2rshape(shape):
    """Complex Hermitian shape -> real shape, last real dim assumed even (utils.py:769-776)."""
    shape = tuple(int(s) for s in shape)
    return shape[:-1] + (2 * (shape[-1] - 1),)


def r2chshape(shape):
    """Real shape -> complex Hermitian shape (utils.py:778-782)."""
    shape = tuple(int(s) for s in shape)
    return shape[:-1] + (shape[-1] // 2 + 1,)


def scale_shape(shape, scale=1.):
    """Valid (even) scaled mesh shape (utils.py:1163-1168)."""
    return tuple(int(2 * np.rint(s * scale / 2)) for s in shape)


def chreshape(mesh, shape):
    """Reshape a complex Hermitian tensor to the half-spectrum shape `shape`, truncating or padding so that the
    Hermitian symmetry and the mean (hence the average power) are preserved (utils.py:981-1013).  HIP kernel
    `mcpm_chreshape_c64`; returns a complex64 device tensor."""
    import ctypes as C
    import torch
    from . import nbody
    from ._lib import lib, check
    x = nbody._c64(mesh)
    ishape, oshape = ch2rshape(x.shape), ch2rshape(shape)
    out = torch.empty(tuple(int(v) for v in shape), dtype=torch.complex64, device=x.device)
    check(lib.mcpm_chreshape_c64(C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream), nbody._ptr(x), *ishape,
                                 nbody._ptr(out), *oshape), None, "mcpm_chreshape_c64")
    return out


def chreshape_vjp(out_bar, in_shape):
    """VJP of `chreshape`: cotangent of the reshaped spectrum -> cotangent of the input of half-spectrum shape
    `in_shape` (real-pair convention dL = Re sum conj(bar) dz)."""
    import ctypes as C
    import torch
    from . import nbody
    from ._lib import lib, check
    ob = nbody._c64(out_bar)
    ishape, oshape = ch2rshape(in_shape), ch2rshape(ob.shape)
    ib = torch.empty(tuple(int(v) for v in in_shape), dtype=torch.complex64, device=ob.device)
    check(lib.mcpm_chreshape_vjp_c64(C.c_void_p(torch.cuda.current_stream(ob.device).cuda_stream), nbody._ptr(ob), *oshape,
                                     nbody._ptr(ib), *ishape), None, "mcpm_chreshape_vjp_c64")
    return ib
