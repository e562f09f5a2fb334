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


def _stream_of(t):
    import ctypes as C
    import torch
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _norm_factor(norm, shape):
    """rg2cgh(x, norm) / rg2cgh(x, "backward"): the norms of utils.py:826-835 differ by a constant only
    (backward sqrt(M/2), ortho 1/sqrt(2), forward 1/sqrt(2 M))."""
    M = float(shape[0]) * float(shape[1]) * float(shape[2])
    if norm == "ortho":
        return M ** -0.5
    if norm == "forward":
        return 1.0 / M
    raise ValueError(f"unknown norm {norm!r}: 'backward', 'ortho', 'forward' (and 'amp' for cgh2rg)")


def rg2cgh(mesh, norm="backward"):
    """Permute and reweight a real Gaussian tensor (3D, even sizes) into a complex Gaussian Hermitian tensor
    distributed as rfftn of a real Gaussian tensor (utils.py:892-906).  HIP kernel mcpm_rg2cgh_f32."""
    import torch
    from . import nbody
    from ._lib import lib, check
    x = nbody._f32(mesh)
    out = torch.empty(r2chshape(x.shape), dtype=torch.complex64, device=x.device)
    check(lib.mcpm_rg2cgh_f32(_stream_of(x), nbody._ptr(x), *x.shape, nbody._ptr(out)), None, "mcpm_rg2cgh_f32")
    return out * _norm_factor(norm, x.shape) if norm != "backward" else out


def rg2cgh_vjp(meshk_bar):
    """VJP of rg2cgh: cotangent of the complex tensor (real-pair convention) -> cotangent of the real tensor."""
    import torch
    from . import nbody
    from ._lib import lib, check
    kb = nbody._c64(meshk_bar)
    shape = ch2rshape(kb.shape)
    out = torch.empty(shape, dtype=torch.float32, device=kb.device)
    check(lib.mcpm_rg2cgh_vjp_f32(_stream_of(kb), nbody._ptr(kb), *shape, nbody._ptr(out)), None, "mcpm_rg2cgh_vjp_f32")
    return out


def cgh2rg(meshk, norm="backward"):
    """Permute and reweight a complex Gaussian Hermitian tensor into a real Gaussian tensor (utils.py:909-921): the
    inverse of rg2cgh.  norm="amp" lays a per-mode amplitude (the real part of `meshk`) out like the real tensor.
    HIP kernels mcpm_cgh2rg_f32 / mcpm_cgh2rg_amp_f32."""
    import torch
    from . import nbody
    from ._lib import lib, check
    k = nbody._c64(meshk)
    shape = ch2rshape(k.shape)
    out = torch.empty(shape, dtype=torch.float32, device=k.device)
    fn = "mcpm_cgh2rg_amp_f32" if norm == "amp" else "mcpm_cgh2rg_f32"
    check(getattr(lib, fn)(_stream_of(k), nbody._ptr(k), *shape, nbody._ptr(out)), None, fn)
    return out / _norm_factor(norm, shape) if norm not in ("backward", "amp") else out
