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
