"""The oracle's multi-threaded back end (scipy.fft workers + the OpenMP float64 kernels of oracle/csrc/pm_kernels.c,
`pm_oracle.set_threads(n)`) against its single-threaded numpy path, which is the one pinned by the known answers and
the one that generated the golden fixtures.  CPU only."""
import os

import numpy as np
import pytest

from oracle import pm_oracle as o, background as obg


@pytest.fixture()
def threads():
    prev = o.set_threads(4)
    yield 4
    o.set_threads(prev)


def _both(fn):
    prev = o.set_threads(1)
    a = fn()
    o.set_threads(4)
    b = fn()
    o.set_threads(prev)
    return a, b


@pytest.mark.parametrize("order", [1, 2, 3, 4])
def test_particle_kernels_match_numpy_path(order):
    rng = np.random.default_rng(order)
    shape = (6, 8, 10)
    pos = rng.uniform(-25, 25, (3000, 3))
    pos[:50] = np.round(pos[:50] * 2) / 2                       # exact integers / half-integers: floor, round-half-even, sign(0)
    w, m, ob = rng.standard_normal(3000), rng.standard_normal(shape), rng.standard_normal(3000)
    for fn in (lambda: o.paint(pos, shape, w, order), lambda: o.paint(pos, shape, 2.5, order), lambda: o.read(pos, m, order),
               lambda: o.cell_index(pos, shape, order)):
        a, b = _both(fn)
        assert a.dtype == b.dtype and np.allclose(a, b, rtol=1e-13, atol=1e-13)
    for fn in (lambda: o.paint_vjp(pos, shape, w, m, order), lambda: o.paint_vjp(pos, shape, 1.5, m, order),
               lambda: o.read_vjp(pos, m, ob, order)):
        (a0, a1), (b0, b1) = _both(fn)
        assert np.allclose(a0, b0, rtol=1e-12, atol=1e-12) and np.allclose(a1, b1, rtol=1e-12, atol=1e-12)


def test_int16_wrap_of_the_base_cell():
    """|pos| beyond the int16 range wraps exactly as the reference's `.astype(int16)` does (nbody.py:369, :375)."""
    pos = np.array([[32766.7, -32768.4, 40000.2], [1.5, 2.5, -0.5]])
    a, b = _both(lambda: o.cell_index(pos, (16, 16, 16), 2))
    assert np.array_equal(a, b)


def test_nbody_and_gradient_match_golden_fixture(threads):
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nbody_16.npz"))
    n, n_steps, a0 = int(g["n"]), int(g["n_steps"]), float(g["a0"])
    shape = (n, n, n)
    cos, pos = obg.Planck18(), o.regular_pos(shape)
    p, v = o.nbody_bf(cos, g["init_mesh"].astype(np.complex128), pos, a0, 1., n_steps)
    assert np.allclose(p[0] - pos, g["final_disp"], rtol=1e-10, atol=1e-11)
    assert np.allclose(v[0], g["final_vel"], rtol=1e-10, atol=1e-11)
    assert np.array_equal(o.cell_index(p[0], shape), g["final_cell"])
    mb, sb = o.nbody_bf_vjp(cos, g["init_mesh"].astype(np.complex128), pos, g["pos_bar"].astype(np.float64),
                            g["vel_bar"].astype(np.float64), a0, 1., n_steps)
    assert np.allclose(mb, g["init_mesh_bar"], rtol=1e-9, atol=1e-10 * np.abs(g["init_mesh_bar"]).max())
    assert np.allclose(sb["alpha"], g["alpha_bar"], rtol=1e-9) and np.allclose(sb["beta"], g["beta_bar"], rtol=1e-9)
