"""CPU-side checks: the C-ABI library loads and exports every symbol include/mcpm.h declares (no compute
calls without a GPU), error paths return codes instead of crashing, and the host-side float64 logic (growth
tables in libmcpm.so, integrator coefficients, wavevector helpers) agrees with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import pm_oracle as o, background as obg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libmod():
    from montecosmo_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(libmod):
    header = open(os.path.join(ROOT, "include", "mcpm.h")).read()
    declared = set(re.findall(r"\b(mcpm_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 35
    raw = C.CDLL(libmod.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"{name} declared in include/mcpm.h but not exported"
    assert declared == set(libmod.SIGNATURES), declared ^ set(libmod.SIGNATURES)


@pytest.mark.parametrize("compiler,flags", [("gcc", ["-std=c99", "-pedantic-errors"]), ("g++", ["-x", "c++", "-std=c++11"])])
def test_header_compiles_as_c_and_cpp(tmp_path, compiler, flags):
    """include/mcpm.h is the boundary of a C ABI: it has to be valid C99 and C++ on its own, and every declared function has to
    be a FREE function (round 3 left two prototypes inside `struct mcpm_comm_ops`, which the symbol regex above cannot see)."""
    import shutil
    import subprocess
    if shutil.which(compiler) is None:
        pytest.skip(f"{compiler} absent")
    header = open(os.path.join(ROOT, "include", "mcpm.h")).read()
    names = sorted(set(re.findall(r"\b(mcpm_[a-z0-9_]+)\s*\(", header)))
    src = tmp_path / "use_mcpm.c"
    # taking the address of every function fails to compile if one of them is not declared at file scope
    src.write_text('#include "mcpm.h"\nvoid *table[] = {\n' + "".join(f"    (void *) {n},\n" for n in names) +
                   "};\nint main(void) { mcpm_comm_ops ops; ops.ctx = 0; return table[0] == 0 || ops.ctx != 0; }\n")
    flags = [f for f in flags if f != "-pedantic-errors"]       # function -> void * casts are not strictly ISO
    r = subprocess.run([compiler, *flags, "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_error_codes_without_crash(libmod):
    lib = libmod.lib
    h = C.c_void_p()
    assert lib.mcpm_plan_create(16, 16, 15, 16, 16, 15, None, C.byref(h)) == -1       # MCPM_E_SHAPE: odd nz
    assert b"nz even" in lib.mcpm_last_error(None)
    assert lib.mcpm_plan_create(16, 16, 40000, 16, 16, 16, None, C.byref(h)) == -1    # int16 index range
    assert lib.mcpm_plan_create(16, 16, 16, 16, 16, 16, None, None) == -6             # MCPM_E_ARG
    assert lib.mcpm_plan_destroy(None) == 0
    assert lib.mcpm_fft_r2c(None, None, None, 1) == -6
    assert lib.mcpm_growth_table(0.3, 0.7, 0., -1., 0., -3., 1, *([None] * 7)) == -6
    assert lib.mcpm_version().startswith(b"mcpm")


def test_product_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from montecosmo_amd import nbody
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        nbody.paint(np.zeros((4, 3), np.float32), (8, 8, 8))
    # and nothing in the product imports the oracle
    pkg = os.path.join(ROOT, "montecosmo_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            assert "oracle" not in open(os.path.join(pkg, f)).read().replace("no oracle", ""), f


def test_growth_tables_match_oracle(libmod):
    from montecosmo_amd import nbody, bricks
    for kw in ({}, {"Omega_c": 0.5, "w0": -0.9, "wa": 0.1, "Omega_k": 0.02}):
        cg, co = bricks.Planck18(**kw), obg.Planck18(**kw)
        tg, to = nbody._growth_cache(cg), o.growth_table(co)
        for k in ("a", "g", "f", "g2", "f2", "h2"):
            assert np.allclose(tg[k], to[k], rtol=1e-11, atol=1e-14), k
        a = np.linspace(0, 1, 31)
        for name in ("a2g", "a2g2", "a2f", "a2f2", "a2dg2dg", "a2chi"):
            assert np.allclose(getattr(nbody, name)(cg, a), getattr(o, name)(co, a), rtol=1e-10, atol=1e-13), name
        gg = np.linspace(0.002, 1, 13)
        for name in ("g2a", "g2g2", "g2f", "g2f2", "g2dg2dg"):
            assert np.allclose(getattr(nbody, name)(cg, gg), getattr(o, name)(co, gg), rtol=1e-10, atol=1e-13), name
        assert np.allclose(nbody.chi2a(cg, nbody.a2chi(cg, a[3:])), a[3:], rtol=1e-6)
        for g0 in (0.01, 0.4, 0.85):
            assert np.isclose(nbody.alpha_bf(cg, g0, 0.1), o.alpha_bf(co, g0, 0.1), rtol=1e-10)
            assert np.isclose(nbody.alpha_fpm(cg, g0, 0.1), o.alpha_fpm(co, g0, 0.1), rtol=1e-10)


def test_product_growth_table_matches_valid_precond_notebook(libmod):
    """The product's host float64 RK4 (csrc/growth.cpp, mcpm_growth_table) against the reference-printed numbers of
    tests_old/valid_precond.ipynb:76-84 (see tests/test_oracle_known_answers.py for how they pin the table)."""
    from test_oracle_known_answers import PRECOND_DG
    steps = 256
    arrs = [np.zeros(steps) for _ in range(7)]
    rc = libmod.lib.mcpm_growth_table(0.31, 0.69, 0., -1., 0., -4., steps, *[a.ctypes.data_as(C.POINTER(C.c_double)) for a in arrs])
    assert rc == 0
    for a_obs, dg in PRECOND_DG.items():
        assert np.isclose(np.interp(a_obs, arrs[0], arrs[1]) / 20, dg, rtol=1e-13, atol=0), a_obs


def test_step_scalars_follow_the_euler_time_grid():
    from montecosmo_amd import nbody, bricks
    cg, co = bricks.Planck18(), obg.Planck18()
    n = 10
    dg, al, be, ls = nbody._step_scalars(cg, 0.0, 1.0, n, "bullfrog")
    g0, g1 = float(o.a2g(co, 0.0)), float(o.a2g(co, 1.0))
    ts = o.euler_times(g0, g1, (g1 - g0) / n, n)
    assert np.isclose(dg, (g1 - g0) / n, rtol=1e-14)
    assert np.allclose(al, [o.alpha_bf(co, t, dg) for t in ts[:-1]], rtol=1e-10)
    assert np.allclose(be, [(1 - o.alpha_bf(co, t, dg)) / (t + dg / 2) for t in ts[:-1]], rtol=1e-10)
    assert np.allclose(ls, [o.a2g(co, 0.0), o.a2g2(co, 0.0), o.a2dg2dg(co, 0.0)], rtol=1e-10)


def test_host_kernels_and_helpers_match_oracle():
    from montecosmo_amd import nbody, bricks, utils
    for shape, box in (((8, 6, 10), None), ((4, 4, 8), (100., 50., 25.))):
        for a, b in zip(nbody.rfftk(shape, box), o.rfftk(shape, box)):
            assert a.shape == b.shape and np.array_equal(a, b)
        for a, b in zip(nbody.fftk(shape, box), o.fftk(shape, box)):
            assert a.shape == b.shape and np.array_equal(a, b)
    kv = nbody.rfftk((8, 8, 8))
    assert [k.shape for k in kv] == [(8, 1, 1), (1, 8, 1), (1, 1, 5)]            # docstring example nbody.py:57-61
    for fd in (np.inf, 2, 4):
        assert np.allclose(nbody.invlaplace_hat(kv, fd), o.invlaplace_hat(kv, fd))
        for i in range(3):
            assert np.allclose(nbody.gradient_hat(kv, i, fd), o.gradient_hat(kv, i, fd))
    assert nbody.invlaplace_hat(kv)[0, 0, 0] == 0
    assert np.allclose(nbody.gaussian_hat(kv, 2.0), o.gaussian_hat(kv, 2.0)) and nbody.gaussian_hat(kv) == 1.
    for order in (1, 2, 3, 4):
        s = np.linspace(-2, 2, 41)
        assert np.allclose(nbody.rectangular(s, order), o.rectangular(s, order))
        assert np.allclose(nbody.rectangular_hat(kv, order), o.rectangular_hat(kv, order))
    assert utils.ch2rshape((8, 8, 5)) == (8, 8, 8) and utils.r2chshape((8, 8, 8)) == (8, 8, 5)
    assert utils.scale_shape((64, 64, 64), 7 / 4) == (112, 112, 112) == o.scale_shape((64, 64, 64), 7 / 4)
    assert np.array_equal(utils.safe_div(np.array([1., 2.]), np.array([0., 4.])), [0., 0.5])
    assert np.array_equal(bricks.regular_pos((4, 6, 8), (2, 3, 8)), o.regular_pos((4, 6, 8), (2, 3, 8)))


def test_synthetic_initial_conditions():
    from montecosmo_amd import synth
    s = synth.init_mesh(16, seed=0, rms_disp=2.0)
    assert s.shape == (16, 16, 9) and s.dtype == np.complex64
    F = o.pm_forces(o.regular_pos((16,) * 3), s.astype(np.complex128), 1)
    assert abs(np.sqrt((F ** 2).sum(-1).mean()) - 2.0) < 0.02
    assert np.array_equal(s, synth.init_mesh(16, seed=0, rms_disp=2.0))


def test_eisenstein_hu_power_table_matches_restatement():
    """montecosmo_amd/power.py (host numpy, vectorised) against the scalar restatement of the published fit in
    oracle/power_oracle.py: the sigma8 = 1 table on logspace(-4, 1, 256) h/Mpc (bricks.py:69-79 with kpow = None)."""
    from montecosmo_amd import power, bricks
    from oracle import power_oracle as po, background as obg
    for kw in (dict(), dict(Omega_c=0.20, h=0.72, n_s=0.93), dict(Omega_b=0.03, Omega_c=0.35)):
        ks, pows = power.lin_power_table(bricks.Planck18(**kw))
        ko, po_ = po.lin_power_table(obg.Planck18(**kw))
        assert np.array_equal(ks, ko) and np.allclose(pows, po_, rtol=1e-8)
        assert abs(power.sigma_r(ks, pows) - 1.0) < 1e-3
    c = bricks.Planck18()
    assert np.allclose(power.eisenstein_hu_transfer(c, [0.0, 1e-7]), 1.0, atol=1e-6)


def test_std2trunc_body_and_12_sigma_tails_match_restatement():
    """utils.py:189-226 incl. the lowtail / hightail soft-max forms beyond 12 sigma (the tails raised):
    product host code (value + two analytic derivatives) against the oracle restatement and central differences; the map
    is continuous across the switch at 12 sigma to the accuracy the reference's temperature gives (~1e-7 sigma)."""
    from montecosmo_amd.logdensity import std2trunc_and_derivs as f
    from oracle import bias_oracle as bo
    loc, sc = 0.8102, 0.01
    for low, high in ((0., np.inf), (-np.inf, np.inf), (0.6, 0.95), (0.75, np.inf)):
        for x in (-40., -13., -12.5, -12.0001, -11.9, -3., 0., 2., 11.9, 12.0001, 12.5, 30.):
            y, d1, d2 = f(x, loc, sc, low, high)
            assert abs(y - bo.std2trunc(x, loc, sc, low, high)) < 1e-13
            h = 1e-5
            assert abs(d1 - (f(x + h, loc, sc, low, high)[0] - f(x - h, loc, sc, low, high)[0]) / (2 * h)) < 1e-6 * abs(d1) + 1e-13
            assert abs(d2 - (f(x + h, loc, sc, low, high)[1] - f(x - h, loc, sc, low, high)[1]) / (2 * h)) < 1e-4 * abs(d2) + 1e-10
            assert low <= y <= high
        assert abs(f(-12.0 - 1e-9, loc, sc, low, high)[0] - f(-12.0 + 1e-9, loc, sc, low, high)[0]) < 1e-6 * sc


def test_window_walk_float_divisions_are_exact():
    """paint_tiled.hip::small_div: the set-up of a tile's window walk divides the thread index and the workgroup size by the run-time window
    widths as (int)((n + 0.5) * rcp(d)) instead of an integer division.  Exact for every case the kernels can meet -- n below 1024 by a
    width of 17 .. 29, 512 or 1024 by a product of two widths and the remainder by a width -- with the reciprocal one ulp off either way
    (v_rcp_f32's accuracy)."""
    f32 = np.float32

    def check(n, d):
        r = f32(1) / f32(d)
        for rr in (np.nextafter(r, f32(0)), r, np.nextafter(r, f32(2))):
            assert int(f32(f32(n) + f32(0.5)) * rr) == n // d, (n, d)

    for d in range(17, 30):
        for n in range(1024):
            check(n, d)
    for wy in range(17, 30):
        for r in range(61):
            check(r, wy)
        for wz in range(17, 30):
            for threads in (512, 1024):
                check(threads, wy * wz)
                check(threads - (threads // (wy * wz)) * wy * wz, wz)
