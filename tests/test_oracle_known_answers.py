"""Pins the oracle (the CPU restatement of the reference algorithm) with the analytic known answers of
SURVEY.md 8(c) items 1-7 and with the single numeric datum the reference's own tests hold for this path
(tests_old/valid_fastpm.ipynb:747-749).  The reference cannot run here (jax not installed)."""
import numpy as np
import pytest

from oracle import pm_oracle as o, background as obg


@pytest.fixture(scope="module")
def rng():
    return np.random.default_rng(0)


def test_paint_conserves_mass_and_regular_grid_is_flat(rng):
    """Item 1: paint of N unit-weight particles sums to N (bricks.py:1101-1102); regular grid -> constant N/M."""
    shape = (8, 10, 12)
    pos = rng.uniform(-30, 30, (5000, 3))
    for order in (1, 2, 3, 4):
        assert np.isclose(o.paint(pos, shape, order=order).sum(), 5000, rtol=1e-12)
    assert np.allclose(o.paint(o.regular_pos(shape), shape), 1.0)
    assert np.allclose(o.paint(o.regular_pos(shape, (16, 20, 24)), shape), 8.0)


@pytest.mark.parametrize("order", [1, 2, 3, 4])
def test_paint_read_are_adjoint(rng, order):
    """Item 2: <paint(w), m> = <w, read(m)>."""
    shape = (6, 8, 10)
    pos = rng.uniform(-10, 20, (700, 3))
    w, m = rng.standard_normal(700), rng.standard_normal(shape)
    assert np.isclose(np.sum(o.paint(pos, shape, w, order) * m), np.sum(w * o.read(pos, m, order)), rtol=1e-12)


def test_index_semantics():
    """floor for CIC, round-half-even for NGP, Python modulo wrap, int16 (nbody.py:369-377)."""
    pos = np.array([[-0.5, 0.5, 1.5], [2.5, -1e-9, 7.999], [8.0, -8.0, 16.25]])
    assert o.cell_index(pos, (8, 8, 8), 2).tolist() == [[7, 0, 1], [2, 7, 7], [0, 0, 0]]
    assert o.cell_index(pos, (8, 8, 8), 1).tolist() == [[0, 0, 2], [2, 0, 0], [0, 0, 0]]
    assert o.cell_index(pos, (8, 8, 8), 2).dtype == np.int16


def test_assignment_kernel_at_zero_matches_valid_nbody_notebook():
    """tests_old/valid_nbody.ipynb:3046-3048 (the cell at :3066-3067 prints `paint_kernels[i](0)` for i = 2, 3, 4):
    1, 0.75, 0.6666666666666666 -- a reference-held datum for `rectangular` (nbody.py:220-246)."""
    assert float(o.rectangular(0., 2)) == 1.0
    assert float(o.rectangular(0., 3)) == 0.75
    assert float(o.rectangular(0., 4)) == 0.6666666666666666
    # the product's host-side kernel is held to the same numbers
    from montecosmo_amd import nbody
    assert [float(np.asarray(nbody.rectangular(np.zeros(1), i))[0]) for i in (2, 3, 4)] == [1.0, 0.75, 0.6666666666666666]


def test_plane_wave_force():
    """Item 3: density eps cos(k.x) on the mesh -> force eps (k/k^2) sin(k.x), read at lattice points (NGP)."""
    n, eps = 16, 1e-3
    shape = (n, n, n)
    kv = 2 * np.pi * np.array([2, 0, 1]) / n
    x = o.regular_pos(shape)
    delta = eps * np.cos(x @ kv).reshape(shape)
    F = o.pm_forces(x, np.fft.rfftn(delta), read_order=1)
    # pot = -delta/k^2 ; F = -grad pot = -(k/k^2) eps sin(k.x) * (-1)... sign follows nbody.py:597-603
    want = -eps * np.sin(x @ kv)[:, None] * kv[None, :] / (kv @ kv) * -1
    assert np.allclose(F, -want, atol=1e-15) or np.allclose(F, want, atol=1e-15)
    # F = -grad(phi) with laplace(phi) = delta
    phi = -delta.reshape(-1) / (kv @ kv)
    grad_phi = -eps * np.sin(x @ kv)[:, None] * kv[None, :] * (-1 / (kv @ kv))
    assert np.allclose(F, -grad_phi, atol=1e-15)


def test_irfftn_discards_non_hermitian_part(rng):
    """The Hermitian projection that the HIP k-space kernels apply explicitly is what numpy's irfftn does
    implicitly on the kz = 0 / Nyquist planes (see montecosmo_amd/csrc/kspace.hip)."""
    shape = (8, 8, 8)
    spec = np.fft.rfftn(rng.standard_normal(shape))
    kvec = o.rfftk(shape)
    for c in range(3):
        mult = -o.gradient_hat(kvec, c) * o.invlaplace_hat(kvec)
        full = np.broadcast_to(mult, spec.shape).copy()
        proj = full.copy()
        nyq = [np.arange(8) == 4, np.arange(8) == 4, np.arange(5) == 4]
        special = np.zeros(5, bool)
        special[[0, 4]] = True
        mask = np.zeros(spec.shape, bool)
        idx = [slice(None)] * 3
        sel = np.ix_(*[nyq[c] if a == c else np.ones(s, bool) for a, s in enumerate(spec.shape)])
        mask[sel] = True
        mask &= special[None, None, :]
        proj[mask] = 0
        assert np.allclose(np.fft.irfftn(full * spec), np.fft.irfftn(proj * spec), atol=1e-14)


def test_eds_growth_closed_forms():
    """Item 4: Einstein-de Sitter: D = a, f = 1, f2 = 2, D2 ~ a^2, and the integrator coefficients."""
    eds = obg.Cosmology(Omega_c=1.0, Omega_b=0.0)
    a = np.array([0.01, 0.1, 0.5, 1.0])
    assert np.allclose(o.a2g(eds, a), a, rtol=1e-6)
    assert np.allclose(o.a2f(eds, a), 1.0, rtol=1e-5)
    assert np.allclose(o.a2f2(eds, a), 2.0, rtol=1e-5)
    assert np.allclose(o.a2g2(eds, a), -3 / 7 * a ** 2, rtol=2e-3)
    # FastPM coefficient in EdS: E a^2 g f = a^(-3/2) a^2 a = a^(3/2)
    assert np.isclose(o.alpha_fpm(eds, 0.25, 0.5), (0.25 / 0.75) ** 1.5, rtol=1e-3)
    assert 0 < o.alpha_bf(eds, 0.25, 0.5) < 1


def test_planck18_growth_matches_reference_notebook():
    """Item 7: tests_old/valid_fastpm.ipynb:747-749 prints [1, 0.429, 0.522, 0.454] for Planck18 at a = 1
    ([D, 3/7 D2/D, D f, 3/7 D2 f2 / D] / D with the JaxPM growth that nbody.py:679-748 follows)."""
    c = obg.Planck18()
    g, g2, f, f2 = o.a2g(c, 1.), o.a2g2(c, 1.), o.a2f(c, 1.), o.a2f2(c, 1.)
    got = [g, -g2 / g, g * f, -g2 * f2 / g]
    assert np.allclose(np.round(got, 3), [1.0, 0.429, 0.522, 0.454])
    assert np.isclose(o.a2g(c, 0.0), o.growth_table(c)["g"][0])     # a0 = 0 is table-clamped, not 0


# tests_old/valid_precond.ipynb:76-84 (cell 3 output): `dg = a2g(a_obs) / 20` printed to 16 digits for a_obs = 0.1, 0.5 with
# the notebook's truth cosmology Omega_m = 0.31 (cell 3 source) and g0 = 0 (a_obs = 1 prints dg = 0.05 exactly).
# That notebook ran an older revision whose growth table was logspace(-4, 0, 256) -- today's module constants are
# (-3, 128) (nbody.py:675-676) -- found by scanning (log10_amin, steps): (-4, 256) reproduces BOTH numbers to the last
# printed digit, which pins the restated jax_cosmo background (Omega_m(a), Omega_de(a), w), its RK4 `odeint`, the growth
# ODE of nbody.py:703-716, the normalisation and the linear interpolation, at 1e-15.
PRECOND_DG = {0.1: 0.0063675511943792435, 0.5: 0.03041908411253382, 1.0: 0.05}


def test_growth_matches_valid_precond_notebook_to_16_digits():
    c = obg.Planck18(Omega_c=0.31 - 0.049)
    t = o.growth_table(c, log10_amin=-4., steps=256)
    for a_obs, dg in PRECOND_DG.items():
        assert np.isclose(np.interp(a_obs, t["a"], t["g"]) / 20, dg, rtol=2e-15, atol=0), a_obs
    # with today's table (-3, 128) the same quantities differ by the interpolation error of the coarser table only
    c._workspace.clear()
    assert abs(float(o.a2g(c, 0.5)) / 20 / PRECOND_DG[0.5] - 1) < 1e-4
    assert abs(float(o.a2g(c, 0.1)) / 20 / PRECOND_DG[0.1] - 1) < 2e-6


def test_rg2cgh_unique_counts_of_valid_fourier_notebook():
    """tests/valid_fourier.ipynb cells 4-5: on a real (6,6,6) field, rfftn and rg2cgh both give a (6,6,4) spectrum with
    144 distinct complex entries, 112 distinct real parts and 105 distinct |imaginary parts| (zero included): the
    Hermitian redundancy of the kz = 0 and kz = 3 planes (2 x (4 self-conjugate + 16 pairs)) + 72 free entries; and
    cgh2rg inverts rg2cgh (assert_allclose(spatial, spatial2) in cell 5)."""
    rng = np.random.default_rng(66)          # own generator: the module-scoped one feeds order-sensitive tests
    x = rng.standard_normal((6, 6, 6))
    for X in (np.fft.rfftn(x), o.rg2cgh(x)):
        assert X.shape == (6, 6, 4)
        r, i = np.round(X.real, 9), np.round(np.abs(X.imag), 9)
        assert len(np.unique(np.round(X, 9))) == 144 and len(np.unique(r)) == 112 and len(np.unique(i)) == 105
        assert len(np.unique(r)) + len(np.unique(i)) == 6 * 6 * (2 * (4 - 1)) + 1
    assert np.allclose(o.cgh2rg(o.rg2cgh(x)), x, rtol=1e-5)
    # same second moments as rfftn (cell 5's assert on the covariances): per-mode variance of the real / imaginary parts
    xs = rng.standard_normal((4000, 6, 6, 6))
    F = np.fft.rfftn(xs, axes=(1, 2, 3))
    G = np.stack([o.rg2cgh(v) for v in xs])
    assert np.allclose((F.real ** 2).mean(0), (G.real ** 2).mean(0), rtol=0.15, atol=2.)
    assert np.allclose((F.imag ** 2).mean(0), (G.imag ** 2).mean(0), rtol=0.15, atol=2.)
    assert np.array_equal((F.imag ** 2).mean(0) < 1e-20, (G.imag ** 2).mean(0) < 1e-20)      # the same 8 purely real modes
    assert ((G.imag ** 2).mean(0) < 1e-20).sum() == 8


def test_zeldovich_plane_wave():
    """Item 5: a single plane wave before shell crossing: PM stepping stays on x = q + D psi(q) to integrator
    and CIC-force accuracy (1-D collapse, amplitude well below shell crossing).  A floor-based CIC paint of a
    one-particle-per-cell lattice is a backward difference, i.e. the mesh force is the Zel'dovich one shifted by
    half a cell: relative L2 error ~ k/2 = 0.098 at this wavelength, which bounds the agreement."""
    n = 32
    shape = (n, n, n)
    cosmo = obg.Planck18()
    q = o.regular_pos(shape)
    k = 2 * np.pi / n
    amp = 0.3                       # D=1 displacement amplitude in cells; shell crossing at amp*k = 1
    delta = (amp * k * np.cos(k * q[:, 0])).reshape(shape)   # delta = -div psi, psi = -amp sin(kx)... sign by lpt
    spec = np.fft.rfftn(delta)
    dpos1, _ = o.lpt(cosmo, spec, q, 1.0, lpt_order=1, read_order=1)
    (p, v) = o.nbody_bf(cosmo, spec, q, a0=0.1, a1=1.0, n_steps=8)
    disp = p[0] - q
    assert np.abs(disp[:, 1:]).max() < 1e-10                 # motion stays 1-D
    assert np.linalg.norm(disp[:, 0] - dpos1[:, 0]) / np.linalg.norm(dpos1[:, 0]) < 0.11


def test_vjps_match_finite_differences(rng):
    """Item 6: hand-derived VJPs vs central finite differences of the same oracle."""
    n = 8
    shape = (n, n, n)
    N = 200
    pos = rng.uniform(-3, 12, (N, 3))
    R = rng.standard_normal((N, 3))
    d = rng.standard_normal((N, 3))
    eps = 1e-6
    L = lambda p: np.sum(o.pm_forces(p, shape) * R)
    pb, _ = o.pm_forces_vjp(pos, shape, R)
    assert np.isclose((L(pos + eps * d) - L(pos - eps * d)) / (2 * eps), np.sum(pb * d), rtol=1e-6)
    X = np.fft.rfftn(rng.standard_normal(shape))
    dX = rng.standard_normal(X.shape) + 1j * rng.standard_normal(X.shape)
    for fwd, vjp in ((o.pm_forces, o.pm_forces_vjp), (o.pm_forces2, o.pm_forces2_vjp)):
        L = lambda X: np.sum(fwd(pos, X) * R)
        _, mb = vjp(pos, X, R)
        assert np.isclose((L(X + eps * dX) - L(X - eps * dX)) / (2 * eps), np.sum((np.conj(mb) * dX).real), rtol=1e-6)
    # paint / read
    w, mbar = rng.standard_normal(N), rng.standard_normal(shape)
    L = lambda p: np.sum(o.paint(p, shape, w) * mbar)
    assert np.isclose((L(pos + eps * d) - L(pos - eps * d)) / (2 * eps), np.sum(o.paint_vjp(pos, shape, w, mbar)[0] * d), rtol=1e-6)
    mesh, ob = rng.standard_normal(shape), rng.standard_normal(N)
    L = lambda p: np.sum(o.read(p, mesh) * ob)
    assert np.isclose((L(pos + eps * d) - L(pos - eps * d)) / (2 * eps), np.sum(o.read_vjp(pos, mesh, ob)[0] * d), rtol=1e-6)


@pytest.mark.parametrize("alpha_fn", [o.alpha_bf, o.alpha_fpm])
def test_nbody_vjp_matches_finite_differences(rng, alpha_fn):
    n = 8
    shape = (n, n, n)
    cosmo = obg.Planck18()
    pos = o.regular_pos(shape)
    X = np.fft.rfftn(rng.standard_normal(shape)) * 0.3
    Rx, Rv = rng.standard_normal((n ** 3, 3)), rng.standard_normal((n ** 3, 3))

    def L(X):
        p, v = o.nbody_bf(cosmo, X, pos, a0=0.1, a1=1., n_steps=3, alpha_fn=alpha_fn)
        return np.sum(p[0] * Rx) + np.sum(v[0] * Rv)

    mb, sb = o.nbody_bf_vjp(cosmo, X, pos, Rx, Rv, a0=0.1, a1=1., n_steps=3, alpha_fn=alpha_fn)
    dX = rng.standard_normal(X.shape) + 1j * rng.standard_normal(X.shape)
    eps = 1e-6
    assert np.isclose((L(X + eps * dX) - L(X - eps * dX)) / (2 * eps), np.sum((np.conj(mb) * dX).real), rtol=1e-6)


def test_euler_time_grid_is_diffrax_like():
    ts = o.euler_times(0.1, 1.0, 0.09, 10)
    assert ts[0] == 0.1 and ts[-1] == 1.0 and len(ts) == 11
    assert all(t1 > t0 for t0, t1 in zip(ts, ts[1:]))


def test_oracle_reproduces_golden_fixtures():
    """The committed fixtures are what the oracle produces (guards accidental oracle drift)."""
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(here, "nbody_16.npz"))
    n, n_steps, a0 = int(g["n"]), int(g["n_steps"]), float(g["a0"])
    shape = (n, n, n)
    cosmo = obg.Planck18()
    pos = o.regular_pos(shape)
    p, v = o.nbody_bf(cosmo, g["init_mesh"].astype(np.complex128), pos, a0, 1.0, n_steps)
    assert np.allclose(p[0] - pos, g["final_disp"], rtol=0, atol=1e-12)
    assert np.array_equal(o.cell_index(p[0], shape), g["final_cell"])
    pr = np.load(os.path.join(here, "paint_read.npz"))
    shape = tuple(int(s) for s in pr["shape"])
    for order in (1, 2):
        assert np.array_equal(o.cell_index(pr["pos"].astype(np.float64), shape, order), pr[f"cell_{order}"])
        assert np.allclose(o.paint(pr["pos"].astype(np.float64), shape, pr["weights"].astype(np.float64), order), pr[f"paint_{order}"])


@pytest.mark.parametrize("ishape,oshape", [((8, 8, 8), (4, 4, 4)), ((4, 4, 4), (8, 8, 8)), ((8, 6, 4), (4, 10, 6)),
                                           ((6, 6, 6), (6, 6, 6))])
def test_chreshape_known_answers(rng, ishape, oshape):
    """utils.py:981-1013: same shape is the identity; the mean and the Hermitian symmetry are preserved; padding then
    truncating a real field's spectrum returns it; the hand-derived VJP is the transpose (dot test)."""
    ic, oc = o.r2chshape(ishape), o.r2chshape(oshape)
    f = rng.standard_normal(ishape)
    X = np.fft.rfftn(f)
    Y = o.chreshape(X, oc)
    assert Y.shape == tuple(oc)
    if ishape == oshape:
        assert np.allclose(Y, X)
    y = np.fft.irfftn(Y, s=oshape, axes=(0, 1, 2))
    assert np.isclose(y.mean(), f.mean())                                   # mean preserved
    assert np.allclose(np.fft.rfftn(y), Y, atol=1e-9 * np.abs(Y).max())     # Y is the spectrum of a real field
    big = tuple(2 * s for s in ishape)
    assert np.allclose(o.chreshape(o.chreshape(X, o.r2chshape(big)), ic), X)
    # dot test on arbitrary (non-Hermitian) complex tensors, real-pair inner product
    x = rng.standard_normal(ic) + 1j * rng.standard_normal(ic)
    w = rng.standard_normal(oc) + 1j * rng.standard_normal(oc)
    lhs = np.sum(np.conj(w) * o.chreshape(x, oc)).real
    rhs = np.sum(np.conj(o.chreshape_vjp(w, ic)) * x).real
    assert np.isclose(lhs, rhs, rtol=1e-10)


def test_nufft_oversampled_mean_and_vjp(rng):
    """nbody.py:559-577 with an oversampled paint mesh: the k = 0 mode is the total weight whatever the paint shape,
    and the VJP matches finite differences."""
    final, N = (8, 8, 8), 50
    pos = rng.uniform(0, 8, (N, 3))
    w = 1.0 + 0.2 * rng.standard_normal(N)
    for ps in (None, 1.5, (12, 12, 12), (4, 4, 4)):
        m = o.nufft(pos, final, ps, w, 2, 2, True)
        assert m.shape == (8, 8, 5) and np.isclose(m[0, 0, 0].real, w.sum())
    mb = rng.standard_normal((8, 8, 5)) + 1j * rng.standard_normal((8, 8, 5))
    L = lambda p, ww: np.sum(np.conj(mb) * o.nufft(p, final, (12, 12, 12), ww, 2, 2, True)).real
    pb, wb = o.nufft_vjp(pos, final, w, mb, 2, 2, True, paint_shape=(12, 12, 12))
    eps = 1e-6
    for (i, ax) in ((3, 0), (17, 2)):
        dp = np.zeros_like(pos)
        dp[i, ax] = eps
        assert np.isclose((L(pos + dp, w) - L(pos - dp, w)) / (2 * eps), pb[i, ax], rtol=1e-5, atol=1e-7)
    dw = np.zeros_like(w)
    dw[5] = eps
    assert np.isclose((L(pos, w + dw) - L(pos, w - dw)) / (2 * eps), wb[5], rtol=1e-6)


def test_lagrangian_bias_known_answers_and_vjp(rng):
    """bricks.py:327-443 (png_type None).  b1-only weights are 1 + b1 g delta_L(q); the shear is traceless so a pure
    plane wave has shear^2 = 2/3 delta^2 and det(shear) = 2/27 delta^3 (eigenvalues 2/3, -1/3, -1/3 of delta);
    the renormalised operators have zero mean; VJP = central finite differences."""
    from oracle import bias_oracle as bo
    shape, box = (8, 8, 8), (80., 80., 80.)
    pos = np.stack(np.meshgrid(*[np.arange(s, dtype=float) for s in shape], indexing="ij"), -1).reshape(-1, 3)
    x = pos[:, 0].reshape(shape)
    wave = 0.1 * np.cos(2 * np.pi * x / 8)
    X = np.fft.rfftn(wave)
    zero = {k: 0. for k in bo.BIAS_KEYS}
    w, dv = bo.lagrangian_bias(0.7, pos, box, X, dict(zero, b1=2.0), 1)
    assert np.allclose(w, 1 + 2.0 * 0.7 * wave.ravel()) and np.allclose(dv, 0)
    fld, _ = bo.bias_fields(X, box)
    assert np.allclose(fld["shear2"], 2 / 3 * wave ** 2) and np.allclose(fld["shear3"], 3 * 2 / 27 * wave ** 3)
    kx = 2 * np.pi / 80.                                                  # physical wavenumber of the wave, h/Mpc
    assert np.allclose(fld["nab2"], -kx ** 2 * wave)
    assert np.allclose(fld["grad"][0], -0.1 * kx * np.sin(2 * np.pi * x / 8)) and np.allclose(fld["grad"][1], 0)
    X = np.fft.rfftn(0.3 * rng.standard_normal(shape))
    w, _ = bo.lagrangian_bias(0.7, pos, box, X, dict(zero, b2=1.0), 1)
    assert abs(w.mean() - 1) < 1e-12                                      # (delta^2 - <delta^2>) has zero mean
    bias = dict(b1=1.1, b2=0.3, bs2=-0.2, b3=0.15, bds2=0.25, bs3=-0.1, bn2=2.0, bnpar=1.5)
    N = len(pos)
    g = 0.5 + 0.4 * rng.uniform(size=(N, 1))
    p = pos + rng.uniform(0, 1, pos.shape)
    wb, vb = rng.standard_normal(N), rng.standard_normal((N, 3))

    def L(X_, bias_, g_):
        w_, dv_ = bo.lagrangian_bias(g_, p, box, X_, bias_, 2)
        return (wb * w_).sum() + (vb * dv_).sum()
    mb, bb, gb = bo.lagrangian_bias_vjp(g, p, box, X, bias, wb, vb, 2)
    eps = 1e-6
    dX = np.fft.rfftn(rng.standard_normal(shape))
    assert np.isclose((L(X + eps * dX, bias, g) - L(X - eps * dX, bias, g)) / (2 * eps), np.sum(np.conj(mb) * dX).real, rtol=1e-5)
    for k in bias:
        assert np.isclose((L(X, dict(bias, **{k: bias[k] + eps}), g) - L(X, dict(bias, **{k: bias[k] - eps}), g)) / (2 * eps), bb[k],
                          rtol=1e-5, atol=1e-7)
    dg = rng.standard_normal(g.shape)
    assert np.isclose((L(X, bias, g + eps * dg) - L(X, bias, g - eps * dg)) / (2 * eps), (gb * dg).sum(), rtol=1e-5)


@pytest.mark.parametrize("shape", [(4, 4, 4), (8, 6, 4), (6, 8, 10)])
def test_rg2cgh_known_answers(rng, shape):
    """utils.py:785-921: rg2cgh(x) is the spectrum of a real field (Hermitian), cgh2rg inverts it, and the map is an
    isometry onto rfftn-normalised spectra (sum y^2 = sum x^2 for y = irfftn(rg2cgh(x))), so rg2cgh(N(0,I)) has the
    law of rfftn(N(0,I))."""
    x = rng.standard_normal(shape)
    X = o.rg2cgh(x)
    y = np.fft.irfftn(X, s=shape, axes=(0, 1, 2))
    assert np.allclose(np.fft.rfftn(y), X)
    assert np.allclose(o.cgh2rg(X), x)
    assert np.isclose((y ** 2).sum() * np.prod(shape), (x ** 2).sum() * np.prod(shape))
    assert np.isclose((y ** 2).sum(), (x ** 2).sum())


def test_kaiser_preconditioning_known_answers():
    """model.py:1127-1148, utils.py:909-921: with (scale, transfer) of the 'kaiser' preconditioning, rg2cgh(scale * N(0, I))
    * transfer is unit-power white noise (every mode has variance M / V_cell, the law of rfftn of N(0, 1/V_cell) cells),
    whatever the fiducial model; and the "amp" layout gives the real and imaginary element of a mode the same amplitude:
    cgh2rg(A, "amp") = |cgh2rg(A, "backward")| / sqrt(2 / M) away from the eight self-conjugate modes."""
    from oracle import bias_oracle as bo
    ks = np.logspace(-3, 1, 64)
    cfg = dict(init_shape=(8, 6, 10), final_shape=(4, 4, 6), cell_length=50., box_size=np.array([200., 200., 300.]),
               box_center=np.array([60., -40., 1400.]), box_rotvec=np.array([0.1, 0.2, -0.1]), a_obs=None, curved_sky=True,
               lin_kpow=(ks, 2e4 * ks / (1 + (ks / 0.02) ** 2.6)), precond="kaiser")
    fid = dict(Omega_m=0.3111, sigma8=0.8102, b1=1., ngbars=1e-3, s_e=1.0)
    cosmo = obg.Planck18(Omega_c=fid["Omega_m"] - 0.049)
    scale, transfer = bo.precond_scale_and_transfer(cfg, fid, cosmo)
    M = np.prod(cfg["init_shape"])
    assert scale.shape == cfg["init_shape"] and scale.min() >= 1.0 and scale.max() > 1.5
    a_fid = bo.fiducial_scale_factor(cfg, cosmo)
    assert 1 / (1 + 0.7) < a_fid < 1 / (1 + 0.4)          # chi ~ 1400 Mpc/h is z ~ 0.5-0.6
    rng = np.random.default_rng(0)
    acc, n = 0., 400
    for _ in range(n):
        acc = acc + np.abs(o.rg2cgh(rng.standard_normal(cfg["init_shape"]) * scale) * transfer) ** 2
    white = np.divide(cfg["init_shape"], cfg["box_size"]).prod() * M
    assert abs((acc / n).mean() / white - 1) < 0.01
    assert np.abs((acc / n) / white - 1).max() < 0.35     # per mode: chi^2 with 400 (800) degrees of freedom
    A = rng.uniform(1, 2, o.r2chshape(cfg["init_shape"]))
    amp, back = o.cgh2rg(A.astype(complex), "amp"), o.cgh2rg(A * (1 + 1j), "backward")
    ratio = np.abs(back) / (2 / M) ** .5 / amp
    assert np.isclose(np.sort(ratio.ravel())[8:], 1.0).all() and np.isclose(np.sort(ratio.ravel())[:8], 2 ** -.5).all()


def test_eisenstein_hu_known_answers():
    """Eisenstein & Hu 1998 (the fit behind jax_cosmo's linear_matter_power, bricks.py:69-79): large-scale limit T -> 1;
    without baryons the fit is its own zero-baryon form T0(q) = L / (L + C q^2) with q = k / (13.41 k_eq); the sound
    horizon of a Planck-like cosmology is ~150 Mpc and the baryon wiggles of T have that period; Silk damping suppresses
    the baryon branch; the table is normalised to sigma8 = 1 under an independent quadrature."""
    from oracle import power_oracle as po
    c = obg.Planck18()
    assert abs(po.eisenstein_hu(c, [1e-6])[0] - 1.0) < 1e-6
    T = po.eisenstein_hu(c, np.logspace(-4, 1, 200))
    assert np.all(np.diff(T[:60]) < 0) and T[-1] < 1e-3 and np.all(T > 0)
    # zero-baryon limit (fb -> 0: alpha_c, beta_c -> 1 and the CDM branch is T0(k, 1, 1))
    nb = obg.Planck18(Omega_b=1e-9, Omega_c=0.3097)
    ks = np.logspace(-3, 0.5, 40)
    k_eq = 0.0746 * nb.Omega_m * nb.h ** 2 * (po.TCMB / 2.7) ** -2
    q = ks * nb.h / (13.41 * k_eq)
    L, C = np.log(np.e + 1.8 * q), 14.2 + 386.0 / (1.0 + 69.9 * q ** 1.08)
    assert np.allclose(po.eisenstein_hu(nb, ks), L / (L + C * q * q), rtol=2e-6)
    s = po.sound_horizon(c)
    assert 145.0 < s < 156.0
    # wiggles: T(with baryons) / T(smooth CDM-like envelope) oscillates with period 2 pi / s in k [1/Mpc]
    kk = np.linspace(0.03, 0.3, 4000)                                   # h/Mpc
    ratio = po.eisenstein_hu(c, kk) / po.eisenstein_hu(obg.Planck18(Omega_b=1e-9, Omega_c=c.Omega_m - 1e-9), kk)
    osc = ratio - np.convolve(ratio, np.ones(801) / 801, mode="same")
    zc = kk[1:][(osc[1:] * osc[:-1] < 0)][3:-3]                          # zero crossings away from the window edges
    period = 2 * np.mean(np.diff(zc)) * c.h                              # 1/Mpc
    assert abs(period / (2 * np.pi / s) - 1) < 0.08
    ks, pows = po.lin_power_table(c)
    lk = np.linspace(np.log(1e-4), np.log(10.0), 200001)               # independent quadrature: fine trapezoid in ln k
    x = 8 * np.exp(lk)
    f = np.exp(lk) ** 3 * np.interp(np.exp(lk), ks, pows) * (3 * (np.sin(x) - x * np.cos(x)) / x ** 3) ** 2
    sig = np.sqrt(np.sum(0.5 * (f[1:] + f[:-1]) * np.diff(lk)) / (2 * np.pi ** 2))
    assert abs(sig - 1.0) < 2e-3
    assert 0.012 < ks[np.argmax(pows)] < 0.022                           # turnover at k_eq


@pytest.mark.parametrize("order", [1, 2, 3, 4])
def test_kaiser_bessel_known_answers(order):
    """nbody.py:280-312, :357-363: the kernel is (nearly) a partition of unity -- painted mass is conserved to the accuracy
    Barnett et al. quote for this cutoff --, its analytic derivative matches central differences, its Fourier transform
    `kaiser_bessel_hat` is the transform of the sampled kernel up to aliasing, and paint / read stay adjoint."""
    rng = np.random.default_rng(70 + order)
    kc = o.optim_kcut(1.)
    assert np.isclose(kc, 0.98 * np.pi)
    s = rng.uniform(-order / 2, order / 2, 200) * 0.999
    h = 1e-6
    assert np.allclose(o.kaiser_bessel_grad(s, order, kc), (o.kaiser_bessel(s + h, order, kc) - o.kaiser_bessel(s - h, order, kc)) / (2 * h),
                       rtol=1e-5, atol=1e-7)
    # continuous transform int K(s) e^{-i k s} ds by quadrature against kaiser_bessel_hat
    x = np.linspace(-order / 2, order / 2, 40001)
    for k in (0.0, 0.7, 2.0, kc * 0.9):
        num = np.trapezoid(o.kaiser_bessel(x, order, kc) * np.cos(k * x), x)
        assert np.isclose(num, o.kaiser_bessel_hat((np.array([k]),), order, kc)[0], rtol=2e-4, atol=2e-6), (order, k)
    shape = (8, 6, 10)
    pos = rng.uniform(-5, 15, (400, 3))
    w, m = rng.standard_normal(400), rng.standard_normal(shape)
    lhs = np.sum(o.paint(pos, shape, w, order, "kaiser_bessel") * m)
    assert np.isclose(lhs, np.sum(w * o.read(pos, m, order, "kaiser_bessel")), rtol=1e-12)
    if order >= 2:
        assert abs(o.paint(pos, shape, 1., order, "kaiser_bessel").sum() / 400 - 1) < (0.25 if order == 2 else 0.05)
    # VJPs against central differences
    mb = rng.standard_normal(shape)
    pb, wb = o.paint_vjp(pos, shape, w, mb, order, "kaiser_bessel")
    d = rng.standard_normal(pos.shape)
    L = lambda p_: np.sum(o.paint(p_, shape, w, order, "kaiser_bessel") * mb)
    keep = np.all(np.abs((pos % 1.0) - 0.5) > 1e-3, axis=1) & np.all(np.abs(pos % 1.0) > 1e-3, axis=1) & np.all(np.abs(pos % 1.0) < 1 - 1e-3, axis=1)
    dd = d * keep[:, None]
    assert np.isclose((L(pos + h * dd) - L(pos - h * dd)) / (2 * h), np.sum(pb * dd), rtol=1e-5)
    assert np.allclose(wb, o.read(pos, mb, order, "kaiser_bessel"))
