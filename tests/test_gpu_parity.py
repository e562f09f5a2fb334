"""GPU parity tests: the HIP path (through the C ABI, via montecosmo_amd.nbody) against the float64 oracle
on the same seeded inputs.  Tolerances: integer work (cell indices) bit-exact; floating point fields
relative L2 <= 1e-5 (north-star tolerance for the final density), tighter where fp32 round-off allows."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import pm_oracle as o, background as obg  # noqa: E402  (checker only)


def rel_l2(a, b):
    a, b = np.asarray(a), np.asarray(b)
    dt = np.complex128 if (np.iscomplexobj(a) or np.iscomplexobj(b)) else np.float64
    a, b = a.astype(dt), b.astype(dt)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def to_np(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def nb(gpu):
    from montecosmo_amd import nbody
    return nbody


def random_pos(n, N, seed, spread=3.0):
    rng = np.random.default_rng(seed)
    pos = rng.uniform(-spread * n, spread * n, (N, 3)).astype(np.float32)
    # exact integers, half-integers and tiny negatives exercise floor / round-half-even / wrap
    pos[:64] = np.round(pos[:64])
    pos[64:128] = np.round(pos[64:128]) + 0.5
    pos[128:160] = -1e-7
    return pos


@pytest.mark.parametrize("order", [1, 2])
@pytest.mark.parametrize("shape", [(16, 16, 16), (12, 20, 6), (64, 64, 64)])
def test_cell_index_bit_exact_absolute(nb, order, shape):
    pos = random_pos(max(shape), 50000, 1)
    got = to_np(nb.cell_index(pos, shape, order))
    want = o.cell_index(pos.astype(np.float64), shape, order)
    assert got.dtype == np.int16
    assert np.array_equal(got, want)


@pytest.mark.parametrize("order", [1, 2])
@pytest.mark.parametrize("mesh,ptcl", [((32, 32, 32), None), ((16, 16, 16), (8, 8, 8)), ((24, 16, 12), (16, 16, 16))])
def test_cell_index_bit_exact_lattice(nb, order, mesh, ptcl):
    ptcl_ = mesh if ptcl is None else ptcl
    N = int(np.prod(ptcl_))
    rng = np.random.default_rng(2)
    disp = (rng.standard_normal((N, 3)) * 3).astype(np.float32)
    disp[:32] = 0.0
    disp[32:64] = 0.5
    disp[64:96] = -1e-9
    lp = nb.LatticePos(disp, mesh, ptcl)
    got = to_np(nb.cell_index(lp, mesh, order))
    # the oracle sees the same positions: lattice point (exact rational, power-of-two ratios here) + float32 disp
    pos64 = o.regular_pos(mesh, ptcl_) + disp.astype(np.float64)
    if ptcl is not None and any(m % p for m, p in zip(mesh, ptcl_)):
        # non-integer lattice spacing: the kernel adds fraction and displacement in float32
        q = o.regular_pos(mesh, ptcl_)
        qi = np.floor(q)
        pos64 = qi + ((q - qi).astype(np.float32) + disp).astype(np.float64)
    want = o.cell_index(pos64, mesh, order)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("order", [1, 2])
def test_paint_read_absolute(nb, order):
    shape = (16, 24, 8)
    pos = random_pos(24, 20000, 3, spread=1.5)
    rng = np.random.default_rng(4)
    w = rng.standard_normal(len(pos)).astype(np.float32)
    got = to_np(nb.paint(pos, shape, w, order))
    want = o.paint(pos.astype(np.float64), shape, w.astype(np.float64), order)
    assert rel_l2(got, want) < 2e-6
    got1 = to_np(nb.paint(pos, shape, 1., order))
    assert abs(got1.sum() / len(pos) - 1) < 1e-6          # bricks.py:1101-1102 invariant
    mesh = rng.standard_normal(shape).astype(np.float32)
    r = to_np(nb.read(pos, mesh, order))
    assert rel_l2(r, o.read(pos.astype(np.float64), mesh.astype(np.float64), order)) < 2e-6


@pytest.mark.parametrize("order", [1, 2, 3, 4])
def test_paint_read_edge_cases(nb, order):
    """Empty input, particles exactly on mesh points and on cell faces (round-half-to-even for the odd orders,
    nbody.py:369-371), many periods outside the box on both sides, and all particles in one cell."""
    shape = (8, 12, 16)
    rng = np.random.default_rng(11)
    empty = np.zeros((0, 3), np.float32)
    assert not to_np(nb.paint(empty, shape, 1., order)).any()
    assert to_np(nb.read(empty, rng.standard_normal(shape).astype(np.float32), order)).shape == (0,)
    grid = np.stack(np.meshgrid(*[np.arange(-2, 4) * 0.5] * 3, indexing="ij"), -1).reshape(-1, 3)      # 0, +-0.5, +-1, 1.5
    far = grid + np.array([8 * 37, -12 * 41, 16 * 29])                                                  # whole periods away
    pile = np.tile(np.array([[3.25, 7.75, 9.5]]), (500, 1))
    pos = np.concatenate([grid, far, pile]).astype(np.float32)
    assert np.array_equal(to_np(nb.cell_index(pos, shape, order)), o.cell_index(pos.astype(np.float64), shape, order))
    w = rng.standard_normal(len(pos)).astype(np.float32)
    got = to_np(nb.paint(pos, shape, w, order))
    want = o.paint(pos.astype(np.float64), shape, w.astype(np.float64), order)
    assert np.abs(got - want).max() < 2e-5 * np.abs(want).max()
    mesh = rng.standard_normal(shape).astype(np.float32)
    r = to_np(nb.read(pos, mesh, order))
    assert np.allclose(r, o.read(pos.astype(np.float64), mesh.astype(np.float64), order), atol=2e-6 * np.abs(mesh).max() * 8)


def expected_bucketed(disp, n, H, centre=True):
    """paint_tiled.hip restated in numpy: window offsets o_T = clip(rint(mean displacement of four z rows -- (x, y) = (4, 4), (4, 12),
    (12, 4), (12, 12) -- of lattice block T), +-8), and the number of (particle, tile) pairs whose lattice point lies outside the window
    T - o_T - (H+1) ... T - o_T + 15 + H of a tile T the particle's CIC stencil touches."""
    from itertools import product
    nt = n // 16
    d = disp.reshape(nt, 16, nt, 16, nt, 16, 3)[:, 4::8, :, 4::8, :, :, :].astype(np.float64)
    off = np.clip(np.rint(d.mean(axis=(1, 3, 5))), -8, 8).astype(int) if centre else np.zeros((nt, nt, nt, 3), int)
    fl = np.floor(disp).astype(int)
    q = np.indices((n, n, n)).reshape(3, -1).T
    c = q + fl
    count = 0
    for alt in product((0, 1), repeat=3):
        alt = np.array(alt)
        valid = np.all((alt == 0) | (c % 16 == 15), axis=1)
        T = (c // 16 + alt) % nt
        rl = np.where(alt == 0, c % 16 - fl, -1 - fl)
        k = rl + off[T[:, 0], T[:, 1], T[:, 2]] + H + 1
        count += int((valid & ~np.all((k >= 0) & (k < 16 + 2 * H + 1), axis=1)).sum())
    return count


@pytest.mark.parametrize("n,halo,sigma", [(32, 4, 1.0), (64, 2, 1.5), (64, 4, 2.5), (16, 4, 1.0), (48, 2, 1.0), (80, 1, 0.8)])
def test_paint_tiled_lattice(nb, n, halo, sigma):
    """LDS-tiled paint (paint_tiled.hip; meshes below 48 per axis take the generic path) against the oracle, weighted form
    included: white-noise displacements of the order of the window halo, so that many particles travel through the
    per-tile buckets, and a few far ones wrapping several tiles."""
    shape = (n, n, n)
    N = n ** 3
    rng = np.random.default_rng(5)
    disp = (rng.standard_normal((N, 3)) * sigma).astype(np.float32)
    disp[:100] *= 8.0   # far particles, wrapping several tiles
    lp = nb.LatticePos(disp, shape)
    plan = nb.get_plan(shape)
    plan.call("mcpm_plan_set_halo", halo)
    try:
        pos64 = o.regular_pos(shape) + disp.astype(np.float64)
        got = to_np(nb.paint(lp, shape))
        want = o.paint(pos64, shape)
        assert rel_l2(got, want) < 2e-6
        assert abs(got.sum() / N - 1) < 1e-6
        if n >= 48:
            assert plan.last_outliers() == 0                  # nothing needed f32 global atomics
            want_b = expected_bucketed(disp, n, halo)
            assert plan.last_bucketed() == want_b > 0         # (particle, tile) pairs outside the tile's window
            assert np.array_equal(got, to_np(nb.paint(lp, shape)))          # integer sums: bit for bit
        w = rng.standard_normal(N).astype(np.float32)
        gotw = to_np(nb.paint(lp, shape, w))
        assert rel_l2(gotw, o.paint(pos64, shape, w.astype(np.float64))) < 2e-6
        assert np.array_equal(gotw, to_np(nb.paint(lp, shape, w)))
    finally:
        plan.call("mcpm_plan_set_halo", 0)      # back to the default


def expected_boxes(disp, shape):
    """The device's window rule (paint_tiled.hip: tile_prologue_kernel + box_tile_kernel) restated in numpy: per 16^3 block the range of
    floor(d) over 64 samples (four z rows), per tile the hull of the 27 blocks around it, extents above 8 cut back to +-4 around the
    block's rounded mean offset, corners kept within +-12.  Returns (lo, hi), each (ntx, nty, ntz, 3)."""
    nt = [n // 16 for n in shape]
    d = disp.reshape(nt[0], 16, nt[1], 16, nt[2], 16, 3)[:, 4::8, :, 4::8, :, :, :].astype(np.float64)
    off = np.clip(np.rint(d.mean(axis=(1, 3, 5))), -8, 8).astype(int)
    f = np.clip(np.floor(d), -100, 100).astype(int)
    lo, hi = f.min(axis=(1, 3, 5)), f.max(axis=(1, 3, 5))
    LO, HI = lo.copy(), hi.copy()
    for a in (-1, 0, 1):
        for b in (-1, 0, 1):
            for e in (-1, 0, 1):
                LO = np.minimum(LO, np.roll(lo, (a, b, e), axis=(0, 1, 2)))
                HI = np.maximum(HI, np.roll(hi, (a, b, e), axis=(0, 1, 2)))
    wide = HI - LO > 8
    LO = np.where(wide, np.maximum(LO, off - 4), LO)
    HI = np.where(wide, np.minimum(HI, off + 4), HI)
    LO = np.clip(LO, -12, 12)
    HI = np.maximum(np.clip(HI, -12, 12), LO)
    return LO, HI


def expected_bucketed_boxes(disp, shape, LO, HI):
    """(particle, tile) pairs whose lattice point lies outside the window of a tile the particle's CIC stencil touches."""
    from itertools import product
    n = np.array(shape)
    nt = n // 16
    fl = np.floor(disp).astype(int)
    q = np.indices(shape).reshape(3, -1).T
    c = q + fl
    count = 0
    for alt in product((0, 1), repeat=3):
        alt = np.array(alt)
        valid = np.all((alt == 0) | (c % 16 == 15), axis=1)
        T = (c // 16 + alt) % nt
        lo, hi = LO[T[:, 0], T[:, 1], T[:, 2]], HI[T[:, 0], T[:, 1], T[:, 2]]
        rl = np.where(alt == 0, c % 16 - fl, -1 - fl)          # the lattice point relative to T; T's window: -1 - hi <= rl <= 15 - lo
        count += int((valid & ~np.all((rl >= -1 - hi) & (rl <= 15 - lo), axis=1)).sum())
    return count


def test_paint_windows_are_boxes_per_tile_and_axis(nb):
    """On meshes of 2048 tiles or more every tile's window is a box per axis, sized on the device from the sampled floor(d) ranges of
    the 27 Lagrangian blocks around it (DESIGN finding 46).  An anisotropic field -- a bulk flow, rough along x, smooth along y, a
    long wave along z, and a slab of tiles whose x range exceeds the largest window -- against the oracle, the device's rule against
    its numpy restatement (window points and bucketed pairs, exactly), bit for bit from call to call, weighted and three-component
    forms included."""
    import ctypes as C
    import torch
    shape = (128, 256, 256)
    N = int(np.prod(shape))
    rng = np.random.default_rng(11)
    q = o.regular_pos(shape)
    disp = np.empty((N, 3), np.float32)
    disp[:, 0] = 3.3 + 0.9 * rng.standard_normal(N)
    disp[:, 1] = -6.2 + 0.05 * rng.standard_normal(N)
    disp[:, 2] = 0.4 + 2.5 * np.sin(2 * np.pi * q[:, 0] / shape[0]) + 0.3 * rng.standard_normal(N)
    band = (q[:, 1] >= 96) & (q[:, 1] < 128)
    disp[band, 0] += 1.2 * rng.standard_normal(int(band.sum()))      # x extents above 8 there: the window is cut back, buckets fill
    lp = nb.LatticePos(disp, shape)
    plan = nb.get_plan(shape)
    pos64 = q + disp.astype(np.float64)
    got = to_np(nb.paint(lp, shape))
    assert rel_l2(got, o.paint(pos64, shape)) < 2e-6
    assert abs(got.sum() / N - 1) < 1e-6
    st = (C.c_int64 * 13)()
    plan.call("mcpm_plan_last_paint_stats", st)
    LO, HI = expected_boxes(disp, shape)
    assert st[7] == int(np.prod(17 + HI - LO, axis=-1).sum())                       # the windows the device chose
    assert st[7] < 0.8 * (N // 4096) * 25 ** 3                                       # ... are far smaller than the static halo's
    ext = (HI - LO).reshape(-1, 3)
    assert ext[:, 1].max() <= 1 and ext[:, 0].min() >= 4                             # ... and anisotropic: tight along y, wide along x
    assert plan.last_outliers() == 0
    assert plan.last_bucketed() == expected_bucketed_boxes(disp, shape, LO, HI) > 100
    assert np.array_equal(got, to_np(nb.paint(lp, shape)))
    w = rng.standard_normal(N).astype(np.float32)
    gotw = to_np(nb.paint(lp, shape, w))
    assert rel_l2(gotw, o.paint(pos64, shape, w.astype(np.float64))) < 2e-6
    assert np.array_equal(gotw, to_np(nb.paint(lp, shape, w)))
    w3 = rng.standard_normal((N, 3)).astype(np.float32)
    out = torch.empty((3,) + shape, dtype=torch.float32, device="cuda")
    wt = torch.from_numpy(w3).cuda()
    args = (C.c_void_p(lp.disp.data_ptr()), N, 1, C.c_void_p(wt.data_ptr()), 2, C.c_void_p(out.data_ptr()), 0)
    plan.call("mcpm_paint3_f32", *args)
    first = out.clone()
    for c in range(3):
        assert rel_l2(to_np(out[c]), o.paint(pos64, shape, w3[:, c].astype(np.float64))) < 2e-6
    plan.call("mcpm_paint3_f32", *args)
    assert torch.equal(out, first)


@pytest.mark.parametrize("n", [64, 96])
def test_paint_windows_follow_the_bulk_displacement(nb, n):
    """paint_tiled.hip: a tile's window is centred on the mean displacement of the particles around it.  A coherent flow of
    several cells (uniform shift + a long wave) plus a small dispersion needs (almost) no bucket even at halo 2; with
    windows centred on the tile itself (mcpm_plan_set_centre(0)) every window misses its particles, the buckets overflow
    and the repair pass deposits them with global atomics: slower, counted, and still the same mesh (no input can lose
    mass)."""
    shape = (n, n, n)
    N = n ** 3
    rng = np.random.default_rng(9)
    q = o.regular_pos(shape)
    bulk = np.array([5.3, -7.1, 2.2]) + 1.5 * np.sin(2 * np.pi * q[:, [1, 2, 0]] / n)
    disp = (bulk + 0.4 * rng.standard_normal((N, 3))).astype(np.float32)
    lp = nb.LatticePos(disp, shape)
    plan = nb.get_plan(shape)
    want = o.paint(q + disp.astype(np.float64), shape)
    plan.call("mcpm_plan_set_centre", 1)
    plan.call("mcpm_plan_set_halo", 2)
    try:
        got = to_np(nb.paint(lp, shape))
        assert rel_l2(got, want) < 2e-6
        b_on = plan.last_bucketed()
        assert plan.last_outliers() == 0 and b_on == expected_bucketed(disp, n, 2) and b_on < 2e-3 * N
        # three-component form (adjoint of the force read) on the same field
        import ctypes as C
        import torch
        w3 = rng.standard_normal((N, 3)).astype(np.float32)
        out = torch.empty((3,) + shape, dtype=torch.float32, device="cuda")
        wt = torch.from_numpy(w3).cuda()
        plan.call("mcpm_paint3_f32", C.c_void_p(lp.disp.data_ptr()), N, 1, C.c_void_p(wt.data_ptr()), 2, C.c_void_p(out.data_ptr()), 0)
        for c in range(3):
            assert rel_l2(to_np(out[c]), o.paint(q + disp.astype(np.float64), shape, w3[:, c].astype(np.float64))) < 2e-6
        plan.call("mcpm_plan_set_centre", 0)
        got0 = to_np(nb.paint(lp, shape))
        assert expected_bucketed(disp, n, 2, centre=False) > 0.5 * N
        assert plan.last_outliers() > 0.25 * N            # appends that found their bucket full
        assert rel_l2(got0, want) < 2e-6 and abs(got0.sum() / N - 1) < 1e-6
    finally:
        plan.call("mcpm_plan_set_centre", 1)
        plan.call("mcpm_plan_set_halo", 0)      # back to the default


def test_paint_non_finite_displacement_is_visible(nb):
    """A NaN / absurd displacement names no cell: the particle is left to the global-atomic kernel, which makes the first
    cell non-finite instead of indexing with garbage; every other particle is painted as usual."""
    shape = (64, 64, 64)
    N = 64 ** 3
    disp = np.zeros((N, 3), np.float32)
    disp[777, 1] = np.nan
    disp[4242, 0] = 1e9
    plan = nb.get_plan(shape)
    got = to_np(nb.paint(nb.LatticePos(disp, shape), shape))
    assert plan.last_outliers() == 2
    assert np.isnan(got[0, 0, 0]) and np.isfinite(got.reshape(-1)[1:]).all()
    assert abs(got.reshape(-1)[1:].sum() - (N - 3)) < 1e-3


def test_paint_regular_grid_is_constant(nb):
    shape = (32, 32, 32)
    got = to_np(nb.paint(nb.LatticePos.regular(shape), shape))
    assert np.array_equal(got, np.ones(shape, dtype=np.float32))
    got = to_np(nb.paint(nb.LatticePos.regular(shape, (16, 16, 16)), shape, order=1))
    assert got.sum() == 16 ** 3


def test_fft_matches_numpy(nb):
    rng = np.random.default_rng(6)
    x = rng.standard_normal((16, 24, 32)).astype(np.float32)
    X = to_np(nb.rfftn(x))
    assert rel_l2(X, np.fft.rfftn(x.astype(np.float64))) < 1e-6
    assert rel_l2(to_np(nb.irfftn(X)), x) < 1e-6


@pytest.mark.parametrize("fd", [np.inf, 2, 4])
def test_pm_forces_spectrum_and_hermitian_projection(nb, fd):
    shape = (16, 16, 16)
    rng = np.random.default_rng(7)
    spec = np.fft.rfftn(rng.standard_normal(shape))
    pos = random_pos(16, 5000, 8, spread=1.0)
    got = to_np(nb.pm_forces(pos, spec.astype(np.complex64), 2, grad_fd=fd, lap_fd=fd))
    want = o.pm_forces(pos.astype(np.float64), spec, 2, grad_fd=fd, lap_fd=fd)
    assert rel_l2(got, want) < 1e-5
    got1 = to_np(nb.pm_forces(pos, spec.astype(np.complex64), 1))
    assert rel_l2(got1, o.pm_forces(pos.astype(np.float64), spec, 1)) < 1e-5


def test_pm_forces_painted_and_pm_forces2(nb):
    shape = (32, 32, 32)
    N = 32 ** 3
    rng = np.random.default_rng(9)
    disp = (rng.standard_normal((N, 3)) * 1.0).astype(np.float32)
    lp = nb.LatticePos(disp, shape)
    pos64 = o.regular_pos(shape) + disp.astype(np.float64)
    got = to_np(nb.pm_forces(lp, shape, 2))
    assert rel_l2(got, o.pm_forces(pos64, shape, 2)) < 1e-5
    got = to_np(nb.pm_forces(lp, shape, 2, paint_deconv=True, kcut=2.0))
    assert rel_l2(got, o.pm_forces(pos64, shape, 2, paint_deconv=True, kcut=2.0)) < 1e-5
    spec = np.fft.rfftn(rng.standard_normal(shape)) * 0.05
    got2 = to_np(nb.pm_forces2(lp, spec.astype(np.complex64), 2))
    assert rel_l2(got2, o.pm_forces2(pos64, spec.astype(np.complex64).astype(np.complex128), 2)) < 1e-5


def _ics(n, rms=1.0, seed=0):
    from montecosmo_amd import synth
    return synth.init_mesh(n, seed=seed, rms_disp=rms)


@pytest.mark.parametrize("lpt_order", [1, 2])
def test_lpt(nb, lpt_order):
    from montecosmo_amd import bricks
    n = 32
    shape = (n, n, n)
    spec = _ics(n)
    pos = bricks.regular_pos(shape)
    cos_o, cos_g = obg.Planck18(), bricks.Planck18()
    dpos, vel = nb.lpt(cos_g, spec, pos, a=0.5, lpt_order=lpt_order, read_order=1)
    dpos_o, vel_o = o.lpt(cos_o, spec.astype(np.complex128), pos, 0.5, lpt_order=lpt_order, read_order=1)
    assert rel_l2(to_np(dpos), dpos_o) < 1e-5
    assert rel_l2(to_np(vel), vel_o) < 1e-5


def test_growth_and_coefficients(nb):
    from montecosmo_amd import bricks
    cos_o, cos_g = obg.Planck18(), bricks.Planck18()
    a = np.linspace(0.0, 1.0, 23)
    for name in ["a2g", "a2g2", "a2f", "a2f2", "a2dg2dg", "a2chi"]:
        assert np.allclose(getattr(nb, name)(cos_g, a), getattr(o, name)(cos_o, a), rtol=1e-10, atol=1e-13), name
    g = np.linspace(0.002, 1.0, 17)
    for name in ["g2a", "g2g2", "g2f", "g2f2", "g2dg2dg"]:
        assert np.allclose(getattr(nb, name)(cos_g, g), getattr(o, name)(cos_o, g), rtol=1e-10, atol=1e-13), name
    for g0 in (0.01, 0.3, 0.8):
        assert np.isclose(nb.alpha_bf(cos_g, g0, 0.1), o.alpha_bf(cos_o, g0, 0.1), rtol=1e-9)
        assert np.isclose(nb.alpha_fpm(cos_g, g0, 0.1), o.alpha_fpm(cos_o, g0, 0.1), rtol=1e-9)


@pytest.mark.parametrize("n,n_steps,integrator", [(32, 5, "bullfrog"), (64, 10, "bullfrog"), (32, 4, "fastpm")])
def test_nbody_bf_final_density(nb, n, n_steps, integrator):
    """North-star gate: final density within 1e-5 relative L2 of the float64 oracle; cell indices of the final
    particles identical except where a particle sits within fp32 round-off of a cell face."""
    from montecosmo_amd import bricks
    shape = (n, n, n)
    spec = _ics(n, rms=1.5)
    pos = bricks.regular_pos(shape)
    cos_o, cos_g = obg.Planck18(), bricks.Planck18()
    alpha_fn = o.alpha_bf if integrator == "bullfrog" else o.alpha_fpm
    (p_o, v_o) = o.nbody_bf(cos_o, spec.astype(np.complex128), pos, 0., 1., n_steps, alpha_fn=alpha_fn)
    lp, vel = nb.nbody_bf(cos_g, spec, pos, a0=0., a1=1., n_steps=n_steps, integrator=integrator, lattice_out=True)
    p_g = to_np(lp.to_absolute())
    assert rel_l2(p_g - pos, p_o[0] - pos) < 1e-5
    assert rel_l2(to_np(vel), v_o[0]) < 1e-5
    dens_g = to_np(nb.paint(lp, shape))
    dens_o = o.paint(p_o[0], shape)
    assert rel_l2(dens_g, dens_o) < 1e-5
    idx_g = to_np(nb.cell_index(lp, shape))
    idx_o = o.cell_index(p_o[0], shape)
    mismatch = np.any(idx_g != idx_o, axis=1).mean()
    assert mismatch < 1e-4
    # default return structure: absolute positions with a leading snapshot axis (nbody.py:988-989, :1000)
    p1, v1 = nb.nbody_bf(cos_g, spec, pos, a0=0., a1=1., n_steps=n_steps, integrator=integrator)
    assert tuple(p1.shape) == (1, n ** 3, 3) and tuple(v1.shape) == (1, n ** 3, 3)


def test_bullfrog_vf_matches_oracle(nb):
    from montecosmo_amd import bricks
    n = 16
    shape = (n, n, n)
    rng = np.random.default_rng(11)
    disp = rng.standard_normal((n ** 3, 3)).astype(np.float32)
    vel = rng.standard_normal((n ** 3, 3)).astype(np.float32)
    cos_o, cos_g = obg.Planck18(), bricks.Planck18()
    vf_g = nb.bullfrog_vf(cos_g, 0.1, shape)
    vf_o = o.bullfrog_vf(cos_o, 0.1, shape)
    dx, dv = vf_g(0.3, (nb.LatticePos(disp, shape), vel))
    pos64 = o.regular_pos(shape) + disp.astype(np.float64)
    dx_o, dv_o = vf_o(0.3, (pos64, vel.astype(np.float64)))
    assert rel_l2(to_np(dx), dx_o) < 1e-5
    assert rel_l2(to_np(dv), dv_o) < 1e-5


def test_paint_read_vjp(nb):
    shape = (16, 16, 16)
    rng = np.random.default_rng(12)
    N = 4000
    pos = rng.uniform(-20, 40, (N, 3)).astype(np.float32)
    w = rng.standard_normal(N).astype(np.float32)
    mb = rng.standard_normal(shape).astype(np.float32)
    pb, wb = nb.paint_vjp(pos, shape, w, mb)
    pb_o, wb_o = o.paint_vjp(pos.astype(np.float64), shape, w.astype(np.float64), mb.astype(np.float64))
    assert rel_l2(to_np(pb), pb_o) < 1e-5 and rel_l2(to_np(wb), wb_o) < 1e-5
    ob = rng.standard_normal(N).astype(np.float32)
    pb, meshb = nb.read_vjp(pos, mb, ob)
    pb_o, meshb_o = o.read_vjp(pos.astype(np.float64), mb.astype(np.float64), ob.astype(np.float64))
    assert rel_l2(to_np(pb), pb_o) < 1e-5 and rel_l2(to_np(meshb), meshb_o) < 1e-5


@pytest.mark.parametrize("n,n_steps,lpt_order,integrator", [(16, 3, 2, "bullfrog"), (32, 5, 2, "bullfrog"), (32, 3, 1, "fastpm"),
                                                            (64, 2, 2, "bullfrog")])  # 64: hand-written FFT incl. the lpt adjoint
def test_nbody_bf_vjp(nb, n, n_steps, lpt_order, integrator):
    """Hand-written reverse sweep against the oracle's (finite-difference-validated) VJP."""
    from montecosmo_amd import bricks
    shape = (n, n, n)
    spec = _ics(n, rms=1.0)
    pos = bricks.regular_pos(shape)
    cos_o, cos_g = obg.Planck18(), bricks.Planck18()
    rng = np.random.default_rng(1)
    xb = rng.standard_normal((n ** 3, 3)).astype(np.float32)
    vb = rng.standard_normal((n ** 3, 3)).astype(np.float32)
    alpha_fn = o.alpha_bf if integrator == "bullfrog" else o.alpha_fpm
    mb_o, sb_o = o.nbody_bf_vjp(cos_o, spec.astype(np.complex128), pos, xb.astype(np.float64), vb.astype(np.float64),
                                0.1, 1., n_steps, lpt_order=lpt_order, alpha_fn=alpha_fn)
    (_, _), ctx = nb.nbody_bf(cos_g, spec, pos, a0=0.1, a1=1., n_steps=n_steps, lpt_order=lpt_order,
                              integrator=integrator, return_ctx=True)
    mb_g, sb_g = nb.nbody_bf_vjp(ctx, xb, vb)
    mb_g = to_np(mb_g)
    assert rel_l2(mb_g, mb_o) < 1e-4
    # directional derivative along a Hermitian-consistent direction
    d = np.fft.rfftn(rng.standard_normal(shape))
    assert np.isclose(np.sum((np.conj(mb_g) * d).real), np.sum((np.conj(mb_o) * d).real), rtol=1e-4)
    assert np.allclose(sb_g["alpha"], sb_o["alpha"], rtol=2e-4, atol=1e-3 * np.abs(sb_o["alpha"]).max())
    assert np.allclose(sb_g["beta"], sb_o["beta"], rtol=2e-4, atol=1e-3 * np.abs(sb_o["beta"]).max())
    for k in ("g", "g2", "dg2dg"):
        assert np.isclose(sb_g[k], sb_o[k], rtol=1e-3, atol=1e-3 * abs(sb_o["g"])), k


@pytest.mark.parametrize("shape,n_steps,a0", [((16, 16, 16), 1, 0.0), ((24, 24, 24), 2, 0.0), ((16, 32, 48), 3, 0.05),
                                                ((4, 4, 4), 2, 0.1), ((8, 6, 4), 2, 0.1)])
def test_nbody_bf_edge_configurations(nb, shape, n_steps, a0):
    """One single (half-drift) step, the model's default start a0 = 0 (the growth table is clamped there, nbody.py:984-985
    note in SURVEY 8a-12), a mesh that is neither a power of two nor a multiple of the tile size (rocFFT + atomic paint
    fallbacks), a non-cubic mesh and meshes smaller than any tile or stencil halo: forward state and reverse sweep against
    the oracle."""
    from montecosmo_amd import bricks, synth
    spec = synth.init_mesh(shape, seed=8, rms_disp=1.0)
    pos = bricks.regular_pos(shape)
    cos_o, cos_g = obg.Planck18(), bricks.Planck18()
    N = len(pos)
    p_o, v_o = o.nbody_bf(cos_o, spec.astype(np.complex128), pos, a0, 1., n_steps)
    (p_g, v_g), ctx = nb.nbody_bf(cos_g, spec, pos, a0=a0, a1=1., n_steps=n_steps, return_ctx=True)
    assert rel_l2(to_np(p_g).reshape(-1, 3) - pos, p_o.reshape(-1, 3) - pos) < 2e-5
    assert rel_l2(to_np(v_g).reshape(-1, 3), v_o.reshape(-1, 3)) < 2e-5
    rng = np.random.default_rng(3)
    xb, vb = rng.standard_normal((N, 3)).astype(np.float32), rng.standard_normal((N, 3)).astype(np.float32)
    mb_o, _ = o.nbody_bf_vjp(cos_o, spec.astype(np.complex128), pos, xb.astype(np.float64), vb.astype(np.float64), a0, 1., n_steps)
    mb_g, _ = nb.nbody_bf_vjp(ctx, xb, vb)
    assert rel_l2(to_np(mb_g), mb_o) < 2e-4


def test_errors_are_loud(nb):
    from montecosmo_amd._lib import McpmError
    with pytest.raises(McpmError):
        nb.Plan((16, 16, 15))          # odd nz
    with pytest.raises(ValueError):
        nb.paint(np.zeros((4, 3), np.float32), (8, 8, 8), order=5)
    with pytest.raises(ValueError):
        nb.invlaplace_hat(nb.rfftk((8, 8, 8)), fd_order=3)


@pytest.mark.parametrize("n,sigma", [(32, 1.0), (64, 2.0), (48, 1.0), (40, 1.0)])
def test_paint3_matches_three_paints(nb, n, sigma):
    """Three-component weighted paint (adjoint of a three-component read) against the oracle, incl. outliers and
    the non-tiled fallback (n = 40)."""
    import ctypes as C
    import torch
    shape = (n, n, n)
    N = n ** 3
    rng = np.random.default_rng(21)
    disp = (rng.standard_normal((N, 3)) * sigma).astype(np.float32)
    disp[:50] *= 6.0
    w3 = rng.standard_normal((N, 3)).astype(np.float32)
    lp = nb.LatticePos(disp, shape)
    plan = nb.get_plan(shape)
    out = torch.empty((3,) + shape, dtype=torch.float32, device="cuda")
    wt = torch.from_numpy(w3).cuda()
    plan.call("mcpm_paint3_f32", C.c_void_p(lp.disp.data_ptr()), N, 1, C.c_void_p(wt.data_ptr()), 2, C.c_void_p(out.data_ptr()), 0)
    pos64 = o.regular_pos(shape) + disp.astype(np.float64)
    for c in range(3):
        assert rel_l2(to_np(out[c]), o.paint(pos64, shape, w3[:, c].astype(np.float64))) < 2e-6


def test_paint3_fixed_point_tiles(nb):
    """The fixed-point accumulator of the three-component paint (particles.hip, paint3_fx_kernel) against the float64
    restatement and against the f64-tile kernel: heavy-tailed weights (the scale follows max|w|), bitwise reproducibility
    (integer sums do not depend on the arrival order), accumulation, the overflow proof (a cell holding more than 32
    maximal weights flags its tile, which the f64 kernel repaints), all-zero weights and NaN propagation."""
    import ctypes as C
    import torch
    from montecosmo_amd._lib import lib
    n = 64
    shape, N = (n, n, n), n ** 3
    rng = np.random.default_rng(33)
    plan = nb.get_plan(shape)
    disp = np.clip(rng.standard_normal((N, 3)) * 1.5, -3.9, 3.9).astype(np.float32)   # within the tile halo: no particle
    pos64 = o.regular_pos(shape) + disp.astype(np.float64)                             # takes the float-atomic outlier path
    lp = nb.LatticePos(disp, shape)
    out = torch.empty((3,) + shape, dtype=torch.float32, device="cuda")

    def paint3(w3, fixed, accumulate=0, d=lp):
        assert lib.mcpm_plan_set_paint3_fixed(plan.h, fixed) == 0
        wt = torch.from_numpy(np.ascontiguousarray(w3, dtype=np.float32)).cuda()
        plan.call("mcpm_paint3_f32", C.c_void_p(d.disp.data_ptr()), N, 1, C.c_void_p(wt.data_ptr()), 2, C.c_void_p(out.data_ptr()),
                  accumulate)
        redo = C.c_int64()
        assert lib.mcpm_plan_last_redo(plan.h, C.byref(redo)) == 0
        return to_np(out).copy(), redo.value

    try:
        for tail in (0.0, 1.5):      # max|w| / rms|w| ~ 5 and ~ 1e3
            w3 = (rng.standard_normal((N, 3)) * np.exp(tail * rng.standard_normal((N, 1)))).astype(np.float32)
            fx, redo = paint3(w3, 1)
            assert redo == 0
            f64, _ = paint3(w3, 0)
            ref = np.stack([o.paint(pos64, shape, w3[:, c].astype(np.float64)) for c in range(3)])
            assert rel_l2(f64, ref) < 2e-6
            assert rel_l2(fx, ref) < (2e-6 if tail == 0 else 2e-5), (tail, rel_l2(fx, ref))
            assert np.array_equal(fx, paint3(w3, 1)[0])                       # reproducible bit for bit
            acc, _ = paint3(w3, 1, accumulate=1)                               # out held fx: accumulate doubles it
            assert rel_l2(acc, 2 * ref) < (2e-6 if tail == 0 else 2e-5)
        # overflow proof: the 64 particles of every 4^3 block of lattice sites land on one mesh point with weight 1.5:
        # 96 > 64 (the capacity of a cell in units of 2^floor(log2 max|w|)) / 2 flags every tile; half of it does not
        g = np.indices(shape).reshape(3, -1).T
        dc = (-(g % 4)).astype(np.float32)
        lc = nb.LatticePos(dc, shape)
        refc = o.paint(o.regular_pos(shape) + dc.astype(np.float64), shape, 1.0)
        fx, redo = paint3(np.full((N, 3), 1.5, np.float32), 1, d=lc)
        assert redo == (n // 16) ** 3
        assert rel_l2(fx[0], 1.5 * refc) < 2e-6 and rel_l2(fx[2], 1.5 * refc) < 2e-6
        wh = np.full((N, 3), 1.5, np.float32)
        wh[g.sum(1) % 2 == 1] = 0.0                                            # 32 x 1.5 = 48 < 64: stays fixed point
        fx, redo = paint3(wh, 1, d=lc)
        assert redo == 0
        assert rel_l2(fx[1], o.paint(o.regular_pos(shape) + dc.astype(np.float64), shape, wh[:, 1].astype(np.float64))) < 2e-6
        # zero weights, then a NaN weight: every tile goes to the f64 kernel and the NaN reaches its 8 cells
        z, redo = paint3(np.zeros((N, 3), np.float32), 1)
        assert redo == 0 and not z.any()
        wn = rng.standard_normal((N, 3)).astype(np.float32)
        wn[12345, 1] = np.nan
        fx, redo = paint3(wn, 1)
        assert redo == (n // 16) ** 3
        assert np.isnan(fx[1]).sum() == 8 and not np.isnan(fx[0]).any() and not np.isnan(fx[2]).any()
    finally:
        lib.mcpm_plan_set_paint3_fixed(plan.h, 1)


@pytest.mark.parametrize("mesh,ptcl", [((32, 32, 32), (16, 16, 16)), ((16, 32, 16), (32, 32, 32))])
def test_nbody_bf_particle_lattice_differs_from_mesh(nb, mesh, ptcl):
    """ptcl_shape != mesh_shape (model.py:738 with ptcl_oversamp != evol_oversamp): generic lattice path, forward
    and gradient, against the oracle."""
    from montecosmo_amd import bricks, synth
    spec = synth.init_mesh(mesh, seed=4, rms_disp=1.0)
    pos = bricks.regular_pos(mesh, ptcl)
    cos_o, cos_g = obg.Planck18(), bricks.Planck18()
    n_steps = 3
    (p_o, v_o) = o.nbody_bf(cos_o, spec.astype(np.complex128), pos, 0.1, 1., n_steps)
    rng = np.random.default_rng(2)
    N = len(pos)
    xb, vb = rng.standard_normal((N, 3)).astype(np.float32), rng.standard_normal((N, 3)).astype(np.float32)
    mb_o, _ = o.nbody_bf_vjp(cos_o, spec.astype(np.complex128), pos, xb.astype(np.float64), vb.astype(np.float64), 0.1, 1., n_steps)
    # This lattice takes the generic paint, whose deposits are fixed-point integer atomics (particles.hip,
    # paint_atomic_kernel): order-independent, so the trajectory and its gradient are the same bits on every run.
    def run():
        (lp, vel), ctx = nb.nbody_bf(cos_g, spec, pos, a0=0.1, a1=1., n_steps=n_steps, lattice_out=True, return_ctx=True)
        mb_g, _ = nb.nbody_bf_vjp(ctx, xb, vb)
        return lp, to_np(lp.disp), to_np(vel), to_np(mb_g)
    lp, d1, v1, g1 = run()
    assert lp.ptcl_shape == tuple(ptcl)
    assert rel_l2(to_np(lp.to_absolute()) - pos, p_o[0] - pos) < 1e-5
    assert rel_l2(v1, v_o[0]) < 1e-5
    assert rel_l2(g1, mb_o) < 1e-4
    _, d2, v2, g2 = run()
    assert np.array_equal(d1, d2) and np.array_equal(v1, v2) and np.array_equal(g1, g2)   # bit for bit


@pytest.mark.parametrize("order", [1, 2, 3, 4])
def test_generic_paint_is_order_independent(nb, order):
    """The generic paint (absolute positions, any order, any mesh) sums fixed-point integers: two launches agree bit
    for bit, heavy collisions included, and the result matches the float64 oracle (nbody.py:365-396)."""
    shape = (12, 10, 8)
    rng = np.random.default_rng(order)
    N = 20000
    pos = (rng.random((N, 3)) * 0.7 + 3.0).astype(np.float32)        # ~40 particles per cell of a small clump
    for w in (None, (rng.standard_normal(N) * 10.0 ** rng.integers(-3, 4, N)).astype(np.float32)):
        ref = o.paint(pos.astype(np.float64), shape, 1.0 if w is None else w.astype(np.float64), order)
        a = to_np(nb.paint(pos, shape, 1.0 if w is None else w, order=order))
        b = to_np(nb.paint(pos, shape, 1.0 if w is None else w, order=order))
        assert np.array_equal(a, b)
        assert rel_l2(a, ref) < 2e-6
    # scalar weight, zero weights, a NaN weight (float-atomic fallback: the NaN reaches exactly its order^3 cells)
    assert rel_l2(to_np(nb.paint(pos, shape, 2.5, order=order)), o.paint(pos.astype(np.float64), shape, 2.5, order)) < 2e-6
    assert not to_np(nb.paint(pos, shape, np.zeros(N, np.float32), order=order)).any()
    wn = np.ones(N, np.float32)
    wn[7] = np.nan
    assert np.isnan(to_np(nb.paint(pos, shape, wn, order=order))).sum() == order ** 3
    # tiny weights keep their relative precision (the scale follows max|w|)
    wt = (rng.random(N) * 1e-30).astype(np.float32)
    assert rel_l2(to_np(nb.paint(pos, shape, wt, order=order)), o.paint(pos.astype(np.float64), shape, wt.astype(np.float64), order)) < 2e-6


def test_lpt_vjp_standalone(nb):
    from montecosmo_amd import bricks, synth
    n = 16
    shape = (n, n, n)
    spec = synth.init_mesh(n, seed=6, rms_disp=1.0)
    pos = bricks.regular_pos(shape)
    rng = np.random.default_rng(3)
    xb, vb = rng.standard_normal((n ** 3, 3)), rng.standard_normal((n ** 3, 3))
    mb_o, _, sb_o = o.lpt_vjp(obg.Planck18(), spec.astype(np.complex128), pos, 0.3, xb, vb, lpt_order=2, read_order=1)
    mb_g, sb_g = nb.lpt_vjp(bricks.Planck18(), spec, pos, 0.3, xb.astype(np.float32), vb.astype(np.float32), lpt_order=2)
    assert rel_l2(to_np(mb_g), mb_o) < 1e-5
    for k in ("g", "g2", "dg2dg"):
        assert np.isclose(sb_g[k], sb_o[k], rtol=1e-4, atol=1e-4 * abs(sb_o["g"])), k


@pytest.mark.parametrize("lpt_order", [1, 2])
def test_lpt_light_cone(nb, lpt_order):
    """nbody.py:634-667 with `a` of shape (N,1): per-particle growth (mcpm_lpt_combine_f32) and its VJP."""
    from montecosmo_amd import bricks, synth
    n = 16
    shape = (n, n, n)
    spec = synth.init_mesh(n, seed=8, rms_disp=1.0)
    pos = bricks.regular_pos(shape)
    rng = np.random.default_rng(4)
    a = 0.3 + 0.6 * rng.uniform(size=(n ** 3, 1))
    dp, v = nb.lpt(bricks.Planck18(), spec, nb.LatticePos.regular(shape), a, lpt_order=lpt_order, read_order=1)
    dp_o, v_o = o.lpt(obg.Planck18(), spec.astype(np.complex128), pos, a, lpt_order=lpt_order, read_order=1)
    assert rel_l2(to_np(dp), dp_o) < 1e-5 and rel_l2(to_np(v), v_o) < 1e-5
    xb, vb = rng.standard_normal((n ** 3, 3)), rng.standard_normal((n ** 3, 3))
    mb_o, _, sb_o = o.lpt_vjp(obg.Planck18(), spec.astype(np.complex128), pos, a, xb, vb, lpt_order=lpt_order, read_order=1)
    mb_g, sb_g = nb.lpt_vjp(bricks.Planck18(), spec, pos, a, xb.astype(np.float32), vb.astype(np.float32), lpt_order=lpt_order)
    assert rel_l2(to_np(mb_g), mb_o) < 1e-5
    assert rel_l2(to_np(sb_g["g"]), sb_o["g"]) < 1e-5
    if lpt_order == 2:
        assert rel_l2(to_np(sb_g["g2"]), sb_o["g2"]) < 1e-5 and rel_l2(to_np(sb_g["dg2dg"]), sb_o["dg2dg"]) < 1e-5


@pytest.mark.parametrize("n,lpt_order,light_cone", [(16, 2, False), (64, 2, True), (64, 1, True), (64, 2, False)])
def test_adjoints_that_keep_the_forward_meshes_equal_the_recomputing_ones(nb, n, lpt_order, light_cone):
    """The forward passes can leave their force / Hessian meshes for the adjoint instead of having it recompute them (DESIGN finding 48):
    `lpt(..., return_ctx=True)` -> `lpt_vjp(..., ctx=...)` (mcpm_lpt_save_f32 / mcpm_lpt_vjp_saved_f32), the checkpoint of nbody_bf, the
    bias context (mcpm_bias_fields_save_f32 / mcpm_bias_fields_vjp_saved_f32).  Same meshes, same arithmetic: every output is bitwise
    equal to the recomputing path's."""
    import os
    import torch
    from montecosmo_amd import bricks, synth
    shape = (n, n, n)
    spec = synth.init_mesh(n, seed=12, rms_disp=1.0)
    lat = nb.LatticePos.regular(shape)
    rng = np.random.default_rng(6)
    a = (0.3 + 0.6 * rng.uniform(size=(n ** 3, 1))) if light_cone else 0.4
    cosmo = bricks.Planck18()
    xb, vb = rng.standard_normal((n ** 3, 3)).astype(np.float32), rng.standard_normal((n ** 3, 3)).astype(np.float32)
    (dp, v), ctx = nb.lpt(cosmo, spec, lat, a, lpt_order=lpt_order, read_order=1, return_ctx=True)
    dp0, v0 = nb.lpt(cosmo, spec, lat, a, lpt_order=lpt_order, read_order=1)
    assert torch.equal(dp, dp0) and torch.equal(v, v0) and ctx.save is not None
    mb1, sb1 = nb.lpt_vjp(cosmo, spec, lat, a, xb, vb, lpt_order=lpt_order, ctx=ctx)
    mb0, sb0 = nb.lpt_vjp(cosmo, spec, lat, a, xb, vb, lpt_order=lpt_order)
    assert torch.equal(mb1, mb0)
    for k in ("g", "g2", "dg2dg"):
        assert np.array_equal(to_np(sb1[k]) if torch.is_tensor(sb1[k]) else sb1[k], to_np(sb0[k]) if torch.is_tensor(sb0[k]) else sb0[k]), k
    mb2, _ = nb.lpt_vjp(cosmo, spec, lat, a, xb, vb, lpt_order=lpt_order, ctx=ctx)       # the context is read, not consumed
    assert torch.equal(mb2, mb1)
    # the trajectory's checkpoint against the stand-alone adjoint chain: nbody_bf_vjp twice on one context (read only, again)
    if not light_cone:
        (lp, vel), nctx = nb.nbody_bf(cosmo, spec, lat, a0=0.1, a1=0.5, n_steps=2, lpt_order=lpt_order, return_ctx=True, lattice_out=True)
        g1 = nb.nbody_bf_vjp(nctx, xb, vb)
        g2 = nb.nbody_bf_vjp(nctx, xb, vb)
        assert torch.equal(g1[0], g2[0]) and all(np.array_equal(np.asarray(g1[1][k]), np.asarray(g2[1][k])) for k in g1[1])
    # bias fields: kept Hessian meshes against recomputed ones
    bias = dict(b1=1.3, b2=0.4, bs2=-0.3, bn2=0.2)
    wb_, dvb_ = rng.standard_normal(n ** 3).astype(np.float32), rng.standard_normal((n ** 3, 3)).astype(np.float32)
    outs = []
    try:
        for keep in ("1", "0"):
            os.environ["MCPM_BIAS_KEEP"] = keep
            (w, dvel, _), bctx = bricks.lagrangian_bias(cosmo, lat, 0.5, (100.,) * 3, spec, bias, read_order=1, return_ctx=True)
            assert (bctx.hess6 is not None) == (keep == "1")
            mbar, bbar, gbar = bricks.lagrangian_bias_vjp(bctx, wb_, dvb_)
            outs.append((w, dvel, mbar, bbar, np.asarray(gbar)))
    finally:
        os.environ.pop("MCPM_BIAS_KEEP", None)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    assert outs[0][3] == outs[1][3] and np.array_equal(outs[0][4], outs[1][4])


@pytest.mark.parametrize("opts", [dict(paint_deconv=True), dict(grad_fd=4, lap_fd=2), dict(paint_deconv=True, grad_fd=2, lap_fd=4)])
def test_nbody_bf_deconv_and_fd_kernels(nb, opts):
    """nbody.py:967-1002 with the options the model leaves at their defaults: paint_deconv (:590-593) and the
    finite-difference Laplace / gradient kernels (:125-163): forward state, reverse sweep, snapshots and a save function
    against the oracle."""
    from montecosmo_amd import bricks, synth
    n, n_steps = 16, 4
    shape = (n, n, n)
    spec = synth.init_mesh(n, seed=2, rms_disp=1.0)
    pos = bricks.regular_pos(shape)
    (p_g, v_g), ctx = nb.nbody_bf(bricks.Planck18(), spec, pos, a0=0.1, a1=0.9, n_steps=n_steps, return_ctx=True, **opts)
    p_o, v_o = o.nbody_bf(obg.Planck18(), spec.astype(np.complex128), pos, a0=0.1, a1=0.9, n_steps=n_steps, **opts)
    assert rel_l2(to_np(p_g)[0] - pos, p_o[0] - pos) < 1e-5 and rel_l2(to_np(v_g)[0], v_o[0]) < 1e-5
    # reverse sweep (the reference differentiates through both options with jax.grad, model.py:362-363)
    rng = np.random.default_rng(8)
    xb, vb = rng.standard_normal((n ** 3, 3)), rng.standard_normal((n ** 3, 3))
    mb_g, sb_g = nb.nbody_bf_vjp(ctx, xb.astype(np.float32), vb.astype(np.float32))
    mb_o, sb_o = o.nbody_bf_vjp(obg.Planck18(), spec.astype(np.complex128), pos, xb, vb, 0.1, 0.9, n_steps, **opts)
    assert rel_l2(to_np(mb_g), mb_o) < 1e-4
    for k in ("alpha", "beta"):
        assert np.allclose(sb_g[k], sb_o[k], rtol=1e-3, atol=1e-3 * np.abs(sb_o[k]).max()), k
    for k in ("g", "g2", "dg2dg", "dg"):
        assert np.isclose(sb_g[k], sb_o[k], rtol=1e-3, atol=1e-3 * abs(sb_o["g"])), k
    # snapshots and a save function on this branch (nbody.py:964-969: fn maps every saved state)
    p3, v3 = nb.nbody_bf(bricks.Planck18(), spec, pos, a0=0.1, a1=0.9, n_steps=n_steps, snapshots=3, **opts)
    p3o, v3o = o.nbody_bf(obg.Planck18(), spec.astype(np.complex128), pos, a0=0.1, a1=0.9, n_steps=n_steps, snapshots=3, **opts)
    assert rel_l2(to_np(p3) - pos, p3o - pos) < 1e-5 and rel_l2(to_np(v3), v3o) < 1e-5
    ke = nb.nbody_bf(bricks.Planck18(), spec, pos, a0=0.1, a1=0.9, n_steps=n_steps, snapshots=3, fn=lambda t, y, args: (y[1] ** 2).sum(), **opts)
    assert np.allclose(to_np(ke), (v3o ** 2).sum(axis=(1, 2)), rtol=1e-4)


@pytest.mark.parametrize("snapshots", [3, [0.3, 0.55, 1.0]])
def test_nbody_bf_snapshots(nb, snapshots):
    """SaveAt(ts=...) (nbody.py:990-997): linear interpolation of the Euler solution between steps."""
    from montecosmo_amd import bricks
    n = 16
    shape = (n, n, n)
    spec = _ics(n, rms=1.0)
    pos = bricks.regular_pos(shape)
    p_o, v_o = o.nbody_bf(obg.Planck18(), spec.astype(np.complex128), pos, 0.1, 1., 4, snapshots=snapshots)
    p_g, v_g = nb.nbody_bf(bricks.Planck18(), spec, pos, a0=0.1, a1=1., n_steps=4, snapshots=snapshots)
    assert tuple(p_g.shape) == p_o.shape == (3, n ** 3, 3) and tuple(v_g.shape) == v_o.shape
    assert rel_l2(to_np(p_g) - pos, p_o - pos) < 1e-5
    assert rel_l2(to_np(v_g), v_o) < 1e-5


@pytest.mark.parametrize("order", [3, 4])
def test_tsc_pcs_paint_read_and_vjps(nb, order):
    """Higher-order assignment (TSC, PCS; nbody.py:243-244): paint, read, cell indices and both VJPs, absolute and
    lattice positions."""
    shape = (16, 12, 20)
    rng = np.random.default_rng(40 + order)
    N = 6000
    pos = random_pos(20, N, 41, spread=1.5)
    w = rng.standard_normal(N).astype(np.float32)
    p64, w64 = pos.astype(np.float64), w.astype(np.float64)
    assert np.array_equal(to_np(nb.cell_index(pos, shape, order)), o.cell_index(p64, shape, order))
    assert rel_l2(to_np(nb.paint(pos, shape, w, order)), o.paint(p64, shape, w64, order)) < 2e-6
    assert abs(float(nb.paint(pos, shape, 1., order).double().sum()) / N - 1) < 1e-6
    mesh = rng.standard_normal(shape).astype(np.float32)
    assert rel_l2(to_np(nb.read(pos, mesh, order)), o.read(p64, mesh.astype(np.float64), order)) < 2e-6
    mb = rng.standard_normal(shape).astype(np.float32)
    pb, wb = nb.paint_vjp(pos, shape, w, mb, order)
    pb_o, wb_o = o.paint_vjp(p64, shape, w64, mb.astype(np.float64), order)
    assert rel_l2(to_np(pb), pb_o) < 1e-5 and rel_l2(to_np(wb), wb_o) < 1e-5
    ob = rng.standard_normal(N).astype(np.float32)
    pb, meshb = nb.read_vjp(pos, mb, ob, order)
    pb_o, meshb_o = o.read_vjp(p64, mb.astype(np.float64), ob.astype(np.float64), order)
    assert rel_l2(to_np(pb), pb_o) < 1e-5 and rel_l2(to_np(meshb), meshb_o) < 1e-5
    # lattice displacements
    n = 16
    disp = (rng.standard_normal((n ** 3, 3)) * 1.5).astype(np.float32)
    lp = nb.LatticePos(disp, (n, n, n))
    pos64 = o.regular_pos((n, n, n)) + disp.astype(np.float64)
    assert rel_l2(to_np(nb.paint(lp, (n, n, n), order=order)), o.paint(pos64, (n, n, n), order=order)) < 2e-6
    assert rel_l2(to_np(nb.pm_forces(lp, (n, n, n), order)), o.pm_forces(pos64, (n, n, n), order)) < 1e-5


@pytest.mark.parametrize("order", [3, 4])
def test_nbody_bf_tsc_pcs(nb, order):
    from montecosmo_amd import bricks
    n, n_steps = 16, 3
    shape = (n, n, n)
    spec = _ics(n, rms=1.0)
    pos = bricks.regular_pos(shape)
    cos_o, cos_g = obg.Planck18(), bricks.Planck18()
    p_o, v_o = o.nbody_bf(cos_o, spec.astype(np.complex128), pos, 0.1, 1., n_steps, paint_order=order)
    (lp, vel), ctx = nb.nbody_bf(cos_g, spec, pos, a0=0.1, a1=1., n_steps=n_steps, paint_order=order, lattice_out=True,
                                 return_ctx=True)
    assert rel_l2(to_np(lp.disp), p_o[0] - pos) < 1e-5 and rel_l2(to_np(vel), v_o[0]) < 1e-5
    rng = np.random.default_rng(7)
    xb, vb = rng.standard_normal((n ** 3, 3)).astype(np.float32), rng.standard_normal((n ** 3, 3)).astype(np.float32)
    mb_o, _ = o.nbody_bf_vjp(cos_o, spec.astype(np.complex128), pos, xb.astype(np.float64), vb.astype(np.float64), 0.1, 1.,
                             n_steps, paint_order=order)
    mb_g, _ = nb.nbody_bf_vjp(ctx, xb, vb)
    assert rel_l2(to_np(mb_g), mb_o) < 1e-4


@pytest.mark.parametrize("n", [16, 64])
def test_pm_forces_vjp(nb, n):
    """pm_forces_vjp, painted and spectrum cases (n = 64 runs the hand-written FFT), against the oracle's VJP."""
    shape = (n, n, n)
    rng = np.random.default_rng(50)
    N = 20000
    pos = rng.uniform(0, n, (N, 3)).astype(np.float32)
    R = rng.standard_normal((N, 3)).astype(np.float32)
    p64, R64 = pos.astype(np.float64), R.astype(np.float64)
    pb, none = nb.pm_forces_vjp(pos, shape, R)
    pb_o, _ = o.pm_forces_vjp(p64, shape, R64)
    assert none is None and rel_l2(to_np(pb), pb_o) < 2e-5
    spec = np.fft.rfftn(rng.standard_normal(shape)).astype(np.complex64)
    pb, mb = nb.pm_forces_vjp(pos, spec, R)
    pb_o, mb_o = o.pm_forces_vjp(p64, spec.astype(np.complex128), R64)
    assert rel_l2(to_np(pb), pb_o) < 2e-5 and rel_l2(to_np(mb), mb_o) < 2e-5


def test_cosmology_gradient_through_growth_tables(nb):
    """dL/dOmega_c: GPU scalar cotangents (alpha_i, beta_i, dg, lpt growth scalars) chained through the host growth
    tables (nb.cosmo_vjp) against a finite difference of the float64 oracle's forward model."""
    from montecosmo_amd import bricks
    n, n_steps, a0 = 16, 3, 0.1
    shape = (n, n, n)
    spec = _ics(n, rms=1.0)
    pos = bricks.regular_pos(shape)
    rng = np.random.default_rng(9)
    Rx, Rv = rng.standard_normal((n ** 3, 3)), rng.standard_normal((n ** 3, 3))
    (_, _), ctx = nb.nbody_bf(bricks.Planck18(), spec, pos, a0=a0, a1=1., n_steps=n_steps, return_ctx=True)
    _, bars = nb.nbody_bf_vjp(ctx, Rx.astype(np.float32), Rv.astype(np.float32))
    got = nb.cosmo_vjp(ctx, bars, params=("Omega_c",))["Omega_c"]

    def L(oc):
        p, v = o.nbody_bf(obg.Planck18(Omega_c=oc), spec.astype(np.complex128), pos, a0, 1., n_steps)
        return np.sum(p[0] * Rx) + np.sum(v[0] * Rv)

    h = 1e-4
    want = (L(0.2607 + h) - L(0.2607 - h)) / (2 * h)
    assert np.isclose(got, want, rtol=2e-3, atol=1e-3 * abs(want)), (got, want)
    _, sb_o = o.nbody_bf_vjp(obg.Planck18(), spec.astype(np.complex128), pos, Rx, Rv, a0, 1., n_steps)
    assert np.isclose(bars["dg"], sb_o["dg"], rtol=1e-3, atol=1e-3 * abs(sb_o["dg"]))


@pytest.mark.parametrize("order", [1, 2, 3, 4])
def test_kaiser_bessel_paint_read_and_vjps(nb, order):
    """kernel_type='kaiser_bessel' (nbody.py:280-312, :357-363, :381-382, :411-412): paint, read, both VJPs, the
    deconvolution and the nufft (forward and VJP) against the oracle; absolute and lattice positions."""
    shape = (16, 12, 20)
    rng = np.random.default_rng(80 + order)
    N = 5000
    pos = random_pos(20, N, 81, spread=1.5)
    w = rng.standard_normal(N).astype(np.float32)
    p64, w64 = pos.astype(np.float64), w.astype(np.float64)
    kb = dict(kernel_type="kaiser_bessel", oversamp=1.5)
    okb = ("kaiser_bessel", 1.5)
    got = to_np(nb.paint(pos, shape, w, order, **kb))
    assert rel_l2(got, o.paint(p64, shape, w64, order, *okb)) < 3e-6
    assert np.array_equal(got, to_np(nb.paint(pos, shape, w, order, **kb)))            # fixed-point sums: bit for bit
    assert rel_l2(to_np(nb.paint(pos, shape, 2.0, order, **kb)), o.paint(p64, shape, 2.0, order, *okb)) < 3e-6
    mesh = rng.standard_normal(shape).astype(np.float32)
    assert rel_l2(to_np(nb.read(pos, mesh, order, **kb)), o.read(p64, mesh.astype(np.float64), order, *okb)) < 3e-6
    pb, wb = nb.paint_vjp(pos, shape, w, mesh, order, **kb)
    pb_o, wb_o = o.paint_vjp(p64, shape, w64, mesh.astype(np.float64), order, *okb)
    assert rel_l2(to_np(pb), pb_o) < 2e-5 and rel_l2(to_np(wb), wb_o) < 3e-6
    ob = rng.standard_normal(N).astype(np.float32)
    pb, mb = nb.read_vjp(pos, mesh, ob, order, **kb)
    pb_o, mb_o = o.read_vjp(p64, mesh.astype(np.float64), ob.astype(np.float64), order, *okb)
    assert rel_l2(to_np(pb), pb_o) < 2e-5 and rel_l2(to_np(mb), mb_o) < 3e-6
    assert rel_l2(to_np(nb.deconv_paint(mesh, order, **kb)), o.deconv_paint(mesh.astype(np.float64), order, *okb)) < 1e-5
    # lattice displacements
    lshape = (16, 16, 16)
    disp = (rng.standard_normal((16 ** 3, 3)) * 1.2).astype(np.float32)
    lp = nb.LatticePos(disp, lshape)
    pl64 = o.regular_pos(lshape) + disp.astype(np.float64)
    assert rel_l2(to_np(nb.paint(lp, lshape, 1., order, **kb)), o.paint(pl64, lshape, 1., order, *okb)) < 3e-6
    # nufft with an oversampled paint mesh, and its VJP
    fshape = (8, 8, 8)
    posf = (p64 * 0.4) % 8
    spec = nb.nufft(posf.astype(np.float32), fshape, 1.5, w, order, 2, kernel_type="kaiser_bessel")
    spec_o = o.nufft(posf.astype(np.float32).astype(np.float64), fshape, 1.5, w64, order, 2, kernel_type="kaiser_bessel")
    assert rel_l2(to_np(spec), spec_o) < 2e-5
    sb = (rng.standard_normal(spec_o.shape) + 1j * rng.standard_normal(spec_o.shape)).astype(np.complex64)
    pb, wb = nb.nufft_vjp(posf.astype(np.float32), fshape, w, sb, order, 2, paint_shape=1.5, kernel_type="kaiser_bessel")
    pb_o, wb_o = o.nufft_vjp(posf.astype(np.float32).astype(np.float64), fshape, w64, sb.astype(np.complex128), order, 2, paint_shape=1.5,
                             kernel_type="kaiser_bessel")
    assert rel_l2(to_np(pb), pb_o) < 5e-5 and rel_l2(to_np(wb), wb_o) < 2e-5


def test_adjoint_step_reading_its_cotangents_elsewhere_equals_the_in_place_form(gpu):
    """`mcpm_bullfrog_step_vjp_from_f32` (incoming cotangents read from one pair of arrays, outgoing written to another) against
    a copy followed by the in-place `mcpm_bullfrog_step_vjp_f32`: same kernels, same arithmetic -> bitwise equal cotangents and
    scalar cotangents over a whole reverse sweep (bench.py uses the former for the first reverse step of every trajectory)."""
    import ctypes as C
    import sys, os
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    K = 4
    r = bench.Runner(64, K, torch.device("cuda", 0))
    r.forward(K)
    r.sbar.zero_()
    r.backward(K)                                  # first step out of place, the rest in place
    torch.cuda.synchronize()
    xb1, vb1, sb1 = r.xb.clone(), r.vb.clone(), r.sbar.clone()
    assert bool(torch.isfinite(xb1).all()) and float(xb1.abs().max()) > 0 and not torch.equal(xb1, r.pos_bar)
    r.sbar.zero_()
    r.xb.copy_(r.pos_bar)
    r.vb.copy_(r.vel_bar)
    for i in reversed(range(K)):
        tau = r.dg / 2 if i == K - 1 else r.dg
        if i > 0:
            r.plan.call("mcpm_plan_hint_next_adjoint", float(r.betas[i - 1]), float(r.dg))
        r.plan.call("mcpm_bullfrog_step_vjp_f32", r.p(r.states[i, 0]), r.p(r.states[i, 1]), r.p(r.fmesh[i]), float(r.alphas[i]),
                    float(r.betas[i]), float(tau), 2, r.p(r.xb), r.p(r.vb), C.c_void_p(r.sbar.data_ptr() + 8 * i),
                    C.c_void_p(r.sbar.data_ptr() + 8 * (K + i)), 0.5 if i == K - 1 else 1.0, C.c_void_p(r.sbar.data_ptr() + 8 * 2 * K))
    torch.cuda.synchronize()
    assert torch.equal(r.xb, xb1) and torch.equal(r.vb, vb1) and torch.equal(r.sbar, sb1)


def test_streaming_store_data_hazard_is_padded(gpu):
    """The hand-written 12-byte streaming store (`global_store_dwordx3 ... nt` by inline asm in store3_nt and in the interleaved
    z C2R pass) reads its data registers after issue; on gfx940-class parts a VALU write of them needs two wait states behind
    the store (LLVM GCNHazardRecognizer, VMEM store data wider than 64 bits), which the compiler cannot insert inside inline
    asm, so the asm carries `s_nop 2` itself (csrc/mcpm_internal.h MCPM_STORE_DATA_HAZARD_NOP).  Round 3's corruption was caught
    by a golden fixture by luck of register allocation; this test forces the situation: the instructions right behind the store
    overwrite all three data registers, over 2^26 records, against a plain-store twin."""
    import ctypes as C
    import torch
    from montecosmo_amd._lib import lib, check
    n = 1 << 26
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ref = torch.empty(3 * n, dtype=torch.float32, device=gpu)
    check(lib.mcpm_selftest_store3_nt(st, C.c_void_p(ref.data_ptr()), n, 0), None, "mcpm_selftest_store3_nt")
    head = (np.arange(3 * 4096, dtype=np.int64) & 0xFFFFFF).astype(np.float32)       # the twin itself against the definition
    assert np.array_equal(ref[:3 * 4096].cpu().numpy(), head)
    tail = ((3 * (n - 4096) + np.arange(3 * 4096, dtype=np.int64)) & 0xFFFFFF).astype(np.float32)
    assert np.array_equal(ref[-3 * 4096:].cpu().numpy(), tail)
    out = torch.empty_like(ref)
    for mode in (1, 3):                 # pinned registers + the padding macro; the library's own helper
        for rep in range(3):
            out.fill_(-7.0)
            check(lib.mcpm_selftest_store3_nt(st, C.c_void_p(out.data_ptr()), n, mode), None, "mcpm_selftest_store3_nt")
            bad = int((out != ref).sum())
            assert bad == 0, f"mode {mode}: {bad} of {3 * n} floats differ from the plain-store twin"
    out.fill_(-7.0)                     # without the wait states: informational (printed with -s), not asserted
    check(lib.mcpm_selftest_store3_nt(st, C.c_void_p(out.data_ptr()), n, 2), None, "mcpm_selftest_store3_nt")
    print(f"store3 hazard, no wait states: {int((out != ref).sum())} of {3 * n} floats corrupted")


def test_particle_pitch_changes_the_layout_not_the_results(gpu, nb):
    """mcpm_plan_set_particle_pitch / mcpm_plan_probe_particle_pitch (the layout fix for C-ABI callers, not
    just for bench.py): the composite forward + reverse sweep gives bitwise the same states and gradients whatever the pitch
    between its checkpoint arrays; the probe keeps the plain layout for a mesh whose arrays live in the caches and zeroes the
    buffer it is given; NbodyCtx.state(i) follows the pitch."""
    import ctypes as C
    import torch
    from montecosmo_amd import bricks, synth
    n, K = 32, 3
    shape = (n, n, n)
    spec = synth.init_mesh(n, seed=3, rms_disp=1.5)
    pos = bricks.regular_pos(shape)
    plan = nb.get_plan(shape)
    rng = np.random.default_rng(2)
    xb, vb = rng.standard_normal((n ** 3, 3)).astype(np.float32), rng.standard_normal((n ** 3, 3)).astype(np.float32)
    res = {}
    try:
        for pitch in (0, 3 * n ** 3 + 1088, 3 * n ** 3 + 17472):
            plan._pitch_probed = True              # this test sets the pitch itself
            plan.call("mcpm_plan_set_particle_pitch", pitch)
            (lp, vel), ctx = nb.nbody_bf(bricks.Planck18(), spec, pos, a0=0.1, a1=1., n_steps=K, return_ctx=True, lattice_out=True)
            assert ctx.pitch == (pitch or 3 * n ** 3)
            mb, sb = nb.nbody_bf_vjp(ctx, xb, vb)
            x1, v1 = ctx.state(1)
            res[pitch] = (lp.disp.clone(), vel.clone(), mb.clone(), sb, x1.clone(), v1.clone())
        ref = res[0]
        for pitch, r in res.items():
            for a, b in zip(r[:3] + r[4:], ref[:3] + ref[4:]):
                assert torch.equal(a, b), pitch
            assert all(np.array_equal(r[3][k], ref[3][k]) for k in ref[3]), pitch
        # out-of-range pitches are refused
        from montecosmo_amd._lib import lib
        assert lib.mcpm_plan_set_particle_pitch(plan.h, 3 * n ** 3 - 4) == -6 and lib.mcpm_plan_set_particle_pitch(plan.h, 3 * n ** 3 + 17476) == -6
        # the probe: small mesh -> plain layout, buffer zeroed or untouched but usable
        nck = lib.mcpm_nbody_ckpt_floats(plan.h, K, 2)
        buf = torch.full((nck,), 7.0, dtype=torch.float32, device=gpu)
        got = C.c_int64()
        plan.call("mcpm_plan_probe_particle_pitch", C.c_void_p(buf.data_ptr()), nck, C.byref(got))
        assert got.value == 3 * n ** 3
    finally:
        plan.call("mcpm_plan_set_particle_pitch", 0)
