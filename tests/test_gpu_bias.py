"""Lagrangian bias expansion (the step before the PM path inside evolve, bricks.py:327-443): HIP path
against the float64 oracle, forward and VJP."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import pm_oracle as o, bias_oracle as bo  # noqa: E402  (checker only)


def rel_l2(a, b):
    a, b = np.asarray(a), np.asarray(b)
    dt = np.complex128 if (np.iscomplexobj(a) or np.iscomplexobj(b)) else np.float64
    return float(np.linalg.norm(a.astype(dt) - b.astype(dt)) / np.linalg.norm(b.astype(dt)))


BIAS = dict(b1=1.1, b2=0.3, bs2=-0.2, b3=0.15, bds2=0.25, bs3=-0.1, bn2=2.0, bnpar=1.5)


class FixedGrowth:
    """cosmology stand-in: lagrangian_bias only needs a2g(cosmo, a); Planck18 tables are used on both sides."""


@pytest.mark.parametrize("shape,box,read_order,per_particle", [
    ((16, 16, 16), (160., 160., 160.), 1, False),
    ((16, 16, 16), (160., 160., 160.), 1, True),
    ((16, 12, 8), (200., 120., 100.), 2, True),       # different cell lengths per axis, CIC reads off the lattice
    ((32, 32, 32), (640., 640., 640.), 2, False),
])
def test_lagrangian_bias_forward_and_vjp(gpu, shape, box, read_order, per_particle):
    from montecosmo_amd import bricks, nbody
    rng = np.random.default_rng(11)
    cosmo = bricks.Planck18()
    X = np.fft.rfftn(0.4 * rng.standard_normal(shape))
    pos = bricks.regular_pos(shape)
    if read_order == 2:
        pos = pos + rng.uniform(0, 1, pos.shape)
    N = len(pos)
    a = (0.3 + 0.6 * rng.uniform(size=(N, 1))) if per_particle else 0.6
    g = o.a2g(cosmo, a)
    # read_order 1 at the mesh's own lattice: the identity-read path (LatticePos.regular), as model.py:738-744 calls it
    pos_in = nbody.LatticePos.regular(shape) if read_order == 1 else pos.astype(np.float32)
    (w, dvel, phi), ctx = bricks.lagrangian_bias(cosmo, pos_in, a, box, X.astype(np.complex64), BIAS,
                                                  read_order=read_order, return_ctx=True)
    assert bool(ctx.gcs) == (read_order == 1)
    p64 = pos.astype(np.float32).astype(np.float64)
    w_o, dv_o = bo.lagrangian_bias(g, p64, box, X, BIAS, read_order)
    assert phi == 0.
    assert rel_l2(w.cpu().numpy(), w_o) < 2e-5 and rel_l2(dvel.cpu().numpy(), dv_o) < 2e-5
    wb = rng.standard_normal(N)
    vb = rng.standard_normal((N, 3))
    mb, bb, gb = bricks.lagrangian_bias_vjp(ctx, wb.astype(np.float32), vb.astype(np.float32))
    mb_o, bb_o, gb_o = bo.lagrangian_bias_vjp(g, p64, box, X, BIAS, wb, vb, read_order)
    assert rel_l2(mb.cpu().numpy(), mb_o) < 1e-4
    scale = max(abs(v) for v in bb_o.values())
    for k in bo.BIAS_KEYS:
        assert abs(bb[k] - bb_o[k]) < 2e-4 * scale, (k, bb[k], bb_o[k])
    gb = gb.cpu().numpy() if hasattr(gb, "cpu") else gb
    assert rel_l2(np.asarray(gb, dtype=np.float64).reshape(-1), np.asarray(gb_o).reshape(-1)) < 2e-4


@pytest.mark.parametrize("curved,lightcone,lattice", [(True, True, True), (True, False, False), (False, True, False),
                                                      (False, False, True)])
def test_observe_pos_forward_and_vjp(gpu, curved, lightcone, lattice):
    """model.py:780-797 (los/scale factor, cell2phys, rsd with the bias velocity term, phys2cell on another mesh) fused
    in mcpm_observe_pos_f32, against the float64 oracle chain; the VJP against central differences of that chain."""
    from montecosmo_amd import bricks, nbody
    rng = np.random.default_rng(21)
    cosmo = bricks.Planck18()
    evol, paint = (16, 16, 16), (24, 20, 16)
    box, center, rotvec = (640., 640., 640.), (100., -50., 1500.), (0.2, -0.1, 0.3)
    R = bo.rotvec_matrix(rotvec)
    N = 16 ** 3
    disp = (1.5 * rng.standard_normal((N, 3))).astype(np.float32)
    vel = (3.0 * rng.standard_normal((N, 3))).astype(np.float32)
    dvel = (0.5 * rng.standard_normal((N, 3))).astype(np.float32)
    a_obs = None if lightcone else 0.7
    lp = nbody.LatticePos(disp, evol)
    x64 = lp.to_absolute().cpu().numpy()
    pos_in = lp if lattice else x64.astype(np.float32)
    if not lattice:
        x64 = pos_in.astype(np.float64)
    got, ctx = bricks.observe_pos(cosmo, pos_in, vel, center, rotvec, box, evol, paint, a_obs=a_obs, curved_sky=curved, dvel=dvel,
                                  return_ctx=True)
    got_abs = got.to_absolute().cpu().numpy() if lattice else got.cpu().numpy().astype(np.float64)
    f = lambda x, v, dv: bo.observe_pos(cosmo, x, v, center, R, box, evol, paint, a_obs, curved, dv)
    ref = f(x64, vel.astype(np.float64), dvel.astype(np.float64))
    assert np.abs(got_abs - ref).max() < 2e-4 and rel_l2(got_abs - x64 * np.divide(paint, evol), ref - x64 * np.divide(paint, evol)) < 2e-5
    ob = rng.standard_normal((N, 3))
    pb, vb, db, gfb = bricks.observe_pos_vjp(ctx, ob.astype(np.float32))
    eps = 1e-4
    for name, bar, idx in (("pos", pb, 0), ("vel", vb, 1), ("dvel", db, 2)):
        d = rng.standard_normal((N, 3))
        args_p = [x64, vel.astype(np.float64), dvel.astype(np.float64)]
        args_m = [a.copy() for a in args_p]
        args_p[idx] = args_p[idx] + eps * d
        args_m[idx] = args_m[idx] - eps * d
        fd = ((f(*args_p) - f(*args_m)) * ob).sum() / (2 * eps)
        an = float((bar.double().cpu().numpy() * d).sum())
        assert abs(fd - an) < 2e-3 * max(abs(fd), np.linalg.norm(ob) * np.linalg.norm(d) * 1e-2), (name, fd, an)
    if not lightcone:    # scalar growth product: d/d(gf) by scaling the velocities
        gf = float(o.a2g(cosmo, 0.7) * o.a2f(cosmo, 0.7))
        v64 = vel.astype(np.float64)
        fd = ((f(x64, v64 * (1 + eps), dvel.astype(np.float64)) - f(x64, v64 * (1 - eps), dvel.astype(np.float64))) * ob).sum() / (2 * eps * gf)
        assert abs(fd - gfb) < 2e-3 * abs(fd)
    else:
        assert gfb == 0.0


def test_lightcone_table_cotangents(gpu):
    """The cotangents of the look-up TABLES themselves (how the cosmology enters on the light cone, model.py:740, :781):
    mcpm_lightcone_tables_vjp_f32 (a_q = chi2a(r0_q) -> a2g, a2g2, a2dg2dg) and mcpm_observe_pos_tables_vjp_f32 (a2g a2f at the
    evolved positions) against float64 directional finite differences of the same chains in numpy, perturbing every table along
    a random direction."""
    import torch
    from montecosmo_amd import bricks, nbody
    rng = np.random.default_rng(23)
    cosmo = bricks.Planck18()
    d, gt = nbody._dist_cache(cosmo), nbody._growth_cache(cosmo)
    chi, aoc = d["chi"][::-1].copy(), d["a"][::-1].copy()
    ag = gt["a"]
    T0 = {"chi": chi, "g": gt["g"].copy(), "g2": gt["g2"].copy(), "f": gt["f"].copy(), "f2": gt["f2"].copy()}
    nchi, ng = len(chi), len(ag)
    N = 20000
    r0 = rng.uniform(800., 2600., N).astype(np.float32)          # 0.45 < a < 0.8: inside both tables
    r0[:4] = [1e-3, 5e5, 9e5, 0.5 * (chi[5] + chi[6])]      # the first bracket and the clamped far end (a node itself, r0 = 0 = chi(a = 1) included, is a kink: no derivative to test)
    gB, g2B, dB = (rng.standard_normal(N).astype(np.float32) for _ in range(3))

    def L_lagr(T, round32=True):
        a = np.interp(r0.astype(np.float64), T["chi"], aoc)
        if round32:
            a = a.astype(np.float32).astype(np.float64)
        g, g2, f, f2 = (np.interp(a, ag, T[k]) for k in ("g", "g2", "f", "f2"))
        g2 = -3 / 7 * g2
        return float((gB * g).sum() + (g2B * g2).sum() + (dB * o.safe_div(g2 * f2, g * f)).sum())

    dev = gpu
    tabs = torch.from_numpy(np.concatenate([chi, aoc, ag, T0["g"], T0["g2"], T0["f"], T0["f2"]])).to(dev)
    out = torch.empty(nchi + 4 * ng, dtype=torch.float64, device=dev)
    plan = nbody.get_plan((16, 16, 16))
    t = lambda x: torch.from_numpy(x).to(dev)
    r0d, gBd, g2Bd, dBd = t(r0), t(gB), t(g2B), t(dB)
    plan.call("mcpm_lightcone_tables_vjp_f32", nbody._ptr(r0d), N, nbody._ptr(tabs), nchi, ng, nbody._ptr(gBd), nbody._ptr(g2Bd),
              nbody._ptr(dBd), nbody._ptr(out))
    bar = out.cpu().numpy()
    bars = {"chi": bar[:nchi], "g": bar[nchi:nchi + ng], "g2": bar[nchi + ng:nchi + 2 * ng], "f": bar[nchi + 2 * ng:nchi + 3 * ng],
            "f2": bar[nchi + 3 * ng:]}

    def check(L, bars, keys, tag):
        for k in keys:
            dirn = rng.standard_normal(len(T0[k])) * (np.abs(np.gradient(T0[k])) if k == "chi" else np.abs(T0[k]))
            eps = 1e-6
            Tp, Tm = dict(T0), dict(T0)
            Tp[k], Tm[k] = T0[k] + eps * dirn, T0[k] - eps * dirn
            fd = (L(Tp) - L(Tm)) / (2 * eps)
            an = float(np.dot(bars[k], dirn))
            scale = np.linalg.norm(bars[k] * dirn) + 1e-30
            assert abs(fd - an) < 2e-3 * max(abs(fd), scale), (tag, k, fd, an)

    check(L_lagr, bars, ("g", "g2", "f", "f2"), "lagrangian")
    # chi moves a itself: differentiate the chain without the float32 rounding of a (which makes L piecewise constant in chi; the
    # kernel's slopes are taken at the rounded a, 3e-8 away)
    check(lambda T: L_lagr(T, round32=False), bars, ("chi",), "lagrangian chi")
    # NULL g2 / dg2dg cotangents = zeros
    plan.call("mcpm_lightcone_tables_vjp_f32", nbody._ptr(r0d), N, nbody._ptr(tabs), nchi, ng, nbody._ptr(gBd), None, None, nbody._ptr(out))
    only_g = out.cpu().numpy()
    assert not np.any(only_g[nchi + ng:]) and np.any(only_g[nchi:nchi + ng])

    # observation side: positions on a 16^3 lattice far enough for 0.4 < a < 0.8
    evol, paint = (16, 16, 16), (24, 20, 16)
    box, center, rotvec = (640., 640., 640.), (100., -50., 1500.), (0.2, -0.1, 0.3)
    R = bo.rotvec_matrix(rotvec)
    Np = 16 ** 3
    disp = (1.5 * rng.standard_normal((Np, 3))).astype(np.float32)
    vel = (3.0 * rng.standard_normal((Np, 3))).astype(np.float32)
    dvel = (0.5 * rng.standard_normal((Np, 3))).astype(np.float32)
    lp = nbody.LatticePos(disp, evol)
    x64 = lp.to_absolute().cpu().numpy()
    _, octx = bricks.observe_pos(cosmo, lp, vel, center, rotvec, box, evol, paint, a_obs=None, curved_sky=True, dvel=dvel, return_ctx=True)
    ob = rng.standard_normal((Np, 3)).astype(np.float32)
    outO = torch.empty(nchi + 2 * ng, dtype=torch.float64, device=dev)
    obd = t(ob)
    octx.plan.call("mcpm_observe_pos_tables_vjp_f32", nbody._ptr(octx.p), nbody._ptr(octx.v), nbody._ptr(octx.dv), Np, octx.mode, octx.geom,
                   octx.flags, nbody._ptr(octx.tables), nchi, ng, nbody._ptr(obd), nbody._ptr(outO))
    bo_ = outO.cpu().numpy()
    barsO = {"chi": bo_[:nchi], "g": bo_[nchi:nchi + ng], "f": bo_[nchi + ng:]}

    def L_obs(T):
        # bo.observe_pos with chi2a / a2g / a2f evaluated on the perturbed tables
        P = bo.cell2phys_pos(x64, center, R, box, evol)
        r = np.linalg.norm(P, axis=-1)
        los = P / r[:, None]
        a = np.interp(r, T["chi"], aoc)
        gf = np.interp(a, ag, T["g"]) * np.interp(a, ag, T["f"])
        V = bo.cell2phys_vel(vel.astype(np.float64), R, box, evol) * gf[:, None] + dvel
        dpos = (V * los).sum(-1, keepdims=True) * los
        out_ = bo.phys2cell_pos(P + dpos, center, R, box, paint)
        return float((out_ * ob).sum())
    check(L_obs, barsO, ("chi", "g", "f"), "observe")
