"""FieldLevelModel.evolve on the HIP path (montecosmo_amd/model.py) against the float64 oracle composition
(oracle/bias_oracle.py::evolve), forward and reverse sweep.  model.py:686-838, lagrangian bias, lpt / nbody."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import pm_oracle as o, bias_oracle as bo, background as obg  # noqa: E402  (checker only)


def rel_l2(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


BIAS = dict(b1=0.8, b2=0.2, bs2=-0.15, b3=0.1, bds2=0.1, bs3=-0.05, bn2=20.0, bnpar=5.0)


def _kpow():
    ks = np.logspace(-3, 1, 128)
    return ks, 3.0e4 * (ks / 0.02) / (1 + (ks / 0.02) ** 2.6)


@pytest.mark.parametrize("evolution,a_obs,curved,ptcl", [("lpt", None, True, 2.), ("lpt", 0.6, False, 2.), ("nbody", 0.7, True, 2.),
                                                          ("nbody", 0.7, False, 1.5), ("lpt", None, True, 1.5)])
def test_evolve_forward_and_vjp(gpu, evolution, a_obs, curved, ptcl):
    from montecosmo_amd import bricks, model
    rng = np.random.default_rng(31)
    fwd = model.FieldLevelForward(final_shape=(8, 8, 8), cell_length=40., box_center=(60., -40., 1400.), box_rotvec=(0.1, 0.2, -0.1),
                                  evolution=evolution, nbody_n_steps=3, lpt_order=2, init_oversamp=1.5, evol_oversamp=2.,
                                  ptcl_oversamp=ptcl, paint_oversamp=2., a_obs=a_obs, curved_sky=curved, lin_kpow=_kpow(),
                                  nbody_a_start=0.1)      # ptcl 1.5: particle lattice (12^3) coarser than the meshes (16^3)
    cfg = fwd.config()
    assert cfg["init_shape"] == (12, 12, 12) and cfg["evol_shape"] == (16, 16, 16)
    cosmo, cosmo_o = bricks.Planck18(), obg.Planck18()
    cosmo_o.sigma8 = cosmo.sigma8
    white = np.fft.rfftn(rng.standard_normal((12, 12, 12))) * (12 ** 3 / np.prod(cfg["box_size"])) ** .5   # bricks.py:138-146
    gxy, ctx = fwd.evolve(cosmo, BIAS, white.astype(np.complex64), return_ctx=True)
    ref, aux = bo.evolve(cfg, cosmo_o, BIAS, white)
    d_lin = np.fft.irfftn(aux["init_mesh"], s=(16, 16, 16), axes=(0, 1, 2))
    assert 0.05 < d_lin.std() < 1.0                                   # a clustered but perturbative field
    assert gxy.shape == (16, 16, 16)
    assert rel_l2(gxy.cpu().numpy(), ref) < 2e-4
    assert abs(float(gxy.mean()) - 1.0) < 0.1                         # 1 + delta_obs
    gb = rng.standard_normal((16, 16, 16))
    grads = fwd.evolve_vjp(ctx, gb.astype(np.float32))
    L = lambda wh, bias_, s8: float((gb * bo.evolve(cfg, _with_s8(cosmo_o, s8), bias_, wh)[0]).sum())
    eps = 1e-5
    dW = np.fft.rfftn(rng.standard_normal((12, 12, 12))) * np.abs(white).mean() / 40.
    fd = (L(white + eps * dW, BIAS, cosmo.sigma8) - L(white - eps * dW, BIAS, cosmo.sigma8)) / (2 * eps)
    an = float(np.sum(np.conj(grads["white_mesh"].cpu().numpy().astype(np.complex128)) * dW).real)
    assert abs(fd - an) < 3e-3 * abs(fd), ("white_mesh", fd, an)
    for k in ("b1", "bs2", "bnpar"):
        h = 1e-4 * max(1.0, abs(BIAS[k]))
        fdk = (L(white, dict(BIAS, **{k: BIAS[k] + h}), cosmo.sigma8) - L(white, dict(BIAS, **{k: BIAS[k] - h}), cosmo.sigma8)) / (2 * h)
        assert abs(fdk - grads["bias"][k]) < 3e-3 * max(abs(fdk), 1e-3 * abs(fd)), (k, fdk, grads["bias"][k])
    h = 1e-4
    fds = (L(white, BIAS, cosmo.sigma8 + h) - L(white, BIAS, cosmo.sigma8 - h)) / (2 * h)
    assert abs(fds - grads["sigma8"]) < 3e-3 * abs(fds), ("sigma8", fds, grads["sigma8"])


    # cosmology: through the growth tables at fixed a_obs (finite-difference Jacobian of host scalars), and on the light cone
    # (a_obs = None, the reference's default, model.py:62) through the per-particle chi2a / a2g / a2g2 / a2f look-ups, whose table
    # cotangents are formed on the device (mcpm_lightcone_tables_vjp_f32, mcpm_observe_pos_tables_vjp_f32)
    got = fwd.cosmo_vjp(ctx, grads, params=("Omega_m",))["Omega_m"]

    def L_om(dom):
        c = obg.Planck18(Omega_c=cosmo.Omega_c + dom)
        c.sigma8 = cosmo.sigma8
        return float((gb * bo.evolve(cfg, c, BIAS, white)[0]).sum())
    h = 1e-4
    fdo = (L_om(h) - L_om(-h)) / (2 * h)
    assert abs(fdo - got) < 1e-2 * abs(fdo), ("Omega_m", fdo, got)


def _with_s8(c, s8):
    c.sigma8 = s8
    return c


@pytest.mark.parametrize("evolution,s_e2,precond,a_obs,survey", [("lpt", 0.0, "fourier", 0.65, False), ("nbody", 0.02, "fourier", 0.65, False),
                                                                 ("nbody", 0.02, "kaiser", 0.65, False), ("lpt", 0.0, "kaiser", None, False),
                                                                 ("lpt", 0.02, "kaiser", 0.65, True), ("nbody", 0.0, "fourier", 0.65, True),
                                                                 ("lpt", 0.0, "kaiser", 0.65, "eh"),
                                                                 ("kaiser", 0.02, "kaiser", 0.65, "flat"),
                                                                 ("lpt", 0.02, "kaiser", 0.65, "ngbars"), ("lpt", 0.0, "fourier", 0.65, "ngbars+survey")])
def test_log_density_and_gradient(gpu, evolution, s_e2, precond, a_obs, survey):
    """Prior + evolve + 'quad_gauss' likelihood (model.py:640-679, :840-908) on the HIP path against the float64
    restatement; gradient w.r.t. every sampled parameter against central differences of that restatement.  'kaiser':
    the reference's default preconditioning (model.py:1134-1147), once at fixed a_obs and once on the light cone (a_obs = None, the
    reference's default, with Omega_m sampled: its gradient runs through the light-cone look-up tables).
    "eh": lin_kpow = None, the linear power is the Eisenstein-Hu fit of the sampled cosmology.  evolution "kaiser": the linear
    Kaiser model on the flat sky (bricks.py:170-198), gradients w.r.t. b1 and the cosmology through its growth and growth rate.
    survey: a selection mesh on the paint mesh, a mask over the final cells and two radial shells with their own mean
    densities (model.py:855-866, :1087-1098; bricks.py:1106-1122)."""
    from montecosmo_amd import model, logdensity
    rng = np.random.default_rng(41)
    fwd = model.FieldLevelForward(final_shape=(8, 8, 8), cell_length=40., box_center=(60., -40., 1400.), box_rotvec=(0.1, 0.2, -0.1),
                                  evolution=evolution, nbody_n_steps=3, lpt_order=2, init_oversamp=1.5, evol_oversamp=2.,
                                  ptcl_oversamp=2., paint_oversamp=2., a_obs=a_obs, curved_sky=survey != "flat",
                                  lin_kpow=None if survey == "eh" else _kpow(), nbody_a_start=0.1)
    cfg = dict(fwd.config(), final_shape=(8, 8, 8), cell_length=40., precond=precond)
    extra = {}
    sampled_ngbars = isinstance(survey, str) and survey.startswith("ngbars")      # ngbars as a per-shell latent (model.py:205-214, :1099-1103)
    if sampled_ngbars:
        survey = survey.endswith("survey")
    if survey in ("eh", "flat"):      # "eh": no tabulated power, Eisenstein-Hu of the sampled cosmology (bricks.py:69-79), Omega_m
        survey = False                # moves its shape; "flat": flat sky (the Kaiser model's built branch)
    if survey:
        gsel = np.indices(fwd.paint_shape).astype(float)
        extra["selec_mesh"] = 0.7 + 0.3 * np.cos(2 * np.pi * gsel[0] / fwd.paint_shape[0]) * np.sin(2 * np.pi * gsel[2] / fwd.paint_shape[2]) \
            + 0.1 * rng.uniform(size=fwd.paint_shape)
        extra["mask_mesh"] = rng.uniform(size=(8, 8, 8)) < 0.8
        cfg.update(extra)
    lat = {"Omega_m": dict(loc=0.3111, scale=0.1, loc_fid=0.3111, scale_fid=1e-2, low=0.05, high=1.),   # model.py:76-83
           "sigma8": dict(loc=0.8102, scale=0.1, loc_fid=0.8102, scale_fid=1e-2, low=0., high=np.inf),   # model.py:100-111
           "b1": dict(loc=1., scale=1e2, loc_fid=1., scale_fid=1e-2), "b2": dict(loc=0., scale=1e2, loc_fid=0., scale_fid=3e-2),
           "bs2": dict(loc=0., scale=1e2, loc_fid=0., scale_fid=1e-1), "bn2": dict(loc=0., scale=1e3, loc_fid=0., scale_fid=1.),
           "s_ed": dict(loc=0., scale=1e1, loc_fid=0., scale_fid=1e-2),
           "bnpar": dict(low=-10., high=20., loc_fid=5., scale_fid=30. / 12 ** .5)}                                                            # uniform (DetruncUnif)
    fixed = dict(b3=0.1, bds2=0.1, bs3=-0.05, ngbars=(np.array([1e-3, 1.4e-3]) if survey else 1e-3), s_e=1.0, s_e2=s_e2)
    if sampled_ngbars:
        fixed.pop("ngbars")
        lat["ngbars"] = dict(loc=np.array([1e-3, 1.4e-3]), scale=1e-2, loc_fid=np.array([1e-3, 1.4e-3]), scale_fid=1e-5, low=0., high=np.inf)
        extra["n_rbins"] = 2
    make_cosmo = lambda base: _cos(obg.Planck18(Omega_c=base["Omega_m"] - 0.0490), base["sigma8"])
    sample = {k + "_": (float(rng.normal(0, 1.0)) if k != "ngbars" else rng.normal(0, 1.0, 2)) for k in lat}
    sample["white_mesh_"] = rng.standard_normal((12, 12, 12))
    truth = dict(sample, **{"b1_": 20.0})
    base_t = dict(fixed, **{k: (bo.std2trunc(truth[k + "_"], c["loc_fid"], c["scale_fid"], c["low"], c["high"]) if "low" in c
                                else truth[k + "_"] * c["scale_fid"] + c["loc_fid"]) for k, c in lat.items() if k != "ngbars"})
    base_t.setdefault("ngbars", 1e-3)
    white_t = o.rg2cgh(truth["white_mesh_"]) * np.divide(cfg["init_shape"], cfg["box_size"]).prod() ** .5
    gxy_t, _ = bo.evolve(cfg, make_cosmo(base_t), {k: base_t[k] for k in bo.BIAS_KEYS}, white_t)
    rc = float(np.mean(base_t["ngbars"])) * 40. ** 3
    cm_t = rc * np.fft.irfftn(o.chreshape(np.fft.rfftn(gxy_t), o.r2chshape((8, 8, 8))), s=(8, 8, 8), axes=(0, 1, 2))
    obs = cm_t + rc ** .5 * rng.standard_normal((8, 8, 8))
    ld = logdensity.FieldLevelLogDensity(fwd, obs, lat, fixed, precond=precond, **extra)
    lp, grad = ld.logdensity_and_grad({k: (v if (np.ndim(v) == 0 or k == "ngbars_") else v.astype(np.float32)) for k, v in sample.items()})
    ref = lambda s: bo.log_density(cfg, lat, fixed, s, obs, make_cosmo)
    lp_o = ref(sample)
    assert np.isfinite(lp_o) and abs(lp - lp_o) < 2e-4 * abs(lp_o) + 0.05, (lp, lp_o)
    for k in lat:
        h = 1e-4
        if k == "ngbars":
            for i in range(2):
                e = np.eye(2)[i] * h
                fd = (ref(dict(sample, ngbars_=sample["ngbars_"] + e)) - ref(dict(sample, ngbars_=sample["ngbars_"] - e))) / (2 * h)
                assert abs(fd - grad["ngbars_"][i]) < 1e-2 * abs(fd) + 1e-3, (k, i, fd, grad["ngbars_"])
            continue
        fd = (ref(dict(sample, **{k + "_": sample[k + "_"] + h})) - ref(dict(sample, **{k + "_": sample[k + "_"] - h}))) / (2 * h)
        assert abs(fd - grad[k + "_"]) < 1e-2 * abs(fd) + 1e-3, (k, fd, grad[k + "_"])
    d = rng.standard_normal((12, 12, 12))
    h = 1e-4
    fd = (ref(dict(sample, white_mesh_=sample["white_mesh_"] + h * d)) - ref(dict(sample, white_mesh_=sample["white_mesh_"] - h * d))) / (2 * h)
    gw = grad["white_mesh_"].double().cpu().numpy()
    an = float((gw * d).sum())
    typical = np.linalg.norm(gw) * np.linalg.norm(d) / np.sqrt(d.size)      # |<g, d>| for a random direction
    assert abs(fd - an) < 5e-3 * max(abs(fd), typical), ("white_mesh_", fd, an, typical)


def _cos(c, s8):
    c.sigma8 = s8
    return c


@pytest.mark.parametrize("s_e2", [0.0, 0.005])
def test_cut_sky_selection_exactly_zero_outside_the_mask(gpu, s_e2):
    """ADVICE r1 (logdensity.py): a cut-sky selection is exactly 0 in the unobserved cells (the normal output of
    cutsky2selection, bricks.py:1054-1103); the reference extracts the observed cells first (mesh2masked, model.py:856-863).
    Here count / selec in the unobserved cells must not poison the sum or the gradient (NaN * 0 = NaN): value against the
    float64 restatement, gradient against its central differences, everything finite."""
    from montecosmo_amd import model, logdensity, samplers
    rng = np.random.default_rng(43)
    fwd = model.FieldLevelForward(final_shape=(8, 8, 8), cell_length=40., box_center=(60., -40., 1400.), box_rotvec=(0.1, 0.2, -0.1),
                                  evolution="lpt", lpt_order=2, init_oversamp=1.5, evol_oversamp=2., ptcl_oversamp=2.,
                                  paint_oversamp=1., a_obs=0.65, curved_sky=True, lin_kpow=_kpow())
    assert tuple(fwd.paint_shape) == (8, 8, 8)          # the selection reaches the final mesh unchanged: exact zeros stay exact
    g = np.indices((8, 8, 8)).astype(float)
    sel = (0.6 + 0.3 * np.cos(2 * np.pi * g[1] / 8)) * (g[0] < 5)              # exactly 0 for x >= 5
    mask = sel > 0
    extra = dict(selec_mesh=sel, mask_mesh=mask)
    cfg = dict(fwd.config(), final_shape=(8, 8, 8), cell_length=40., precond="fourier", **extra)
    lat = {"sigma8": dict(loc=0.8102, scale=0.1, loc_fid=0.8102, scale_fid=1e-2, low=0., high=np.inf),
           "b1": dict(loc=1., scale=1e2, loc_fid=1., scale_fid=1e-2), "s_ed": dict(loc=0., scale=1e1, loc_fid=0., scale_fid=1e-2)}
    fixed = dict(Omega_m=0.3111, b2=0.1, bs2=-0.1, bn2=0., b3=0., bds2=0., bs3=0., bnpar=0., ngbars=1e-3, s_e=1.0, s_e2=s_e2)
    make_cosmo = lambda base: _cos(obg.Planck18(Omega_c=base["Omega_m"] - 0.0490), base["sigma8"])
    sample = {k + "_": float(rng.normal(0, 1.0)) for k in lat}
    sample["white_mesh_"] = rng.standard_normal((12, 12, 12))
    obs = 60. + 8. * rng.standard_normal((8, 8, 8))
    obs[~mask] = np.nan                                                          # never read: a NaN here must not matter
    ld = logdensity.FieldLevelLogDensity(fwd, np.where(mask, obs, 0.), lat, fixed, precond="fourier", **extra)
    lp, grad = ld.logdensity_and_grad({k: (v if np.ndim(v) == 0 else v.astype(np.float32)) for k, v in sample.items()})
    with np.errstate(all="ignore"):
        ref = lambda s: bo.log_density(cfg, lat, fixed, s, obs, make_cosmo)
        lp_o = ref(sample)
        assert np.isfinite(lp_o) and np.isfinite(lp) and abs(lp - lp_o) < 2e-4 * abs(lp_o) + 0.05, (lp, lp_o)
        assert all(np.isfinite(grad[k + "_"]) for k in lat) and bool(torch_isfinite(grad["white_mesh_"]))
        for k in lat:
            h = 1e-4
            fd = (ref(dict(sample, **{k + "_": sample[k + "_"] + h})) - ref(dict(sample, **{k + "_": sample[k + "_"] - h}))) / (2 * h)
            assert abs(fd - grad[k + "_"]) < 1e-2 * abs(fd) + 1e-3, (k, fd, grad[k + "_"])
        d = rng.standard_normal((12, 12, 12))
        h = 1e-4
        fd = (ref(dict(sample, white_mesh_=sample["white_mesh_"] + h * d)) - ref(dict(sample, white_mesh_=sample["white_mesh_"] - h * d))) / (2 * h)
    gw = grad["white_mesh_"].double().cpu().numpy()
    typical = np.linalg.norm(gw) * np.linalg.norm(d) / np.sqrt(d.size)
    assert abs(fd - float((gw * d).sum())) < 5e-3 * max(abs(fd), typical)
    # and the flat adapter refuses a NaN log density instead of turning it into a frozen chain
    flat = samplers.FlatLogDensity(ld)
    q = flat.pack({k: (v if np.ndim(v) == 0 else v.astype(np.float32)) for k, v in sample.items()})
    assert abs(flat(q)[0] - lp) < 1e-3 * abs(lp) + 0.05       # float32 packing of the scalars
    ld.count_obs[0, 0, 0] = float("nan")                 # an observed cell this time
    if s_e2 == 0.0:
        with pytest.raises(FloatingPointError):
            flat(q)
    else:      # the quadratic branch maps a NaN discriminant to "outside the support" (utils.py:507): rejected and counted
        assert flat(q)[0] == -np.inf and flat.n_nonfinite == 1


def torch_isfinite(t):
    import torch
    return torch.isfinite(t).all()
