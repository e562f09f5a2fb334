"""BASELINE config 5 in miniature under -m gpu: short seeded NUTS and MCLMC chains over
`FieldLevelLogDensity` -- the HIP log density and its hand-written gradient (montecosmo/model.py:350-363, the contract
samplers.py:44 / :311-315 consumes) -- checking reproducibility of a seeded chain, bounded energy errors, and the log
density of the states the chain visits against the float64 restatement (oracle/bias_oracle.py::log_density)."""
import math

import numpy as np
import pytest

from oracle import bias_oracle as bo, background as obg  # checker only

pytestmark = pytest.mark.gpu


def _setup(evolution="nbody"):
    import torch
    from montecosmo_amd import model, logdensity, samplers
    rng = np.random.default_rng(7)
    ks = np.logspace(-3, 1, 128)
    kpow = (ks, 3.0e4 * (ks / 0.02) / (1 + (ks / 0.02) ** 2.6))
    fwd = model.FieldLevelForward(final_shape=(8, 8, 8), cell_length=40., box_center=(60., -40., 1400.), box_rotvec=(0.1, 0.2, -0.1),
                                  evolution=evolution, nbody_n_steps=3, lpt_order=2, init_oversamp=1.5, evol_oversamp=2.,
                                  ptcl_oversamp=2., paint_oversamp=2., a_obs=0.65, curved_sky=True, lin_kpow=kpow, nbody_a_start=0.1)
    cfg = dict(fwd.config(), final_shape=(8, 8, 8), cell_length=40., precond="kaiser")
    lat = {"sigma8": dict(loc=0.8102, scale=0.1, loc_fid=0.8102, scale_fid=1e-2, low=0., high=np.inf),
           "b1": dict(loc=1., scale=1e2, loc_fid=1., scale_fid=1e-2), "b2": dict(loc=0., scale=1e2, loc_fid=0., scale_fid=3e-2)}
    fixed = dict(Omega_m=0.3111, bs2=0., bn2=0., b3=0., bds2=0., bs3=0., bnpar=0., ngbars=1e-3, s_e=1.0, s_ed=0., s_e2=0.)
    make_cosmo = lambda base: _cos(obg.Planck18(Omega_c=base["Omega_m"] - 0.0490), base["sigma8"])
    obs = 64. + 8. * rng.standard_normal((8, 8, 8))
    ld = logdensity.FieldLevelLogDensity(fwd, obs, lat, fixed, precond="kaiser")
    flat = samplers.FlatLogDensity(ld)
    start = {k + "_": 0.0 for k in lat}
    start["white_mesh_"] = (0.3 * rng.standard_normal(fwd.init_shape)).astype(np.float32)
    q0 = flat.pack(start)
    ref = lambda q: bo.log_density(cfg, lat, fixed, {k: (np.asarray(v.cpu(), np.float64) if torch.is_tensor(v) else v)
                                                     for k, v in flat.unpack(q).items()}, obs, make_cosmo)
    return samplers, flat, q0, ref


def _cos(c, s8):
    c.sigma8 = s8
    return c


def test_log_density_and_gradient_are_bitwise_reproducible(gpu):
    """The same state gives the same log density and the same gradient, bit for bit, call after call: every sum on the path
    is order-independent or taken in a fixed order (integer accumulators in the paints; the adjoint of chreshape is a gather -- its
    float atomics once changed the last bit of a few Nyquist-plane elements from call to call -- enough for a seeded NUTS chain to end
    1 % away from its twin)."""
    import torch
    samplers, flat, q0, ref = _setup()
    rng = np.random.default_rng(0)
    for trial in range(3):
        q = q0 + 0.2 * trial * torch.from_numpy(rng.standard_normal(tuple(q0.shape)).astype(np.float32)).to(q0.device)
        lp0, g0 = flat(q)
        g0 = g0.clone()
        for _ in range(60):
            lp, g = flat(q)
            assert lp == lp0 and torch.equal(g, g0)
        # ... and the float64 scalar cotangents before they are packed to float32: float64 sums of per-workgroup partials, added up in a
        # fixed order (reduce_dev.h; added with float64 atomics until round 4, their last bit moved in about one call in four)
        sample = flat.unpack(q)
        _, d0 = flat.ld.logdensity_and_grad(sample)
        for _ in range(40):
            _, d = flat.ld.logdensity_and_grad(sample)
            assert all(np.array_equal(np.asarray(d[k]), np.asarray(d0[k])) for k in flat.scalars)


def test_nuts_chain_over_the_hip_log_density(gpu):
    samplers, flat, q0, ref = _setup()
    run = lambda: samplers.nuts_sample(flat, q0, n_warmup=12, n_samples=8, max_tree_depth=4, seed=3)
    r1 = run()
    infos = r1["infos"]
    assert len(r1["samples"]) == 8 and all(math.isfinite(i["logdensity"]) for i in infos)
    assert not any(i["diverging"] for i in infos[12:])                      # bounded energy error over the sampling phase
    assert np.mean([i["accept_stat"] for i in infos[12:]]) > 0.4
    assert r1["samples"][-1].ne(q0).any()                                   # the chain moved
    # log density of visited states against the float64 restatement
    for q in (q0, r1["samples"][0], r1["samples"][-1]):
        lp, lp_o = flat(q)[0], ref(q)
        assert abs(lp - lp_o) < 2e-4 * abs(lp_o) + 0.05, (lp, lp_o)
    assert abs(infos[-1]["logdensity"] - ref(r1["samples"][-1])) < 2e-4 * abs(infos[-1]["logdensity"]) + 0.05
    # a seeded chain is reproducible: same decisions, same states
    r2 = run()
    assert [i["n_leapfrog"] for i in r2["infos"]] == [i["n_leapfrog"] for i in infos]
    assert np.array_equal(r2["samples"][-1].cpu().numpy(), r1["samples"][-1].cpu().numpy())      # bit for bit
    assert r2["step_size"] == r1["step_size"]


def test_mclmc_chain_over_the_hip_log_density(gpu):
    samplers, flat, q0, ref = _setup("lpt")
    d = q0.numel()
    run = lambda: samplers.mclmc_sample(flat, q0, n_warmup=40, n_samples=20, seed=5)
    r1 = run()
    infos = r1["infos"]
    assert len(r1["samples"]) == 20 and all(math.isfinite(i["logdensity"]) for i in infos)
    de = np.array([i["energy_change"] for i in infos[40:]])
    assert np.all(np.isfinite(de))
    assert np.mean(de ** 2) / d < 20 * 5e-4                                 # energy-error variance per dimension near its target
    lp, lp_o = flat(r1["samples"][-1])[0], ref(r1["samples"][-1])
    assert abs(lp - lp_o) < 2e-4 * abs(lp_o) + 0.05, (lp, lp_o)
    r2 = run()
    assert np.array_equal(r2["samples"][-1].cpu().numpy(), r1["samples"][-1].cpu().numpy())      # bit for bit
    assert r2["step_size"] == r1["step_size"]


def _mock_posterior(nf=16, seed=0, evolution="lpt", a_obs=0.65, sample_om=False):
    """A 16^3 field-level inference problem whose truth is known: the truth is a prior draw (sigma8_ = b1_ = 0 in sample
    space), the observation the model's own mean at the truth plus its Gaussian noise (model.py:893-908).  sample_om: Omega_m
    is a latent too, with the reference's truncated-normal prior (model.py:76-83); a_obs = None: the light cone (model.py:62)."""
    import torch
    from montecosmo_amd import model, logdensity, samplers, bricks, utils, nbody
    ks = np.logspace(-3, 1, 128)
    kpow = (ks, 3.0e4 * (ks / 0.02) / (1 + (ks / 0.02) ** 2.6))
    fwd = model.FieldLevelForward(final_shape=(nf,) * 3, cell_length=40., box_center=(0., 0., 2500.), evolution=evolution,
                                  lpt_order=2, a_obs=a_obs, lin_kpow=kpow, nbody_n_steps=3, nbody_a_start=0.1)
    lat = {"sigma8": dict(loc=0.8102, scale=0.1, loc_fid=0.8102, scale_fid=1e-2, low=0., high=np.inf),
           "b1": dict(loc=1., scale=1e2, loc_fid=1., scale_fid=1e-2)}
    fixed = dict(Omega_m=0.3111, b2=0., bs2=0., bn2=0., b3=0., bds2=0., bs3=0., bnpar=0., ngbars=1e-3, s_e=1.0, s_ed=0., s_e2=0.)
    if sample_om:
        lat = dict({"Omega_m": dict(loc=0.3111, scale=0.1, loc_fid=0.3111, scale_fid=1e-2, low=0.05, high=1.)}, **lat)
        fixed.pop("Omega_m")
    g = torch.Generator(device="cuda").manual_seed(seed)
    truth = {k + "_": 0.0 for k in lat}
    ld0 = logdensity.FieldLevelLogDensity(fwd, torch.zeros(fwd.final_shape), lat, fixed, precond="kaiser")
    prior_std = 1.0 if ld0.scale is None else ld0.scale
    truth["white_mesh_"] = torch.randn(fwd.init_shape, device="cuda", generator=g) * prior_std
    base = ld0.base_params(truth)
    white = utils.rg2cgh(truth["white_mesh_"]) * ld0.transfer
    gxy = fwd.evolve(ld0.make_cosmo(base), {k: base[k] for k in bricks.BIAS_KEYS}, white)
    rc = fixed["ngbars"] * fwd.cell_length ** 3
    cm = rc * nbody.irfftn(utils.chreshape(nbody.rfftn(gxy), utils.r2chshape(fwd.final_shape)))
    obs = cm + rc ** .5 * torch.randn(fwd.final_shape, device="cuda", generator=g)
    ld = logdensity.FieldLevelLogDensity(fwd, obs, lat, fixed, precond="kaiser")
    flat = samplers.FlatLogDensity(ld)
    start = dict(truth)
    start["white_mesh_"] = 0.1 * torch.randn(fwd.init_shape, device="cuda", generator=g) * prior_std
    return samplers, ld, flat, flat.pack(start), truth


def test_posterior_of_sigma8_covers_the_truth(gpu):
    """adapted chains over the HIP log density of a 16^3 mock land on a posterior that covers the truth:
    the posterior mean of sigma8_ (and b1_) is within 3 posterior standard deviations of the true value.  MCLMC with the step
    size and L tuned as the reference does (samplers.py:322-331), NUTS with the windowed diagonal mass matrix (samplers.py:44)."""
    samplers, ld, flat, q0, truth = _mock_posterior()
    ns = len(flat.scalars)
    assert flat.scalars == ["sigma8_", "b1_"]
    lp_truth = ld(truth)
    r = samplers.mclmc_sample(flat, q0, n_warmup=600, n_samples=600, seed=1, keep=lambda q: q[:ns].tolist())
    d = np.array(r["samples"])
    mean, std = d.mean(0), d.std(0)
    assert np.all(np.abs(mean) < 3 * std), (mean, std)              # truth = 0 in sample space
    assert 0.5 < std[0] < 30 and std[1] > 0.5, std                   # a posterior, not a stuck chain; sigma8 to a few 1e-2
    assert r["L"] != math.sqrt(q0.numel()) and 0.2 * math.sqrt(q0.numel()) < r["L"] < 20 * math.sqrt(q0.numel())     # L was tuned
    assert abs(r["infos"][-1]["logdensity"] - lp_truth) < 0.02 * abs(lp_truth)   # the chain sits at the truth's level
    # NUTS with window adaptation (short: each transition is up to 63 gradients)
    n = samplers.nuts_sample(flat, q0, n_warmup=80, n_samples=80, max_tree_depth=6, seed=1, keep=lambda q: q[:ns].tolist())
    dn = np.array(n["samples"])
    assert np.all(np.abs(dn.mean(0)) < 3 * dn.std(0) + 3 * std), (dn.mean(0), dn.std(0))
    minv = n["inverse_mass"]
    assert minv is not None and float(minv[0]) > 1.5 and float(minv[1]) > 1.5     # the scalar latents are wider than unit scale
    assert np.mean([i["accept_stat"] for i in n["infos"][80:]]) > 0.5


@pytest.mark.parametrize("evolution,a_obs", [("lpt", None), ("nbody", 0.65)])
def test_posterior_with_omega_m_covers_the_truth(gpu, evolution, a_obs):
    """the same check with Omega_m SAMPLED (truncated-normal prior, model.py:76-83), at a 32^3 evolution mesh
    (final 18^3), for the reference's default configuration -- 'lpt' on the light cone (model.py:45, :62), where d logp / d Omega_m
    runs through the per-particle chi2a / growth look-up tables -- and for 'nbody' at fixed a_obs through the BullFrog step
    coefficients.  A wrong Omega_m gradient biases exactly this: the long 256^3 chains of rounds 1-3 sat at Omega_m_ = 47 in
    sample space, unconverged; here 1500 + 1500 MCLMC transitions (20 s) converge and the posterior means of Omega_m_, sigma8_ and
    b1_ are within 3 posterior standard deviations of the truth (0)."""
    samplers, ld, flat, q0, truth = _mock_posterior(18, 0, evolution, a_obs, sample_om=True)
    assert ld.fwd.evol_shape == (32, 32, 32)
    ns = len(flat.scalars)
    assert flat.scalars == ["Omega_m_", "sigma8_", "b1_"]
    lp_truth = ld(truth)
    r = samplers.mclmc_sample(flat, q0, n_warmup=1500, n_samples=1500, seed=1, keep=lambda q: q[:ns].tolist())
    d = np.array(r["samples"])
    mean, std = d.mean(0), d.std(0)
    print(f"\n[{evolution}, a_obs={a_obs}] posterior mean {mean.round(2)} std {std.round(2)} (sample space; truth 0), L {r['L']:.0f}")
    assert np.all(np.abs(mean) < 3 * std), (mean, std)
    assert np.all(std > 0.5) and std[0] < 30, std                    # a posterior, not a stuck chain; Omega_m to better than the prior (10)
    assert abs(r["infos"][-1]["logdensity"] - lp_truth) < 0.02 * abs(lp_truth)
    assert getattr(flat, "n_nonfinite", 0) <= 5                     # (warm-up may touch the edge of the likelihood's support)
