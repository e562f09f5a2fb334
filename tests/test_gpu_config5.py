"""BASELINE config 5 at its own size under -m gpu: the call a field-level NUTS chain times -- log density +
gradient of the field-level model at a 256^3 evolution mesh (final 146^3, initial 220^3, particles and paint mesh 256^3, 10-step
BullFrog N-body, 'kaiser' preconditioning; montecosmo/model.py:350-363 -> :640-679, :686-838, :840-908) -- through the HIP path
against the float64 restatement run on the threaded oracle back end, and its scalar gradients against central differences.

At this size the same Python calls take other code paths than in the 8^3 / 16^3 model tests: the tiled paints with weights and the
device-chosen window halo, `nufft` on the 256^3 paint mesh, `bias.hip` on 256^3, `reshape.hip` between 220^3 and 256^3 and 146^3,
the hand-written Poisson solve, the 2 x 2 row-patch gathers.

Tolerances: forward (galaxy mesh and mean counts) relative L2 < 2e-4 (the model tests' gate; measured ~1e-5); log density: the
difference from the float64 value is held against the size of the likelihood term's own round-off, not against |lp| ~ 2e7;
scalar gradients: central differences of the SAME log density at an offset point where they are well conditioned (h = 0.5 in
sample space, relative 2e-3; profiles/r01_scalar_grads_256.txt did this by hand with round-1 kernels), Omega_m included.
"""
import os
import time

import numpy as np
import pytest

from oracle import pm_oracle as o, bias_oracle as bo, background as obg  # checker only

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def _cos(c, s8):
    c.sigma8 = s8
    return c


LAT = {"Omega_m": dict(loc=0.3111, scale=0.1, loc_fid=0.3111, scale_fid=1e-2),
       "sigma8": dict(loc=0.8102, scale=0.1, loc_fid=0.8102, scale_fid=1e-2),
       "b1": dict(loc=1., scale=1e2, loc_fid=1., scale_fid=1e-2), "b2": dict(loc=0., scale=1e2, loc_fid=0., scale_fid=3e-2),
       "bs2": dict(loc=0., scale=1e2, loc_fid=0., scale_fid=1e-1), "bn2": dict(loc=0., scale=1e3, loc_fid=0., scale_fid=1.)}
FIXED = dict(b3=0., bds2=0., bs3=0., bnpar=0., ngbars=1e-3, s_e=1.0, s_ed=0., s_e2=0.)


@pytest.fixture(scope="module")
def threads(gpu):
    try:
        nthr = len(os.sched_getaffinity(0))
    except AttributeError:
        nthr = os.cpu_count() or 1
    prev = o.set_threads(max(1, min(nthr, 64)))
    yield
    o.set_threads(prev)


def _problem(final_n, evolution, a_obs):
    """The problem of tools/run_nuts_field.py (the config-5 driver): synthetic truth -> observed counts -> log density."""
    import torch
    from montecosmo_amd import model, logdensity, bricks, utils, nbody
    ks = np.logspace(-3, 1, 128)
    kpow = (ks, 3.0e4 * (ks / 0.02) / (1 + (ks / 0.02) ** 2.6))
    fwd = model.FieldLevelForward(final_shape=(final_n,) * 3, cell_length=10., box_center=(0., 0., 2500.), evolution=evolution,
                                  nbody_n_steps=10, a_obs=a_obs, lin_kpow=kpow)
    gen = torch.Generator(device="cuda").manual_seed(0)
    ld0 = logdensity.FieldLevelLogDensity(fwd, torch.zeros(fwd.final_shape), LAT, FIXED, precond="kaiser")
    truth = {k + "_": 0.0 for k in LAT}
    truth["white_mesh_"] = torch.randn(fwd.init_shape, device="cuda", generator=gen) * ld0.scale
    base = ld0.base_params(truth)
    gxy = fwd.evolve(ld0.make_cosmo(base), {k: base[k] for k in bricks.BIAS_KEYS}, utils.rg2cgh(truth["white_mesh_"]) * ld0.transfer)
    rc = FIXED["ngbars"] * fwd.cell_length ** 3
    cm = rc * nbody.irfftn(utils.chreshape(nbody.rfftn(gxy), utils.r2chshape(fwd.final_shape)))
    obs = cm + rc ** .5 * torch.randn(fwd.final_shape, device="cuda", generator=gen)
    ld = logdensity.FieldLevelLogDensity(fwd, obs, LAT, FIXED, precond="kaiser")
    offset = dict(truth, **{"b1_": 30.0, "sigma8_": -8.0, "Omega_m_": 10.0, "b2_": 5.0, "white_mesh_": 0.7 * truth["white_mesh_"]})
    return fwd, ld, obs, truth, offset


@pytest.mark.parametrize("evolution,a_obs", [("nbody", 0.7), ("lpt", None)])
def test_log_density_and_gradient_at_config5_size(gpu, threads, evolution, a_obs):
    """'nbody' at a_obs = 0.7: BASELINE config 5 (what tools/run_nuts_field.py runs).  'lpt' on the light cone: the reference's
    default configuration (model.py:45, :62) at the same size, Omega_m through the light-cone look-up tables."""
    import torch
    fwd, ld, obs, truth, point = _problem(146, evolution, a_obs)
    assert fwd.evol_shape == (256, 256, 256) and fwd.paint_shape == (256, 256, 256) and fwd.init_shape == (220, 220, 220)
    t0 = time.perf_counter()
    lp, g = ld.logdensity_and_grad(point)
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    lp2, g2 = ld.logdensity_and_grad(point)
    torch.cuda.synchronize()
    t_grad = time.perf_counter() - t0
    # Repeatable call after call, bit for bit, scalars included: every grid sum on the gradient path is either an integer sum (paints, light-cone
    # table cotangents) or per-workgroup float64 partials added up in a fixed order (reduce_dev.h: det_fold_kernel; round 3's float64 atomics
    # moved the last bit of the scalar cotangents in 3 of 11 repeats).
    same = lambda ga, gb: all(ga[k] == gb[k] for k in gb if k != "white_mesh_") and torch.equal(ga["white_mesh_"], gb["white_mesh_"])
    assert lp2 == lp and same(g2, g)
    for _ in range(10):
        lp3, g3 = ld.logdensity_and_grad(point)
        assert lp3 == lp and same(g3, g)
    # ---- forward against the float64 restatement (threaded back end) -------------------------------------------------
    cfg = dict(fwd.config(), final_shape=fwd.final_shape, cell_length=fwd.cell_length, precond="kaiser")
    make_cosmo = lambda base: _cos(obg.Planck18(Omega_c=base["Omega_m"] - 0.0490), base["sigma8"])
    sample = {k: (v.double().cpu().numpy() if torch.is_tensor(v) else v) for k, v in point.items()}
    aux = {}
    t0 = time.perf_counter()
    lp_o = bo.log_density(cfg, LAT, FIXED, sample, obs.double().cpu().numpy(), make_cosmo, aux=aux)
    t_oracle = time.perf_counter() - t0
    from montecosmo_amd import bricks, utils, nbody
    base = ld.base_params(point)
    white = utils.rg2cgh(point["white_mesh_"]) * ld.transfer
    gxy = fwd.evolve(ld.make_cosmo(base), {k: base[k] for k in bricks.BIAS_KEYS}, white)
    e_white = rel_l2(white.cpu().numpy().view(np.float32), np.asarray(aux["white"], np.complex128).view(np.float64))
    e_gxy = rel_l2(gxy.cpu().numpy(), aux["gxy"])
    rc = FIXED["ngbars"] * fwd.cell_length ** 3
    cm = rc * nbody.irfftn(utils.chreshape(nbody.rfftn(gxy), utils.r2chshape(fwd.final_shape)))
    e_cm = rel_l2(cm.cpu().numpy(), aux["count"])
    contrast = float(np.std(aux["count"] / rc - 1))
    print(f"\n[{evolution}, a_obs={a_obs}] lp {lp:.2f} oracle {lp_o:.2f} diff {lp - lp_o:+.3f}; white {e_white:.2e} gxy {e_gxy:.2e} count {e_cm:.2e} "
          f"(count contrast std {contrast:.3f}); first call {t_first * 1e3:.0f} ms, gradient {t_grad * 1e3:.1f} ms, oracle value {t_oracle:.0f} s")
    assert e_white < 2e-6 and e_gxy < 2e-4 and e_cm < 2e-4
    assert 0.2 < contrast < 5.0                                   # a clustered field, not noise and not a blow-up
    # the float32 path's log density against the float64 one: N cells each off by ~z dz with dz = d count / sigma; bound it by
    # 5 sigma of that sum for the measured count error, plus the float32 sum of the prior term
    ncell = float(np.prod(fwd.final_shape))
    dz = e_cm * float(np.linalg.norm(aux["count"])) / ncell ** .5 / rc ** .5
    zs = float(np.sqrt(np.mean(((obs.double().cpu().numpy() - aux["count"]) / rc ** .5) ** 2)))
    bound = 5.0 * ncell ** .5 * dz * zs + 1e-7 * abs(lp_o) + ncell * dz ** 2
    assert abs(lp - lp_o) < bound, (lp, lp_o, bound)
    # ---- scalar gradients against central differences of the same log density -----------------------------------------
    h = 0.5
    worst = 0.0
    for k in LAT:
        fd = (ld(dict(point, **{k + "_": point[k + "_"] + h})) - ld(dict(point, **{k + "_": point[k + "_"] - h}))) / (2 * h)
        err = abs(fd - g[k + "_"]) / abs(fd)
        worst = max(worst, err)
        print(f"  d lp / d {k + '_':9s}: analytic {g[k + '_']:14.4f}   central difference (h = {h}) {fd:14.4f}   rel {err:.1e}")
        assert err < 2e-3, (k, fd, g[k + "_"])
    # ---- the field gradient along a random direction, same way ---------------------------------------------------------
    gen = torch.Generator(device="cuda").manual_seed(5)
    d = torch.randn(fwd.init_shape, device="cuda", generator=gen) * ld.scale
    eps = 1e-2
    fd = (ld(dict(point, white_mesh_=point["white_mesh_"] + eps * d)) - ld(dict(point, white_mesh_=point["white_mesh_"] - eps * d))) / (2 * eps)
    an = float((g["white_mesh_"].double() * d.double()).sum())
    print(f"  d lp / d white_mesh_ . d: analytic {an:.4f}   central difference {fd:.4f}   rel {abs(fd - an) / abs(fd):.1e}")
    assert abs(fd - an) < 5e-3 * abs(fd)
