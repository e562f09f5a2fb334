"""Scalar-parameter gradients of the field-level log density at production size (evolution mesh 256^3) against central
differences of the SAME log density (the 8^3 parity tests check them against the float64 restatement; this checks that
nothing degrades with size: long reductions, fp32 sums).  usage: python tools/check_scalar_grads.py [final_n=146] [precond]"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from montecosmo_amd import model, logdensity, bricks, utils, nbody

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 146
precond = sys.argv[2] if len(sys.argv) > 2 else "kaiser"
ks = np.logspace(-3, 1, 128)
kpow = (ks, 3.0e4 * (ks / 0.02) / (1 + (ks / 0.02) ** 2.6))
fwd = model.FieldLevelForward(final_shape=(nf,) * 3, cell_length=10., box_center=(0., 0., 2500.), evolution="nbody",
                              nbody_n_steps=10, a_obs=0.7, lin_kpow=kpow)
lat = {"Omega_m": dict(loc=0.3111, scale=0.1, loc_fid=0.3111, scale_fid=1e-2),
       "sigma8": dict(loc=0.8102, scale=0.1, loc_fid=0.8102, scale_fid=1e-2),
       "b1": dict(loc=1., scale=1e2, loc_fid=1., scale_fid=1e-2), "b2": dict(loc=0., scale=1e2, loc_fid=0., scale_fid=3e-2),
       "bs2": dict(loc=0., scale=1e2, loc_fid=0., scale_fid=1e-1), "bn2": dict(loc=0., scale=1e3, loc_fid=0., scale_fid=1.)}
fixed = dict(b3=0., bds2=0., bs3=0., bnpar=0., ngbars=1e-3, s_e=1.0, s_ed=0., s_e2=0.)
torch.manual_seed(0)
ld0 = logdensity.FieldLevelLogDensity(fwd, torch.zeros(fwd.final_shape), lat, fixed, precond=precond)
std = 1.0 if ld0.scale is None else ld0.scale
truth = {k + "_": 0.0 for k in lat}
truth["white_mesh_"] = torch.randn(fwd.init_shape, device="cuda") * std
base = ld0.base_params(truth)
gxy = fwd.evolve(ld0.make_cosmo(base), {k: base[k] for k in bricks.BIAS_KEYS}, utils.rg2cgh(truth["white_mesh_"]) * ld0.transfer)
rc = fixed["ngbars"] * fwd.cell_length ** 3
cm = rc * nbody.irfftn(utils.chreshape(nbody.rfftn(gxy), utils.r2chshape(fwd.final_shape)))
obs = cm + rc ** .5 * torch.randn(fwd.final_shape, device="cuda")
ld = logdensity.FieldLevelLogDensity(fwd, obs, lat, fixed, precond=precond)
for label, point in (("truth", truth), ("offset", dict(truth, **{"b1_": 30.0, "sigma8_": -8.0, "Omega_m_": 10.0, "b2_": 5.0,
                                                                  "white_mesh_": 0.7 * truth["white_mesh_"]}))):
    lp, g = ld.logdensity_and_grad(point)
    print(f"{label}: lp {lp:.2f}", flush=True)
    for k in lat:
        h = 0.5
        fd = (ld(dict(point, **{k + "_": point[k + "_"] + h})) - ld(dict(point, **{k + "_": point[k + "_"] - h}))) / (2 * h)
        print(f"  d lp / d {k+'_':9s}: analytic {g[k + '_']:14.4f}   central difference (h = {h}) {fd:14.4f}", flush=True)
