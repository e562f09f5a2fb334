"""The call a field-level NUTS chain of BASELINE config 5 times -- log density + gradient of the field-level model at a 256^3 evolution
mesh (final 146^3, 'kaiser' preconditioning, 10-step BullFrog; the problem of tests/test_gpu_config5.py and tools/run_nuts_field.py) --
repeated, for rocprofv3 --kernel-trace --stats and for a host-side wall clock.  usage: python tools/profile_config5.py [evolution=nbody] [reps=20]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from montecosmo_amd import model, logdensity, bricks, utils, nbody

evolution = sys.argv[1] if len(sys.argv) > 1 else "nbody"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
LAT = {"Omega_m": dict(loc=0.3111, scale=0.1, loc_fid=0.3111, scale_fid=1e-2), "sigma8": dict(loc=0.8102, scale=0.1, loc_fid=0.8102, scale_fid=1e-2),
       "b1": dict(loc=1., scale=1e2, loc_fid=1., scale_fid=1e-2), "b2": dict(loc=0., scale=1e2, loc_fid=0., scale_fid=3e-2),
       "bs2": dict(loc=0., scale=1e2, loc_fid=0., scale_fid=1e-1), "bn2": dict(loc=0., scale=1e3, loc_fid=0., scale_fid=1.)}
FIXED = dict(b3=0., bds2=0., bs3=0., bnpar=0., ngbars=1e-3, s_e=1.0, s_ed=0., s_e2=0.)
ks = np.logspace(-3, 1, 128)
kpow = (ks, 3.0e4 * (ks / 0.02) / (1 + (ks / 0.02) ** 2.6))
fwd = model.FieldLevelForward(final_shape=(146,) * 3, cell_length=10., box_center=(0., 0., 2500.), evolution=evolution, nbody_n_steps=10,
                              a_obs=0.7 if evolution == "nbody" else None, lin_kpow=kpow)
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
point = dict(truth, **{"b1_": 30.0, "sigma8_": -8.0, "Omega_m_": 10.0, "b2_": 5.0, "white_mesh_": 0.7 * truth["white_mesh_"]})
for _ in range(3):
    ld.logdensity_and_grad(point)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    lp, g = ld.logdensity_and_grad(point)
torch.cuda.synchronize()
print(f"{evolution}: {1e3 * (time.perf_counter() - t0) / reps:.2f} ms per log density + gradient ({reps} calls), lp {lp:.2f}", flush=True)
