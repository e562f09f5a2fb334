"""Small field-level posterior (16^3 final mesh): does the chain's sigma8 posterior cover the truth?  Exploration script behind
tests/test_gpu_samplers.py::test_posterior_of_sigma8_covers_the_truth and ::test_posterior_with_omega_m_covers_the_truth.
usage: python tools/posterior_probe.py [nuts|mclmc] [n_warm] [n_samp] [depth] [final_n=16] [lpt|nbody] [a_obs|lightcone] [om] [seed]"""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
from montecosmo_amd import model, logdensity, samplers, bricks, utils, nbody


def build(nf=16, seed=0, evolution="lpt", a_obs=0.65, sample_om=False):
    ks = np.logspace(-3, 1, 128)
    kpow = (ks, 3.0e4 * (ks / 0.02) / (1 + (ks / 0.02) ** 2.6))
    fwd = model.FieldLevelForward(final_shape=(nf,) * 3, cell_length=40., box_center=(0., 0., 2500.), evolution=evolution,
                                  nbody_n_steps=3, lpt_order=2, a_obs=a_obs, lin_kpow=kpow, nbody_a_start=0.1)
    lat = {"sigma8": dict(loc=0.8102, scale=0.1, loc_fid=0.8102, scale_fid=1e-2, low=0., high=np.inf),
           "b1": dict(loc=1., scale=1e2, loc_fid=1., scale_fid=1e-2)}
    fixed = dict(Omega_m=0.3111, b2=0., bs2=0., bn2=0., b3=0., bds2=0., bs3=0., bnpar=0., ngbars=1e-3, s_e=1.0, s_ed=0., s_e2=0.)
    if sample_om:      # montecosmo/model.py:76-83: truncated normal on [0.05, 1]
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
    return fwd, ld, flat, flat.pack(start), truth


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "nuts"
    nw = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    nsamp = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    depth = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    nf = int(sys.argv[5]) if len(sys.argv) > 5 else 16
    evolution = sys.argv[6] if len(sys.argv) > 6 else "lpt"
    a_obs = None if (len(sys.argv) > 7 and sys.argv[7] == "lightcone") else (float(sys.argv[7]) if len(sys.argv) > 7 else 0.65)
    sample_om = len(sys.argv) > 8 and sys.argv[8] == "om"
    seed = int(sys.argv[9]) if len(sys.argv) > 9 else 0
    fwd, ld, flat, q0, truth = build(nf, seed, evolution, a_obs, sample_om)
    ns = len(flat.scalars)
    print("scalars", flat.scalars, "dimension", q0.numel(), "shapes", fwd.final_shape, fwd.init_shape, fwd.evol_shape, flush=True)
    t0 = time.perf_counter()
    if which == "nuts":
        res = samplers.nuts_sample(flat, q0, n_warmup=nw, n_samples=nsamp, max_tree_depth=depth, seed=1, keep=lambda q: q[:ns].tolist())
    else:
        res = samplers.mclmc_sample(flat, q0, n_warmup=nw, n_samples=nsamp, seed=1, keep=lambda q: q[:ns].tolist())
    wall = time.perf_counter() - t0
    d = np.array(res["samples"])
    inf = res["infos"]
    out = {"sampler": which, "wall_s": round(wall, 1), "n_eval": flat.n_eval, "mean": d.mean(0).round(3).tolist(), "std": d.std(0).round(3).tolist(),
           "step_size": res["step_size"], "lp_truth": ld(truth), "lp_end": inf[-1]["logdensity"]}
    if which == "nuts":
        out.update(mean_depth=float(np.mean([i["depth"] for i in inf[nw:]])), mean_leap=float(np.mean([i["n_leapfrog"] for i in inf[nw:]])),
                   accept=float(np.mean([i["accept_stat"] for i in inf[nw:]])), div=int(sum(i["diverging"] for i in inf)),
                   minv_scalars=res["inverse_mass"][:ns].tolist() if res["inverse_mass"] is not None else None,
                   minv_field_mean=float(res["inverse_mass"][ns:].mean()) if res["inverse_mass"] is not None else None)
    else:
        out.update(L=res["L"], sqrt_d=q0.numel() ** .5)
    print(json.dumps(out), flush=True)
