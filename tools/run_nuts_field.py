"""End-to-end field-level NUTS on one MI355X (BASELINE config 5 in miniature or at size): synthetic truth -> observed
counts -> `FieldLevelLogDensity` -> `samplers.nuts_sample`.
usage: python tools/run_nuts_field.py [final_n=146] [n_warmup=200] [n_samples=200] [max_depth=6] [evolution=nbody] [out.json] [nuts|mclmc] [precond=fourier]"""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from montecosmo_amd import model, logdensity, samplers, bricks, utils, nbody

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 146
n_warm = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n_samp = int(sys.argv[3]) if len(sys.argv) > 3 else 200
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 6
evolution = sys.argv[5] if len(sys.argv) > 5 else "nbody"
out_path = sys.argv[6] if len(sys.argv) > 6 else None
precond = sys.argv[8] if len(sys.argv) > 8 else "fourier"

ks = np.logspace(-3, 1, 128)
kpow = (ks, 3.0e4 * (ks / 0.02) / (1 + (ks / 0.02) ** 2.6))
fwd = model.FieldLevelForward(final_shape=(nf,) * 3, cell_length=10., box_center=(0., 0., 2500.), evolution=evolution,
                              nbody_n_steps=10, a_obs=0.7, lin_kpow=kpow)
print("shapes: final", fwd.final_shape, "init", fwd.init_shape, "evol", fwd.evol_shape, "paint", fwd.paint_shape, flush=True)
lat = {"Omega_m": dict(loc=0.3111, scale=0.1, loc_fid=0.3111, scale_fid=1e-2),
       "sigma8": dict(loc=0.8102, scale=0.1, loc_fid=0.8102, scale_fid=1e-2),
       "b1": dict(loc=1., scale=1e2, loc_fid=1., scale_fid=1e-2), "b2": dict(loc=0., scale=1e2, loc_fid=0., scale_fid=3e-2),
       "bs2": dict(loc=0., scale=1e2, loc_fid=0., scale_fid=1e-1), "bn2": dict(loc=0., scale=1e3, loc_fid=0., scale_fid=1.)}
fixed = dict(b3=0., bds2=0., bs3=0., bnpar=0., ngbars=1e-3, s_e=1.0, s_ed=0., s_e2=0.)
torch.manual_seed(0)
truth = {k + "_": 0.0 for k in lat}
truth["white_mesh_"] = torch.randn(fwd.init_shape, device="cuda")
# observed counts: the model's own mean at the truth + its Gaussian noise (model.py:893-908 with s_ed = s_e2 = 0)
ld0 = logdensity.FieldLevelLogDensity(fwd, torch.zeros(fwd.final_shape), lat, fixed, precond=precond)
prior_std = 1.0 if ld0.scale is None else ld0.scale          # white_mesh_ ~ Normal(0, scale): the truth is a prior draw
truth["white_mesh_"] = truth["white_mesh_"] * prior_std
base = ld0.base_params(truth)
white = utils.rg2cgh(truth["white_mesh_"]) * ld0.transfer
gxy = fwd.evolve(ld0.make_cosmo(base), {k: base[k] for k in bricks.BIAS_KEYS}, white)
rc = fixed["ngbars"] * fwd.cell_length ** 3
cm = rc * nbody.irfftn(utils.chreshape(nbody.rfftn(gxy), utils.r2chshape(fwd.final_shape)))
obs = cm + rc ** .5 * torch.randn(fwd.final_shape, device="cuda")
print(f"truth: mean count {float(cm.mean()):.3f}, count contrast std {float((cm / rc - 1).std()):.3f}", flush=True)
ld = logdensity.FieldLevelLogDensity(fwd, obs, lat, fixed, precond=precond)
flat = samplers.FlatLogDensity(ld)
start = dict(truth)
start["white_mesh_"] = 0.1 * torch.randn(fwd.init_shape, device="cuda") * prior_std     # away from the truth, near the prior mode
q0 = flat.pack(start)
lp_truth = ld(truth)
t0 = time.perf_counter()
lp0, g0 = flat(q0)
torch.cuda.synchronize()
print(f"dimension {q0.numel()}, log density at start {lp0:.1f}, at truth {lp_truth:.1f}, one gradient {1e3 * (time.perf_counter() - t0):.0f} ms (first call)", flush=True)

def cb(it, info):
    if it % 10 == 0 or it == n_warm + n_samp - 1:
        print(f"iter {it:4d} {'warm' if info['warmup'] else 'samp'} lp {info['logdensity']:.1f} eps {info['step_size']:.4f} "
              f"leapfrogs {info['n_leapfrog']:3d} accept {info['accept_stat']:.2f} div {info['diverging']} "
              f"[{time.perf_counter() - t_run:.0f} s]", flush=True)

ns = len(flat.scalars)
t_run = time.perf_counter()
sampler = sys.argv[7] if len(sys.argv) > 7 else "nuts"
if sampler == "mclmc":       # montecosmo/samplers.py:273-398 (the production drivers' sampler); 2 gradients per transition
    def cb(it, info):
        if it % 50 == 0 or it == n_warm + n_samp - 1:
            print(f"iter {it:4d} {'warm' if info['warmup'] else 'samp'} lp {info['logdensity']:.1f} eps {info['step_size']:.4f} "
                  f"mse/dim {info['mse_per_dim']:.2e} [{time.perf_counter() - t_run:.0f} s]", flush=True)
    res = samplers.mclmc_sample(flat, q0, n_warmup=n_warm, n_samples=n_samp, seed=1, callback=cb,
                                keep=lambda q: q[:ns].tolist() + [float(q[ns:].std())])
    for i in res["infos"]:
        i.update(n_leapfrog=2, accept_stat=float("nan"), diverging=False)
else:
    res = samplers.nuts_sample(flat, q0, n_warmup=n_warm, n_samples=n_samp, max_tree_depth=depth, seed=1, callback=cb,
                               keep=lambda q: q[:ns].tolist() + [float(q[ns:].std())])
torch.cuda.synchronize()
wall = time.perf_counter() - t_run
infos = res["infos"]
draws = np.array(res["samples"])


def field_correlation(q_end):
    """Cross-correlation of the last state's initial field with the truth's, over all modes and over |k| < k_Nyquist / 4
    (in units of the prior width of each mode, i.e. of white_mesh_ / scale)."""
    w_end = q_end[ns:].reshape(fwd.init_shape) / prior_std
    w_tr = truth["white_mesh_"] / prior_std
    a, b = utils.rg2cgh(w_end), utils.rg2cgh(w_tr)
    kv = nbody.rfftk(fwd.init_shape)
    low = torch.from_numpy((sum(k ** 2 for k in kv) ** .5 < np.pi / 4)).to(a.device)
    cc = lambda m: float(((a.conj() * b).real * m).sum() / (((a.abs() ** 2) * m).sum() * ((b.abs() ** 2) * m).sum()).sqrt())
    return {"all_modes": round(cc(torch.ones_like(low)), 4), "k_below_quarter_nyquist": round(cc(low), 4)}


corr = field_correlation(res["last_state"]["q"])
print("correlation of the last state's initial field with the truth:", corr, flush=True)
summary = {"sampler": sampler, "precond": precond, "final_shape": fwd.final_shape, "evol_shape": fwd.evol_shape, "evolution": evolution, "dimension": int(q0.numel()),
           "n_warmup": n_warm, "n_samples": n_samp, "max_tree_depth": depth, "wall_s": round(wall, 1),
           "gradient_evals": flat.n_eval, "ms_per_gradient": round(1e3 * wall / max(flat.n_eval - 1, 1), 2),
           "mean_leapfrogs": float(np.mean([i["n_leapfrog"] for i in infos])), "step_size": res["step_size"],
           "accept_stat_sampling": float(np.mean([i["accept_stat"] for i in infos[n_warm:]])) if n_samp else None,
           "divergences": int(sum(i["diverging"] for i in infos)), "logdensity_start": lp0, "logdensity_truth": lp_truth,
           "logdensity_end": infos[-1]["logdensity"], "field_correlation_with_truth": corr,
           "posterior_mean_sample_space": dict(zip(flat.scalars + ["white_std"], draws.mean(0).round(4).tolist())) if n_samp else None}
print(json.dumps(summary), flush=True)
if out_path:
    open(out_path, "w").write(json.dumps(summary, indent=1))
