"""Convergence run of the field-level NUTS sampler: the configuration of BASELINE config 5 (10-step BullFrog
N-body, Lagrangian bias, 'kaiser' preconditioning, the six scalar latents Omega_m, sigma8, b1, b2, bs2, bn2 sampled next to the
initial field) on a problem small enough to CONVERGE in the budget -- a 64^3 evolution mesh (final 36^3) -- with the tree depth
left at the reference's value (blackjax's default max_num_doublings = 10: montecosmo/samplers.py:184-200; window adaptation of a
diagonal mass matrix, samplers.py:44), several independent chains from dispersed starting points, split-R-hat and effective
sample sizes of the scalar latents.  The 256^3 run of tools/run_nuts_field.py stays what it is labelled: a throughput figure.

usage: python tools/run_nuts_convergence.py [final_n=36] [chains: 4 | c2 | 0,1,2,3] [n_warmup=200] [n_samples=200] [max_depth=10] [out.json] [draws_dir]
A gpurun call lasts 20 minutes at most and a chain about ten: with `draws_dir` every chain's draws are kept as <draws_dir>/chain<c>.npz,
chains found there are not run again -- so "c0", "c1", "c2", "c3" and then "0,1,2,3" in five calls give the four-chain summary."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from montecosmo_amd import model, logdensity, samplers, bricks, utils, nbody

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 36
arg = sys.argv[2] if len(sys.argv) > 2 else "4"      # "4": chains 0..3; "c2": chain 2 alone; "0,1,2,3": these chains
chains = [int(arg[1:])] if arg.startswith("c") else ([int(c) for c in arg.split(",")] if "," in arg else list(range(int(arg))))
n_chains = len(chains)
n_warm = int(sys.argv[3]) if len(sys.argv) > 3 else 200
n_samp = int(sys.argv[4]) if len(sys.argv) > 4 else 200
depth = int(sys.argv[5]) if len(sys.argv) > 5 else 10
out_path = sys.argv[6] if len(sys.argv) > 6 else None
draws_dir = sys.argv[7] if len(sys.argv) > 7 else None


def split_rhat(x):
    """Split R-hat (Gelman et al. 2013, BDA3 11.4) of draws x[chain, draw]."""
    c, n = x.shape
    h = n // 2
    s = np.concatenate([x[:, :h], x[:, h:2 * h]], axis=0)
    m, n2 = s.shape
    W = s.var(axis=1, ddof=1).mean()
    B = n2 * s.mean(axis=1).var(ddof=1)
    return float(np.sqrt(((n2 - 1) / n2 * W + B / n2) / W))


def ess(x):
    """Effective sample size of x[chain, draw]: multi-chain autocorrelation estimate with Geyer's initial positive sequence
    (Stan reference manual 16.4)."""
    c, n = x.shape
    xc = x - x.mean(axis=1, keepdims=True)
    acov = np.stack([np.correlate(r, r, mode="full")[n - 1:] / n for r in xc])      # biased autocovariance per chain
    W = (acov[:, 0] * n / (n - 1)).mean()
    B_over_n = x.mean(axis=1).var(ddof=1) if c > 1 else 0.0
    var_plus = (n - 1) / n * W + B_over_n
    rho = 1.0 - (W - acov.mean(axis=0) * n / (n - 1)) / var_plus
    tau, t = 1.0, 1
    while t + 1 < n:
        pair = rho[t] + rho[t + 1]
        if pair < 0:
            break
        tau += 2.0 * pair
        t += 2
    return float(c * n / max(tau, 1.0 / np.log10(max(c * n, 10))))


ks = np.logspace(-3, 1, 128)
kpow = (ks, 3.0e4 * (ks / 0.02) / (1 + (ks / 0.02) ** 2.6))
fwd = model.FieldLevelForward(final_shape=(nf,) * 3, cell_length=10. * 146 / nf, box_center=(0., 0., 2500.), evolution="nbody",
                              nbody_n_steps=10, a_obs=0.7, lin_kpow=kpow)       # the config-5 box (1460 Mpc/h), coarser cells
print("shapes: final", fwd.final_shape, "init", fwd.init_shape, "evol", fwd.evol_shape, "paint", fwd.paint_shape, flush=True)
lat = {"Omega_m": dict(loc=0.3111, scale=0.1, loc_fid=0.3111, scale_fid=1e-2, low=0.05, high=1.),       # model.py:76-83
       "sigma8": dict(loc=0.8102, scale=0.1, loc_fid=0.8102, scale_fid=1e-2, low=0., high=np.inf),       # model.py:100-111
       "b1": dict(loc=1., scale=1e2, loc_fid=1., scale_fid=1e-2), "b2": dict(loc=0., scale=1e2, loc_fid=0., scale_fid=3e-2),
       "bs2": dict(loc=0., scale=1e2, loc_fid=0., scale_fid=1e-1), "bn2": dict(loc=0., scale=1e3, loc_fid=0., scale_fid=1.)}
fixed = dict(b3=0., bds2=0., bs3=0., bnpar=0., ngbars=1e-3, s_e=1.0, s_ed=0., s_e2=0.)
gen = torch.Generator(device="cuda").manual_seed(0)
ld0 = logdensity.FieldLevelLogDensity(fwd, torch.zeros(fwd.final_shape), lat, fixed, precond="kaiser")
prior_std = ld0.scale
truth = {k + "_": 0.0 for k in lat}
truth["white_mesh_"] = torch.randn(fwd.init_shape, device="cuda", generator=gen) * prior_std
base = ld0.base_params(truth)
gxy = fwd.evolve(ld0.make_cosmo(base), {k: base[k] for k in bricks.BIAS_KEYS}, utils.rg2cgh(truth["white_mesh_"]) * ld0.transfer)
rc = fixed["ngbars"] * fwd.cell_length ** 3
cm = rc * nbody.irfftn(utils.chreshape(nbody.rfftn(gxy), utils.r2chshape(fwd.final_shape)))
obs = cm + rc ** .5 * torch.randn(fwd.final_shape, device="cuda", generator=gen)
print(f"truth: mean count {float(cm.mean()):.2f}, count contrast std {float((cm / rc - 1).std()):.3f}", flush=True)
ld = logdensity.FieldLevelLogDensity(fwd, obs, lat, fixed, precond="kaiser")
flat = samplers.FlatLogDensity(ld)
ns = len(flat.scalars)
lp_truth = ld(truth)

draws, stats = [], []
t_all = time.perf_counter()
import os
for c in chains:
    saved = os.path.join(draws_dir, f"chain{c}.npz") if draws_dir else None
    if saved and os.path.exists(saved):
        z = np.load(saved, allow_pickle=True)
        draws.append(z["draws"])
        stats.append(json.loads(str(z["stats"])))
        continue
    g = torch.Generator(device="cuda").manual_seed(100 + c)
    start = {k + "_": float(2.0 * torch.randn(1, generator=g, device="cuda")) for k in lat}      # dispersed: 2 fiducial scales
    start["white_mesh_"] = 0.3 * torch.randn(fwd.init_shape, device="cuda", generator=g) * prior_std
    q0 = flat.pack(start)
    n0, t0 = flat.n_eval, time.perf_counter()
    res = samplers.nuts_sample(flat, q0, n_warmup=n_warm, n_samples=n_samp, max_tree_depth=depth, seed=1 + c,
                               keep=lambda q: q[:ns].tolist() + [float((q[ns:].reshape(fwd.init_shape) / prior_std).std())])
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    inf = res["infos"]
    d = np.array(res["samples"])
    draws.append(d)
    st = {"chain": c, "wall_s": round(wall, 1), "gradients": flat.n_eval - n0, "ms_per_gradient": round(1e3 * wall / (flat.n_eval - n0), 3),
          "step_size": res["step_size"], "mean_depth": float(np.mean([i["depth"] for i in inf[n_warm:]])),
          "mean_leapfrogs": float(np.mean([i["n_leapfrog"] for i in inf[n_warm:]])),
          "at_depth_cap": int(sum(i["depth"] >= depth for i in inf[n_warm:])),
          "accept": float(np.mean([i["accept_stat"] for i in inf[n_warm:]])),
          "divergences_warmup": int(sum(i["diverging"] for i in inf[:n_warm])), "divergences_sampling": int(sum(i["diverging"] for i in inf[n_warm:])),
          "logdensity_end": inf[-1]["logdensity"], "mean": d.mean(0).round(3).tolist()}
    stats.append(st)
    print(json.dumps(st), flush=True)
    if saved:
        os.makedirs(draws_dir, exist_ok=True)
        np.savez(saved, draws=d, stats=json.dumps(st))

x = np.stack(draws)                      # (chains, draws, scalars + 1)
names = flat.scalars + ["white_std"]
summary = {"final_shape": fwd.final_shape, "evol_shape": fwd.evol_shape, "dimension": int(ns + np.prod(fwd.init_shape)), "n_chains": n_chains, "n_warmup": n_warm,
           "n_samples": n_samp, "max_tree_depth": depth, "wall_s_total": round(time.perf_counter() - t_all, 1), "logdensity_truth": lp_truth,
           "truth_sample_space": 0.0,
           "scalars": {n: {"mean": float(x[:, :, i].mean()), "std": float(x[:, :, i].std()), "split_rhat": round(split_rhat(x[:, :, i]), 3),
                           "ess": round(ess(x[:, :, i]), 1),
                           "truth_within_3_sigma": bool(abs(x[:, :, i].mean() - (1.0 if n == "white_std" else 0.0)) < 3 * x[:, :, i].std())}
                       for i, n in enumerate(names)},
           "chains": stats}
print(json.dumps(summary), flush=True)
if out_path:
    open(out_path, "w").write(json.dumps(summary, indent=1))
