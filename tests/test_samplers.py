"""The host NUTS loop (montecosmo_amd/samplers.py) on targets with known answers (CPU tensors)."""
import math
import numpy as np
import pytest
import torch

from montecosmo_amd import samplers


def test_dual_averaging_converges_to_target():
    da = samplers.DualAveraging(1.0, target=0.8)
    eps = 1.0
    for _ in range(400):                      # acceptance falls with the step size: a = exp(-eps)
        eps = da.update(math.exp(-eps))
    assert abs(math.exp(-da.final()) - 0.8) < 0.03


def test_nuts_recovers_gaussian_moments():
    torch.manual_seed(0)
    d = 40
    sd = torch.linspace(0.5, 2.0, d, dtype=torch.float64)
    mu = torch.linspace(-1.0, 1.0, d, dtype=torch.float64)

    def fn(q):
        z = (q - mu) / sd
        return float(-0.5 * (z * z).sum()), -(z / sd)

    out = samplers.nuts_sample(fn, torch.zeros(d, dtype=torch.float64), n_warmup=300, n_samples=1500, seed=3)
    x = torch.stack(out["samples"]).numpy()
    acc = np.mean([i["accept_stat"] for i in out["infos"][300:]])
    assert 0.6 < acc < 0.95 and not any(i["diverging"] for i in out["infos"][300:])
    se = sd.numpy() / np.sqrt(150)            # generous effective-sample-size allowance
    assert np.all(np.abs(x.mean(0) - mu.numpy()) < 5 * se)
    assert np.all(np.abs(x.std(0) / sd.numpy() - 1) < 0.2)
    assert all(i["n_leapfrog"] <= 2 ** 10 for i in out["infos"])


def test_nuts_is_reproducible_and_respects_depth():
    fn = lambda q: (float(-0.5 * (q * q).sum()), -q)
    a = samplers.nuts_sample(fn, torch.ones(5, dtype=torch.float64), n_warmup=20, n_samples=20, seed=1, max_tree_depth=3)
    b = samplers.nuts_sample(fn, torch.ones(5, dtype=torch.float64), n_warmup=20, n_samples=20, seed=1, max_tree_depth=3)
    assert all(torch.equal(x, y) for x, y in zip(a["samples"], b["samples"]))
    assert all(i["n_leapfrog"] <= 2 ** 3 - 1 + 2 ** 3 for i in a["infos"])


def test_mclmc_recovers_gaussian_moments():
    d = 100
    sd = torch.linspace(0.5, 2.0, d, dtype=torch.float64)

    def fn(q):
        z = q / sd
        return float(-0.5 * (z * z).sum()), -(z / sd)

    out = samplers.mclmc_sample(fn, torch.ones(d, dtype=torch.float64), n_warmup=2000, n_samples=20000, seed=5, L=10.0)
    x = torch.stack(out["samples"]).numpy()
    mse = np.nanmean([i["mse_per_dim"] for i in out["infos"][2000:]])
    assert 5e-5 < mse < 5e-3                              # the step size was tuned to the energy-error target
    assert np.all(np.abs(x.std(0) / sd.numpy() - 1) < 0.25)
    assert np.all(np.abs(x.mean(0)) < 0.5 * sd.numpy())


@pytest.mark.parametrize("which", ["nuts", "mclmc"])
def test_chain_save_and_resume_continue_exactly(tmp_path, which):
    """montecosmo/samplers.py:596-660 (`save_run`, `sample_and_save`): runs are written one .npz each plus the last
    state; a chain resumed from the saved state repeats, bit for bit, what the uninterrupted chain draws."""
    from montecosmo_amd import samplers
    scale = torch.tensor([1.0, 0.5, 2.0, 1.5], dtype=torch.float64)

    def logdf(q):
        return float(-0.5 * ((q / scale) ** 2).sum()), -q / scale ** 2

    sampler = samplers.nuts_sample if which == "nuts" else samplers.mclmc_sample
    q0 = torch.zeros(4, dtype=torch.float64)
    path = str(tmp_path / "chain")
    whole = sampler(logdf, q0, n_warmup=30, n_samples=40, seed=3)
    first = sampler(logdf, q0, n_warmup=30, n_samples=15, seed=3)
    samplers.save_run(first, 0, path)
    rest = sampler(logdf, None, n_warmup=0, n_samples=25, state=samplers.load_state(path))
    samplers.save_run(rest, 1, path)
    a = np.load(path + "_0.npz")
    b = np.load(path + "_1.npz")
    joined = np.concatenate([a["samples"], b["samples"]])
    assert joined.shape == (40, 4) and a["warmup"].sum() == 30 and not b["warmup"].any()
    assert np.array_equal(joined, torch.stack(whole["samples"]).numpy())
    assert "n_evals" in a.files and len(a["n_evals"]) == 45
    # the driver: warm-up run, two more runs, then a resumed fourth run
    res = samplers.sample_and_save(sampler, logdf, q0, str(tmp_path / "drv"), start=0, end=2, n_warmup=20, n_samples=10, seed=1)
    res2 = samplers.sample_and_save(sampler, logdf, q0, str(tmp_path / "drv"), start=3, end=3, n_samples=10, resume=True)
    assert all((tmp_path / f"drv_{i}.npz").exists() for i in range(4))
    assert res2["step_size"] == res["step_size"] and not any(i["warmup"] for i in res2["infos"])


def test_flat_log_density_packs_per_shell_latents():
    """`FlatLogDensity` lays the scalar latents, the per-shell 'ngbars_' array and the white mesh out in one vector and
    routes the gradient back in the same order."""
    class Fwd:
        init_shape = (2, 2, 2)

    class Stub:
        fwd, n_rbins = Fwd(), 3

        def names(self):
            return ["b1_", "ngbars_", "s_e_", "white_mesh_"]

        def logdensity_and_grad(self, s):
            assert isinstance(s["b1_"], float) and len(s["ngbars_"]) == 3 and tuple(s["white_mesh_"].shape) == (2, 2, 2)
            lp = -0.5 * (s["b1_"] ** 2 + sum(v ** 2 for v in s["ngbars_"]) + s["s_e_"] ** 2 + float((s["white_mesh_"] ** 2).sum()))
            return lp, {"b1_": -s["b1_"], "ngbars_": [-v for v in s["ngbars_"]], "s_e_": -s["s_e_"], "white_mesh_": -s["white_mesh_"]}

    flat = samplers.FlatLogDensity(Stub())
    import montecosmo_amd.nbody as nbody
    nbody_f32 = nbody._f32
    nbody._f32 = lambda x, shape=None: torch.as_tensor(np.asarray(x), dtype=torch.float32).reshape(shape)   # CPU stand-in
    try:
        q = flat.pack({"b1_": 1.0, "ngbars_": [2.0, 3.0, 4.0], "s_e_": 5.0, "white_mesh_": np.arange(8.).reshape(2, 2, 2)})
    finally:
        nbody._f32 = nbody_f32
    assert q.tolist() == [1.0, 2.0, 3.0, 4.0, 5.0] + list(np.arange(8.))
    lp, g = flat(q)
    assert np.isclose(lp, -0.5 * float((q ** 2).sum())) and torch.allclose(g, -q)
    assert flat.unpack(q)["ngbars_"] == [2.0, 3.0, 4.0]


def test_warmup_windows_follow_stans_schedule():
    """`blackjax.window_adaptation` (reference samplers.py:44) builds Stan's schedule: 75 fast, slow windows 25, 50, 100, ...
    (the last one stretched to the terminal buffer), 50 fast."""
    assert samplers.warmup_windows(1000) == [(75, 100), (100, 150), (150, 250), (250, 450), (450, 950)]
    assert samplers.warmup_windows(200) == [(75, 100), (100, 150)]
    assert samplers.warmup_windows(100) == [(15, 90)]              # short warm-up: 15 % / 75 % / 10 %
    assert samplers.warmup_windows(10) == []


def test_nuts_window_adaptation_learns_the_scales():
    """A Gaussian whose standard deviations span 1e-2 .. 1e1: the adapted diagonal metric recovers the variances, the draws have
    the right moments, and trajectories are an order of magnitude shorter than with the identity metric."""
    d = 30
    sd = torch.logspace(-2, 1, d, dtype=torch.float64)

    def fn(q):
        z = q / sd
        return float(-0.5 * (z * z).sum()), -(z / sd)

    q0 = 0.5 * sd
    out = samplers.nuts_sample(fn, q0, n_warmup=400, n_samples=600, seed=2)
    ratio = (out["inverse_mass"] / sd ** 2).numpy()
    assert np.all(ratio > 0.4) and np.all(ratio < 2.5), ratio
    x = torch.stack(out["samples"]).numpy()
    assert np.all(np.abs(x.std(0) / sd.numpy() - 1) < 0.2)
    assert np.all(np.abs(x.mean(0)) < 0.3 * sd.numpy())
    plain = samplers.nuts_sample(fn, q0, n_warmup=400, n_samples=100, seed=2, adapt_mass=False)
    leap = lambda o, a: np.mean([i["n_leapfrog"] for i in o["infos"][a:]])
    assert leap(out, 400) < 16 and leap(plain, 400) > 8 * leap(out, 400), (leap(out, 400), leap(plain, 400))
    # a resumed chain keeps the metric
    st = out["last_state"]
    more = samplers.nuts_sample(fn, None, n_warmup=0, n_samples=5, state=st)
    assert torch.equal(more["inverse_mass"], out["inverse_mass"])


def test_mclmc_tunes_L_to_the_size_of_the_typical_set():
    """Second warm-up stage of mclmc_find_L_and_step_size (reference samplers.py:322-331): L = sqrt(sum of position variances)."""
    d = 100
    sd = torch.linspace(0.5, 2.0, d, dtype=torch.float64)

    def fn(q):
        z = q / sd
        return float(-0.5 * (z * z).sum()), -(z / sd)

    out = samplers.mclmc_sample(fn, torch.ones(d, dtype=torch.float64), n_warmup=6000, n_samples=10, seed=5)
    want = float(torch.sqrt((sd ** 2).sum()))
    assert 0.6 * want < out["L"] < 1.5 * want, (out["L"], want)
    assert out["infos"][0]["L"] == math.sqrt(d) and out["infos"][-1]["L"] == out["L"]
    fixed = samplers.mclmc_sample(fn, torch.ones(d, dtype=torch.float64), n_warmup=200, n_samples=10, seed=5, L=7.0)
    assert fixed["L"] == 7.0
