"""Host sampler loop for the field-level log density (the role of montecosmo/samplers.py:17-269, which drives blackjax
NUTS): a from-scratch multinomial No-U-Turn sampler (Hoffman & Gelman 2014; Betancourt 2017) with dual-averaging
step-size adaptation, operating on one flat tensor (device or host) and a callable returning (log density, gradient).
Third-party algorithm: what is montecosmo's is the call contract `logdensity_fn(position) -> scalar`, kept here as
`logdensity_and_grad(q) -> (float, tensor)`.  Warm-up follows `blackjax.window_adaptation` (samplers.py:44: Stan's windowed
scheme): dual averaging of the step size throughout, and a DIAGONAL mass matrix estimated over doubling slow windows
(`adapt_mass=True`, the default; identity metric otherwise -- the model's sample space is already roughly standardised by
`scale_fid` and the 'kaiser' preconditioning, so the estimate mostly corrects the scalar latents).  MCLMC tunes its step
size to the energy-error target and, in the second half of the warm-up, L from the running variances of the position
(`mclmc_find_L_and_step_size` as the reference calls it, samplers.py:322-331: frac_tune1 = frac_tune2 = 0.5, no
ESS-based third stage)."""
from __future__ import annotations

import math
import time

import torch


class DualAveraging:
    """Nesterov dual averaging of log(step size) towards a target acceptance statistic (Hoffman & Gelman 2014, alg. 5)."""

    def __init__(self, eps0, target=0.8, gamma=0.05, t0=10.0, kappa=0.75):
        self.mu = math.log(10.0 * eps0)
        self.target, self.gamma, self.t0, self.kappa = target, gamma, t0, kappa
        self.hbar, self.log_eps_bar, self.t = 0.0, 0.0, 0
        self.eps = eps0

    def update(self, accept_stat):
        self.t += 1
        eta = 1.0 / (self.t + self.t0)
        self.hbar = (1 - eta) * self.hbar + eta * (self.target - accept_stat)
        log_eps = self.mu - math.sqrt(self.t) / self.gamma * self.hbar
        w = self.t ** (-self.kappa)
        self.log_eps_bar = w * log_eps + (1 - w) * self.log_eps_bar
        self.eps = math.exp(log_eps)
        return self.eps

    def final(self):
        return math.exp(self.log_eps_bar)


class _State:
    __slots__ = ("q", "p", "lp", "g")

    def __init__(self, q, p, lp, g):
        self.q, self.p, self.lp, self.g = q, p, lp, g


def _leapfrog(fn, s, eps, minv=None):
    p = s.p + (0.5 * eps) * s.g
    q = s.q + eps * (p if minv is None else minv * p)
    lp, g = fn(q)
    p = p + (0.5 * eps) * g
    return _State(q, p, lp, g)


def _energy(s, minv=None):
    return -s.lp + 0.5 * float(torch.dot(s.p, s.p) if minv is None else torch.dot(s.p, minv * s.p))


def _logaddexp(a, b):
    if a == -math.inf:
        return b
    if b == -math.inf:
        return a
    m = max(a, b)
    return m + math.log(math.exp(a - m) + math.exp(b - m))


def _build_tree(fn, edge, direction, depth, eps, H0, rng, max_delta=1000.0, minv=None):
    """Recursively doubles from `edge` in `direction`.  Returns (new edge, proposal, log_w, rho, n_leapfrog,
    sum_accept, diverging, turning) where log_w = log sum exp(-H) over the subtree and rho its summed momentum."""
    if depth == 0:
        s = _leapfrog(fn, edge, direction * eps, minv)
        H = _energy(s, minv)
        if not math.isfinite(H):
            H = math.inf
        diverging = (H - H0) > max_delta
        acc = min(1.0, math.exp(min(0.0, H0 - H))) if math.isfinite(H) else 0.0
        return s, s, (-H if math.isfinite(H) else -math.inf), s.p.clone(), 1, acc, diverging, False, s
    e1, prop1, lw1, rho1, n1, a1, div1, turn1, first1 = _build_tree(fn, edge, direction, depth - 1, eps, H0, rng, max_delta, minv)
    if div1 or turn1:
        return e1, prop1, lw1, rho1, n1, a1, div1, turn1, first1
    e2, prop2, lw2, rho2, n2, a2, div2, turn2, _ = _build_tree(fn, e1, direction, depth - 1, eps, H0, rng, max_delta, minv)
    lw = _logaddexp(lw1, lw2)
    prop = prop1
    if not (div2 or turn2) and lw2 > -math.inf:
        if math.log(max(rng.random(), 1e-300)) < lw2 - lw:      # multinomial sampling within the subtree
            prop = prop2
    rho = rho1 + rho2
    # generalised no-U-turn criterion between the two ends of this subtree (p_sharp = M^-1 p)
    p_first, p_last = first1.p, e2.p
    if direction < 0:
        p_first, p_last = p_last, p_first
    rs = rho if minv is None else minv * rho
    turning = turn2 or (float(torch.dot(rs, p_first)) <= 0.0) or (float(torch.dot(rs, p_last)) <= 0.0)
    return e2, prop, lw, rho, n1 + n2, a1 + a2, div2, turning, first1


def nuts_step(fn, q, lp, g, eps, rng, max_tree_depth=10, minv=None):
    """One NUTS transition; `minv`: diagonal inverse mass matrix (None = identity).  Returns (q, lp, g, info)."""
    p0 = torch.randn(q.shape, dtype=q.dtype, device=q.device, generator=rng.torch_gen(q.device))
    if minv is not None:
        p0 = p0 / torch.sqrt(minv)                      # p ~ N(0, M)
    start = _State(q, p0, lp, g)
    H0 = _energy(start, minv)
    left = right = start
    sample = start
    log_w = -H0
    rho = p0.clone()
    n_leap, sum_acc, depth, diverging = 0, 0.0, 0, False
    while depth < max_tree_depth:
        direction = 1 if rng.random() < 0.5 else -1
        edge = right if direction > 0 else left
        new_edge, prop, lw_sub, rho_sub, n, acc, div, turn, _ = _build_tree(fn, edge, direction, depth, eps, H0, rng, minv=minv)
        n_leap += n
        sum_acc += acc
        if div:
            diverging = True
            break
        if turn:
            break
        if lw_sub > -math.inf and math.log(max(rng.random(), 1e-300)) < lw_sub - log_w:   # biased progressive sampling
            sample = prop
        log_w = _logaddexp(log_w, lw_sub)
        rho = rho + rho_sub
        if direction > 0:
            right = new_edge
        else:
            left = new_edge
        depth += 1
        rs = rho if minv is None else minv * rho
        if float(torch.dot(rs, left.p)) <= 0.0 or float(torch.dot(rs, right.p)) <= 0.0:
            break
    info = {"n_leapfrog": n_leap, "accept_stat": sum_acc / max(n_leap, 1), "depth": depth, "diverging": diverging,
            "energy": H0}
    return sample.q, sample.lp, sample.g, info


class _Rng:
    def __init__(self, seed):
        import random
        self._r = random.Random(seed)
        self._seed = seed
        self._gens = {}

    def random(self):
        return self._r.random()

    def torch_gen(self, device):
        key = str(device)
        if key not in self._gens:
            g = torch.Generator(device=device)
            g.manual_seed(self._seed + 12345)
            self._gens[key] = g
        return self._gens[key]

    def get_state(self):
        """Everything needed to continue the streams exactly (a chain resumed from a saved state repeats the draws an
        uninterrupted chain would have made)."""
        return {"seed": self._seed, "py": self._r.getstate(), "gens": {k: g.get_state() for k, g in self._gens.items()}}

    @classmethod
    def from_state(cls, st):
        r = cls(st["seed"])
        r._r.setstate(st["py"])
        for k, gs in st["gens"].items():
            r.torch_gen(torch.device(k)).set_state(gs)
        return r


def find_reasonable_step_size(fn, q, lp, g, rng, eps=1.0, minv=None):
    """Heuristic of Hoffman & Gelman 2014, alg. 4: double / halve until the one-step acceptance crosses 1/2."""
    p = torch.randn(q.shape, dtype=q.dtype, device=q.device, generator=rng.torch_gen(q.device))
    if minv is not None:
        p = p / torch.sqrt(minv)
    s0 = _State(q, p, lp, g)
    H0 = _energy(s0, minv)
    s1 = _leapfrog(fn, s0, eps, minv)
    dH = H0 - _energy(s1, minv)
    direction = 1.0 if (math.isfinite(dH) and dH > math.log(0.5)) else -1.0
    for _ in range(50):
        eps *= 2.0 ** direction
        s1 = _leapfrog(fn, s0, eps, minv)
        dH = H0 - _energy(s1, minv)
        ok = math.isfinite(dH) and dH > math.log(0.5)
        if (direction > 0 and not ok) or (direction < 0 and ok):
            break
    return eps


def warmup_windows(n_warmup, init_buffer=75, term_buffer=50, base_window=25):
    """Stan's warm-up schedule (what `blackjax.window_adaptation` builds): [fast initial buffer | slow windows, each twice the
    last, the final one stretched to the terminal buffer | fast terminal buffer].  Returns the list of (start, end) of the slow
    windows (the mass matrix is re-estimated at every `end`); short warm-ups use the 15 % / 75 % / 10 % split."""
    if n_warmup < 20:
        return []
    if init_buffer + base_window + term_buffer > n_warmup:
        init_buffer, term_buffer = int(0.15 * n_warmup), int(0.10 * n_warmup)
        base_window = n_warmup - init_buffer - term_buffer
    wins, start, size, slow_end = [], init_buffer, base_window, n_warmup - term_buffer
    while start < slow_end:
        end = start + size
        if end + 2 * size > slow_end:          # the next window would not fit: this one runs to the terminal buffer
            end = slow_end
        wins.append((start, end))
        start, size = end, 2 * size
    return wins


class _Welford:
    """Running mean / variance of the chain's positions on their own device (two vectors, whatever the dimension)."""

    def __init__(self):
        self.n, self.mean, self.m2 = 0, None, None

    def add(self, q):
        q = q.double()
        if self.mean is None:
            self.mean, self.m2 = torch.zeros_like(q), torch.zeros_like(q)
        self.n += 1
        d = q - self.mean
        self.mean += d / self.n
        self.m2 += d * (q - self.mean)

    def inverse_mass(self, dtype):
        """Stan's regularised variance estimate: (n / (n + 5)) var + 1e-3 (5 / (n + 5))."""
        var = self.m2 / max(self.n - 1, 1)
        return ((self.n / (self.n + 5.0)) * var + 1e-3 * (5.0 / (self.n + 5.0))).to(dtype)


def nuts_sample(logdensity_and_grad, q0, n_warmup=200, n_samples=200, max_tree_depth=10, target_accept=0.8, seed=0,
                step_size=None, callback=None, keep=None, state=None, adapt_mass=True, inverse_mass=None):
    """Runs warm-up (step-size adaptation by dual averaging) then sampling.  `logdensity_and_grad(q) -> (float, tensor)`.
    `keep(q)` maps a state to what is stored per draw (default: the state itself; pass a reducer for 10^7-dimensional
    states).  Returns dict(samples=[...], step_size, infos=[...], seconds, last_state).  `state` (a previous call's
    `last_state`, see `save_run` / `load_state`) continues that chain: position, step size and random streams are taken
    from it (q0, seed and step_size are ignored; pass n_warmup=0 to keep the adapted step size, as the reference does
    with `post_warmup_state`, samplers.py:618-660)."""
    if state is not None:
        rng = _Rng.from_state(state["rng"])
        q = state["q"].clone().to(q0.device if q0 is not None else state["q"].device)
        step_size = state["step_size"]
    else:
        rng = _Rng(seed)
        q = q0.clone()
    minv = inverse_mass
    if state is not None and state.get("inverse_mass") is not None:
        minv = state["inverse_mass"].to(q.device)
    lp, g = logdensity_and_grad(q)
    eps = step_size if step_size is not None else find_reasonable_step_size(logdensity_and_grad, q, lp, g, rng, minv=minv)
    da = DualAveraging(eps, target=target_accept)
    wins = warmup_windows(n_warmup) if adapt_mass else []
    win_end = {e: s_ for s_, e in wins}
    wf = _Welford()
    infos, samples = [], []
    t0 = time.perf_counter()
    for it in range(n_warmup + n_samples):
        warm = it < n_warmup
        q, lp, g, info = nuts_step(logdensity_and_grad, q, lp, g, eps, rng, max_tree_depth, minv)
        if warm:
            eps = da.update(info["accept_stat"])
            if wins and wins[0][0] <= it < wins[-1][1]:
                wf.add(q)
            if (it + 1) in win_end and wf.n > 1:
                # end of a slow window: new metric from this window's draws, step size re-initialised and its averaging
                # restarted (window_adaptation does the same)
                minv = wf.inverse_mass(q.dtype)
                wf = _Welford()
                eps = find_reasonable_step_size(logdensity_and_grad, q, lp, g, rng, eps=da.final(), minv=minv)
                da = DualAveraging(eps, target=target_accept)
            if it == n_warmup - 1:
                eps = da.final()
        else:
            samples.append(keep(q) if keep is not None else q.clone())
        info.update(step_size=eps, warmup=warm, logdensity=lp)
        infos.append(info)
        if callback is not None:
            callback(it, info)
    return {"samples": samples, "step_size": eps, "inverse_mass": minv, "infos": infos, "seconds": time.perf_counter() - t0,
            "last_state": {"sampler": "nuts", "q": q.clone(), "step_size": eps, "rng": rng.get_state(),
                           "inverse_mass": None if minv is None else minv.clone()}}


# ---- chains on disk (montecosmo/samplers.py:596-660: one .npz of draws per run + the last state, overwritten) ----
def save_run(result, i_run, path, extra_fields=("n_evals", "accept_stat", "step_size", "logdensity", "diverging")):
    """Writes `path_{i_run}.npz` (the kept draws stacked as 'samples' + the requested per-transition fields; n_evals is
    the number of gradient evaluations, the reference's renamed `num_steps`) and overwrites `path_last_state.pt`."""
    import numpy as np
    infos = result["infos"]
    out = {}
    draws = result["samples"]
    if len(draws):
        out["samples"] = np.stack([np.asarray(d.detach().cpu()) if isinstance(d, torch.Tensor) else np.asarray(d) for d in draws])
    for f in extra_fields:
        key = "n_leapfrog" if (f == "n_evals" and infos and "n_evals" not in infos[0]) else f
        if infos and key in infos[0]:
            out[f] = np.asarray([i[key] for i in infos])
    out["warmup"] = np.asarray([bool(i["warmup"]) for i in infos])
    np.savez(f"{path}_{i_run}.npz", **out)
    st = dict(result["last_state"])
    st["q"] = st["q"].detach().cpu()
    if "u" in st:
        st["u"] = st["u"].detach().cpu()
    if st.get("inverse_mass") is not None:
        st["inverse_mass"] = st["inverse_mass"].detach().cpu()
    torch.save(st, f"{path}_last_state.pt")


def load_state(path, device=None):
    """The state `save_run` left at `path_last_state.pt`, on `device`."""
    st = torch.load(f"{path}_last_state.pt", weights_only=False)
    if device is not None:
        st["q"] = st["q"].to(device)
        if "u" in st:
            st["u"] = st["u"].to(device)
    return st


def sample_and_save(sampler, logdensity_and_grad, q0, path, start=0, end=1, n_warmup=200, n_samples=200, resume=False, **kw):
    """Run `start` = warm-up (+ its draws), runs start+1..end = `n_samples` draws each, every run saved with `save_run`
    (samplers.py:618-660).  `sampler` is nuts_sample or mclmc_sample.  resume=True continues from `path_last_state.pt`
    without warm-up."""
    import os
    state = None
    if resume and os.path.exists(f"{path}_last_state.pt"):
        state = load_state(path, q0.device if q0 is not None else None)
    res = None
    for i_run in range(start, end + 1):
        warm = n_warmup if (state is None) else 0
        res = sampler(logdensity_and_grad, q0, n_warmup=warm, n_samples=n_samples, state=state, **kw)
        save_run(res, i_run, path)
        state = res["last_state"]
    return res


# ---- packing the model's parameter dict into one flat vector -----------------------------------------------------
class FlatLogDensity:
    """Adapts `FieldLevelLogDensity` (dict of scalars + 'white_mesh_') to the flat-vector contract of `nuts_sample`:
    q = [scalars in ld.names() order ..., white_mesh_.ravel()] as one float32 device tensor."""

    def __init__(self, ld):
        self.ld = ld
        self.scalars = [n for n in ld.names() if n != "white_mesh_"]
        # 'ngbars_' is one value per radial shell (model.py:1099-1103); every other scalar latent is one number
        self.sizes = [getattr(ld, "n_rbins", 1) if n == "ngbars_" else 1 for n in self.scalars]
        self.ns = int(sum(self.sizes))
        self.shape = tuple(ld.fwd.init_shape)
        self.n_eval = 0

    def pack(self, sample):
        from . import nbody
        import numpy as np
        w = nbody._f32(sample["white_mesh_"], self.shape).reshape(-1)
        vals = np.concatenate([np.atleast_1d(np.asarray(sample[n], dtype=np.float64)).reshape(-1) for n in self.scalars]) \
            if self.scalars else np.zeros(0)
        return torch.cat([torch.tensor(vals, dtype=torch.float32, device=w.device), w])

    def unpack(self, q):
        vals = q[:self.ns].tolist()
        out, k = {}, 0
        for n, sz in zip(self.scalars, self.sizes):
            out[n] = vals[k] if n != "ngbars_" else vals[k:k + sz]
            k += sz
        out["white_mesh_"] = q[self.ns:].reshape(self.shape)
        return out

    def __call__(self, q):
        import numpy as np
        self.n_eval += 1
        lp, grad = self.ld.logdensity_and_grad(self.unpack(q))
        if math.isnan(lp):      # never a legitimate value: a silent -inf would freeze the chain at its current state
            raise FloatingPointError("log density is NaN at this state (non-finite inputs to the likelihood?)")
        if not math.isfinite(lp):      # outside the support (QuadGaussian's is bounded, utils.py:476-477): rejected, and counted
            self.n_nonfinite = getattr(self, "n_nonfinite", 0) + 1
            return -math.inf, torch.zeros_like(q)
        gs = np.concatenate([np.atleast_1d(np.asarray(grad[n], dtype=np.float64)).reshape(-1) for n in self.scalars]) \
            if self.scalars else np.zeros(0)
        g = torch.cat([torch.tensor(gs, dtype=torch.float32, device=q.device), grad["white_mesh_"].reshape(-1)])
        return lp, g


# ------------------------------------------------------------------------------------------------------------------
# MCLMC: microcanonical Langevin Monte Carlo (Robnik, De Luca, Silverstein & Seljak 2023), the sampler the reference's
# production drivers run through blackjax (montecosmo/samplers.py:273-398: isokinetic McLachlan integrator, step size
# tuned to an energy-error variance per dimension of 5e-4, L initialised at sqrt(d)).  From scratch, identity metric.
_MCLACHLAN = 0.1931833275037836


def _mclmc_B(u, g, eps, d):
    """Isokinetic velocity update for the gradient g of the LOG density (force direction e = g / |g|).  Returns the new
    unit velocity and the kinetic energy change (d - 1) log(cosh delta + e.u sinh delta)."""
    gn = float(torch.linalg.vector_norm(g))
    if gn == 0.0:
        return u, 0.0
    e = g / gn
    delta = eps * gn / (d - 1)
    ue = float(torch.dot(u, e))
    if delta > 30.0:       # sinh / cosh overflow: the update saturates at u -> e
        return e.clone(), (d - 1) * (delta + math.log(0.5 * (1.0 + ue) + 1e-300))
    sh, ch = math.sinh(delta), math.cosh(delta)
    un = (u + e * (sh + ue * (ch - 1.0))) / (ch + ue * sh)
    return un / torch.linalg.vector_norm(un), (d - 1) * math.log(ch + ue * sh)


def mclmc_step(fn, q, lp, g, u, eps, L, rng):
    """One MCLMC transition (McLachlan minimal-norm splitting + partial velocity refreshment).
    Returns (q, lp, g, u, energy_change)."""
    d = q.numel()
    lam = _MCLACHLAN
    u, k1 = _mclmc_B(u, g, lam * eps, d)
    q = q + (0.5 * eps) * u
    lp1, g1 = fn(q)
    u, k2 = _mclmc_B(u, g1, (1 - 2 * lam) * eps, d)
    q = q + (0.5 * eps) * u
    lp2, g2 = fn(q)
    u, k3 = _mclmc_B(u, g2, lam * eps, d)
    dE = (k1 + k2 + k3) - (lp2 - lp)           # kinetic + potential (V = -log p) change
    nu = math.sqrt((math.exp(2.0 * eps / L) - 1.0) / d)
    z = torch.randn(q.shape, dtype=q.dtype, device=q.device, generator=rng.torch_gen(q.device))
    u = u + nu * z
    u = u / torch.linalg.vector_norm(u)
    return q, lp2, g2, u, dE


def mclmc_sample(logdensity_and_grad, q0, n_warmup=200, n_samples=200, desired_energy_var=5e-4, L=None, step_size=None,
                 seed=0, callback=None, keep=None, state=None, tune_L=True):
    """Warm-up (the two stages of `blackjax.mclmc_find_L_and_step_size` the reference runs, samplers.py:322-331, each half of
    `n_warmup`): throughout, the step size is driven to an energy-error variance per dimension of `desired_energy_var`
    (E[dE^2] / d = C eps^6 for this second-order splitting: C is a down-weighted running average, eps = C^(-1/6)); in the
    second half the same weights also average x and x^2, and at its end L = sqrt(sum_i Var(x_i)) (the size of the typical set;
    `tune_L=False` or an explicit `L` keep L, default sqrt(d), samplers.py:285-287).  Then `n_samples` transitions are
    recorded.  Two gradient evaluations per transition."""
    if state is not None:      # continue a chain: position, direction, step size, L and random streams (see nuts_sample)
        rng = _Rng.from_state(state["rng"])
        dev = q0.device if q0 is not None else state["q"].device
        q, step_size, L = state["q"].clone().to(dev), state["step_size"], state["L"]
    else:
        rng = _Rng(seed)
        q = q0.clone()
    d = q.numel()
    lp, g = logdensity_and_grad(q)
    tune_L = bool(tune_L) and L is None and state is None
    L = float(L) if L is not None else math.sqrt(d)
    eps = float(step_size) if step_size is not None else math.sqrt(d) / 1e4 * 10.0
    n_tune1 = n_warmup // 2                      # stage 1: step size only; stage 2: step size + position variances
    x_avg = x2_avg = None
    wx_sum = 0.0
    if state is not None:
        u = state["u"].clone().to(q.device)
    else:
        u = torch.randn(q.shape, dtype=q.dtype, device=q.device, generator=rng.torch_gen(q.device))
        u = u / torch.linalg.vector_norm(u)
    infos, samples = [], []
    # step-size predictor (the scheme of blackjax's mclmc adaptation): the energy-error variance per dimension scales as
    # xi = C eps^6; C is tracked as a weighted running average (outliers down-weighted in log space) and
    # eps = C^(-1/6) puts xi at the target
    neff, sigma_xi = 150.0, 1.5
    gamma = (neff - 1.0) / (neff + 1.0)
    c_avg, w_sum = 0.0, 0.0
    t0 = time.perf_counter()
    for it in range(n_warmup + n_samples):
        warm = it < n_warmup
        qn, lpn, gn, un, dE = mclmc_step(logdensity_and_grad, q, lp, g, u, eps, L, rng)
        bad = not (math.isfinite(dE) and math.isfinite(lpn))
        if bad:                  # reject a blown-up step, shrink the step
            eps *= 0.5
            dE = float("nan")
        else:
            q, lp, g, u = qn, lpn, gn, un
        if warm and not bad:
            xi = dE * dE / (d * desired_energy_var) + 1e-8
            w = math.exp(-0.5 * (math.log(xi) / (6.0 * sigma_xi)) ** 2)
            c_avg = gamma * c_avg + w * xi / eps ** 6
            w_sum = gamma * w_sum + w
            eps = (c_avg / w_sum) ** (-1.0 / 6.0)
            if tune_L and it >= n_tune1:
                qd = q.double()
                if x_avg is None:
                    x_avg, x2_avg = torch.zeros_like(qd), torch.zeros_like(qd)
                x_avg = gamma * x_avg + w * qd
                x2_avg = gamma * x2_avg + w * qd * qd
                wx_sum = gamma * wx_sum + w
        if tune_L and it == n_warmup - 1 and x_avg is not None and wx_sum > 0:
            var = x2_avg / wx_sum - (x_avg / wx_sum) ** 2
            Lnew = math.sqrt(max(float(var.clamp_min(0).sum()), 0.0))
            if math.isfinite(Lnew) and Lnew > 0:
                L = Lnew
        if not warm:
            samples.append(keep(q) if keep is not None else q.clone())
        info = {"energy_change": dE, "mse_per_dim": dE * dE / d if not bad else float("nan"), "step_size": eps, "L": L,
                "warmup": warm, "logdensity": lp, "n_evals": 2}
        infos.append(info)
        if callback is not None:
            callback(it, info)
    return {"samples": samples, "step_size": eps, "L": L, "infos": infos, "seconds": time.perf_counter() - t0,
            "last_state": {"sampler": "mclmc", "q": q.clone(), "u": u.clone(), "step_size": eps, "L": L, "rng": rng.get_state()}}
