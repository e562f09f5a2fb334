"""montecosmo_amd.jax_bridge (custom_vjp + DLPack wrappers for the reference's jax.grad / blackjax callers).  jax is an
optional dependency: without it the module must fail with a clear ImportError and nothing else in the package may need it;
with it (and a GPU) the wrapped operators must agree with the explicit (forward, vjp) pairs they are built from."""
import importlib
import sys

import numpy as np
import pytest


def test_bridge_is_optional_and_says_so():
    try:
        import jax  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="optional dependency"):
            importlib.import_module("montecosmo_amd.jax_bridge")
        assert "montecosmo_amd.jax_bridge" not in sys.modules
        import montecosmo_amd.nbody  # noqa: F401   the torch / numpy surface does not need jax
    else:
        mod = importlib.import_module("montecosmo_amd.jax_bridge")
        assert all(hasattr(mod, n) for n in ("lpt", "nbody_bf", "logdensity_fn"))


@pytest.mark.gpu
def test_bridge_gradients_match_the_explicit_vjps(gpu):
    jax = pytest.importorskip("jax")
    import jax.numpy as jnp
    from montecosmo_amd import jax_bridge as jb, nbody, bricks, synth
    n = 16
    shape = (n, n, n)
    spec = synth.init_mesh(n, seed=2, rms_disp=1.0)
    pos = bricks.regular_pos(shape)
    cosmo = bricks.Planck18()
    rng = np.random.default_rng(0)
    xb, vb = rng.standard_normal((n ** 3, 3)).astype(np.float32), rng.standard_normal((n ** 3, 3)).astype(np.float32)

    def loss(m):
        p, v = jb.nbody_bf(cosmo, m, jnp.asarray(pos), 0.1, 1.0, 3, 2, 2)
        return jnp.sum(jnp.asarray(xb) * (p[0] - jnp.asarray(pos))) + jnp.sum(jnp.asarray(vb) * v[0])

    g = np.asarray(jax.grad(loss)(jnp.asarray(spec)))
    (lp, v), ctx = nbody.nbody_bf(cosmo, spec, nbody.LatticePos.regular(shape), a0=0.1, a1=1.0, n_steps=3, return_ctx=True, lattice_out=True)
    mb, _ = nbody.nbody_bf_vjp(ctx, xb, vb)
    ref = np.conj(mb.cpu().numpy())
    assert np.linalg.norm(g - ref) < 1e-5 * np.linalg.norm(ref)


@pytest.mark.gpu
def test_bridge_rules_execute_under_a_jax_stand_in(gpu):
    """jax is absent from this image, so the bridge had never run.  Here every rule of it executes against
    tests/_jax_standin.py, which implements the documented semantics of the entry points the bridge uses (custom_vjp with
    nondiff_argnums and defaults, pure_callback with CHECKED result shapes / dtypes): forward values, and the cotangents its
    backward rules return (jax's conjugate convention for the complex half-spectrum), against the explicit pairs
    nbody.lpt / lpt_vjp and nbody.nbody_bf / nbody_bf_vjp; logdensity_fn's value and gradient against FieldLevelLogDensity."""
    try:
        import jax  # noqa: F401
        if not getattr(jax, "_montecosmo_standin", False):
            pytest.skip("real jax present: the real-jax test above covers the bridge")
    except ImportError:
        pass
    import _jax_standin as js
    js.install()
    try:
        sys.modules.pop("montecosmo_amd.jax_bridge", None)
        jb = importlib.import_module("montecosmo_amd.jax_bridge")
        from montecosmo_amd import nbody, bricks, synth
        n = 16
        shape = (n, n, n)
        spec = synth.init_mesh(n, seed=2, rms_disp=1.0)
        pos = bricks.regular_pos(shape)
        cosmo = bricks.Planck18()
        rng = np.random.default_rng(0)
        xb, vb = rng.standard_normal((n ** 3, 3)).astype(np.float32), rng.standard_normal((n ** 3, 3)).astype(np.float32)
        lat = nbody.LatticePos.regular(shape)
        # nbody_bf: forward through the primal, cotangent through the bridge's own fwd / bwd rules
        p, v = jb.nbody_bf(cosmo, spec, pos, 0.1, 1.0, 3)                       # paint_order, lpt_order by default
        (lp, vr), ctx = nbody.nbody_bf(cosmo, spec, lat, a0=0.1, a1=1.0, n_steps=3, return_ctx=True, lattice_out=True)
        assert p.shape == (1, n ** 3, 3) and p.dtype == np.float32 and v.shape == (1, n ** 3, 3)
        assert np.array_equal(p[0], lp.to_absolute().cpu().numpy().astype(np.float32)) and np.array_equal(v[0], vr.cpu().numpy())   # x64 off: f32 out
        _, (mb_j, pos_bar) = js.vjp_of_call(jb.nbody_bf, (cosmo, spec, pos, 0.1, 1.0, 3, 2, 2), (xb[None], vb[None]))
        mb, _ = nbody.nbody_bf_vjp(ctx, xb, vb)
        assert mb_j.dtype == spec.dtype and np.array_equal(mb_j, np.conj(mb.cpu().numpy())) and not np.any(pos_bar)
        # lpt
        d, vv = jb.lpt(cosmo, spec, pos, 0.3)
        dr, vvr = nbody.lpt(cosmo, spec, lat, 0.3, lpt_order=2, read_order=1)
        assert np.array_equal(d, dr.cpu().numpy()) and np.array_equal(vv, vvr.cpu().numpy())
        _, (mb_j, _) = js.vjp_of_call(jb.lpt, (cosmo, spec, pos, 0.3, 2), (xb, vb))
        mbl, _ = nbody.lpt_vjp(cosmo, spec, lat, 0.3, xb, vb, lpt_order=2)
        assert np.array_equal(mb_j, np.conj(mbl.cpu().numpy()))
        # logdensity_fn: the scalar blackjax / jax.grad see, and the gradient its backward rule hands back
        import torch
        from test_gpu_samplers import _setup
        _, flat, q0, _ = _setup("lpt")
        ld = flat.ld
        sample = {k: (v.cpu().numpy() if torch.is_tensor(v) else float(v)) for k, v in flat.unpack(q0).items()}
        f = jb.logdensity_fn(ld)
        lp_ref, g_ref = ld.logdensity_and_grad(flat.unpack(q0))
        assert float(f(sample)) == np.float32(lp_ref)
        _, (g,) = js.vjp_of_call(f, (sample,), np.float32(2.0))
        assert set(g) == set(ld.names())
        for k in g:
            ref = g_ref[k].cpu().numpy() if torch.is_tensor(g_ref[k]) else np.float32(g_ref[k])
            assert np.array_equal(np.asarray(g[k]), 2.0 * np.asarray(ref, dtype=np.float32)), k
        # a proposal far in a saturated tail: the value is -inf and the gradient keeps its structure (zeros), so
        # the chain rejects the proposal instead of dying in the callback
        far = dict(sample)
        far["sigma8_"] = -4000.0      # 40 prior sigmas below the lower bound of the truncated-normal latent
        assert float(f(far)) == -np.inf
        _, (g,) = js.vjp_of_call(f, (far,), np.float32(1.0))
        assert set(g) == set(ld.names()) and all(not np.any(np.asarray(g[k])) for k in g)
    finally:
        js.uninstall()
