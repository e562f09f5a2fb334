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
