"""A stand-in for the handful of jax entry points `montecosmo_amd.jax_bridge` uses (jax is absent from the build image, so the
bridge had never executed): `custom_vjp` (with `nondiff_argnums`, `defvjp`, defaults bound by signature), `pure_callback`
(the callee gets host arrays; the results are CHECKED against the declared `ShapeDtypeStruct`s, as jax does), `jax.numpy` = numpy,
`dlpack`, `config`; `vjp_of_call` runs a custom_vjp's forward and backward RULES on given output cotangents, i.e. what
`jax.vjp(f, *args)[1](bars)` evaluates for one such call.  It implements the DOCUMENTED semantics of those entry points, nothing
of jax's tracing; tests/test_jax_bridge.py keeps its real-jax test (skipped where jax is missing).  Test infrastructure only."""
import inspect
import sys
import types

import numpy as np


class ShapeDtypeStruct:
    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)


def _check(res, spec):
    if isinstance(spec, (tuple, list)):
        assert isinstance(res, (tuple, list)) and len(res) == len(spec), "pure_callback: result structure differs from result_shapes"
        return tuple(_check(r, s) for r, s in zip(res, spec))
    a = np.asarray(res)
    assert a.shape == spec.shape, f"pure_callback: shape {a.shape} where {spec.shape} was declared"
    assert a.dtype == spec.dtype, f"pure_callback: dtype {a.dtype} where {spec.dtype} was declared"
    return a


def pure_callback(fn, result_shapes, *args, vmap_method=None):
    return _check(fn(*[np.asarray(a) for a in args]), result_shapes)


class custom_vjp:
    def __init__(self, fun, nondiff_argnums=()):
        self.fun, self.nondiff = fun, tuple(nondiff_argnums)
        self.sig = inspect.signature(fun)
        self.fwd = self.bwd = None
        self.__doc__ = fun.__doc__

    def defvjp(self, fwd, bwd):
        self.fwd, self.bwd = fwd, bwd

    def _bind(self, args, kwargs):
        b = self.sig.bind(*args, **kwargs)
        b.apply_defaults()
        return list(b.args)

    def __call__(self, *args, **kwargs):
        return self.fun(*self._bind(args, kwargs))


def vjp_of_call(cv, args, out_bars):
    """(outputs, cotangents of the differentiable arguments) of ONE custom_vjp call: what jax.vjp(cv, *args)[1](out_bars) gives."""
    a = cv._bind(args, {})
    out, res = cv.fwd(*a)
    cots = cv.bwd(*[a[i] for i in cv.nondiff], res, out_bars)
    return out, cots


def install():
    """Registers the stand-in as `jax` / `jax.numpy` (only if the real one is absent); returns the module."""
    if "jax" in sys.modules and not getattr(sys.modules["jax"], "_montecosmo_standin", False):
        return sys.modules["jax"]
    m = types.ModuleType("jax")
    m._montecosmo_standin = True
    m.custom_vjp, m.pure_callback, m.ShapeDtypeStruct = custom_vjp, pure_callback, ShapeDtypeStruct
    m.default_backend = lambda: "cpu"
    m.config = types.SimpleNamespace(jax_enable_x64=False)
    m.dlpack = types.SimpleNamespace(from_dlpack=lambda t: np.asarray(t.detach().cpu().numpy()))
    jnp = types.ModuleType("jax.numpy")
    for k in ("float32", "float64", "complex64", "asarray", "zeros_like", "sum", "conj"):
        setattr(jnp, k, getattr(np, k))
    m.numpy = jnp
    sys.modules["jax"], sys.modules["jax.numpy"] = m, jnp
    return m


def uninstall():
    for k in ("jax", "jax.numpy", "montecosmo_amd.jax_bridge"):
        if k in sys.modules and (k.startswith("montecosmo") or getattr(sys.modules[k], "_montecosmo_standin", False) or k == "jax.numpy"):
            del sys.modules[k]
