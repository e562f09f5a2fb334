"""
oracle/ -- TEST INFRASTRUCTURE ONLY.

A CPU (numpy, float64) restatement of the hot path of hsimonfroy/montecosmo
(`montecosmo/nbody.py`: paint -> FFT Poisson solve -> read -> BullFrog/FastPM kick-drift,
started from 1LPT/2LPT initial conditions) and of its hand-derived adjoint (VJP).

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
this package, and only as the checker / reported CPU baseline.  The product
(`montecosmo_amd/`) never imports it and has no CPU fallback.

PARITY PINNING STATUS (see DESIGN.md "Oracle"):
  * The reference is Python on JAX; jax, jax_cosmo and diffrax are not installed in the build
    container (ModuleNotFoundError, no network), so the reference itself cannot be run and
    holds no golden vectors / fixtures for this path.
  * The oracle is therefore pinned by the analytic known answers of SURVEY.md 8(c) items 1-7
    (tests/test_oracle_*.py) and by the one numeric datum the reference's tests hold
    (tests_old/valid_fastpm.ipynb:747-749, Planck18 growth ratios to 3 digits).
  * Everything the reference delegates to un-vendored third-party code (jax_cosmo==0.1.0
    background + RK4 odeint, diffrax==0.5.0 Euler stepping, jax.numpy FFT/interp/round) is
    restated from the published algorithm and flagged "parity unpinned" where no reference
    datum exists.
"""
