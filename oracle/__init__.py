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
    container (ModuleNotFoundError, no network), so the reference itself cannot be run to
    generate vectors.  The printed outputs of its notebooks pin the oracle:
      - tests_old/valid_precond.ipynb:76-84: dg = a2g(a_obs)/20 to 16 digits for a_obs = 0.1, 0.5, 1
        (Omega_m = 0.31, growth table logspace(-4, 0, 256)): reproduced to the last digit, which pins the
        restated jax_cosmo background, its RK4 odeint, the growth ODE, normalisation and interpolation;
      - tests_old/valid_fastpm.ipynb:747-749: Planck18 growth ratios to 3 digits;
      - tests/valid_fourier.ipynb cells 4-5: unique-entry counts of rg2cgh on a (6,6,6) field, cgh2rg inverse;
    plus the analytic known answers of SURVEY.md 8(c) items 1-7 (tests/test_oracle_known_answers.py).
  * Still "parity unpinned" (no reference datum exists): diffrax==0.5.0's Euler time grid, jax.numpy
    FFT / round semantics (numpy's are used), jax_cosmo's Eisenstein-Hu sigma8 quadrature.

THREADS: pm_oracle.set_threads(n) switches paint / read / FFT to scipy.fft workers and the OpenMP kernels
of oracle/csrc/pm_kernels.c (oracle/Makefile -> oracle/_build/libpmo.so); the numpy path is the definition,
tests/test_oracle_threads.py holds the two together.
"""
