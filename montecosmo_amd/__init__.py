"""MI355X-native differentiable particle-mesh forward model behind montecosmo's nbody.py surface.

Importing `montecosmo_amd.nbody` loads libmcpm.so (hand-written HIP kernels + rocFFT) and fails loudly if
the library has not been built.  Host-only helpers live in `utils`, `bricks` and `synth`.
"""
__version__ = "0.1.0"
