"""Slab code path with one rank (local-copy communicator) against the plain path at mesh n.  usage: check_slab_p1.py n [n_steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from montecosmo_amd import nbody, bricks, synth, dist
n = int(sys.argv[1]); n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
shape = (n, n, n)
spec = synth.init_mesh(n, seed=3, rms_disp=1.5)
cosmo = bricks.Planck18()
(d, v), ctx = dist.nbody_bf_slab(cosmo, spec, a0=0.1, a1=1.0, n_steps=n_steps, ghost=8, return_ctx=True)
(lp, v1), c1 = nbody.nbody_bf(cosmo, spec, nbody.LatticePos.regular(shape), a0=0.1, a1=1.0, n_steps=n_steps, return_ctx=True, lattice_out=True)
rel = lambda a, b: float((a - b).norm() / b.norm())
rng = np.random.default_rng(5)
xb = rng.standard_normal((n ** 3, 3)).astype(np.float32); vb = rng.standard_normal((n ** 3, 3)).astype(np.float32)
mb, sb = dist.nbody_bf_slab_vjp(ctx, xb, vb)
mb1, sb1 = nbody.nbody_bf_vjp(c1, xb, vb)
print(n, "P=1 slab vs plain: disp", rel(d, lp.disp), "vel", rel(v, v1), "grad", rel(mb, mb1),
      "alpha", float(np.abs(sb["alpha"] - sb1["alpha"]).max() / np.abs(sb1["alpha"]).max()),
      "beta", float(np.abs(sb["beta"] - sb1["beta"]).max() / np.abs(sb1["beta"]).max()),
      "lpt", [abs(sb[k] - sb1[k]) / abs(sb1["g"]) for k in ("g", "g2", "dg2dg")], "oob", ctx.pm.out_of_ghost())
