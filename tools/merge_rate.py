"""Fraction of z-adjacent lattice particle pairs that land in z-adjacent cells of the same (x, y) row, i.e. whose CIC
deposits could be combined across lanes before the LDS atomics (evolved BullFrog state, workload of bench.py)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from montecosmo_amd import nbody, bricks, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
spec = synth.init_mesh(n, seed=0, rms_disp=2.0)
cosmo = bricks.Planck18()
for a1 in (0.3, 0.6, 1.0):
    pos, vel = nbody.nbody_bf(cosmo, spec, nbody.LatticePos.regular((n,) * 3), a0=0.0, a1=a1, n_steps=10, lattice_out=True)
    d = pos.disp.reshape(n, n, n, 3)
    f = torch.floor(d)
    same = (f[:, :, 1:, :] == f[:, :, :-1, :]).all(-1).float().mean().item()
    print(f"a1={a1}: rms disp {float(d.pow(2).sum(-1).mean().sqrt()):.2f}  z-adjacent pairs with equal floor(d): {same:.3f}", flush=True)
