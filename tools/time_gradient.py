"""Wall time of one full log-prob gradient of the PM part: nbody_bf (2LPT start + n_steps) + its reverse sweep."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from montecosmo_amd import nbody, bricks, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
shape = (n, n, n)
spec = torch.from_numpy(synth.init_mesh(n, seed=0, rms_disp=2.0)).cuda()
lat = nbody.LatticePos.regular(shape)
cosmo = bricks.Planck18()
xb = torch.randn(n ** 3, 3, device="cuda")
vb = torch.randn(n ** 3, 3, device="cuda")
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    (lp, v), ctx = nbody.nbody_bf(cosmo, spec, lat, n_steps=n_steps, return_ctx=True, lattice_out=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    mb, sb = nbody.nbody_bf_vjp(ctx, xb, vb)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{n}^3 {n_steps} steps: forward {1e3*(t1-t0):.1f} ms, reverse {1e3*(t2-t1):.1f} ms, total {1e3*(t2-t0):.1f} ms")
    del ctx
