"""One-off parity check at a larger mesh than the test suite uses: GPU nbody_bf (+ reverse sweep) against the float64
oracle with its threaded back end.  usage: python tools/validate_large.py [n] [n_steps] [grad 0|1] [threads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from montecosmo_amd import nbody, bricks, synth
from oracle import pm_oracle as o, background as obg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
grad = int(sys.argv[3]) if len(sys.argv) > 3 else 1
o.set_threads(int(sys.argv[4]) if len(sys.argv) > 4 else min(16, os.cpu_count()))
shape = (n, n, n)
spec = synth.init_mesh(n, seed=0, rms_disp=2.0)
pos = bricks.regular_pos(shape)
rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.complex128) - b) / np.linalg.norm(b))
res = nbody.nbody_bf(bricks.Planck18(), spec, pos, a0=0., a1=1., n_steps=n_steps, lattice_out=True, return_ctx=bool(grad))
(lp, vel), ctx = res if grad else (res, None)
t0 = time.time()
(p_o, v_o) = o.nbody_bf(obg.Planck18(), spec.astype(np.complex128), pos, 0., 1., n_steps)
print(f"oracle forward {time.time()-t0:.0f} s", flush=True)
dens_g = nbody.paint(lp, shape).cpu().numpy().astype(np.float64)
dens_o = o.paint(p_o[0], shape)
idx_g = nbody.cell_index(lp, shape).cpu().numpy()
idx_o = o.cell_index(p_o[0], shape)
print(f"{n}^3, {n_steps} steps, rms displacement 2 cells")
print("final displacement rel L2 :", rel(lp.disp.cpu().numpy(), p_o[0] - pos))
print("final velocity rel L2     :", rel(vel.cpu().numpy(), v_o[0]))
print("final density rel L2      :", rel(dens_g, dens_o), " (north-star gate 1e-5)")
print("density max               :", dens_o.max())
print("cell-index mismatches     :", float(np.any(idx_g != idx_o, axis=1).mean()), "(particles within fp32 round-off of a cell face)", flush=True)
if grad:
    rng = np.random.default_rng(1)
    xb, vb = rng.standard_normal((n ** 3, 3)), rng.standard_normal((n ** 3, 3))
    mb_g, sb_g = nbody.nbody_bf_vjp(ctx, xb.astype(np.float32), vb.astype(np.float32))
    t0 = time.time()
    mb_o, sb_o = o.nbody_bf_vjp(obg.Planck18(), spec.astype(np.complex128), pos, xb, vb, 0., 1., n_steps)
    print(f"oracle reverse {time.time()-t0:.0f} s")
    print("init_mesh_bar rel L2      :", rel(mb_g.cpu().numpy(), mb_o))
    print("alpha_bar max rel err     :", float(np.abs(sb_g['alpha'] - sb_o['alpha']).max() / np.abs(sb_o['alpha']).max()))
    print("beta_bar max rel err      :", float(np.abs(sb_g['beta'] - sb_o['beta']).max() / np.abs(sb_o['beta']).max()))
