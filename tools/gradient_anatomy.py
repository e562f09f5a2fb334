"""Where the fp32 gradient's deviation from the float64 oracle comes from.
usage: python tools/gradient_anatomy.py [n] [n_steps] [threads]

Compares, at n^3 / n_steps BullFrog steps (bench inputs):
  A  GPU gradient vs the pure float64 oracle (forward AND reverse in float64);
  B  GPU gradient vs the oracle's float64 reverse sweep run ON THE GPU'S fp32 TRAJECTORY (checkpoints x'_i, v_i cast to
     float64): isolates the arithmetic of the hand-written VJP kernels from the sensitivity of the gradient to the
     linearisation point;
  C  the per-step fraction of particles whose CIC base cell differs between the two trajectories (a particle within fp32
     round-off of a cell face: the CIC gradient is discontinuous there)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from montecosmo_amd import nbody, bricks, synth
from oracle import pm_oracle as o, background as obg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
o.set_threads(int(sys.argv[3]) if len(sys.argv) > 3 else min(16, os.cpu_count()))
shape = (n, n, n)
N = n ** 3
spec = synth.init_mesh(n, seed=0, rms_disp=2.0)
pos = bricks.regular_pos(shape)
cos_o = obg.Planck18()
rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.complex128) - b) / np.linalg.norm(b))
(lp, vel), ctx = nbody.nbody_bf(bricks.Planck18(), spec, pos, a0=0., a1=1., n_steps=K, lattice_out=True, return_ctx=True)
rng = np.random.default_rng(1)
xb, vb = rng.standard_normal((N, 3)), rng.standard_normal((N, 3))
mb_g = nbody.nbody_bf_vjp(ctx, xb.astype(np.float32), vb.astype(np.float32))[0].cpu().numpy()
t0 = time.time()
(_, traj, ts, dg) = o.nbody_bf(cos_o, spec.astype(np.complex128), pos, 0., 1., K, return_traj=True)
mb_o, _ = o.nbody_bf_vjp(cos_o, spec.astype(np.complex128), pos, xb, vb, 0., 1., K)
print(f"{n}^3, {K} steps; oracle forward + reverse {time.time() - t0:.0f} s", flush=True)
print("A  GPU gradient vs float64 oracle              rel L2 =", rel(mb_g, mb_o))
# B: float64 reverse sweep on the GPU's trajectory
ck = ctx.ckpt
state = lambda i: tuple(t.double().cpu().numpy() for t in ctx.state(i))
xbb, vbb = xb.copy(), vb.copy()
flips = []
for i in reversed(range(K)):
    xh, v = state(i)                      # x'_i = x_i + v_i dg/2 (displacement from the lattice), v_i
    x = pos + xh - v * (dg / 2)
    r = (ts[i + 1] - ts[i]) / dg
    alpha = float(o.alpha_bf(cos_o, ts[i], dg))
    xb2, vb2, _, _, _ = o.dkd_vjp(x, v, r * xbb, r * vbb, dg, alpha, ts[i] + dg / 2, shape)
    xbb, vbb = (1 - r) * xbb + xb2, (1 - r) * vbb + vb2
    xo = traj[i][0] + traj[i][1] * (dg / 2)
    flips.append(float(np.any(o.cell_index(pos + xh, shape) != o.cell_index(xo, shape), axis=1).mean()))
mb_mixed, _, _ = o.lpt_vjp(cos_o, spec.astype(np.complex128), pos, 0., xbb, vbb, lpt_order=2, read_order=1)
print("B  GPU gradient vs float64 VJP on GPU trajectory rel L2 =", rel(mb_g, mb_mixed))
print("   float64 VJP on GPU trajectory vs pure oracle  rel L2 =", rel(mb_mixed, mb_o))
print("C  fraction of particles in another cell, steps 0..K-1:", " ".join(f"{f:.2e}" for f in reversed(flips)))
print("   sum over steps:", f"{sum(flips):.3e}", " sqrt:", f"{np.sqrt(sum(flips)):.3e}")
