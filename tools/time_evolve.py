"""Times FieldLevelForward.evolve + evolve_vjp (one log-density gradient of the field-level model without the
likelihood) at evolution-mesh 256^3 (final 146^3, the BASELINE 'full field-level' configuration's mesh).
usage: python tools/time_evolve.py [nbody|lpt] [final_n]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from montecosmo_amd import model, bricks

evolution = sys.argv[1] if len(sys.argv) > 1 else "nbody"
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 146
ks = np.logspace(-3, 1, 128)
kpow = (ks, 3.0e4 * (ks / 0.02) / (1 + (ks / 0.02) ** 2.6))
fwd = model.FieldLevelForward(final_shape=(nf,) * 3, cell_length=10., box_center=(0., 0., 2500.), evolution=evolution,
                              nbody_n_steps=10, a_obs=0.7 if evolution == "nbody" else None, lin_kpow=kpow)
print("shapes: init", fwd.init_shape, "evol", fwd.evol_shape, "ptcl", fwd.ptcl_shape, "paint", fwd.paint_shape, flush=True)
cosmo = bricks.Planck18()
bias = dict(b1=0.8, b2=0.2, bs2=-0.15, b3=0.1, bds2=0.1, bs3=-0.05, bn2=5.0, bnpar=2.0)
rng = np.random.default_rng(0)
ni = fwd.init_shape[0]
white = torch.fft.rfftn(torch.randn(fwd.init_shape, device="cuda")) * float((ni ** 3 / np.prod(fwd.box_size)) ** .5)
gb = torch.randn(fwd.paint_shape, device="cuda")
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gxy, ctx = fwd.evolve(cosmo, bias, white, return_ctx=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    g = fwd.evolve_vjp(ctx, gb)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"iter {it}: evolve {1e3*(t1-t0):.1f} ms, evolve_vjp {1e3*(t2-t1):.1f} ms; gxy mean {float(gxy.mean()):.4f} std {float(gxy.std()):.3f}", flush=True)
    del ctx, g
