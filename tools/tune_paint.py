"""Times the tiled paint variants (threads x unroll, halo) on a realistically clustered state. GPU only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 3, 4, 5, 6]
halos = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [4]
from montecosmo_amd import nbody, bricks, synth
shape = (n, n, n)
spec = synth.init_mesh(n, seed=0, rms_disp=2.0)
lp, vel = nbody.nbody_bf(bricks.Planck18(), spec, nbody.LatticePos.regular(shape), n_steps=10, lattice_out=True)
disp = lp.disp
print("disp rms per axis", float(disp.std()), "max", float(disp.abs().max()))
mesh = torch.empty(shape, dtype=torch.float32, device=disp.device)
w = torch.randn(n ** 3 * 3, device=disp.device)
for var in variants:
    for H in halos:
        os.environ["MCPM_PAINT_VARIANT"] = str(var)
        nbody.clear_plans()
        plan = nbody.get_plan(shape)
        plan.call("mcpm_plan_set_halo", H)
        for weighted in (0, 1):
            args = (C.c_void_p(disp.data_ptr()), n ** 3, 1, C.c_void_p(w.data_ptr() if weighted else 0), 3, 1.0, 2, C.c_void_p(mesh.data_ptr()), 0)
            for _ in range(3):
                plan.call("mcpm_paint_f32", *args)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            R = 10
            for _ in range(R):
                plan.call("mcpm_paint_f32", *args)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / R
            print(f"variant {var} halo {H} weighted {weighted}: {dt*1e3:.3f} ms  ({16*n**3/dt/1e9:.0f} GB/s algorithmic), outliers {plan.last_outliers()}, sum {float(mesh.double().sum())/n**3:.6f}")
