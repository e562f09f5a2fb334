"""Kernel time of ONE rank's share of the 512^3 slab run, measured on one GPU: a slab plan of (512 / P) x 512 x 512 planes
with the local transport (no communication at all), fed the displacements of the first 512 / P planes of the bench
trajectory's LPT start.  This is the compute term T_rank(P) of DESIGN section 6's cost model (what a rank does between its
exchanges), including the ghost planes' share and the chunked launches.  usage: python tools/slab_rank_shape.py [P ...]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from montecosmo_amd import nbody, bricks, synth, dist

n, K = 512, 10
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
spec = torch.from_numpy(synth.init_mesh(n, seed=0, rms_disp=2.0)).to(dev)
cosmo = bricks.Planck18()
dg, alphas, betas, lpt_s = nbody._step_scalars(cosmo, 0.0, 1.0, K, "bullfrog")
plan = nbody.get_plan((n, n, n))
x0 = torch.empty((n ** 3, 3), device=dev)
v0 = torch.empty((n ** 3, 3), device=dev)
p_ = lambda t: C.c_void_p(t.data_ptr())
plan.call("mcpm_lpt_f32", p_(spec), 2, float(lpt_s[0]), float(lpt_s[1]), float(lpt_s[2]), 0, 0, p_(x0), p_(v0))
x0 += v0 * (dg / 2)
torch.cuda.synchronize()
nbody.clear_plans()
for P in [int(a) for a in sys.argv[1:]] or [8, 4, 2]:
    nxl = n // P
    Nl = nxl * n * n
    for chunks in ((1, 2) if nxl < 128 else (1, 4)):
        pm = dist.SlabPM((nxl, n, n), dist.LocalComm(), 8, dev, adaptive_ghost=True, chunks=chunks, native=True)
        st = torch.empty((K + 1, 2, Nl, 3), device=dev)
        st[0, 0].copy_(x0[:Nl] * 0.6)         # |d_x| must stay inside the 8 ghost planes of a slab that is periodic over nxl planes
        st[0, 1].copy_(v0[:Nl] * 0.6)
        f3 = torch.zeros((K, pm.nxe, n, n, 3), device=dev)
        xb, vb = torch.randn((Nl, 3), device=dev), torch.randn((Nl, 3), device=dev)
        sbar = torch.zeros(2 * K + 1, dtype=torch.float64, device=dev)

        def run():
            pm.reset_depth()
            depths = []
            for i in range(K):
                tau = dg / 2 if i == K - 1 else dg
                pm.step(st[i, 0], st[i, 1], alphas[i], betas[i], tau, f3[i], st[i + 1, 0], st[i + 1, 1])
                depths.append(pm.ge)
            pm.finish_depth()
            for i in reversed(range(K)):
                tau = dg / 2 if i == K - 1 else dg
                pm.step_vjp(st[i, 0], st[i, 1], f3[i], alphas[i], betas[i], tau, xb, vb, C.c_void_p(sbar.data_ptr() + 8 * i),
                            C.c_void_p(sbar.data_ptr() + 8 * (K + i)), 0.5 if i == K - 1 else 1.0, C.c_void_p(sbar.data_ptr() + 16 * K),
                            depth=depths[i], next_beta_tau=(betas[i - 1], dg) if i > 0 else None)
            return depths

        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            depths = run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (3 * K)
        print(f"P={P}: slab {nxl} x {n} x {n} (+2x8 ghost), chunks {pm.chunks}: {dt * 1e3:.3f} ms per fwd+adj step of one rank "
              f"(depths {depths}, out of ghost {pm.out_of_ghost()})", flush=True)
        del pm, st, f3
        torch.cuda.empty_cache()
