"""Paint time per STEP of the bench trajectory for several window halos (mcpm_plan_set_halo): what a per-call choice of H could gain.
Forward steps only (density paint) and the adjoint's three-component paint at the same checkpoints.  usage: python tools/halo_by_step.py [mesh=512]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from montecosmo_amd._lib import lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
K = 10
r = bench.Runner(n, K, dev)
r.run(K)
torch.cuda.synchronize()


def stage(name):
    ms, by, calls = (C.c_double * 16)(), (C.c_double * 16)(), (C.c_int64 * 16)()
    ns = lib.mcpm_plan_profile_read(r.plan.h, 16, ms, by, calls)
    for i in range(ns):
        if lib.mcpm_stage_name(i).decode() == name and calls[i]:
            return ms[i] / calls[i]
    return float("nan")


for H in tuple(int(c) for c in os.environ.get("HALOS", "234")):
    r.plan.call("mcpm_plan_set_halo", H)
    fw, bw = [], []
    for rep in range(2):
        fw, bw = [], []
        for i in range(K):
            tau = r.dg / 2 if i == K - 1 else r.dg
            r.plan.call("mcpm_plan_profile", 1)
            r.plan.call("mcpm_bullfrog_step_f32", r.p(r.states[i, 0]), r.p(r.states[i, 1]), float(r.alphas[i]), float(r.betas[i]), float(tau), 2,
                        r.p(r.fmesh[i]), r.p(r.states[i + 1, 0]), r.p(r.states[i + 1, 1]))
            fw.append(round(stage("paint"), 3))
            r.plan.call("mcpm_plan_profile", 0)
        for i in reversed(range(K)):
            tau = r.dg / 2 if i == K - 1 else r.dg
            first = i == K - 1
            r.plan.call("mcpm_plan_profile", 1)
            r.plan.call("mcpm_bullfrog_step_vjp_from_f32", r.p(r.states[i, 0]), r.p(r.states[i, 1]), r.p(r.fmesh[i]), float(r.alphas[i]),
                        float(r.betas[i]), float(tau), 2, r.p(r.pos_bar if first else r.xb), r.p(r.vel_bar if first else r.vb), r.p(r.xb), r.p(r.vb),
                        C.c_void_p(r.sbar.data_ptr() + 8 * i), C.c_void_p(r.sbar.data_ptr() + 8 * (K + i)), 0.5 if first else 1.0,
                        C.c_void_p(r.sbar.data_ptr() + 8 * 2 * K))
            bw.append(round(stage("paint3"), 3))
            r.plan.call("mcpm_plan_profile", 0)
    bw.reverse()
    print(f"H = {H}: density paint per step {fw} sum {sum(fw):.3f}; three-component paint per step {bw} sum {sum(bw):.3f}", flush=True)
r.plan.call("mcpm_plan_set_halo", 0)
