"""Does the fused x pass's time depend on WHERE its spectra were allocated, within one process?  Creates and destroys the 512^3
plan several times (optionally with a dummy allocation of varying size in between, to move the plan's buffers) and times the
forward Poisson solve's stages each time.  usage: python tools/placement_probe.py [rounds=8]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from montecosmo_amd import nbody
from montecosmo_amd._lib import lib

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
shape = (512, 512, 512)
rho = torch.randn(shape, device="cuda")
fm = torch.empty((3,) + shape, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
dummies = []
for r in range(rounds):
    if r % 2 == 1:
        dummies.append(torch.empty((r * 97 + 13) * (1 << 20), dtype=torch.uint8, device="cuda"))   # moves the next plan's buffers
    plan = nbody.get_plan(shape)
    for _ in range(3):
        plan.call("mcpm_force_meshes_f32", p(rho), p(fm))
    plan.call("mcpm_plan_profile", 1)
    for _ in range(10):
        plan.call("mcpm_force_meshes_f32", p(rho), p(fm))
    ms, by, calls = (C.c_double * 16)(), (C.c_double * 16)(), (C.c_int64 * 16)()
    ns = lib.mcpm_plan_profile_read(plan.h, 16, ms, by, calls)
    plan.call("mcpm_plan_profile", 0)
    out = {lib.mcpm_stage_name(i).decode(): round(ms[i] / 10, 4) for i in range(ns) if calls[i]}
    print(f"round {r}: dummies {sum(d.numel() for d in dummies) >> 20} MB  {out}", flush=True)
    nbody.clear_plans()
    torch.cuda.synchronize()
