"""Per-pass timings of the hand-written FFT Poisson solve (forward chain and its adjoint) for a few mesh shapes (GPU only).
usage: python tools/tune_fft.py [nx x ny x nz ...]      env knobs: MCPM_FFT_SWEEP, MCPM_[XY]SWEEP_LINES[_ACC], ..."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from montecosmo_amd import nbody
from montecosmo_amd._lib import lib

shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(512, 512, 512)]
for shape in shapes:
    plan = nbody.get_plan(shape)
    rho = torch.randn(shape, device="cuda")
    fm = torch.empty((3,) + shape, device="cuda")
    rb = torch.empty(shape, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    res = {}
    for name, call in (("fwd", lambda: plan.call("mcpm_force_meshes_f32", p(rho), p(fm))),
                       ("adj", lambda: plan.call("mcpm_force_meshes_vjp_f32", p(fm), p(rb)))):
        for _ in range(3):
            call()
        plan.call("mcpm_plan_profile", 1)
        R = 10
        for _ in range(R):
            call()
        ms, by, calls = (C.c_double * 16)(), (C.c_double * 16)(), (C.c_int64 * 16)()
        ns = lib.mcpm_plan_profile_read(plan.h, 16, ms, by, calls)
        plan.call("mcpm_plan_profile", 0)
        out = {lib.mcpm_stage_name(i).decode(): round(ms[i] / R, 4) for i in range(ns) if calls[i]}
        out["total"] = round(sum(out.values()), 4)
        res[name] = out
    print(shape, res, flush=True)
    nbody.clear_plans()
