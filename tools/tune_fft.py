"""Per-pass timings of the hand-written FFT Poisson solve for a few mesh shapes (GPU only)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from montecosmo_amd import nbody
from montecosmo_amd._lib import lib

shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(512, 512, 512), (512, 64, 512), (512, 128, 512)]
for shape in shapes:
    plan = nbody.get_plan(shape)
    rho = torch.randn(shape, device="cuda")
    fm = torch.empty((3,) + shape, device="cuda")
    for _ in range(3):
        plan.call("mcpm_force_meshes_f32", C.c_void_p(rho.data_ptr()), C.c_void_p(fm.data_ptr()))
    plan.call("mcpm_plan_profile", 1)
    R = 10
    for _ in range(R):
        plan.call("mcpm_force_meshes_f32", C.c_void_p(rho.data_ptr()), C.c_void_p(fm.data_ptr()))
    ms, by, calls = (C.c_double * 16)(), (C.c_double * 16)(), (C.c_int64 * 16)()
    ns = lib.mcpm_plan_profile_read(plan.h, 16, ms, by, calls)
    plan.call("mcpm_plan_profile", 0)
    M = shape[0] * shape[1] * shape[2]
    out = {lib.mcpm_stage_name(i).decode(): round(ms[i] / R, 4) for i in range(ns) if calls[i]}
    print(shape, out, "xfused GB/s(4 spectra moved):", round(4 * M * 4.25 / (out["kspace"] * 1e-3) / 1e9), "total ms", round(sum(out.values()), 3))
    nbody.clear_plans()
