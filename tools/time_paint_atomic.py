"""Global-atomic paint (paint_atomic_kernel, 8 f32 atomics per particle) vs the LDS-tiled pull paint on the same smooth
displacement field at 512^3: what a deposit through L2 atomics costs."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from montecosmo_amd import nbody, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
plan = nbody.get_plan((n, n, n))
N = plan.N
p = lambda t: C.c_void_p(t.data_ptr())
spec = torch.from_numpy(synth.init_mesh(n, seed=0, rms_disp=2.0)).cuda()
x = torch.empty(N, 3, device="cuda"); v = torch.empty(N, 3, device="cuda")
mesh = torch.empty(n, n, n, device="cuda")
def timeit(call, reps=3):
    call(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / reps
for g in (0.3, 1.0):
    plan.call("mcpm_lpt_f32", p(spec), 2, g, -3.0 / 7.0 * g * g, 2.0 * g, 0, 0, p(x), p(v))
    lat = nbody.LatticePos(x, (n, n, n))
    xa = lat.to_absolute(torch.float32).contiguous()
    t_tile = timeit(lambda: plan.call("mcpm_paint_f32", p(x), N, 1, None, 1, 1.0, 2, p(mesh), 0))
    t_atom = timeit(lambda: plan.call("mcpm_paint_f32", p(xa), N, 0, None, 1, 1.0, 2, p(mesh), 0))
    print(f"g={g}: tiled {t_tile:.3f} ms   global atomics {t_atom:.3f} ms", flush=True)
