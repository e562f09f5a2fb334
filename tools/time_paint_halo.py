"""paint / paint3 time vs tile halo H on smooth LPT displacement fields of growing amplitude (512^3)."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from montecosmo_amd import nbody, synth
from montecosmo_amd._lib import lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
plan = nbody.get_plan((n, n, n))
N = plan.N
p = lambda t: C.c_void_p(t.data_ptr())
spec = torch.from_numpy(synth.init_mesh(n, seed=0, rms_disp=2.0)).to(dev)
x = torch.empty(N, 3, device=dev)
v = torch.empty(N, 3, device=dev)
w3 = torch.randn(N, 3, device=dev)
mesh = torch.empty(3, n, n, n, device=dev)

def outl():
    c, b = C.c_int64(), C.c_int64()
    lib.mcpm_plan_last_outliers(plan.h, C.byref(c))
    lib.mcpm_plan_last_bucketed(plan.h, C.byref(b))
    return f"{c.value} outliers, {b.value} bucketed"

def timeit(call, reps=3):
    call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps

orders = (0, 1) if len(sys.argv) > 2 else (0,)      # here: 0 = windows on the tile, 1 = centred on the bulk displacement
for g in ((1.0,) if len(sys.argv) > 2 else (0.1, 0.3, 0.5, 0.7, 1.0)):
    plan.call("mcpm_lpt_f32", p(spec), 2, g, -3.0 / 7.0 * g * g, 2.0 * g, 0, 0, p(x), p(v))
    rms = float(x.pow(2).sum(1).mean().sqrt())
    amax = float(x.abs().max())
    for H, order in [(h, o_) for h in ((2, 4) if len(sys.argv) > 2 else (1, 2, 3, 4)) for o_ in orders]:
        if lib.mcpm_plan_set_halo(plan.h, H) != 0:
            continue
        lib.mcpm_plan_set_centre(plan.h, order)
        t1 = timeit(lambda: plan.call("mcpm_paint_f32", p(x), N, 1, None, 1, 1.0, 2, p(mesh), 0))
        out1 = outl()
        t3 = timeit(lambda: plan.call("mcpm_paint3_f32", p(x), N, 1, p(w3), 2, p(mesh), 0))
        out3 = outl()
        print(f"g={g:.1f} rms={rms:.2f} max={amax:.1f} H={H} centre={order}: paint {t1:.3f} ms ({out1})  paint3 {t3:.3f} ms ({out3})", flush=True)
