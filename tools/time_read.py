"""Times the 3-component read at 512^3 for zero / LPT / evolved displacements (is the gather address-divergence bound?)."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from montecosmo_amd import nbody, bricks, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
plan = nbody.get_plan((n, n, n))
N = plan.N
p = lambda t: C.c_void_p(t.data_ptr())
mesh = torch.randn(3, n, n, n, device=dev)
out = torch.empty(N, 3, device=dev)
vel = torch.randn(N, 3, device=dev)

def t_read(x, label, reps=5):
    for name, call in (("read3", lambda: plan.call("mcpm_read_f32", p(x), N, 1, p(mesh), 3, 2, p(out))),
                       ("read1", lambda: plan.call("mcpm_read_f32", p(x), N, 1, p(mesh), 1, 2, p(out))),
                       ("kick_drift", lambda: plan.call("mcpm_kick_drift_f32", p(x), p(vel), N, 1, p(mesh), 2, 0.9, 0.1, 0.0, p(x), p(vel)))):
        call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record(); e1.synchronize()
        print(f"{label:10s} {name:10s} {e0.elapsed_time(e1)/reps:.3f} ms", flush=True)

quick = len(sys.argv) > 2
x = torch.zeros(N, 3, device=dev)
t_read(x, "zero", 2 if quick else 5)
for s in (() if quick else (0.3, 1.0, 2.0)):
    x = torch.randn(N, 3, device=dev) * s / 3 ** 0.5      # uncorrelated jitter, rms s cells
    t_read(x, f"jitter{s}")
spec = torch.from_numpy(synth.init_mesh(n, seed=0, rms_disp=2.0)).to(dev)
x = torch.empty(N, 3, device=dev)
v = torch.empty(N, 3, device=dev)
for g in ((1.0,) if quick else (0.25, 1.0)):
    plan.call("mcpm_lpt_f32", p(spec), 2, g, -3.0 / 7.0 * g * g, 2.0 * g, 0, 0, p(x), p(v))     # smooth field, rms ~ 2 g cells
    print("rms", float(x.pow(2).sum(1).mean().sqrt()))
    t_read(x.clone(), f"lpt g={g}")
