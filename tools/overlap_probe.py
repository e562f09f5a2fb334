"""Do a VALU-bound tiled paint and a memory-bound FFT pass overlap when they are issued on two streams?  (The question behind an
x-chunk pipeline of the adjoint step: paint3 of chunk c+1 beside the z / y passes of chunk c.)  Two plans of the same mesh on two
streams; the three-component paint of the evolved bench particles on one, a batch of three R2C transforms on the other; each alone,
then both at once.  usage: python tools/overlap_probe.py [mesh=512]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
import bench
from montecosmo_amd import nbody

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
r = bench.Runner(n, 10, dev, forward_only=True)
r.run(10)
torch.cuda.synchronize()
x = r.states[10, 0]
N, M = r.N, r.M
sA, sB = torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=0)
nbody.clear_plans()
with torch.cuda.stream(sA):
    pA = nbody.Plan((n, n, n))
with torch.cuda.stream(sB):
    pB = nbody.Plan((n, n, n))
w3 = torch.randn((N, 3), device=dev)
m3 = torch.empty((3, n, n, n), device=dev)
real3 = torch.randn((3, n, n, n), device=dev)
spec3 = torch.empty((3, n, n, n // 2 + 1), dtype=torch.complex64, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
paint = lambda: pA.call("mcpm_paint3_f32", p(x), N, 1, p(w3), 2, p(m3), 0)
fft = lambda: pB.call("mcpm_fft_r2c", p(real3), p(spec3), 3)


def timed(fns, reps=10):
    for f in fns:
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for f in fns:
            f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


tp, tf = timed([paint]), timed([fft])
tb = timed([paint, fft])
print(f"{n}^3: paint3 alone {tp:.3f} ms, 3 x R2C alone {tf:.3f} ms, both on two streams {tb:.3f} ms (sum {tp + tf:.3f}, max {max(tp, tf):.3f})", flush=True)
