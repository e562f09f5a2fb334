"""Host time of one slab forward+adjoint step (64^3, where the GPU is idle most of the time): wall time per step with the
exchanges issued by the library (native) and from Python.  usage: python tools/slab_host_time.py [n=64] [reps=200]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
for native in ("0", "1"):
    os.environ["MCPM_SLAB_NATIVE"] = native
    r = bench.SlabRunner(n, 10, dev, 8)
    r.run(10)
    torch.cuda.synchronize()
    # host time: issue `reps` steps without synchronising in between; the queue never fills at this size, so the time to ISSUE
    # them is host time, and the time until the GPU is done is the step time
    t0 = time.perf_counter()
    r.run(reps)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"native={native} n={n}: host issue {1e3 * (t1 - t0) / reps:.4f} ms per fwd+adj step, wall {1e3 * (t2 - t0) / reps:.4f} ms per step", flush=True)
    del r
