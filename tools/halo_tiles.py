"""Per-tile windows along the bench trajectory (findings 43, 46): for the density paint of every state x'_i, the share of tiles by the
extent of their widest window axis (e_h: 17 + 2h - 1 or 17 + 2h points: what a symmetric halo of h spans), the window visits per particle, the suspects handed to the exact coverage test, the (particle, tile) pairs routed through buckets and the
overflow count, next to the paint's time.  usage: python tools/halo_tiles.py [mesh=512]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
r = bench.Runner(n, 10, dev, forward_only=True)
r.run(10)
torch.cuda.synchronize()
mesh = torch.empty((n, n, n), device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
out = (C.c_int64 * 13)()
for i in range(11):
    x = r.states[i, 0]
    args = (p(x), r.N, 1, None, 1, 1.0, 2, p(mesh), 0)
    r.plan.call("mcpm_paint_f32", *args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        r.plan.call("mcpm_paint_f32", *args)
    e1.record()
    e1.synchronize()
    r.plan.call("mcpm_plan_last_paint_stats", out)
    nt = sum(out[8:13])
    share = " ".join(f"e{h}:{100.0 * out[8 + h] / nt:5.1f}%" for h in range(0, 5))      # widest window axis 17 + 2h - 1 or 17 + 2h points
    visits = out[7] / (nt * 4096.0)
    print(f"{n}^3 state {i:2d}: paint {e0.elapsed_time(e1) / 5:.4f} ms  {share}  visits/particle {visits:.2f}  suspects {out[6]:9d} ({100.0 * out[6] / r.N:.2f} %)  "
          f"bucketed pairs {out[5]:8d}  bucket tiles {out[4]:6d}  overflow appends {out[3]}  wild {out[0]}", flush=True)
