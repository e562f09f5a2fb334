"""pm_forces(pos, mesh_shape) at the bench workload (evolved particles of the 10-step trajectory), timed as bench.py times it,
several batches of 10 calls; prints every batch, the median and the per-stage profile.  usage: python tools/time_pm_forces.py [mesh=512] [batches=6]"""
import os, sys, ctypes as C, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from montecosmo_amd._lib import lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
r = bench.Runner(n, 10, dev)
r.forward(10)
torch.cuda.synchronize()
ts = [r.force_cycle_ms() for _ in range(nb)]
r.plan.call("mcpm_plan_profile", 1)
for _ in range(5):
    r.force_cycle_ms(reps=1)
names, fwd, bwd = None, None, None
ms, by, calls = (C.c_double * 16)(), (C.c_double * 16)(), (C.c_int64 * 16)()
ns = lib.mcpm_plan_profile_read(r.plan.h, 16, ms, by, calls)
r.plan.call("mcpm_plan_profile", 0)
st = {lib.mcpm_stage_name(i).decode(): round(ms[i] / 10, 4) for i in range(ns) if calls[i]}    # 10 calls profiled (5 x (1 warm + 1))
M = float(n) ** 3
med = statistics.median(ts)
import hashlib
r.force_cycle_ms(reps=1)
torch.cuda.synchronize()
digest = hashlib.sha1(r.xb.cpu().numpy().tobytes()).hexdigest()[:16]      # the forces of the last call: A/B runs of a knob must agree bit for bit
print(f"forces sha1 {digest}")
if os.environ.get("MCPM_DUMP"):
    import numpy as np
    np.save(os.environ["MCPM_DUMP"], r.xb.cpu().numpy())
print(f"pm_forces {n}^3: batches {[round(t, 4) for t in ts]} ms, median {med:.4f} ms = {100 * M / (med * 1e-3) / 1e9 / 8000:.4f} of 8 TB/s; stages per call {st}", flush=True)
