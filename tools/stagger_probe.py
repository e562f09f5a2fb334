"""Do the particle streams of the step kernels collide in the memory system because they are all in phase?  bench.Runner keeps
x_i, v_i of every checkpoint in ONE tensor, 12 N bytes = 1.5 GiB apart at 512^3: every stream of the adjoint particle kernel
(x, v, x_bar, v_bar in; x_bar, v_bar, F_bar out) then has the SAME low 29 address bits at the same time, and whether they meet in
the same channel / bank is left to the upper physical bits -- the per-process placement that makes the kernel bimodal (DESIGN
finding 25).  This probe lays the arrays out with a per-array stagger of S bytes (bench.Runner(stagger=S): array j starts j * S bytes later
than in phase) and times the particle stages for several S in ONE process, alternating.  usage: python tools/stagger_probe.py [mesh=512]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
K = 10


def stage_ms(r):
    names, fwd, bwd = r.profile()
    out = {}
    for i, nm in enumerate(names):
        c = fwd[2][i] + bwd[2][i]
        if nm in ("kick_drift", "step_adjoint", "paint3", "axpy") and c:
            out[nm] = round((fwd[0][i] + bwd[0][i]) / c, 4)
    return out


res = {}
for S in (0, 4096 + 256, 0, 4096 + 256, 16384 + 512, 65536 + 4096 + 256, 2048 + 128):
    r = bench.Runner(n, K, dev, stagger=S)      # one flat buffer, array j at j * (12 N + S) bytes (bench.Runner._allocate)
    r.run(K); torch.cuda.synchronize()
    ms = []
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r.run(K); e1.record(); e1.synchronize()
        ms.append(round(e0.elapsed_time(e1) / K, 3))
    st = stage_ms(r)
    print(f"stagger {S:8d} B per array: step {ms} ms, stages {st}", flush=True)
    del r
    torch.cuda.empty_cache()
