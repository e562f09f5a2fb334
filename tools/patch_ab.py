"""In-process A/B of the thread -> particle map of the lattice-mode kernels (mcpm_plan_set_lattice_patch): 2 x 2 patches of lattice
rows per workgroup against one row of 256 z.  Step time, pm_forces and the particle stages per setting, alternating on the same
buffers; the gradients must agree bit for bit (the map is a permutation of the work).  usage: python tools/patch_ab.py [mesh=512] [rounds=2]"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
MODES = (0, 1)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
r = bench.Runner(n, 10, dev)
r.run(10)
torch.cuda.synchronize()
keys = ("kick_drift", "step_adjoint", "read", "paint", "paint3")
digest = {}
for rd in range(rounds):
    for on in MODES:
        r.plan.call("mcpm_plan_set_lattice_patch", on)
        r.run(10)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r.run(10)
        e1.record()
        e1.synchronize()
        ms_step = e0.elapsed_time(e1) / 10
        fc = r.force_cycle_ms()
        names, fwd, bwd = r.profile()
        st = {}
        for i, nm in enumerate(names):
            calls = fwd[2][i] + bwd[2][i]
            if nm in keys and calls:
                st[nm] = round((fwd[0][i] + bwd[0][i]) / calls, 4)
        r.run(10)
        torch.cuda.synchronize()
        digest[on] = hashlib.sha1(r.xb.cpu().numpy().tobytes()).hexdigest()[:16]
        print(f"round {rd} patch={on}: step {ms_step:.3f} ms, pm_forces {fc:.4f} ms, stages {st}, pos_bar sha1 {digest[on]}", flush=True)
print("gradients bitwise equal between the maps:", len(set(digest.values())) == 1)
