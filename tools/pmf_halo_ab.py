"""pm_forces on the evolved particles of the bench trajectory, window halo chosen per input on the device (0) against fixed
halos, alternating in ONE process on the same buffers (process-to-process scatter is +-1 %).  usage: python tools/pmf_halo_ab.py [mesh=512]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
r = bench.Runner(n, 10, dev)
r.run(10)
torch.cuda.synchronize()
for rep in range(3):
    row = []
    for H in (0, 2, 3, 4):
        r.plan.call("mcpm_plan_set_halo", H)
        row.append(f"H={H or 'device'}: {r.force_cycle_ms(reps=20):.4f} ms")
    print(f"{n}^3 pm_forces  " + "   ".join(row), flush=True)
r.plan.call("mcpm_plan_set_halo", 0)
