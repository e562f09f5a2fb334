"""Is the HIP log density (value and gradient) bitwise reproducible call after call?  usage: python tools/determinism_probe.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_gpu_samplers as T
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
samplers, flat, q0, ref = T._setup()
rng = np.random.default_rng(0)
bad = 0
for trial in range(4):
    q = q0 + 0.2 * trial * torch.from_numpy(rng.standard_normal(q0.shape).astype(np.float32)).to(q0.device)
    lp0, g0 = flat(q)
    g0 = g0.clone()
    nd = 0
    for r in range(reps):
        lp, g = flat(q)
        if lp != lp0 or not torch.equal(g, g0):
            nd += 1
            if nd <= 3:
                d = (g - g0).abs()
                print(f"trial {trial} rep {r}: lp {lp!r} vs {lp0!r}; gradient differs in {int((d > 0).sum())} of {d.numel()} entries, max {float(d.max()):.3e} (|g| max {float(g0.abs().max()):.3e})")
    print(f"trial {trial}: {nd} of {reps} repetitions differ")
    bad += nd
print("TOTAL differing:", bad)
