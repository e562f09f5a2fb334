"""What a per-axis, asymmetric window would save over the per-tile symmetric halo (finding 43): for every state of the bench trajectory,
window points per particle of (a) the current rule -- offset o_T = rounded mean of 64 samples, H_T = max |fd - o_T| over the sampled floor(d)
ranges of the 27 blocks around T, clamped to 1..4, window (16 + 2 H + 1)^3 -- and (b) a box [lo_a, hi_a] per axis straight from those ranges,
window prod_a (17 + hi_a - lo_a), extent clamped to 8.  usage: python tools/window_extents.py [mesh=512]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
r = bench.Runner(n, 10, dev, forward_only=True)
r.run(10)
torch.cuda.synchronize()
nt = n // 16


def pool27(x, mode):      # x: (3, nt, nt, nt); min / max over the 27 periodic neighbours
    y = F.pad(x[None], (1, 1, 1, 1, 1, 1), mode="circular")
    y = F.max_pool3d(y if mode == "max" else -y, 3, 1)[0]
    return y if mode == "max" else -y


for i in range(11):
    d = r.states[i, 0][: r.N * 3].view(n, n, n, 3)
    b = d.view(nt, 16, nt, 16, nt, 16, 3)
    s = b[:, [4, 12]][:, :, :, [4, 12]]                          # (nt, 2, nt, 2, nt, 16, 3): the prologue's 64 samples
    s = s.permute(0, 2, 4, 1, 3, 5, 6).reshape(nt, nt, nt, 64, 3)
    o = torch.round(s.mean(3)).permute(3, 0, 1, 2)              # (3, nt, nt, nt)
    f = torch.floor(s)
    lo = pool27(f.amin(3).permute(3, 0, 1, 2).contiguous(), "min")
    hi = pool27(f.amax(3).permute(3, 0, 1, 2).contiguous(), "max")
    need = torch.maximum(hi - o, o - lo).amax(0).clamp(1, 4)
    sym = ((17 + 2 * need) ** 3).sum().item() / (nt ** 3 * 4096.0)
    ext = (hi - lo).clamp(max=8)
    box = (17 + ext).prod(0).sum().item() / (nt ** 3 * 4096.0)
    ext_mean = ext.mean().item()
    print(f"{n}^3 state {i:2d}: visits/particle  per-tile halo {sym:.3f}   per-axis box {box:.3f}  ({100 * (1 - box / sym):.1f} % fewer; mean extent {ext_mean:.2f})", flush=True)
