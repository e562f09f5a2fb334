"""What the tile prologue could know about the halo a paint needs: per step of the bench trajectory, the fraction of tiles whose
NEED = (largest offset difference to a neighbouring tile) + (largest |floor(d) - o_T| among the prologue's 64 samples) exceeds H,
next to the same quantity from ALL particles of the tile.  usage: python tools/halo_stat.py [mesh=256]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
K = 10
r = bench.Runner(n, K, dev)
r.forward(K)
torch.cuda.synchronize()
nt = n // 16
for i in range(K + 1):
    d = r.states[i, 0].view(nt, 16, nt, 16, nt, 16, 3)
    s = d[:, 4::8, :, 4::8, :, :, :]                                   # (nt, 2, nt, 2, nt, 16, 3): the prologue's four z rows
    o = torch.clamp(torch.round(s.mean(dim=(1, 3, 5))), -8, 8)            # (nt, nt, nt, 3)
    fl = torch.floor(d)
    dev_s = (torch.floor(s) - o[:, None, :, None, :, None, :]).abs().amax(dim=(1, 3, 5, 6))
    dev_a = (fl - o[:, None, :, None, :, None, :]).abs().amax(dim=(1, 3, 5, 6))
    D = torch.zeros_like(dev_s)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                if dx or dy or dz:
                    D = torch.maximum(D, (torch.roll(o, (dx, dy, dz), (0, 1, 2)) - o).abs().amax(dim=3))
    need_s, need_a = D + dev_s, D + dev_a
    # fraction of PARTICLES outside the sure interval of half width H - D_T (what becomes a suspect)
    hw = lambda H: (H - D)[:, None, :, None, :, None, None]
    dv = (fl - o[:, None, :, None, :, None, :]).abs()
    sus = {H: float((dv > hw(H)).any(dim=6).float().mean()) for H in (2, 3, 4)}
    f = lambda x, H: float((x > H).float().mean())
    print(f"step {i:2d}: tiles with need > H (samples | all particles)  H=2: {f(need_s, 2):.4f} | {f(need_a, 2):.4f}   H=3: {f(need_s, 3):.4f} | {f(need_a, 3):.4f}"
          f"   H=4: {f(need_s, 4):.4f} | {f(need_a, 4):.4f}   suspect particles H=2 {sus[2]:.5f} H=3 {sus[3]:.5f} H=4 {sus[4]:.5f}   max D {int(D.max())} mean dev_s {float(dev_s.mean()):.2f}", flush=True)
