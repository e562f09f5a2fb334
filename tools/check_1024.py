"""Size-independent properties at the largest mesh the int32 / uint32 index arithmetic is written for (1024^3 = 2^30
cells and particles): mass conservation, paint / read adjointness, FFT round trip, linearity and momentum conservation of
the force cycle, and one fused BullFrog step + its adjoint running to completion.  Not a pytest (about 150 GB of HBM)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from montecosmo_amd import nbody
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
shape = (n, n, n)
N = n ** 3
torch.manual_seed(0)
# smooth displacement field of rms ~2 cells built from a few plane waves (no host-side 1024^3 random field needed)
plan = nbody.get_plan(shape)
q = torch.arange(n, device="cuda", dtype=torch.float32)
dx = (1.2 * torch.sin(2 * np.pi * 3 * q / n))[:, None, None] + (0.8 * torch.cos(2 * np.pi * 5 * q / n))[None, :, None] + torch.zeros(n, device="cuda")[None, None, :]
dy = (1.0 * torch.sin(2 * np.pi * 2 * q / n))[None, :, None] + (0.9 * torch.sin(2 * np.pi * 7 * q / n))[None, None, :] + torch.zeros(n, device="cuda")[:, None, None]
dz = (1.1 * torch.cos(2 * np.pi * 4 * q / n))[None, None, :] + (0.7 * torch.sin(2 * np.pi * 3 * q / n))[:, None, None] + torch.zeros(n, device="cuda")[None, :, None]
disp = torch.stack([dx, dy, dz], dim=-1).reshape(N, 3).contiguous()
del dx, dy, dz
lp = nbody.LatticePos(disp, shape)
print(f"n = {n}: rms displacement {float(disp.pow(2).sum(1).mean().sqrt()):.2f} cells", flush=True)
dens = nbody.paint(lp, shape)
print("mass conservation |sum/N - 1| =", abs(float(dens.double().sum()) / N - 1), " outliers", plan.last_outliers(), flush=True)
w = torch.randn(N, device="cuda")
m = torch.randn(shape, device="cuda")
lhs = float((nbody.paint(lp, shape, w).double() * m.double()).sum())
rhs = float((w.double() * nbody.read(lp, m).double()).sum())
print("paint/read adjointness |lhs - rhs| / sqrt(N) =", abs(lhs - rhs) / np.sqrt(N), flush=True)
del w
back = nbody.irfftn(nbody.rfftn(m))
print("FFT round trip rel L2 =", float((back - m).norm() / m.norm()), flush=True)
del back, m
torch.cuda.empty_cache()
F = nbody.pm_forces(lp, shape)
print("forces finite:", bool(torch.isfinite(F).all()), " momentum |sum F| / sum |F| =", float(F.double().sum(0).norm() / F.double().abs().sum()), flush=True)
F2 = nbody.pm_forces(lp, nbody.rfftn(dens))
print("pm_forces(painted) vs pm_forces(spectrum of the same paint) rel L2 =", float((F - F2).norm() / F.norm()), flush=True)
del F2, dens
torch.cuda.empty_cache()
# one fused step and its adjoint through the C ABI
import ctypes as C
p = lambda t: C.c_void_p(t.data_ptr())
v = 0.1 * F
x1, v1 = torch.empty_like(disp), torch.empty_like(disp)
fm = torch.empty(3 * N, device="cuda")
t0 = time.perf_counter()
plan.call("mcpm_bullfrog_step_f32", p(disp), p(v), 0.9, 0.05, 0.1, 2, p(fm), p(x1), p(v1))
xb, vb = torch.randn_like(disp), torch.randn_like(disp)
sb = torch.zeros(3, dtype=torch.float64, device="cuda")
plan.call("mcpm_bullfrog_step_vjp_f32", p(disp), p(v), p(fm), 0.9, 0.05, 0.1, 2, p(xb), p(vb), C.c_void_p(sb.data_ptr()),
          C.c_void_p(sb.data_ptr() + 8), 1.0, C.c_void_p(sb.data_ptr() + 16))
torch.cuda.synchronize()
print(f"fused step + adjoint: {1e3 * (time.perf_counter() - t0):.0f} ms (first call), finite: {bool(torch.isfinite(xb).all() and torch.isfinite(x1).all())}, "
      f"scalar bars {sb.tolist()}", flush=True)
