"""Where does a slab-vs-plain gradient difference come from?  One adjoint step composed from public calls on both plans, the
whole reverse sweep both ways, four implementations of the first adjoint step, and the particles whose cotangent differs
(at 128^3: ONE particle sitting 5.6e-8 from a cell face, in different cells on the two trajectories).  usage: debug_slab_adj.py n"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from montecosmo_amd import nbody, bricks, synth, dist
from montecosmo_amd._lib import POS_LATTICE
n = int(sys.argv[1]); K = 2
shape = (n, n, n)
spec = synth.init_mesh(n, seed=3, rms_disp=1.5)
cosmo = bricks.Planck18()
(d, v), ctx = dist.nbody_bf_slab(cosmo, spec, a0=0.1, a1=1.0, n_steps=K, ghost=8, return_ctx=True)
pm = ctx.pm
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
p = lambda t: C.c_void_p(t.data_ptr())
rng = np.random.default_rng(5)
xb0 = torch.from_numpy(rng.standard_normal((n ** 3, 3)).astype(np.float32)).cuda(); vb0 = torch.from_numpy(rng.standard_normal((n ** 3, 3)).astype(np.float32)).cuda()
plan = nbody.get_plan(shape)
N = n ** 3
i = K - 1
x, vv = ctx.states[i, 0], ctx.states[i, 1]
alpha, beta, tau = ctx.alphas[i], ctx.betas[i], ctx.dg / 2
# plain: force mesh of this step recomputed on the plain plan (interleaved layout kept by the plan)
fm_plain = torch.empty((n, n, n, 3), device="cuda")
xo, vo = torch.empty_like(x), torch.empty_like(vv)
plan.call("mcpm_bullfrog_step_f32", p(x), p(vv), float(alpha), float(beta), float(tau), 2, p(fm_plain), p(xo), p(vo))
G = pm.G
print("force mesh slab vs plain", rel(ctx.f3s[i][G:G + n], fm_plain), " next state", rel(ctx.states[i + 1, 0], xo))
# --- F_bar
Fb_s, Fb_p = torch.empty((N, 3), device="cuda"), torch.empty((N, 3), device="cuda")
pm.call("mcpm_kick_f32", p(vb0), p(xb0), N, float(beta), float(beta * tau), p(Fb_s))
plan.call("mcpm_kick_f32", p(vb0), p(xb0), N, float(beta), float(beta * tau), p(Fb_p))
print("Fb", rel(Fb_s, Fb_p))
# --- paint3
pm.set_depth(x, 2, ctx.depths[i])
pm.call("mcpm_paint3_f32", p(x), N, POS_LATTICE, p(Fb_s), 2, p(pm.f3), 0)
pm.halo_add(pm.f3)
m3 = torch.empty((3, n, n, n), device="cuda")
plan.call("mcpm_paint3_f32", p(x), N, POS_LATTICE, p(Fb_p), 2, p(m3), 0)
print("paint3 (after halo add) slab vs plain", [rel(pm.f3[c, G:G + n], m3[c]) for c in range(3)], "depth", pm.ge)
# reference paint3 in float64 on the host for one component (small n only)
# --- force_meshes_vjp
pm.force_meshes_vjp(pm.f3, pm.rho, None, fill_ghosts=True)
rb = torch.empty((n, n, n), device="cuda")
plan.call("mcpm_force_meshes_vjp_f32", p(m3), p(rb))
print("rho_bar slab vs plain", rel(pm.rho[G:G + n], rb))
# --- particle kernel
xs, vs = xb0.clone(), vb0.clone()
xp, vp = xb0.clone(), vb0.clone()
sb = torch.zeros(8, dtype=torch.float64, device="cuda")
pm.call("mcpm_step_adjoint_particles_il_f32", p(x), p(vv), p(ctx.f3s[i]), p(pm.rho), float(alpha), float(beta), float(tau), 2, p(xs), p(vs),
        C.c_void_p(sb.data_ptr()), C.c_void_p(sb.data_ptr() + 8), 0.5, C.c_void_p(sb.data_ptr() + 16))
fm3 = fm_plain.permute(3, 0, 1, 2).contiguous()
plan.call("mcpm_step_adjoint_particles_f32", p(x), p(vv), p(fm3), p(rb), float(alpha), float(beta), float(tau), 2, p(xp), p(vp),
          C.c_void_p(sb.data_ptr() + 32), C.c_void_p(sb.data_ptr() + 40), 0.5, C.c_void_p(sb.data_ptr() + 48))
print("xb, vb after the step: slab vs plain", rel(xs, xp), rel(vs, vp), "scalars", sb.cpu().numpy())

# ---- the whole reverse sweep, both ways, up to (not including) the LPT adjoint
print("---- full sweep")
states_p = torch.empty((K + 1, 2, N, 3), device="cuda"); fm_p = torch.empty((K, 3, n, n, n), device="cuda")
states_p[0] = ctx.states[0]
for j in range(K):
    tj = ctx.dg / 2 if j == K - 1 else ctx.dg
    plan.call("mcpm_bullfrog_step_f32", p(states_p[j, 0]), p(states_p[j, 1]), float(ctx.alphas[j]), float(ctx.betas[j]), float(tj), 2, p(fm_p[j]),
              p(states_p[j + 1, 0]), p(states_p[j + 1, 1]))
print("forward states", rel(ctx.states[K], states_p[K]))
xs, vs = xb0.clone(), vb0.clone()
xp, vp = xb0.clone(), vb0.clone()
sbs = torch.zeros(2 * K + 1, dtype=torch.float64, device="cuda"); sbp = torch.zeros(2 * K + 1, dtype=torch.float64, device="cuda")
for j in reversed(range(K)):
    tj = ctx.dg / 2 if j == K - 1 else ctx.dg
    pm.step_vjp(ctx.states[j, 0], ctx.states[j, 1], ctx.f3s[j], ctx.alphas[j], ctx.betas[j], tj, xs, vs,
                C.c_void_p(sbs.data_ptr() + 8 * j), C.c_void_p(sbs.data_ptr() + 8 * (K + j)), 0.5 if j == K - 1 else 1.0,
                C.c_void_p(sbs.data_ptr() + 8 * 2 * K), 2, depth=ctx.depths[j], next_beta_tau=(ctx.betas[j - 1], ctx.dg) if j > 0 else None)
    if j > 0:
        plan.call("mcpm_plan_hint_next_adjoint", float(ctx.betas[j - 1]), float(ctx.dg))
    plan.call("mcpm_bullfrog_step_vjp_f32", p(states_p[j, 0]), p(states_p[j, 1]), p(fm_p[j]), float(ctx.alphas[j]), float(ctx.betas[j]), float(tj), 2,
              p(xp), p(vp), C.c_void_p(sbp.data_ptr() + 8 * j), C.c_void_p(sbp.data_ptr() + 8 * (K + j)), 0.5 if j == K - 1 else 1.0,
              C.c_void_p(sbp.data_ptr() + 8 * 2 * K))
    print("after adjoint of step", j, ": xb", rel(xs, xp), "vb", rel(vs, vp))
print("scalars slab", sbs.cpu().numpy(), "\nscalars plain", sbp.cpu().numpy())
print("---- variants of the first adjoint step (slab) against the plain step without hint")
j = K - 1
tj = ctx.dg / 2
xp, vp = xb0.clone(), vb0.clone()
plan.call("mcpm_bullfrog_step_vjp_f32", p(states_p[j, 0]), p(states_p[j, 1]), p(fm_p[j]), float(ctx.alphas[j]), float(ctx.betas[j]), float(tj), 2,
          p(xp), p(vp), C.c_void_p(sbp.data_ptr()), C.c_void_p(sbp.data_ptr() + 8), 0.5, C.c_void_p(sbp.data_ptr() + 16))
for name, kw in (("no hint", dict(next_beta_tau=None)), ("hint", dict(next_beta_tau=(ctx.betas[j - 1], ctx.dg)))):
    xs, vs = xb0.clone(), vb0.clone()
    pm.step_vjp(ctx.states[j, 0], ctx.states[j, 1], ctx.f3s[j], ctx.alphas[j], ctx.betas[j], tj, xs, vs,
                C.c_void_p(sbs.data_ptr()), C.c_void_p(sbs.data_ptr() + 8), 0.5, C.c_void_p(sbs.data_ptr() + 16), 2, depth=ctx.depths[j], **kw)
    print(name, "xb", rel(xs, xp), "vb", rel(vs, vp))
# the windowed transform path alone
pm.set_depth(ctx.states[j, 0], 2, ctx.depths[j])
pm.call("mcpm_kick_f32", p(vb0), p(xb0), N, float(ctx.betas[j]), float(ctx.betas[j] * tj), p(Fb_s))
pm.call("mcpm_paint3_f32", p(ctx.states[j, 0]), N, POS_LATTICE, p(Fb_s), 2, p(pm.f3), 0)
add = pm.halo_add(pm.f3, async_op=True)
pm.force_meshes_vjp(pm.f3, pm.rho, [add, add, add], fill_ghosts=True)
r_async = pm.rho.clone()
pm.call("mcpm_paint3_f32", p(ctx.states[j, 0]), N, POS_LATTICE, p(Fb_s), 2, p(pm.f3), 0)
pm.halo_add(pm.f3)
pm.force_meshes_vjp(pm.f3, pm.rho, None, fill_ghosts=True)
print("rho_bar: windowed/async vs whole/sync", rel(r_async[G:G + n], pm.rho[G:G + n]), "ghosts", rel(r_async, pm.rho))
print("---- four ways of the first adjoint step")
def plain_fused():
    xp, vp = xb0.clone(), vb0.clone()
    plan.call("mcpm_bullfrog_step_vjp_f32", p(states_p[j, 0]), p(states_p[j, 1]), p(fm_p[j]), float(ctx.alphas[j]), float(ctx.betas[j]), float(tj), 2,
              p(xp), p(vp), C.c_void_p(sbp.data_ptr()), C.c_void_p(sbp.data_ptr() + 8), 0.5, C.c_void_p(sbp.data_ptr() + 16))
    return xp, vp
def plain_composed():
    xp, vp = xb0.clone(), vb0.clone()
    plan.call("mcpm_kick_f32", p(vb0), p(xb0), N, float(ctx.betas[j]), float(ctx.betas[j] * tj), p(Fb_p))
    plan.call("mcpm_paint3_f32", p(states_p[j, 0]), N, POS_LATTICE, p(Fb_p), 2, p(m3), 0)
    plan.call("mcpm_force_meshes_vjp_f32", p(m3), p(rb))
    plan.call("mcpm_step_adjoint_particles_f32", p(states_p[j, 0]), p(states_p[j, 1]), p(fm_p[j].reshape(n, n, n, 3).permute(3, 0, 1, 2).contiguous()), p(rb),
              float(ctx.alphas[j]), float(ctx.betas[j]), float(tj), 2, p(xp), p(vp), C.c_void_p(sbp.data_ptr()), C.c_void_p(sbp.data_ptr() + 8), 0.5, C.c_void_p(sbp.data_ptr() + 16))
    return xp, vp
def slab_fused():
    xs, vs = xb0.clone(), vb0.clone()
    pm.step_vjp(ctx.states[j, 0], ctx.states[j, 1], ctx.f3s[j], ctx.alphas[j], ctx.betas[j], tj, xs, vs,
                C.c_void_p(sbs.data_ptr()), C.c_void_p(sbs.data_ptr() + 8), 0.5, C.c_void_p(sbs.data_ptr() + 16), 2, depth=ctx.depths[j])
    return xs, vs
A, B, Cc = plain_fused(), plain_composed(), slab_fused()
print("plain fused vs plain composed", rel(A[0], B[0]), " slab fused vs plain composed", rel(Cc[0], B[0]), " plain fused vs slab fused", rel(A[0], Cc[0]))
dx = (Cc[0] - B[0]).abs()
bad = (dx > 1e-3 * B[0].abs().mean()).any(dim=1)
idx = bad.nonzero().flatten()
print("particles with a visible xb difference:", int(bad.sum()), "of", N, "; max abs diff", float(dx.max()), "rms xb", float(B[0].pow(2).mean().sqrt()))
if len(idx):
    ii = idx[:20].cpu().numpy()
    print("lattice coords (x, y, z) of the first ones:", [(int(i // (n * n)), int(i // n % n), int(i % n)) for i in ii])
    print("their displacements:", ctx.states[j, 0][idx[:6]].cpu().numpy())
    print("slab xb", Cc[0][idx[:4]].cpu().numpy(), "\nplain xb", B[0][idx[:4]].cpu().numpy())
    xs_ = np.array([int(i // (n * n)) for i in idx.cpu().numpy()])
    print("histogram of their lattice x planes:", np.bincount(xs_, minlength=n)[:n])
