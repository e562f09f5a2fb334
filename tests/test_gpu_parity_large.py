"""Parity at the sizes BASELINE.json's configs name: the HIP path through the C ABI against
the float64 oracle run with its threaded back end (scipy.fft workers + the OpenMP kernels of oracle/csrc/pm_kernels.c,
checked against the pinned single-threaded numpy path in tests/test_oracle_threads.py).

  config 2: 128^3, 10-step forward (+ the gradient, whose error budget is taken apart here);
  config 3: 256^3, 10-step forward + VJP.
Inputs are bench.py's: seed-0 Gaussian field, rms 1LPT displacement 2 cells, Planck18, a 0 -> 1, 2LPT start.

Tolerances.  Forward: north_star's 1e-5 relative L2 on the final density (measured 2.5e-6), cell indices bit-exact except
for particles within fp32 round-off of a cell face (measured 6e-6 of them; asserted < 5e-5).  Gradient: the hand-written
fp32 VJP reproduces the float64 VJP *evaluated on the same (fp32) trajectory* to ~1e-6 (asserted < 1e-5); against the
pure float64 run the gradient differs by 1.1e-4 / 1.8e-4 / 3.1e-4 at 64^3 / 128^3 / 256^3, and ALL of that is the float64
VJP's own sensitivity to the linearisation point (the same number is obtained with no fp32 VJP arithmetic involved): the
CIC gradient is discontinuous across cell faces and ~3e-5 particle-steps sit on the other side of a face in fp32.
The bound asserted for that, 6e-4, is documented in DESIGN.md section 5."""
import os

import numpy as np
import pytest

from oracle import pm_oracle as o, background as obg

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = np.asarray(a), np.asarray(b)
    dt = np.complex128 if (np.iscomplexobj(a) or np.iscomplexobj(b)) else np.float64
    return float(np.linalg.norm(a.astype(dt) - b.astype(dt)) / np.linalg.norm(b.astype(dt)))


@pytest.fixture(scope="module")
def nb(gpu):
    from montecosmo_amd import nbody
    try:
        nthr = len(os.sched_getaffinity(0))
    except AttributeError:
        nthr = os.cpu_count() or 1
    prev = o.set_threads(max(1, min(nthr, 64)))
    yield nbody
    o.set_threads(prev)


def _run(nb, n, K):
    from montecosmo_amd import bricks, synth
    shape = (n, n, n)
    spec = synth.init_mesh(n, seed=0, rms_disp=2.0)
    pos = bricks.regular_pos(shape)
    (lp, vel), ctx = nb.nbody_bf(bricks.Planck18(), spec, pos, a0=0., a1=1., n_steps=K, lattice_out=True, return_ctx=True)
    cos = obg.Planck18()
    (out, traj, ts, dg) = o.nbody_bf(cos, spec.astype(np.complex128), pos, 0., 1., K, return_traj=True)
    return shape, spec, pos, lp, vel, ctx, cos, out, traj, ts, dg


def _check_forward(nb, shape, pos, lp, vel, out):
    p_o, v_o = out
    assert rel_l2(lp.disp.cpu().numpy(), p_o[0] - pos) < 1e-5
    assert rel_l2(vel.cpu().numpy(), v_o[0]) < 1e-5
    assert rel_l2(nb.paint(lp, shape).cpu().numpy(), o.paint(p_o[0], shape)) < 1e-5            # north-star gate
    mism = float(np.any(nb.cell_index(lp, shape).cpu().numpy() != o.cell_index(p_o[0], shape), axis=1).mean())
    assert mism < 5e-5, mism


def _gradient_anatomy(nb, n, K, shape, spec, pos, ctx, cos, traj, ts, dg):
    """fp32 VJP (GPU) against (1) the float64 reverse sweep evaluated on the GPU's own fp32 trajectory -- the error of the
    hand-written kernels -- and (2) the pure float64 run, whose distance from (1) is the float64 VJP's sensitivity to the
    linearisation point.  Returns (e_kernels, e_traj, e_total, flips, sb_g, sb_o)."""
    N = n ** 3
    rng = np.random.default_rng(1)
    xb, vb = rng.standard_normal((N, 3)), rng.standard_normal((N, 3))
    mb_g, sb_g = nb.nbody_bf_vjp(ctx, xb.astype(np.float32), vb.astype(np.float32))
    mb_g = mb_g.cpu().numpy()
    mb_o, sb_o = o.nbody_bf_vjp(cos, spec.astype(np.complex128), pos, xb, vb, 0., 1., K)
    # float64 reverse sweep on the GPU's fp32 trajectory (checkpoints x'_i = x_i + v_i dg/2 and v_i)
    ck = ctx.ckpt
    xbb, vbb, flips = xb.copy(), vb.copy(), 0.0
    for i in reversed(range(K)):
        xh, v = (t.double().cpu().numpy() for t in ctx.state(i))
        r = (ts[i + 1] - ts[i]) / dg
        xb2, vb2, _, _, _ = o.dkd_vjp(pos + xh - v * (dg / 2), v, r * xbb, r * vbb, dg, float(o.alpha_bf(cos, ts[i], dg)),
                                      ts[i] + dg / 2, shape)
        xbb, vbb = (1 - r) * xbb + xb2, (1 - r) * vbb + vb2
        xo = traj[i][0] + traj[i][1] * (dg / 2)
        flips += float(np.any(o.cell_index(pos + xh, shape) != o.cell_index(xo, shape), axis=1).mean())
    mb_mixed, _, _ = o.lpt_vjp(cos, spec.astype(np.complex128), pos, 0., xbb, vbb, lpt_order=2, read_order=1)
    return rel_l2(mb_g, mb_mixed), rel_l2(mb_mixed, mb_o), rel_l2(mb_g, mb_o), flips, sb_g, sb_o


def test_config2_128_forward_and_gradient_anatomy(nb):
    n, K = 128, 10
    shape, spec, pos, lp, vel, ctx, cos, out, traj, ts, dg = _run(nb, n, K)
    _check_forward(nb, shape, pos, lp, vel, out)
    e_kernels, e_traj, e_total, flips, sb_g, sb_o = _gradient_anatomy(nb, n, K, shape, spec, pos, ctx, cos, traj, ts, dg)
    assert e_kernels < 1e-5, e_kernels            # the hand-written VJP itself (measured 1.1e-6)
    assert e_total < 6e-4, e_total                # documented bound (measured 1.8e-4) ...
    assert abs(e_total - e_traj) < 0.1 * e_traj   # ... all of it the float64 VJP's sensitivity to the linearisation point
    assert 0 < flips < 2e-4, flips                # particle-steps on the other side of a cell face (measured 2.5e-5)
    assert np.allclose(sb_g["alpha"], sb_o["alpha"], rtol=1e-3, atol=1e-3 * np.abs(sb_o["alpha"]).max())
    assert np.allclose(sb_g["beta"], sb_o["beta"], rtol=1e-3, atol=1e-3 * np.abs(sb_o["beta"]).max())


def test_config3_256_forward_and_vjp(nb):
    """256^3 with the same decomposition as at 128^3: the kernels' own error stays at the 1e-6 level, and
    the whole distance to the pure float64 gradient is the float64 VJP's sensitivity to the linearisation point."""
    n, K = 256, 10
    shape, spec, pos, lp, vel, ctx, cos, out, traj, ts, dg = _run(nb, n, K)
    _check_forward(nb, shape, pos, lp, vel, out)
    e_kernels, e_traj, e_total, flips, sb_g, sb_o = _gradient_anatomy(nb, n, K, shape, spec, pos, ctx, cos, traj, ts, dg)
    assert e_kernels < 1e-5, e_kernels            # the hand-written VJP itself
    assert e_total < 1e-3, e_total                # measured 3.1e-4 ...
    assert abs(e_total - e_traj) < 0.1 * e_traj   # ... trajectory sensitivity, not kernel arithmetic
    assert 0 < flips < 2e-4, flips
    assert np.allclose(sb_g["alpha"], sb_o["alpha"], rtol=1e-3, atol=1e-3 * np.abs(sb_o["alpha"]).max())
    assert np.allclose(sb_g["beta"], sb_o["beta"], rtol=1e-3, atol=1e-3 * np.abs(sb_o["beta"]).max())


def test_config4_512_forward_parity_and_properties(nb):
    """The bench size itself: 512^3, 10 steps, forward state against the threaded float64 oracle at the
    north-star tolerances, plus properties that do not need an oracle: mass conservation, paint / read adjointness, zero
    net force (same assignment order for paint and read), bitwise repeatability of the force cycle."""
    import torch
    from montecosmo_amd import bricks, synth
    n, K = 512, 10
    N = n ** 3
    shape = (n, n, n)
    spec = synth.init_mesh(n, seed=0, rms_disp=2.0)
    pos = bricks.regular_pos(shape)
    lp, vel = nb.nbody_bf(bricks.Planck18(), spec, pos, a0=0., a1=1., n_steps=K, lattice_out=True)
    out = o.nbody_bf(obg.Planck18(), spec.astype(np.complex128), pos, 0., 1., K)
    _check_forward(nb, shape, pos, lp, vel, out)
    del out, pos
    # properties at full size
    rho = nb.paint(lp, shape)
    assert abs(float(rho.double().sum()) / N - 1.0) < 1e-6                                  # paint conserves mass
    g = torch.Generator(device="cuda").manual_seed(3)
    m = torch.randn(shape, device="cuda", generator=g)
    w = torch.randn(N, device="cuda", generator=g)
    lhs = float((nb.paint(lp, shape, weights=w).double() * m.double()).sum())
    rhs = float((w.double() * nb.read(lp, m).double()).sum())
    # both sides are sums of 8 N products of unit normals (magnitude ~ sqrt(8 N) = 3e4); fp32 terms: error ~ 1e-7 of that
    assert abs(lhs - rhs) < 2e-5 * (8 * N) ** 0.5, (lhs, rhs)                               # paint and read are adjoint
    F1 = nb.pm_forces(lp, shape)
    F2 = nb.pm_forces(lp, shape)
    assert torch.equal(F1, F2)                                                              # order-independent sums
    net = F1.double().sum(0).abs().max().item()
    assert net < 1e-6 * float(F1.double().abs().sum(0).max()), net                          # momentum conservation


def test_config4_512_adjoint_against_finite_differences(gpu):
    """A gradient check AT the bench size (nothing checked one at 512^3): the hand-written reverse sweep's
    cotangents of the step scalars alpha_i, beta_i (nbody.py:933-944 kick coefficients) against central finite differences of
    the loss <x_bar, x'_K> + <v_bar, v_K> over the same 10-step trajectory, which bench.py times.  A scalar perturbation moves
    every particle coherently, so the difference quotient stands far above the fp32 noise of a 1.3e8-particle sum; the
    cotangent itself is a sum over all particles and all later steps of the chain paint3 -> FFT -> read, so a wrong adjoint of
    any kernel at this size shows here.  Size-independent property; no oracle involved.  The loss is piecewise smooth (CIC cell
    crossings), so the difference quotient approaches the cotangent linearly in the step: beta_1 2.1e-2, 1.3e-2, 8e-3, 3.4e-3,
    1.6e-3 at relative steps 4e-3 ... 2e-4, then fp32 noise (1e-2 at 1e-4); late steps agree to 1e-4 .. 1e-3.  Step 5e-4."""
    import sys, os
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    n, K = 512, 10
    dev = torch.device("cuda", 0)
    r = bench.Runner(n, K, dev)

    def loss():
        r.forward(K)
        return float((r.pos_bar.double() * r.states[K, 0].double()).sum() + (r.vel_bar.double() * r.states[K, 1].double()).sum())

    l0 = loss()
    r.sbar.zero_()          # the adjoint steps add into their scalar cotangents
    r.backward(K)
    torch.cuda.synchronize()
    sb = r.sbar.cpu().numpy().copy()
    abar, bbar = sb[:K], sb[K:2 * K]
    assert np.all(np.isfinite(sb)) and np.abs(bbar).max() > 0
    worst = 0.0
    for name, arr, bar, idx in (("beta", r.betas, bbar, (1, 5, 8)), ("alpha", r.alphas, abar, (2, 7))):
        for i in idx:
            keep = float(arr[i])
            eps = 5e-4 * abs(keep)
            arr[i] = keep + eps
            lp = loss()
            arr[i] = keep - eps
            lm = loss()
            arr[i] = keep
            fd = (lp - lm) / (2 * eps)
            rel = abs(fd - bar[i]) / max(abs(bar[i]), 1e-30)
            print(f"{name}[{i}] = {keep:.5f}: cotangent {bar[i]:.6e}, finite difference {fd:.6e}, rel {rel:.2e} (loss {l0:.6e})")
            worst = max(worst, rel)
    assert worst < 8e-3, worst
