"""Generates the committed golden fixtures with the float64 oracle (run here, in the build container):

    python tests/golden/make_golden.py

The reference itself (Python on JAX) cannot be imported in this container (jax / jax_cosmo / diffrax are not
installed, no network), so these vectors come from the oracle restatement, which is pinned by the analytic
known answers of tests/test_oracle_known_answers.py.  Inputs follow SURVEY.md 8(d) (seeded Gaussian field,
Planck18, regular lattice)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pm_oracle as o, background as obg  # noqa: E402
from montecosmo_amd import synth  # noqa: E402  (host-side input synthesis only)

HERE = os.path.dirname(os.path.abspath(__file__))


def make(n, n_steps, a0, rms, seed):
    shape = (n, n, n)
    cosmo = obg.Planck18()
    spec = synth.init_mesh(n, seed=seed, rms_disp=rms)            # complex64: the exact input both sides see
    spec64 = spec.astype(np.complex128)
    pos = o.regular_pos(shape)
    dpos, vel = o.lpt(cosmo, spec64, pos, a0, lpt_order=2, read_order=1)
    (p, v) = o.nbody_bf(cosmo, spec64, pos, a0, 1.0, n_steps)
    dens = o.paint(p[0], shape)
    rng = np.random.default_rng(seed + 100)
    xb = rng.standard_normal((n ** 3, 3)).astype(np.float32)
    vb = rng.standard_normal((n ** 3, 3)).astype(np.float32)
    mb, sb = o.nbody_bf_vjp(cosmo, spec64, pos, xb.astype(np.float64), vb.astype(np.float64), a0, 1.0, n_steps)
    forces = o.pm_forces(p[0], shape)
    np.savez_compressed(
        os.path.join(HERE, f"nbody_{n}.npz"),
        n=n, n_steps=n_steps, a0=a0, init_mesh=spec, lpt_dpos=dpos, lpt_vel=vel,
        final_disp=(p[0] - pos), final_vel=v[0], final_cell=o.cell_index(p[0], shape), final_density=dens.astype(np.float64),
        final_forces=forces, pos_bar=xb, vel_bar=vb, init_mesh_bar=mb.astype(np.complex128),
        alpha_bar=sb["alpha"], beta_bar=sb["beta"], lpt_scalar_bars=np.array([sb["g"], sb["g2"], sb["dg2dg"]]),
        growth_g=o.growth_table(cosmo)["g"], growth_g2=o.growth_table(cosmo)["g2"], growth_f=o.growth_table(cosmo)["f"],
        growth_f2=o.growth_table(cosmo)["f2"],
    )


def make_paint(n, N, seed):
    rng = np.random.default_rng(seed)
    shape = (n, n + 4, n - 4)
    pos = rng.uniform(-2.5 * n, 2.5 * n, (N, 3)).astype(np.float32)
    pos[:32] = np.round(pos[:32])
    pos[32:64] = np.round(pos[32:64]) + 0.5
    w = rng.standard_normal(N).astype(np.float32)
    mesh = rng.standard_normal(shape).astype(np.float32)
    out = {"pos": pos, "weights": w, "mesh": mesh, "shape": np.array(shape)}
    for order in (1, 2):
        out[f"cell_{order}"] = o.cell_index(pos.astype(np.float64), shape, order)
        out[f"paint_{order}"] = o.paint(pos.astype(np.float64), shape, w.astype(np.float64), order)
        out[f"read_{order}"] = o.read(pos.astype(np.float64), mesh.astype(np.float64), order)
    np.savez_compressed(os.path.join(HERE, "paint_read.npz"), **out)


def make_evolve():
    """FieldLevelModel.evolve fixture (model.py:686-838; oracle/bias_oracle.py::evolve): white noise + bias -> galaxy mesh,
    for the light-cone 2LPT and the fixed-a_obs N-body branches, with the log density of one observed count mesh."""
    from oracle import bias_oracle as bo
    ks = np.logspace(-3, 1, 128)
    kpow = (ks, 3.0e4 * (ks / 0.02) / (1 + (ks / 0.02) ** 2.6))
    rng = np.random.default_rng(77)
    out = {"ks": ks, "pows": kpow[1]}
    bias = dict(b1=0.8, b2=0.2, bs2=-0.15, b3=0.1, bds2=0.1, bs3=-0.05, bn2=20.0, bnpar=5.0)
    box = np.array([320., 320., 320.])
    white = np.fft.rfftn(rng.standard_normal((12, 12, 12))) * (12 ** 3 / box.prod()) ** .5
    out["white"] = white
    out["bias_keys"], out["bias_vals"] = np.array(list(bias)), np.array(list(bias.values()))
    for tag, evolution, a_obs in (("lpt_lc", "lpt", None), ("nbody", "nbody", 0.7)):
        cfg = dict(init_shape=(12, 12, 12), evol_shape=(16, 16, 16), ptcl_shape=(16, 16, 16), paint_shape=(16, 16, 16), box_size=box,
                   box_center=np.array([60., -40., 1400.]), box_rotvec=np.array([.1, .2, -.1]), a_obs=a_obs, curved_sky=True,
                   evolution=evolution, nbody_a_start=0.1, nbody_n_steps=3, lpt_order=2, paint_order=2, paint_deconv=True,
                   interlace_order=2, lin_kpow=kpow)
        gxy, aux = bo.evolve(cfg, obg.Planck18(), bias, white)
        out[f"gxy_{tag}"] = gxy
        out[f"weights_{tag}"] = aux["weights"]
    np.savez_compressed(os.path.join(HERE, "evolve_16.npz"), **out)


if __name__ == "__main__":
    make_evolve()
    make(16, 4, 0.1, 1.0, 0)
    make(32, 5, 0.0, 1.5, 1)
    make_paint(12, 3000, 2)
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
