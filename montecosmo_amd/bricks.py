"""The pieces of montecosmo/bricks.py the PM path touches: cosmology presets (bricks.py:16-47) as a
duck-typed object (the path reads Omega_m, Omega_de, Omega_k, w0, wa and uses `_workspace`, nbody.py:699)
and the initial particle lattice (bricks.py:593-603)."""
import os

import numpy as np


class Cosmology:
    """Stand-in for jax_cosmo.Cosmology with the attributes the PM path reads."""

    def __init__(self, Omega_c, Omega_b, h, n_s, sigma8, Omega_k=0.0, w0=-1.0, wa=0.0):
        self.Omega_c, self.Omega_b, self.h, self.n_s, self.sigma8 = Omega_c, Omega_b, h, n_s, sigma8
        self.Omega_k, self.w0, self.wa = Omega_k, w0, wa
        self._workspace = {}

    @property
    def Omega_m(self):
        return self.Omega_b + self.Omega_c

    @property
    def Omega_de(self):
        return 1.0 - self.Omega_k - self.Omega_m


def _preset(**defaults):
    def make(**kw):
        args = dict(defaults)
        args.update(kw)
        return Cosmology(**args)
    return make


Planck15 = _preset(Omega_c=0.2589, Omega_b=0.04860, Omega_k=0.0, h=0.6774, n_s=0.9667, sigma8=0.8159, w0=-1.0, wa=0.0)
Planck18 = _preset(Omega_c=0.2607, Omega_b=0.0490, sigma8=0.8102, Omega_k=0.0, h=0.6766, n_s=0.9665, w0=-1.0, wa=0.0)
AbacusSummit0 = _preset(Omega_c=0.26447041, Omega_b=0.04930169, sigma8=0.8076353990239834, Omega_k=0.0, h=0.6736,
                        n_s=0.9649, w0=-1.0, wa=0.0)


def regular_pos(mesh_shape, ptcl_shape=None):
    """Regularly spaced positions in cell coordinates, x slowest / z fastest (bricks.py:593-603), float64 numpy.
    (`montecosmo_amd.nbody.LatticePos.regular` is the same lattice in the kernels' displacement encoding.)"""
    ptcl_shape = mesh_shape if ptcl_shape is None else ptcl_shape
    axes = [np.arange(p) * (m / p) for m, p in zip(mesh_shape, ptcl_shape)]
    return np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1).reshape(-1, 3)


# ------------------------------------------------------------------------------------------------
# Lagrangian bias expansion (bricks.py:327-443), png_type = None
BIAS_KEYS = ("b1", "b2", "bs2", "b3", "bds2", "bs3", "bn2", "bnpar")


class BiasCtx:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def lagrangian_bias(cosmo, pos, a, box_size, lin_mesh, bias, png=None, png_type=None, kpow=None, read_order: int = 2,
                    return_ctx=False):
    """Lagrangian bias expansion weights (bricks.py:327-443): returns (weights (N,), dvel (N,3), phi) like the
    reference (phi = 0. without primordial non-Gaussianity, the only case built).  `a`: scalar or (N,1) scale
    factor(s); `pos`: the Lagrangian positions in cell units ((N,3) array or LatticePos); `bias`: dict with the keys
    BIAS_KEYS (missing keys = 0).  HIP: mcpm_bias_fields_f32 -> mcpm_read_f32 x5 -> mcpm_bias_weights_f32."""
    import ctypes as C
    import torch
    from . import nbody
    if png_type is not None:
        raise NotImplementedError("primordial non-Gaussianity terms are not built")
    spec = nbody._c64(lin_mesh)
    shape = nbody.ch2rshape(spec.shape)
    plan, p, n, mode = nbody._pos_args(pos, shape)
    M = plan.M
    dev = spec.device
    kphys = [float(s) / float(b) for s, b in zip(shape, box_size)]
    fields = torch.empty((7,) + tuple(shape), dtype=torch.float32, device=dev)
    # with a context for the adjoint, delta and the Hessian meshes stay resident for it (six transforms less per gradient)
    hess6 = torch.empty((6,) + tuple(shape), dtype=torch.float32, device=dev) if (return_ctx and os.environ.get("MCPM_BIAS_KEEP", "1") != "0") else None
    plan.call("mcpm_bias_fields_save_f32", nbody._ptr(spec), kphys[0], kphys[1], kphys[2], nbody._ptr(fields), nbody._ptr(hess6))
    # NGP read at the mesh's own lattice points is the identity: the fields themselves are the reads (no pass at all)
    ident = (int(read_order) == 1 and isinstance(pos, nbody.LatticePos) and pos.is_regular and tuple(pos.ptcl_shape) == tuple(shape))
    if ident:
        reads, gr, gcs = fields[:4].reshape(4, n), fields[4:7], M
    else:
        gcs = 0
        reads = torch.empty((4, n), dtype=torch.float32, device=dev)
        for c in range(4):
            plan.call("mcpm_read_f32", nbody._ptr(p), n, mode, nbody._ptr(fields[c]), 1, int(read_order), nbody._ptr(reads[c]))
        gr = torch.empty((n, 3), dtype=torch.float32, device=dev)
        plan.call("mcpm_read_f32", nbody._ptr(p), n, mode, nbody._ptr(fields[4]), 3, int(read_order), nbody._ptr(gr))
    if isinstance(a, torch.Tensor) and a.is_cuda:      # per-particle scale factors on the device (light cone)
        gp, g_shape = nbody.growth_dev(cosmo, a, "g"), tuple(a.shape)
        if gp.numel() != n:
            raise ValueError("a must have one entry per particle")
        gs = 0.0
    else:
        g = np.asarray(nbody.a2g(cosmo, a), dtype=np.float64)
        g_shape = g.shape
        gp = nbody._f32(g.reshape(-1), (n,)) if g.size == n and n > 1 else None
        gs = float(g.reshape(-1)[0]) if gp is None else 0.0
    b8 = (C.c_float * 8)(*[float(bias.get(k, 0.0)) for k in BIAS_KEYS])
    w = torch.empty(n, dtype=torch.float32, device=dev)
    dvel = torch.empty((n, 3), dtype=torch.float32, device=dev)
    plan.call("mcpm_bias_weights_f32", n, nbody._ptr(reads[0]), nbody._ptr(reads[1]), nbody._ptr(reads[2]), nbody._ptr(reads[3]),
              nbody._ptr(gr), gcs, nbody._ptr(gp), gs, b8, nbody._ptr(w), nbody._ptr(dvel), None)
    if return_ctx:
        ctx = BiasCtx(plan=plan, spec=spec, shape=shape, p=p, n=n, mode=mode, kphys=kphys, reads=reads, gr=gr, gp=gp, gs=gs,
                      g_shape=g_shape, b8=b8, read_order=int(read_order), gcs=gcs, hess6=hess6)
        return (w, dvel, 0.), ctx
    return w, dvel, 0.


def lagrangian_bias_vjp(ctx, weights_bar, dvel_bar):
    """VJP of lagrangian_bias w.r.t. (lin_mesh, bias, growth factor(s) a2g(a)): returns (lin_mesh_bar [complex64,
    real-pair convention], bias_bar dict, growths_bar [shape of a2g(a)]).  Positions are the fixed Lagrangian
    lattice of the model (model.py:738), so no position cotangent is formed."""
    import torch
    from . import nbody
    plan, n, dev = ctx.plan, ctx.n, ctx.spec.device
    wb = nbody._f32(weights_bar, (n,))
    vb = nbody._f32(dvel_bar, (n, 3))
    fb = torch.empty((7,) + tuple(ctx.shape), dtype=torch.float32, device=dev)
    if ctx.gcs:      # identity reads: the read cotangents ARE the mesh cotangents
        rb, grb = fb[:4].reshape(4, n), fb[4:7]
    else:
        rb = torch.empty((4, n), dtype=torch.float32, device=dev)
        grb = torch.empty((n, 3), dtype=torch.float32, device=dev)
    gbar = torch.empty(n, dtype=torch.float32, device=dev) if ctx.gp is not None else None
    scal = torch.zeros(10, dtype=torch.float64, device=dev)
    r = ctx.reads
    plan.call("mcpm_bias_weights_vjp_f32", n, nbody._ptr(r[0]), nbody._ptr(r[1]), nbody._ptr(r[2]), nbody._ptr(r[3]), nbody._ptr(ctx.gr),
              ctx.gcs, nbody._ptr(ctx.gp), ctx.gs, ctx.b8, nbody._ptr(wb), nbody._ptr(vb), nbody._ptr(rb[0]), nbody._ptr(rb[1]),
              nbody._ptr(rb[2]), nbody._ptr(rb[3]), nbody._ptr(grb), nbody._ptr(gbar), nbody._ptr(scal))
    if not ctx.gcs:
        for c in range(4):       # adjoint of a read w.r.t. its mesh = a weighted paint
            plan.call("mcpm_paint_f32", nbody._ptr(ctx.p), n, ctx.mode, nbody._ptr(rb[c]), 1, 0.0, ctx.read_order, nbody._ptr(fb[c]), 0)
        plan.call("mcpm_paint3_f32", nbody._ptr(ctx.p), n, ctx.mode, nbody._ptr(grb), ctx.read_order, nbody._ptr(fb[4]), 0)
    out = torch.empty(tuple(ctx.spec.shape), dtype=torch.complex64, device=dev)
    if getattr(ctx, "hess6", None) is not None:
        plan.call("mcpm_bias_fields_vjp_saved_f32", ctx.kphys[0], ctx.kphys[1], ctx.kphys[2], nbody._ptr(ctx.hess6), nbody._ptr(fb), nbody._ptr(out))
    else:
        plan.call("mcpm_bias_fields_vjp_f32", nbody._ptr(ctx.spec), ctx.kphys[0], ctx.kphys[1], ctx.kphys[2], nbody._ptr(fb), nbody._ptr(out))
    s = scal.cpu().numpy()
    bias_bar = {k: float(s[i]) for i, k in enumerate(BIAS_KEYS)}
    # per-particle growth cotangents stay on the device; a scalar one comes back as a float64 array of the shape of a2g(a)
    growths_bar = gbar.reshape(ctx.g_shape) if gbar is not None else np.asarray(s[8]).reshape(ctx.g_shape)
    return out, bias_bar, growths_bar


# ------------------------------------------------------------------------------------------------
# Cell <-> physical coordinates, line of sight, redshift-space distortions (bricks.py:628-662, :750-803)
def rot_matrix(box_rot):
    """3x3 matrix of `box_rot`: a scipy Rotation (as the reference passes), a rotation vector (3,) or a matrix."""
    if hasattr(box_rot, "as_matrix"):
        return np.asarray(box_rot.as_matrix(), dtype=np.float64)
    r = np.asarray(box_rot, dtype=np.float64)
    if r.shape == (3, 3):
        return r
    th = np.linalg.norm(r)
    if th == 0:
        return np.eye(3)
    k = r / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def cell2phys_pos(pos, box_center, box_rot, box_size, mesh_shape):
    """Cell positions to physical positions (bricks.py:628-636); host numpy (set-up sized arrays)."""
    pos = np.asarray(pos, dtype=np.float64) * np.divide(box_size, mesh_shape) - np.asarray(box_size) / 2
    return pos @ rot_matrix(box_rot).T + np.asarray(box_center)


def phys2cell_pos(pos, box_center, box_rot, box_size, mesh_shape):
    """bricks.py:638-646"""
    pos = (np.asarray(pos, dtype=np.float64) - np.asarray(box_center)) @ rot_matrix(box_rot) + np.asarray(box_size) / 2
    return pos / np.divide(box_size, mesh_shape)


def cell2phys_vel(vel, box_rot, box_size, mesh_shape):
    """bricks.py:648-654"""
    return (np.asarray(vel, dtype=np.float64) * np.divide(box_size, mesh_shape)) @ rot_matrix(box_rot).T


def phys2cell_vel(vel, box_rot, box_size, mesh_shape):
    """bricks.py:656-662"""
    return (np.asarray(vel, dtype=np.float64) @ rot_matrix(box_rot)) / np.divide(box_size, mesh_shape)


class ObsCtx:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def observe_pos(cosmo, pos, vel, box_center, box_rot, box_size, evol_shape, paint_shape, a_obs=None, curved_sky=True,
                dvel=None, return_ctx=False):
    """Evolved particles (cell units of evol_shape) -> redshift-space positions in cell units of paint_shape: the chain
    los_scalefactor_pos -> cell2phys_pos -> + rsd(vel, los, a, dvel) -> phys2cell_pos of model.py:780-797 (no
    Alcock-Paczynski), fused in one HIP pass (mcpm_observe_pos_f32).  a_obs=None is the light cone (a = chi2a(|x|)).
    A LatticePos comes back as a LatticePos on the paint mesh (same particle lattice), an array as an (N,3) tensor."""
    import ctypes as C
    import torch
    from . import nbody
    evol_shape = tuple(int(s) for s in evol_shape)
    paint_shape = tuple(int(s) for s in paint_shape)
    plan, p, n, mode = nbody._pos_args(pos, evol_shape)
    v = nbody._f32(vel, (n, 3))
    dv = nbody._f32(dvel, (n, 3)) if dvel is not None else None
    R = rot_matrix(box_rot)
    lightcone = a_obs is None
    gf = 0.0 if lightcone else float(nbody.a2g(cosmo, a_obs) * nbody.a2f(cosmo, a_obs))
    geom = (C.c_float * 19)(*[float(x) for x in list(R.reshape(-1)) + list(box_size) + list(box_center) + list(paint_shape) + [gf]])
    flags = (1 if curved_sky else 0) | (2 if lightcone else 0)
    tables, nchi, ngrow = None, 0, 0
    if lightcone:
        d, gtab = nbody._dist_cache(cosmo), nbody._growth_cache(cosmo)
        nchi, ngrow = len(d["chi"]), len(gtab["a"])
        tables = torch.from_numpy(np.concatenate([d["chi"][::-1], d["a"][::-1], gtab["a"], gtab["g"], gtab["f"]])).to(p.device)
    out = torch.empty((n, 3), dtype=torch.float32, device=p.device)
    plan.call("mcpm_observe_pos_f32", nbody._ptr(p), nbody._ptr(v), nbody._ptr(dv), n, mode, geom, flags, nbody._ptr(tables), nchi,
              ngrow, nbody._ptr(out))
    res = nbody.LatticePos(out, paint_shape, pos.ptcl_shape) if isinstance(pos, nbody.LatticePos) else out
    if return_ctx:
        return res, ObsCtx(plan=plan, p=p, v=v, dv=dv, n=n, mode=mode, geom=geom, flags=flags, tables=tables, nchi=nchi, ngrow=ngrow)
    return res


def observe_pos_vjp(ctx, out_bar):
    """VJP of observe_pos: cotangent of the returned positions (N,3) -> (pos_bar, vel_bar, dvel_bar or None, gf_bar) where
    gf_bar is the cotangent of the scalar a2g(a_obs) a2f(a_obs) (0.0 on the light cone)."""
    import torch
    from . import nbody
    n, dev = ctx.n, ctx.p.device
    ob = nbody._f32(out_bar, (n, 3))
    pb = torch.empty((n, 3), dtype=torch.float32, device=dev)
    vb = torch.empty((n, 3), dtype=torch.float32, device=dev)
    db = torch.empty((n, 3), dtype=torch.float32, device=dev) if ctx.dv is not None else None
    gfb = torch.zeros(1, dtype=torch.float64, device=dev)
    ctx.plan.call("mcpm_observe_pos_vjp_f32", nbody._ptr(ctx.p), nbody._ptr(ctx.v), nbody._ptr(ctx.dv), n, ctx.mode, ctx.geom, ctx.flags,
                  nbody._ptr(ctx.tables), ctx.nchi, ctx.ngrow, nbody._ptr(ob), nbody._ptr(pb), nbody._ptr(vb), nbody._ptr(db), nbody._ptr(gfb))
    return pb, vb, db, float(gfb.item())


# ------------------------------------------------------------------------------------------------
# Sample mesh -> base mesh (bricks.py:290-320)
def samp2base_mesh(init: dict, precond, transfer, inv=False, temp=1.) -> dict:
    """Transform the sample mesh into the base mesh, i.e. the initial wavevector coefficients (bricks.py:290-320):
    'real': rfftn(mesh) * transfer; 'fourier' / 'kaiser': rg2cgh(mesh) * transfer; and the inverse.  `transfer` is the
    model's (fiducial, fixed) real k-space array; the permutation runs in mcpm_rg2cgh_f32 / mcpm_cgh2rg_f32."""
    import torch
    from . import nbody, utils
    assert len(init) <= 1, "init dict should only have one or zero key"
    for in_name, mesh in init.items():
        out_name = in_name + '_' if inv else in_name[:-1]
        tr = torch.as_tensor(np.asarray(transfer, dtype=np.float32) * temp ** .5, device=nbody._device())
        if not inv:
            mesh = nbody.rfftn(mesh) if precond == 'real' else utils.rg2cgh(mesh)
            mesh = mesh * tr
        else:
            mesh = nbody._c64(mesh)
            mesh = torch.where(tr != 0, mesh / torch.where(tr != 0, tr, torch.ones_like(tr)), torch.zeros_like(mesh))
            mesh = nbody.irfftn(mesh) if precond == 'real' else utils.cgh2rg(mesh)
        return {out_name: mesh}
    return {}


def samp2base_mesh_vjp(base_bar, precond, transfer, temp=1.):
    """VJP of samp2base_mesh (forward direction): cotangent of the base mesh (complex, real-pair convention) -> cotangent
    of the real sample mesh."""
    import torch
    from . import nbody, utils
    tr = torch.as_tensor(np.asarray(transfer, dtype=np.float32) * temp ** .5, device=nbody._device())
    kb = nbody._c64(base_bar) * tr
    if precond == 'real':        # adjoint of rfftn under the real-pair convention: unnormalised C2R of the cotangent
        kb = kb.clone()
        shape = utils.ch2rshape(kb.shape)
        plan = nbody.get_plan(shape)
        # irfftn's multiplicity weights must not be applied: halve the doubly counted modes first
        kb[..., 1:shape[-1] // 2] *= 0.5
        out = torch.empty(shape, dtype=torch.float32, device=kb.device)
        plan.call("mcpm_fft_c2r", nbody._ptr(kb), nbody._ptr(out), 1)
        return out
    return utils.rg2cgh_vjp(kb)
