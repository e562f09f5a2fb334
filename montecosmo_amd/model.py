"""`FieldLevelModel.evolve` (montecosmo/model.py:686-838) on the HIP path, with its hand-written reverse sweep.

Built branch: bias_type 'lagrangian', evolution 'lpt' (scalar a_obs or light cone), 'nbody' (scalar a_obs, as the
reference asserts) or 'kaiser' (flat sky, scalar a_obs: bricks.py:170-198), png_type None, ap_auto None, kernel_type 'rectangular', linear power from a table (`lin_kpow`,
bricks.py:75-77) or, with lin_kpow = None, from the Eisenstein-Hu fit of the current cosmology (bricks.py:72-74; power.py).
Priors, likelihood and samplers: logdensity.py, samplers.py.

    fwd = FieldLevelForward(final_shape=(64, 64, 64), cell_length=20., box_center=(0, 0, 2000.), evolution='nbody',
                            a_obs=0.7, lin_kpow=(ks, pows))
    gxy_mesh, ctx = fwd.evolve(cosmology, bias, white_mesh, return_ctx=True)       # gxy_mesh = 1 + delta_obs
    grads = fwd.evolve_vjp(ctx, gxy_mesh_bar)      # {'white_mesh': ..., 'bias': {...}, 'sigma8': ..., 'growth': ...}

Chain (every arrow is a HIP kernel sequence of libmcpm.so, each with its VJP):
white_mesh -white2lin-> init_mesh -chreshape-> evol mesh -lagrangian_bias-> (weights, dvel); -lpt | nbody_bf-> (pos, vel)
-observe_pos (los, rsd)-> pos on init_shape -nufft(weights, paint_shape)-> spectrum -chreshape-> -irfftn-> gxy_mesh.
"""
from __future__ import annotations

import numpy as np
import torch

from . import nbody, bricks
from .utils import scale_shape, r2chshape, chreshape, chreshape_vjp


class EvolveCtx:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class FieldLevelForward:
    def __init__(self, final_shape=(64, 64, 64), cell_length=20., box_center=(0., 0., 0.), box_rotvec=(0., 0., 0.),
                 evolution='lpt', nbody_a_start=0., nbody_n_steps=10, lpt_order=2, paint_order=2, paint_deconv=True,
                 init_oversamp=3 / 2, evol_oversamp=7 / 4, ptcl_oversamp=7 / 4, paint_oversamp=7 / 4, interlace_order=2,
                 a_obs=None, curved_sky=True, lin_kpow=None):
        if evolution not in ('kaiser', 'lpt', 'nbody'):
            raise ValueError("evolution must be 'kaiser', 'lpt' or 'nbody'")
        if evolution == 'kaiser' and (curved_sky or a_obs is None):
            raise NotImplementedError("the Kaiser model is built for the flat sky at fixed a_obs (bricks.py:194-198); "
                                      "its curved-sky / light-cone forms (bricks.py:200-231) are not")
        self.final_shape = tuple(int(s) for s in final_shape)
        self.cell_length = float(cell_length)
        self.box_center = np.asarray(box_center, dtype=np.float64)
        self.box_rotvec = np.asarray(box_rotvec, dtype=np.float64)
        self.box_size = np.multiply(self.final_shape, self.cell_length)            # model.py:568-573
        self.init_shape = scale_shape(self.final_shape, init_oversamp)
        self.evol_shape = scale_shape(self.final_shape, evol_oversamp)
        self.ptcl_shape = scale_shape(self.final_shape, ptcl_oversamp)
        self.paint_shape = scale_shape(self.final_shape, paint_oversamp)
        self.evolution, self.nbody_a_start, self.nbody_n_steps = evolution, float(nbody_a_start), int(nbody_n_steps)
        self.lpt_order, self.paint_order, self.paint_deconv = int(lpt_order), int(paint_order), bool(paint_deconv)
        self.interlace_order, self.a_obs, self.curved_sky = int(interlace_order), a_obs, bool(curved_sky)
        if evolution == 'nbody' and a_obs is None:
            raise NotImplementedError("N-body light-cone not implemented (model.py:770)")
        # lin_kpow = (ks, pows) normalised to sigma8 = 1, or None: the Eisenstein-Hu power of the CURRENT cosmology
        # (bricks.py:69-79; montecosmo_amd/power.py), re-tabulated whenever Omega_m / Omega_b / h / n_s change
        self.lin_kpow = None if lin_kpow is None else (np.asarray(lin_kpow[0], dtype=np.float64), np.asarray(lin_kpow[1], dtype=np.float64))
        self._dev_kpow = {}
        self._r0 = None

    def config(self):
        """The model attributes as a dict (what the parity tests hand to their float64 checker)."""
        keys = ("init_shape", "evol_shape", "ptcl_shape", "paint_shape", "box_size", "box_center", "box_rotvec", "a_obs",
                "curved_sky", "evolution", "nbody_a_start", "nbody_n_steps", "lpt_order", "paint_order", "paint_deconv",
                "interlace_order", "lin_kpow")
        return {k: getattr(self, k) for k in keys}

    # ---- pieces ------------------------------------------------------------------------------------------
    def _kphys(self, shape):
        return [float(s) / float(b) for s, b in zip(shape, self.box_size)]

    def kpow(self, cosmo):
        """(ks, pows) normalised to sigma8 = 1 for this cosmology: the given table, or Eisenstein-Hu (power.py)."""
        if self.lin_kpow is not None:
            return self.lin_kpow
        from . import power
        return power.lin_power_table(cosmo)

    def _power_mult(self, spec, cosmo, sigma8=None):
        """white2lin (bricks.py:149-154): spec * sqrt(sigma8^2 P(|k|)); real multiplier, self-adjoint.  `sigma8` overrides the
        cosmology's (1.0 gives d init_mesh / d sigma8, which stays finite where a bounded latent has walked to sigma8 = 0)."""
        key = None if self.lin_kpow is not None else (float(cosmo.Omega_c), float(cosmo.Omega_b), float(cosmo.h), float(cosmo.n_s))
        tab = self._dev_kpow.get(key)
        if tab is None:
            if len(self._dev_kpow) > 8:
                self._dev_kpow.clear()
            tab = self._dev_kpow[key] = torch.from_numpy(np.concatenate(self.kpow(cosmo))).to(spec.device)
        nt = tab.numel() // 2
        plan = nbody.get_plan(self.init_shape)
        out = torch.empty_like(spec)
        kp = self._kphys(self.init_shape)
        plan.call("mcpm_power_mult_f32", nbody._ptr(spec), kp[0], kp[1], kp[2], float(cosmo.sigma8 if sigma8 is None else sigma8) ** 2, nbody._ptr(tab),
                  nbody.C.c_void_p(tab.data_ptr() + 8 * nt), nt, nbody._ptr(out))
        return out

    def _scale_factors(self, cosmo):
        """Scale factor(s) of the Lagrangian lattice (model.py:741-742): a_obs, or chi2a(|x|) per particle as a device
        tensor (N,1).  The lattice is fixed, so its physical distances are computed once (host float64) and kept on
        the device; the cosmology-dependent look-up runs there (mcpm_interp_f32)."""
        if self.a_obs is not None:
            return self.a_obs
        if self._r0 is None:
            pos = bricks.regular_pos(self.evol_shape, self.ptcl_shape)
            p = bricks.cell2phys_pos(pos, self.box_center, self.box_rotvec, self.box_size, self.evol_shape)
            if self.curved_sky:
                r0 = np.linalg.norm(p, axis=-1)
            else:
                los = nbody.safe_div(self.box_center, np.linalg.norm(self.box_center))
                r0 = np.abs((p * los).sum(-1))
            self._r0 = nbody._f32(r0)
        d = nbody._dist_cache(cosmo)
        return nbody.interp_dev(self._r0, d["chi"][::-1], d["a"][::-1]).reshape(-1, 1)

    # ---- Kaiser model (bricks.py:170-198, flat sky, fixed a): growth, Eulerian linear bias and RSD, diagonal in k -------
    def _mu2_mesh(self, device):
        """(k . los)^2 / k^2 on the half-spectrum of the evolution mesh, los = the box centre's direction in cell axes."""
        if getattr(self, "_mu2", None) is None:
            los = bricks.rot_matrix(self.box_rotvec).T @ nbody.safe_div(self.box_center, np.linalg.norm(self.box_center))
            kvec = nbody.rfftk(self.evol_shape, self.box_size)
            kk = sum(k ** 2 for k in kvec)
            mu2 = nbody.safe_div(sum(k * l for k, l in zip(kvec, los)) ** 2, kk)
            self._mu2 = torch.from_numpy(np.ascontiguousarray(mu2, dtype=np.float32)).to(device)
        return self._mu2

    def _kaiser(self, cosmo, bias, white, evol_k, return_ctx):
        D, f = float(nbody.a2g(cosmo, self.a_obs)), float(nbody.a2f(cosmo, self.a_obs))
        mu2 = self._mu2_mesh(evol_k.device)
        boost = D * ((1.0 + float(bias["b1"])) + f * mu2)                 # b1E = 1 + b1 (bricks.py:454)
        gxy = nbody.irfftn(evol_k * boost) + 1.0
        cosmo._workspace = {}
        if return_ctx:
            return gxy, EvolveCtx(cosmo=cosmo, white=white, evol_k=evol_k, kaiser=(D, f, float(bias["b1"]), boost))
        return gxy

    def _kaiser_vjp(self, ctx, gxy_bar):
        cosmo = ctx.cosmo
        D, f, b1, boost = ctx.kaiser
        gb = nbody._f32(gxy_bar, self.evol_shape)
        kb = nbody.rfftn(gb) / float(np.prod(self.evol_shape))           # irfftn adjoint (real-pair convention)
        kb[..., 1:self.evol_shape[-1] // 2] *= 2.0
        prod = kb.conj() * ctx.evol_k
        c0 = float(prod.real.double().sum())                              # d/d(D b1E)
        c1 = float((prod.real * self._mu2_mesh(kb.device)).double().sum())    # d/d(D f)
        init_b = chreshape_vjp(kb * boost, r2chshape(self.init_shape))
        white_b = self._power_mult(init_b, cosmo)
        s8b = float((init_b.conj() * self._power_mult(ctx.white, cosmo, sigma8=1.0)).real.sum().item())
        bias_bar = {k: 0.0 for k in bricks.BIAS_KEYS}
        bias_bar["b1"] = D * c0
        return {"white_mesh": white_b, "bias": bias_bar, "sigma8": s8b, "init_bar": init_b,
                "kaiser": {"g": (1.0 + b1) * c0 + f * c1, "f": D * c1}}

    # ---- forward -----------------------------------------------------------------------------------------
    def evolve(self, cosmo, bias, white_mesh, return_ctx=False):
        """cosmo: duck-typed cosmology (Omega_m, Omega_de, Omega_k, w0, wa, sigma8, _workspace); bias: dict of the
        Lagrangian bias parameters; white_mesh: complex half-spectrum of shape r2chshape(init_shape) (what
        samp2base_mesh returns).  Returns gxy_mesh (paint_shape, float32 device tensor) = 1 + delta_obs."""
        white = nbody._c64(white_mesh, r2chshape(self.init_shape))
        init_k = self._power_mult(white, cosmo)
        evol_k = chreshape(init_k, r2chshape(self.evol_shape))
        if self.evolution == 'kaiser':      # gxy_mesh lives on the evolution mesh (model.py:690-696: no oversampling needed)
            return self._kaiser(cosmo, bias, white, evol_k, return_ctx)
        pos0 = getattr(self, "_pos0", None)      # the undisplaced lattice: built once (201 MB of zeros per call at 256^3 otherwise); never written to
        if pos0 is None:
            pos0 = self._pos0 = nbody.LatticePos.regular(self.evol_shape, self.ptcl_shape)
        a = self._scale_factors(cosmo)
        (w, dvel, _), bctx = bricks.lagrangian_bias(cosmo, pos0, a, self.box_size, evol_k, bias, read_order=1, return_ctx=True)
        cosmo._workspace = {}                                                        # model.py:762, :769
        if self.evolution == 'lpt':
            (dpos, vel), lctx = nbody.lpt(cosmo, evol_k, pos0, a, lpt_order=self.lpt_order, read_order=1, return_ctx=True)
            pos, nctx = pos0 + dpos, lctx      # (the LPT context rides in the N-body context's slot)
        else:
            (pos, vel), nctx = nbody.nbody_bf(cosmo, evol_k, pos0, a0=self.nbody_a_start, a1=a, n_steps=self.nbody_n_steps,
                                              paint_order=self.paint_order, lpt_order=self.lpt_order, return_ctx=True,
                                              lattice_out=True)
            vel = vel.reshape(-1, 3)
        # the growth-table Jacobian of cosmo_vjp: a millisecond of host work, done HERE -- the device has the whole evolution queued
        fd = self._cosmo_scalar_fd(cosmo, self.cosmo_fd_params) if (return_ctx and self.a_obs is not None and getattr(self, "cosmo_fd_params", None)) else None
        pos_c, octx = bricks.observe_pos(cosmo, pos, vel, self.box_center, self.box_rotvec, self.box_size, self.evol_shape,
                                         self.init_shape, a_obs=self.a_obs, curved_sky=self.curved_sky, dvel=dvel, return_ctx=True)
        gxy_k = nbody.nufft(pos_c, self.init_shape, self.paint_shape, weights=w, paint_order=self.paint_order,
                            interlace_order=self.interlace_order, paint_deconv=self.paint_deconv)
        jac = float(np.divide(self.init_shape, self.ptcl_shape).prod())
        gxy_k = chreshape(gxy_k * jac, r2chshape(self.paint_shape))
        gxy = nbody.irfftn(gxy_k)
        if return_ctx:
            return gxy, EvolveCtx(cosmo=cosmo, white=white, evol_k=evol_k, pos0=pos0, a=a, bctx=bctx, nctx=nctx, octx=octx,
                                  pos_c=pos_c, w=w, jac=jac, scalar_fd=fd)
        return gxy

    # ---- reverse sweep -----------------------------------------------------------------------------------
    def evolve_vjp(self, ctx, gxy_bar):
        """Cotangent of gxy_mesh (real, paint_shape) -> {'white_mesh': complex64 cotangent (real-pair convention),
        'bias': dict, 'sigma8': float, 'growth': cotangents of the growth scalars (see nbody.lpt_vjp / nbody_bf_vjp),
        'bias_growth': cotangent(s) of a2g(a) through the bias weights, 'gf': cotangent of a2g(a_obs) a2f(a_obs) through rsd}."""
        if self.evolution == 'kaiser':
            return self._kaiser_vjp(ctx, gxy_bar)
        cosmo = ctx.cosmo
        gb = nbody._f32(gxy_bar, self.paint_shape)
        # irfftn adjoint: X_bar = (w / M) rfftn(y_bar)
        Mp = float(np.prod(self.paint_shape))
        kb = nbody.rfftn(gb) / Mp
        kb[..., 1:self.paint_shape[-1] // 2] *= 2.0
        kb = chreshape_vjp(kb, r2chshape(self.init_shape)) * ctx.jac
        pb, wb = nbody.nufft_vjp(ctx.pos_c, self.init_shape, ctx.w, kb, self.paint_order, self.interlace_order, self.paint_deconv,
                                 paint_shape=self.paint_shape)
        xb, vb, dvb, gfb = bricks.observe_pos_vjp(ctx.octx, pb)
        mesh_b, bias_bar, bg_bar = bricks.lagrangian_bias_vjp(ctx.bctx, wb, dvb)
        if self.evolution == 'lpt':
            mb, growth = nbody.lpt_vjp(cosmo, ctx.evol_k, ctx.pos0, ctx.a, xb, vb, lpt_order=self.lpt_order, ctx=ctx.nctx)
        else:
            mb, growth = nbody.nbody_bf_vjp(ctx.nctx, xb, vb)
        mesh_b = mesh_b + mb
        init_b = chreshape_vjp(mesh_b, r2chshape(self.init_shape))
        white_b = self._power_mult(init_b, cosmo)
        # d/d sigma8: init_mesh is linear in sigma8
        s8b = float((init_b.conj() * self._power_mult(ctx.white, cosmo, sigma8=1.0)).real.sum().item())
        return {"white_mesh": white_b, "bias": bias_bar, "sigma8": s8b, "growth": growth, "bias_growth": bg_bar, "gf": gfb,
                "init_bar": init_b, "obs_bar": pb if self.a_obs is None else None}

    def cosmo_vjp(self, ctx, grads, params=("Omega_m",), rel_eps=1e-5):
        """Chains the growth cotangents of `evolve_vjp` to cosmological parameters (fixed a_obs): besides sigma8 and the
        Eisenstein-Hu table (when no `lin_kpow` is given; central differences of the 256-point table, applied to the white
        field on the device) the cosmology enters evolve through host float64 scalars looked up in the 128-point growth tables -- the
        BullFrog coefficients and the 2LPT start (nbody.cosmo_vjp), a2g(a_obs) in the bias weights, a2g a2f in the
        RSD -- so dL/dtheta = sum_s s_bar ds/dtheta with the table Jacobian taken by central finite differences.
        `params`: attribute names of the cosmology object; 'Omega_m' varies Omega_c at fixed Omega_b.  On the light cone
        (a_obs = None) the look-ups are per particle: `_cosmo_vjp_lightcone`."""
        import copy
        if self.a_obs is None:
            return self._cosmo_vjp_lightcone(ctx, grads, params, rel_eps)
        cosmo, a = ctx.cosmo, self.a_obs
        scalars = self._cosmo_scalars
        pre = getattr(ctx, "scalar_fd", None) or {}

        if self.evolution == 'kaiser':
            g, bars = None, [float(grads["kaiser"]["g"]), float(grads["kaiser"]["f"])]
        else:
            g = grads["growth"]
            bars = [float(np.asarray(grads["bias_growth"]).sum()), float(grads["gf"])]
        if self.evolution == 'kaiser':
            pass
        elif self.evolution == 'lpt':
            bars += [float(g["g"]), float(g["g2"]), float(g["dg2dg"])]
        else:
            bars += [float(g["dg"])] + list(g["alpha"]) + list(g["beta"]) + [float(g["g"]), float(g["g2"]), float(g["dg2dg"])]
        bars = np.array(bars)
        out = {}
        for name in params:
            attr = "Omega_c" if name == "Omega_m" else name
            base = float(getattr(cosmo, attr))
            h = rel_eps * max(abs(base), 1e-2)
            vals, inits = [], []
            cached = pre.get((name, rel_eps))      # the table Jacobian's two evaluations, made while the device ran the forward pass
            for sgn in (+1, -1):
                c = copy.copy(cosmo)
                setattr(c, attr, base + sgn * h)
                vals.append(cached[len(vals)] if cached is not None else scalars(c))
                if self.lin_kpow is None:      # the Eisenstein-Hu shape moves with the cosmology: init_mesh = white sqrt(P)
                    inits.append(self._power_mult(ctx.white, c))
            out[name] = float(np.dot(bars, (vals[0] - vals[1]) / (2 * h)))
            if inits:
                out[name] += float((grads["init_bar"].conj() * (inits[0] - inits[1])).real.sum().item()) / (2 * h)
        cosmo._workspace = {}
        return out

    def _cosmo_scalars(self, c):
        """The host float64 scalars through which a cosmology enters `evolve` at fixed a_obs (see cosmo_vjp)."""
        a = self.a_obs
        c._workspace = {}
        if self.evolution == 'kaiser':
            return np.array([float(nbody.a2g(c, a)), float(nbody.a2f(c, a))])
        out = [float(nbody.a2g(c, a)), float(nbody.a2g(c, a) * nbody.a2f(c, a))]
        if self.evolution == 'lpt':
            out += [float(nbody.a2g(c, a)), float(nbody.a2g2(c, a)), float(nbody.a2dg2dg(c, a))]
        else:
            dg, al, be, ls = nbody._step_scalars(c, self.nbody_a_start, a, self.nbody_n_steps, "bullfrog")
            out += [dg] + list(al) + list(be) + list(ls)
        return np.array(out)

    def _cosmo_scalar_fd(self, cosmo, params, rel_eps=1e-5):
        """{(name, rel_eps): (scalars at +h, scalars at -h)}: the two evaluations of cosmo_vjp's central difference.  Host work of about
        a millisecond (two growth-table solves per parameter) that `evolve` does right after it has queued the forward pass, while the
        device runs it; left to cosmo_vjp it sits at the very end of a gradient, with the device idle (`cosmo_fd_params`)."""
        import copy
        out = {}
        for name in params:
            attr = "Omega_c" if name == "Omega_m" else name
            base = float(getattr(cosmo, attr))
            h = rel_eps * max(abs(base), 1e-2)
            pair = []
            for sgn in (+1, -1):
                c = copy.copy(cosmo)
                setattr(c, attr, base + sgn * h)
                pair.append(self._cosmo_scalars(c))
            out[(name, rel_eps)] = tuple(pair)
        return out      # (the copies carry their own tables: `cosmo`'s cached ones stay for the rest of evolve)

    # ---- light cone: the cosmology enters through per-particle table look-ups -----------------------------------------
    _LC_TABLES = ("chi", "g", "g2", "f", "f2")

    def _lightcone_tables(self, cosmo):
        """The five cosmology-dependent tables the light-cone look-ups read (host float64): chi ascending (the nodes of
        chi2a, nbody.py:862-884; its values, the scale-factor grid, do not move) and g, g2 (raw), f, f2 on the growth grid."""
        d, gt = nbody._dist_cache(cosmo), nbody._growth_cache(cosmo)
        return {"chi": d["chi"][::-1].copy(), "g": gt["g"], "g2": gt["g2"], "f": gt["f"], "f2": gt["f2"]}

    def lightcone_table_bars(self, ctx, grads):
        """Cotangents of those tables (dict of float64 arrays): the Lagrangian look-ups a_q = chi2a(r0_q) -> a2g (bias weights,
        model.py:756, and lpt), a2g2, a2dg2dg (lpt, nbody.py:652-666) contracted with the per-particle cotangents of
        `evolve_vjp` (mcpm_lightcone_tables_vjp_f32), plus the observation-side a2g a2f at the evolved positions
        (model.py:781-784; mcpm_observe_pos_tables_vjp_f32)."""
        cosmo = ctx.cosmo
        d, gt = nbody._dist_cache(cosmo), nbody._growth_cache(cosmo)
        nchi, ng = len(d["chi"]), len(gt["a"])
        dev = ctx.evol_k.device
        tabs = torch.from_numpy(np.concatenate([d["chi"][::-1], d["a"][::-1], gt["a"], gt["g"], gt["g2"], gt["f"], gt["f2"]])).to(dev)
        plan = nbody.get_plan(self.evol_shape, self.ptcl_shape)
        n = plan.N
        g = grads["growth"]
        gB = (nbody._f32(grads["bias_growth"]).reshape(-1) + nbody._f32(g["g"]).reshape(-1)).contiguous()
        g2B, dB = nbody._f32(g["g2"]).reshape(-1).contiguous(), nbody._f32(g["dg2dg"]).reshape(-1).contiguous()
        tbL = torch.empty(nchi + 4 * ng, dtype=torch.float64, device=dev)
        plan.call("mcpm_lightcone_tables_vjp_f32", nbody._ptr(self._r0), n, nbody._ptr(tabs), nchi, ng, nbody._ptr(gB), nbody._ptr(g2B),
                  nbody._ptr(dB), nbody._ptr(tbL))
        o = ctx.octx
        tbO = torch.empty(nchi + 2 * ng, dtype=torch.float64, device=dev)
        o.plan.call("mcpm_observe_pos_tables_vjp_f32", nbody._ptr(o.p), nbody._ptr(o.v), nbody._ptr(o.dv), o.n, o.mode, o.geom, o.flags,
                    nbody._ptr(o.tables), o.nchi, o.ngrow, nbody._ptr(grads["obs_bar"]), nbody._ptr(tbO))
        L, O = tbL.cpu().numpy(), tbO.cpu().numpy()
        out = {"chi": L[:nchi] + O[:nchi], "g": L[nchi:nchi + ng] + O[nchi:nchi + ng], "g2": L[nchi + ng:nchi + 2 * ng],
               "f": L[nchi + 2 * ng:nchi + 3 * ng] + O[nchi + ng:], "f2": L[nchi + 3 * ng:]}
        return out

    def _cosmo_vjp_lightcone(self, ctx, grads, params, rel_eps):
        """cosmo_vjp on the light cone (a_obs = None, the reference's default configuration, model.py:45, :62): dL/dtheta =
        sum over the five tables of <table_bar, d table / d theta>, the table Jacobian by central differences of the host
        float64 RK4 tables (256-point distance table, 128-point growth tables), plus the Eisenstein-Hu term as at fixed a_obs."""
        import copy
        if self.evolution != 'lpt':
            raise NotImplementedError("light cone is built for evolution='lpt' (model.py:770 asserts the same for 'nbody')")
        cosmo = ctx.cosmo
        bars = self.lightcone_table_bars(ctx, grads)
        out = {}
        for name in params:
            attr = "Omega_c" if name == "Omega_m" else name
            base = float(getattr(cosmo, attr))
            h = rel_eps * max(abs(base), 1e-2)
            tabs, inits = [], []
            for sgn in (+1, -1):
                c = copy.copy(cosmo)
                c._workspace = {}
                setattr(c, attr, base + sgn * h)
                tabs.append(self._lightcone_tables(c))
                if self.lin_kpow is None:
                    inits.append(self._power_mult(ctx.white, c))
            out[name] = float(sum(np.dot(bars[k], (tabs[0][k] - tabs[1][k]) / (2 * h)) for k in self._LC_TABLES))
            if inits:
                out[name] += float((grads["init_bar"].conj() * (inits[0] - inits[1])).real.sum().item()) / (2 * h)
        return out
