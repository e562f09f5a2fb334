"""
TEST INFRASTRUCTURE (see oracle/__init__.py) -- numpy float64 restatement of the operators either side of the PM path
inside `FieldLevelModel.evolve` (montecosmo/model.py:740-810): the Lagrangian bias expansion weights, the
cell <-> physical coordinate maps, line of sight / scale factor, and redshift-space distortions, with hand-derived
VJPs.  Paths are relative to /root/reference.  Primordial non-Gaussianity (png_type != None), Alcock-Paczynski and the
Eulerian bias branch are not restated.

scipy.spatial.transform.Rotation (box_rot) is replaced by its 3x3 matrix R: `box_rot.apply(x)` = x @ R.T,
`apply(x, inverse=True)` = x @ R (scipy's documented convention).
"""
import numpy as np

from . import pm_oracle as o

BIAS_KEYS = ("b1", "b2", "bs2", "b3", "bds2", "bs3", "bn2", "bnpar")


def rotvec_matrix(rotvec):
    """Rotation matrix of a rotation vector (Rodrigues), = scipy Rotation.from_rotvec(rotvec).as_matrix()."""
    v = np.asarray(rotvec, dtype=float)
    th = np.linalg.norm(v)
    if th == 0:
        return np.eye(3)
    k = v / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


# --------------------------------------------------------------------------- bias
def bias_fields(lin_mesh, box_size):
    """The real-space fields lagrangian_bias reads (bricks.py:350-352, :372-399, :438-441):
    delta, shear^2, 3 det(shear), laplacian(delta) (h/Mpc)^2, grad(delta) (h/Mpc).  Returns (fields dict, parts)."""
    delta = o._irfftn(lin_mesh)
    shape = delta.shape
    kvec = o.rfftk(shape, box_size)
    k2 = sum(ki ** 2 for ki in kvec)
    pot = lin_mesh * o.invlaplace_hat(kvec)
    sh = {}
    for i in range(2):
        nabi = o.gradient_hat(kvec, i)
        sh[(i, i)] = o._irfftn(nabi ** 2 * pot - lin_mesh / 3)
        for j in range(i + 1, 3):
            sh[(i, j)] = o._irfftn(nabi * o.gradient_hat(kvec, j) * pot)
    sh[(2, 2)] = -(sh[(0, 0)] + sh[(1, 1)])
    a, b, c = sh[(0, 0)], sh[(1, 1)], sh[(2, 2)]
    d, e, f = sh[(0, 1)], sh[(0, 2)], sh[(1, 2)]
    shear2 = a ** 2 + b ** 2 + c ** 2 + 2 * (d ** 2 + e ** 2 + f ** 2)
    shear3 = 3 * (a * (b * c - f ** 2) - d * (d * c - e * f) + e * (d * f - b * e))
    nab2 = o._irfftn(-k2 * lin_mesh)
    grad = [o._irfftn(o.gradient_hat(kvec, i) * lin_mesh) for i in range(3)]
    return dict(delta=delta, shear2=shear2, shear3=shear3, nab2=nab2, grad=grad), (a, b, c, d, e, f)


def lagrangian_bias(growths, pos, box_size, lin_mesh, bias, read_order=2):
    """bricks.py:327-443 with png_type=None.  `growths` = a2g(cosmo, a): scalar or (N,1).  Returns (weights (N,),
    dvel (N,3))."""
    fld, _ = bias_fields(lin_mesh, box_size)
    g = np.asarray(growths, dtype=float)
    gs = g.squeeze()
    delta_pos = o.read(pos, fld["delta"], read_order) * gs
    weights = 1. + bias["b1"] * delta_pos
    delta2_pos = delta_pos ** 2
    sigma2 = delta2_pos.mean()
    delta2_pos = delta2_pos - sigma2
    weights = weights + bias["b2"] * delta2_pos / 2
    shear2_pos = o.read(pos, fld["shear2"], read_order) * gs ** 2
    shear2_pos = shear2_pos - 2 / 3 * sigma2
    weights = weights + bias["bs2"] * shear2_pos
    delta3_pos = delta_pos ** 3 - 3 * sigma2 * delta_pos
    weights = weights + bias["b3"] * delta3_pos / 6
    weights = weights + bias["bds2"] * delta_pos * shear2_pos
    shear3_pos = o.read(pos, fld["shear3"], read_order) * gs ** 3
    weights = weights + bias["bs3"] * shear3_pos
    weights = weights + bias["bn2"] * o.read(pos, fld["nab2"], read_order) * gs
    nabpar = np.stack([o.read(pos, fld["grad"][i], read_order) for i in range(3)], axis=-1)
    dvel = bias["bnpar"] * nabpar * (g if g.ndim else gs)
    return weights, dvel


def lagrangian_bias_vjp(growths, pos, box_size, lin_mesh, bias, w_bar, dvel_bar, read_order=2):
    """VJP of lagrangian_bias w.r.t. (lin_mesh, bias dict, growths) -- positions are the fixed Lagrangian lattice.
    Returns (lin_mesh_bar [real-pair convention], bias_bar dict, growths_bar [shape of growths])."""
    fld, (a, b, c, d, e, f) = bias_fields(lin_mesh, box_size)
    shape = fld["delta"].shape
    N = len(pos)
    g = np.asarray(growths, dtype=float)
    gs = np.broadcast_to(g.squeeze(), (N,)).astype(float)
    dr = o.read(pos, fld["delta"], read_order)
    s2r = o.read(pos, fld["shear2"], read_order)
    s3r = o.read(pos, fld["shear3"], read_order)
    lr = o.read(pos, fld["nab2"], read_order)
    gr = np.stack([o.read(pos, fld["grad"][i], read_order) for i in range(3)], axis=-1)
    dl = dr * gs
    sigma2 = (dl ** 2).mean()
    s2 = s2r * gs ** 2 - 2 / 3 * sigma2
    s3 = s3r * gs ** 3
    ll = lr * gs
    B = bias
    wb = np.asarray(w_bar, dtype=float)
    vb = np.asarray(dvel_bar, dtype=float)
    bias_bar = dict(b1=(wb * dl).sum(), b2=(wb * (dl ** 2 - sigma2) / 2).sum(), bs2=(wb * s2).sum(),
                    b3=(wb * (dl ** 3 - 3 * sigma2 * dl) / 6).sum(), bds2=(wb * dl * s2).sum(), bs3=(wb * s3).sum(),
                    bn2=(wb * ll).sum(), bnpar=(vb * gr * gs[:, None]).sum())
    dw_ds2 = B["bs2"] + B["bds2"] * dl
    dw_dsig = -B["b2"] / 2 - B["b3"] * dl / 2 - 2 / 3 * dw_ds2
    sig_bar = (wb * dw_dsig).sum()
    d_bar = wb * (B["b1"] + B["b2"] * dl + B["b3"] * (dl ** 2 - sigma2) / 2 + B["bds2"] * s2) + sig_bar * 2 * dl / N
    dr_bar = d_bar * gs
    s2r_bar = wb * dw_ds2 * gs ** 2
    s3r_bar = wb * B["bs3"] * gs ** 3
    lr_bar = wb * B["bn2"] * gs
    gr_bar = vb * B["bnpar"] * gs[:, None]
    g_bar = d_bar * dr + wb * dw_ds2 * 2 * gs * s2r + wb * B["bs3"] * 3 * gs ** 2 * s3r + wb * B["bn2"] * lr \
        + (vb * gr).sum(-1) * B["bnpar"]
    growths_bar = g_bar.reshape(g.shape) if g.size == N else np.asarray(g_bar.sum()).reshape(g.shape)
    # reads -> meshes
    D_bar = o.read_vjp(pos, fld["delta"], dr_bar, read_order)[1]
    S2_bar = o.read_vjp(pos, fld["shear2"], s2r_bar, read_order)[1]
    S3_bar = o.read_vjp(pos, fld["shear3"], s3r_bar, read_order)[1]
    L_bar = o.read_vjp(pos, fld["nab2"], lr_bar, read_order)[1]
    G_bar = [o.read_vjp(pos, fld["grad"][i], gr_bar[:, i], read_order)[1] for i in range(3)]
    # shear2, shear3 -> a, b (c = -a-b), d, e, f
    cof = dict(a=b * c - f ** 2, b=a * c - e ** 2, c=a * b - d ** 2, d=2 * (e * f - d * c), e=2 * (d * f - b * e),
               f=2 * (d * e - a * f))
    a_bar = S2_bar * (2 * a - 2 * c) + 3 * S3_bar * (cof["a"] - cof["c"])
    b_bar = S2_bar * (2 * b - 2 * c) + 3 * S3_bar * (cof["b"] - cof["c"])
    d_bar_m = S2_bar * 4 * d + 3 * S3_bar * cof["d"]
    e_bar_m = S2_bar * 4 * e + 3 * S3_bar * cof["e"]
    f_bar_m = S2_bar * 4 * f + 3 * S3_bar * cof["f"]
    # irfftn adjoints and the k-space multipliers (all real-linear in lin_mesh; multipliers m_s: y_s = irfftn(m_s X))
    kvec = o.rfftk(shape, box_size)
    k2 = sum(ki ** 2 for ki in kvec)
    il = o.invlaplace_hat(kvec)
    nab = [o.gradient_hat(kvec, i) for i in range(3)]
    out = np.conj(1.0 + 0 * k2) * o.irfftn_vjp(D_bar)
    out = out + np.conj(nab[0] ** 2 * il - 1 / 3) * o.irfftn_vjp(a_bar)
    out = out + np.conj(nab[1] ** 2 * il - 1 / 3) * o.irfftn_vjp(b_bar)
    out = out + np.conj(nab[0] * nab[1] * il) * o.irfftn_vjp(d_bar_m)
    out = out + np.conj(nab[0] * nab[2] * il) * o.irfftn_vjp(e_bar_m)
    out = out + np.conj(nab[1] * nab[2] * il) * o.irfftn_vjp(f_bar_m)
    out = out + np.conj(-k2) * o.irfftn_vjp(L_bar)
    for i in range(3):
        out = out + np.conj(nab[i]) * o.irfftn_vjp(G_bar[i])
    return out, bias_bar, growths_bar


# --------------------------------------------------------------------------- geometry
def cell2phys_pos(pos, box_center, R, box_size, mesh_shape):
    """bricks.py:628-636"""
    pos = pos * np.divide(box_size, mesh_shape)
    pos = pos - np.asarray(box_size) / 2
    pos = pos @ np.asarray(R).T
    return pos + np.asarray(box_center)


def phys2cell_pos(pos, box_center, R, box_size, mesh_shape):
    """bricks.py:638-646"""
    pos = pos - np.asarray(box_center)
    pos = pos @ np.asarray(R)
    pos = pos + np.asarray(box_size) / 2
    return pos / np.divide(box_size, mesh_shape)


def cell2phys_vel(vel, R, box_size, mesh_shape):
    """bricks.py:648-654"""
    return (vel * np.divide(box_size, mesh_shape)) @ np.asarray(R).T


def phys2cell_vel(vel, R, box_size, mesh_shape):
    """bricks.py:656-662"""
    return (vel @ np.asarray(R)) / np.divide(box_size, mesh_shape)


def los_scalefactor_pos(pos, box_center, R, box_size, mesh_shape, cosmo, a_obs=None, curved_sky=True):
    """bricks.py:750-768: line of sight(s) and scale factor(s) of particles given in cell units."""
    p = cell2phys_pos(pos, box_center, R, box_size, mesh_shape)
    if curved_sky:
        rpos = np.linalg.norm(p, axis=-1, keepdims=True)
        los = o.safe_div(p, rpos)
    else:
        los = o.safe_div(np.asarray(box_center, dtype=float), np.linalg.norm(box_center))
        rpos = np.abs((p * los).sum(-1, keepdims=True))
    a = o.chi2a(cosmo, rpos) if a_obs is None else a_obs
    return los, a


def rsd(cosmo, vel, los, a, R, box_size, mesh_shape, dvel=0.):
    """bricks.py:791-803: displacement (Mpc/h) along the line of sight; vel is dq/dg in cell units."""
    v = cell2phys_vel(vel, R, box_size, mesh_shape)
    v = v * (o.a2g(cosmo, a) * o.a2f(cosmo, a))
    v = v + dvel
    return (v * los).sum(-1, keepdims=True) * los


def observe_pos(cosmo, pos, vel, box_center, R, box_size, evol_shape, paint_shape, a_obs=None, curved_sky=True, dvel=0.):
    """model.py:780-784, :797: evolved particles (cell units of evol_shape) -> redshift-space positions in cell units of
    paint_shape (no Alcock-Paczynski)."""
    los, a = los_scalefactor_pos(pos, box_center, R, box_size, evol_shape, cosmo, a_obs, curved_sky)
    p = cell2phys_pos(pos, box_center, R, box_size, evol_shape)
    p = p + rsd(cosmo, vel, los, a, R, box_size, evol_shape, dvel)
    return phys2cell_pos(p, box_center, R, box_size, paint_shape)


# --------------------------------------------------------------------------- evolve (model.py:686-838)
def white2lin(sigma8, white_mesh, init_shape, box_size, kpow):
    """bricks.py:83-100, :149-154 with a tabulated power spectrum normalised to sigma8 = 1."""
    ks, pows = kpow
    kvec = o.rfftk(init_shape, box_size)
    kmesh = sum(ki ** 2 for ki in kvec) ** .5
    pmesh = np.interp(kmesh.reshape(-1), ks, np.asarray(pows) * sigma8 ** 2, left=0., right=0.).reshape(kmesh.shape)
    return white_mesh * pmesh ** .5


def evolve(cfg, cosmo, bias, white_mesh):
    """FieldLevelModel.evolve (model.py:686-838) for bias_type 'lagrangian', evolution 'lpt' or 'nbody', png_type None,
    ap_auto None.  cfg: dict with the model's attributes (shapes, box, a_obs, curved_sky, orders...).  Returns the real
    galaxy mesh 1 + delta_obs of shape paint_shape, and the intermediates a test may want."""
    R = rotvec_matrix(cfg["box_rotvec"])
    box, ctr = cfg["box_size"], cfg["box_center"]
    kpow = cfg["lin_kpow"]
    if kpow is None:      # bricks.py:69-79 with kpow = None: Eisenstein-Hu power of this cosmology (oracle/power_oracle.py)
        from . import power_oracle
        kpow = power_oracle.lin_power_table(cosmo)
    init_mesh = white2lin(cosmo.sigma8, white_mesh, cfg["init_shape"], box, kpow)
    init_mesh = o.chreshape(init_mesh, o.r2chshape(cfg["evol_shape"]))
    if cfg["evolution"] == "kaiser":      # model.py:690-696 with bricks.py:186-198: flat sky, fixed a -> diagonal in k
        assert not cfg["curved_sky"] and cfg["a_obs"] is not None
        c = np.asarray(ctr, float)
        los = R.T @ o.safe_div(c, np.linalg.norm(c))
        boost = kaiser_boost(cosmo, cfg["a_obs"], cfg["evol_shape"], box, 1. + bias["b1"], los)
        cosmo._workspace = {}
        return 1. + o._irfftn(init_mesh * boost, s=tuple(cfg["evol_shape"]), axes=(0, 1, 2)), None
    pos = o.regular_pos(cfg["evol_shape"], cfg["ptcl_shape"])
    _, a = los_scalefactor_pos(pos, ctr, R, box, cfg["evol_shape"], cosmo, cfg["a_obs"], cfg["curved_sky"])
    w, dvel = lagrangian_bias(o.a2g(cosmo, a), pos, box, init_mesh, bias, read_order=1)
    if cfg["evolution"] == "lpt":
        cosmo._workspace = {}
        dpos, vel = o.lpt(cosmo, init_mesh, pos, a, lpt_order=cfg["lpt_order"], read_order=1)
        pos = pos + dpos
    else:
        cosmo._workspace = {}
        assert np.ndim(a) == 0
        p, v = o.nbody_bf(cosmo, init_mesh, pos, a0=cfg["nbody_a_start"], a1=a, n_steps=cfg["nbody_n_steps"],
                          paint_order=cfg["paint_order"], lpt_order=cfg["lpt_order"])
        pos, vel = p[-1], v[-1]
    pos_c = observe_pos(cosmo, pos, vel, ctr, R, box, cfg["evol_shape"], cfg["init_shape"], cfg["a_obs"], cfg["curved_sky"], dvel)
    gxy = o.nufft(pos_c, cfg["init_shape"], tuple(cfg["paint_shape"]), weights=w, paint_order=cfg["paint_order"],
                  interlace_order=cfg["interlace_order"], paint_deconv=cfg["paint_deconv"])
    gxy = gxy * np.divide(cfg["init_shape"], cfg["ptcl_shape"]).prod()
    gxy = o.chreshape(gxy, o.r2chshape(cfg["paint_shape"]))
    return o._irfftn(gxy, s=tuple(cfg["paint_shape"]), axes=(0, 1, 2)), dict(init_mesh=init_mesh, weights=w, pos=pos_c)


# --------------------------------------------------------------------------- log density (model.py:640-679, :840-908)
def quad_gaussian_log_prob(value, loc, scale1, scale2):
    """montecosmo/utils.py:497-510 (QuadGaussian.log_prob)."""
    a, b = np.broadcast_to(scale2, np.shape(value)).astype(float), np.broadcast_to(scale1, np.shape(value)).astype(float)
    r = value - loc + a
    D = b ** 2 + 4.0 * a * r
    D_safe = np.where(D > 0, D, 1.0)
    sq = np.sqrt(D_safe)
    a_safe = np.where(np.abs(a) < 1e-12, 1.0, a)
    ep = (-b + sq) / (2.0 * a_safe)
    em = (-b - sq) / (2.0 * a_safe)
    m = np.maximum(-0.5 * ep ** 2, -0.5 * em ** 2)
    lse = m + np.log(np.exp(-0.5 * ep ** 2 - m) + np.exp(-0.5 * em ** 2 - m))
    lp_quad = -0.5 * np.log(2 * np.pi) - 0.5 * np.log(D_safe) + lse
    lp_quad = np.where(D > 0, lp_quad, -np.inf)
    lp_gauss = -0.5 * np.log(2 * np.pi) - np.log(b) - 0.5 * ((value - loc) / b) ** 2
    return np.where(np.abs(a) < 1e-8, lp_gauss, lp_quad)


def std2trunc(x, loc=0., scale=1., low=-np.inf, high=np.inf):
    """montecosmo/utils.py:189-226: transport a standard normal variable to a truncated normal one, with the 12-sigma
    tail forms lowtail / hightail (temperature 1 / 6.2842226 / 2, utils.py:189-198) where both x and the bound on that
    side lie beyond 12 sigma (utils.py:222-225)."""
    from scipy.stats import norm
    from scipy.special import logsumexp
    lo, hi = (low - loc) / scale, (high - loc) / scale
    temp = 1 / 6.2842226 / 2
    if x < -12 and lo < -12:
        return loc + scale * temp * logsumexp(np.array([x, lo]) / temp)
    if x > 12 and hi > 12:
        return loc + scale * (-temp * logsumexp(-np.array([x, hi]) / temp))
    if x < 0:
        cl, ch = norm.cdf(lo), norm.cdf(hi)
        y = norm.ppf(cl + (ch - cl) * norm.cdf(x))
    else:
        cnl, cnh = norm.cdf(-lo), norm.cdf(-hi)
        y = -norm.ppf(cnh - (cnh - cnl) * norm.cdf(-x))
    return loc + scale * y


def detrunc_truncnorm_log_prob(x, loc, scale, low, high, loc_fid, scale_fid, h=1e-6):
    """montecosmo/utils.py:296-311 (DetruncTruncNorm.log_prob): TruncatedNormal(loc, scale, low, high).log_prob(
    std2trunc(x; fid)) + log |d std2trunc / dx| (the reference differentiates with jax.grad; central difference here)."""
    from scipy.stats import truncnorm
    y = std2trunc(x, loc_fid, scale_fid, low, high)
    jac = (std2trunc(x + h, loc_fid, scale_fid, low, high) - std2trunc(x - h, loc_fid, scale_fid, low, high)) / (2 * h)
    a_, b_ = (low - loc) / scale, (high - loc) / scale
    return truncnorm.logpdf(y, a_, b_, loc=loc, scale=scale) + np.log(abs(jac))


def kaiser_boost(cosmo, a, mesh_shape, box_size, b1E, los=(0., 0., 0.)):
    """Eulerian Kaiser boost: linear growth, linear bias, RSD (bricks.py:170-184, png_type = None)."""
    kvec = o.rfftk(mesh_shape, box_size)
    kmesh = sum(ki ** 2 for ki in kvec) ** .5
    mumesh = o.safe_div(sum(ki * li for ki, li in zip(kvec, los)), kmesh)
    return o.a2g(cosmo, a) * (b1E + o.a2f(cosmo, a) * mumesh ** 2)


def lin_power_mesh(sigma8, mesh_shape, box_size, kpow):
    """bricks.py:69-106 with a tabulated kpow (normalised to sigma8 = 1): linear interpolation, 0 outside the table."""
    kvec = o.rfftk(mesh_shape, box_size)
    kmesh = sum(ki ** 2 for ki in kvec) ** .5
    ks, pows = kpow
    return np.interp(kmesh.reshape(-1), ks, np.asarray(pows) * sigma8 ** 2, left=0., right=0.).reshape(kmesh.shape)


def fiducial_scale_factor(cfg, cosmo_fid):
    """a_fid = g2a(mean a2g(a)) over the final-mesh cells (model.py:604-606; bricks.py:665-686, :760-778)."""
    if cfg["a_obs"] is not None:
        a = cfg["a_obs"]
    else:
        R = rotvec_matrix(cfg["box_rotvec"])
        final = tuple(cfg["final_shape"])
        pos = cell2phys_pos(o.regular_pos(final), cfg["box_center"], R, cfg["box_size"], final)
        if cfg["curved_sky"]:
            r = np.linalg.norm(pos, axis=-1)
        else:
            los = o.safe_div(np.asarray(cfg["box_center"], float), np.linalg.norm(cfg["box_center"]))
            r = np.abs(pos @ los)
        a = o.chi2a(cosmo_fid, r)
    return float(o.g2a(cosmo_fid, np.mean(o.a2g(cosmo_fid, a))))


def precond_scale_and_transfer(cfg, fiduc=None, cosmo_fid=None):
    """Scale (real layout, prior std of white_mesh_) and transfer (per mode) of the white-field preconditioning
    (model.py:1127-1148); 'kaiser' with a uniform selection (selec_fid = 1)."""
    init_shape = tuple(cfg["init_shape"])
    if cfg["precond"] in ("real", "fourier"):
        scale = np.ones(o.r2chshape(init_shape))
    else:
        assert cfg["precond"] == "kaiser"
        a_fid = fiducial_scale_factor(cfg, cosmo_fid)
        c = np.asarray(cfg["box_center"], float)
        los = o.safe_div(c, np.linalg.norm(c))
        los_fid = rotvec_matrix(cfg["box_rotvec"]).T @ los
        boost = kaiser_boost(cosmo_fid, a_fid, init_shape, cfg["box_size"], 1. + fiduc["b1"], los_fid)
        kpow = cfg["lin_kpow"]
        if kpow is None:
            from . import power_oracle
            kpow = power_oracle.lin_power_table(cosmo_fid)
        pmesh = lin_power_mesh(fiduc["sigma8"], init_shape, cfg["box_size"], kpow)
        pmesh = pmesh * np.divide(init_shape, cfg["box_size"]).prod()
        count_fid = np.mean(fiduc["ngbars"]) * cfg["cell_length"] ** 3                  # model.py:602
        sel = cfg.get("selec_mesh")
        selec_fid = 1.0 if sel is None else (np.asarray(sel) ** 2).mean() ** .5 / np.asarray(sel).mean()   # model.py:609
        var_fid = fiduc["s_e"] / (count_fid * selec_fid)
        scale = (1 + boost ** 2 / var_fid * pmesh) ** .5
    transfer = np.divide(init_shape, cfg["box_size"]).prod() ** .5 / scale
    return o.cgh2rg(scale.astype(complex), norm="amp"), transfer


def radius_mesh(cfg, shape):
    """Physical distance of the cells of a mesh of `shape` spanning the box (bricks.py:665-686)."""
    R = rotvec_matrix(cfg["box_rotvec"])
    pos = cell2phys_pos(o.regular_pos(tuple(shape)), cfg["box_center"], R, cfg["box_size"], tuple(shape))
    if cfg["curved_sky"]:
        return np.linalg.norm(pos, axis=-1).reshape(shape)
    los = o.safe_div(np.asarray(cfg["box_center"], float), np.linalg.norm(cfg["box_center"]))
    return np.abs(pos @ los).reshape(shape)


def radial_edges(cfg, n_rbins=None):
    """Radial shell edges over the unmasked final-mesh cells (model.py:1087-1098)."""
    r = radius_mesh(cfg, cfg["final_shape"])
    if cfg.get("mask_mesh") is not None:
        r = r[np.asarray(cfg["mask_mesh"], bool)]
    dr = 3 ** .5 * cfg["cell_length"]
    n = max(int((r.max() - r.min()) / dr), 1) if n_rbins is None else n_rbins
    return np.linspace(r.min() - dr / 1000, r.max() + dr / 1000, n + 1)


def set_radial_count(mesh, rmesh, redges, rcounts):
    """bricks.py:1106-1122: cells of shell i are multiplied by rcounts[i]."""
    out = np.array(mesh, dtype=float)
    for c, lo, hi in zip(rcounts, redges[:-1], redges[1:]):
        m = (lo < rmesh) & (rmesh <= hi)
        out[m] = out[m] * c
    return out


def log_density(cfg, latents, fixed, sample, count_obs, make_cosmo, aux=None):
    """log p(sample params, count_obs) of the field-level model for Normal or truncated-Normal latents (model.py:1105-1125 prior in
    sample space; bricks.py:255-287 affine reparametrisation; model.py:1127-1148 'fourier' / 'real' / 'kaiser'
    preconditioning; model.py:686-838 evolve; model.py:840-908 'quad_gauss' likelihood with no mask, unit selection, one radial
    bin and phi = 0).  `latents`: name -> dict(loc, scale, loc_fid, scale_fid); `fixed`: name -> value of the base
    parameters that are not sampled; `sample`: name_ -> value (scalars) and 'white_mesh_' (real, init_shape).  `aux`: an optional
    dict that receives the intermediates a test may want ('gxy': evolve's output, 'white': its input, 'count': the mean counts)."""
    lp = 0.0
    base = dict(fixed)
    scalar_items = []
    for name, conf in latents.items():      # a per-shell latent (ngbars, model.py:1099-1103): one scalar latent per element
        xs = np.atleast_1d(np.asarray(sample[name + "_"], dtype=float))
        if name == "ngbars":
            n = len(xs)
            el = lambda k, i: None if conf.get(k) is None else float(np.broadcast_to(np.asarray(conf[k], float), (n,))[i])
            for i in range(n):
                scalar_items.append((name, i, {k: el(k, i) for k in ("loc", "scale", "loc_fid", "scale_fid", "low", "high") if k in conf}, xs[i]))
            base[name] = np.zeros(n)
        else:
            scalar_items.append((name, None, conf, float(xs[0])))
    for name, idx, conf, x in scalar_items:
        low, high = conf.get("low", -np.inf), conf.get("high", np.inf)
        if conf.get("loc") is None:      # uniform prior (model.py:1122-1123; utils.py:314-353 DetruncUnif), central-difference Jacobian
            lf, sf, h = conf.get("loc_fid", (low + high) / 2), conf.get("scale_fid", (high - low) / 12 ** .5), 1e-6
            jac = (std2trunc(x + h, lf, sf, low, high) - std2trunc(x - h, lf, sf, low, high)) / (2 * h)
            lp += -np.log(high - low) + np.log(abs(jac))
            val = std2trunc(x, lf, sf, low, high)
        elif low == -np.inf and high == np.inf:
            mu, sd = (conf["loc"] - conf["loc_fid"]) / conf["scale_fid"], conf["scale"] / conf["scale_fid"]
            lp += -0.5 * np.log(2 * np.pi) - np.log(sd) - 0.5 * ((x - mu) / sd) ** 2
            val = x * conf["scale_fid"] + conf["loc_fid"]
        else:      # model.py:1120-1121, bricks.py:271-273
            lp += detrunc_truncnorm_log_prob(x, conf["loc"], conf["scale"], low, high, conf["loc_fid"], conf["scale_fid"])
            val = std2trunc(x, conf["loc_fid"], conf["scale_fid"], low, high)
        if idx is None:
            base[name] = val
        else:
            base[name][idx] = val
    w = np.asarray(sample["white_mesh_"], dtype=float)
    fiduc = cosmo_fid = None
    if cfg["precond"] == "kaiser":      # fiducial values: loc_fid of the latents, else the fixed value (model.py:1214-1223)
        fiduc = dict(fixed)
        fiduc.update({k: (np.asarray(v["loc_fid"], float) if k == "ngbars" else v["loc_fid"]) for k, v in latents.items()})
        cosmo_fid = make_cosmo(fiduc)
    scale, transfer = precond_scale_and_transfer(cfg, fiduc, cosmo_fid)
    lp += np.sum(-0.5 * np.log(2 * np.pi) - np.log(scale) - 0.5 * (w / scale) ** 2)     # model.py:666-672
    white = (o.rg2cgh(w) if cfg["precond"] != "real" else o._rfftn(w)) * transfer
    cosmo = make_cosmo(base)
    bias = {k: base[k] for k in BIAS_KEYS}
    gxy, _ = evolve(cfg, cosmo, bias, white)
    if aux is not None:
        aux.update(gxy=gxy, white=white, base=base)
    # likelihood (model.py:852-866, :893-908): selection mesh (paint_shape or scalar 1), mask over the final cells,
    # radial shells with their own mean densities; count_obs is the full final mesh (only its unmasked cells are used)
    final = tuple(cfg["final_shape"])
    down = lambda m: o._irfftn(o.chreshape(o._rfftn(m), o.r2chshape(final)), s=final, axes=(0, 1, 2))
    sel = cfg.get("selec_mesh")
    mask = np.ones(final, bool) if cfg.get("mask_mesh") is None else np.asarray(cfg["mask_mesh"], bool)
    rcounts = np.atleast_1d(np.asarray(base["ngbars"], float)) * cfg["cell_length"] ** 3
    redges = cfg.get("redges")
    if redges is None:
        redges = radial_edges(cfg, len(rcounts))
    rmesh = radius_mesh(cfg, final)
    cm = set_radial_count(down(gxy if sel is None else gxy * sel), rmesh, redges, rcounts)
    selec = np.mean(rcounts) if sel is None else np.abs(set_radial_count(down(sel), rmesh, redges, rcounts))
    if aux is not None:
        aux.update(count=cm)
    delta = cm / selec - 1
    scale1 = (np.abs(base["s_e"] + base["s_ed"] * delta) + 1e-9) * selec ** .5
    scale2 = base["s_e2"] * selec ** .5 * np.ones(final)
    lp += np.sum(quad_gaussian_log_prob(np.asarray(count_obs)[mask], cm[mask], (scale1 * np.ones(final))[mask], scale2[mask]))
    return float(lp)
