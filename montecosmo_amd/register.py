"""The on-disk "register" of a mock (reference: run/register.py:8-18 writes it, montecosmo/model.py:518-553 loads it): ONE
self-describing file per mock from which `FieldLevelModel(register=path)` takes its geometry, painting parameters, meshes
and fiducial cosmology.  This module reads and writes that schema and turns it into the arguments of this package's
`FieldLevelForward` / `FieldLevelLogDensity`.

Schema (key: meaning; * = mandatory)
  * cell_length, box_center, box_rotvec      geometry (final_shape = count_mesh.shape)
  * init_oversamp, paint_oversamp            mesh oversampling
  * cosmo_fid/{Omega_m, sigma8}              fiducial cosmology of the mock (-> latents loc / loc_fid)
  * count_mesh                               painted tracer counts at final_shape (sum == n_tracers)
    selec_mesh, mask_mesh                    selection at paint_shape / footprint at final_shape (cut sky)
    n_tracers, n_randoms                     weighted catalog sizes
    a_obs, curved_sky                        full sky: 1 / (1 + z), False; cut sky: None (light cone), True
    paint_order, interlace_order, paint_deconv, kernel_type, cell_budget, padding
    lin_kpow                                 (2, N): k and P(k) / sigma8^2
    white_mesh | white_fake                  whitened initial conditions, half-spectrum at r2chshape(init_shape)

Container: HDF5 through h5py, as the reference's `h5save` / `h5load` (utils.py:76-160) lay it out: one dataset per key,
nested dicts as groups, None left out.  h5py is an optional dependency (it is absent from the build image): `.npz` files with
'/'-joined keys carry the same schema everywhere, and `save_register` / `load_register` choose by the file suffix.
"""
import os

import numpy as np

MANDATORY = ("cell_length", "box_center", "box_rotvec", "init_oversamp", "paint_oversamp", "cosmo_fid", "count_mesh")
OPTIONAL = ("selec_mesh", "mask_mesh", "n_tracers", "n_randoms", "a_obs", "curved_sky", "paint_order", "interlace_order",
            "paint_deconv", "kernel_type", "cell_budget", "padding", "lin_kpow", "white_mesh", "white_fake")


def _flatten(d, prefix=""):
    out = {}
    for k, v in d.items():
        if v is None:
            continue                                    # h5save skips None (utils.py:104-106): absent == None on load
        if isinstance(v, dict):
            out.update(_flatten(v, prefix + k + "/"))
        else:
            out[prefix + k] = np.asarray(v)
    return out


def _unflatten(flat):
    out = {}
    for k, v in flat.items():
        node = out
        *groups, leaf = k.split("/")
        for g in groups:
            node = node.setdefault(g, {})
        v = np.asarray(v)
        if v.dtype.kind in ("S", "U") and v.ndim == 0:
            v = str(v.item().decode() if isinstance(v.item(), bytes) else v.item())
        elif v.ndim == 0:
            v = v.item()                                # 0-d scalar -> native python (utils.py:150-153)
        node[leaf] = v
    return out


def validate(reg):
    missing = [k for k in MANDATORY if k not in reg]
    if missing:
        raise KeyError(f"register is missing mandatory keys {missing}")
    unknown = [k for k in reg if k not in MANDATORY + OPTIONAL]
    if unknown:
        raise KeyError(f"register has keys outside the schema: {unknown}")
    for k in ("Omega_m", "sigma8"):
        if k not in reg["cosmo_fid"]:
            raise KeyError(f"cosmo_fid/{k} missing")
    if np.ndim(reg["count_mesh"]) != 3:
        raise ValueError("count_mesh must be a 3-D mesh (its shape is the model's final_shape)")
    return reg


def save_register(path, reg):
    """Writes `reg` (a dict following the schema above) to `path` (.h5 / .hdf5 through h5py, .npz otherwise)."""
    validate(reg)
    flat = _flatten(reg)
    if str(path).endswith((".h5", ".hdf5")):
        try:
            import h5py
        except ImportError as e:
            raise ImportError("writing an HDF5 register needs h5py; use a .npz path (same schema) where it is absent") from e
        with h5py.File(str(path), "w") as f:
            for k, v in flat.items():
                f.create_dataset(k, data=v.astype("S") if v.dtype.kind == "U" else v)
    else:
        np.savez(str(path), **{k.replace("/", "__"): v for k, v in flat.items()})
    return path


def load_register(path):
    """The dict `save_register` (or the reference's run/register.py) wrote."""
    if str(path).endswith((".h5", ".hdf5")):
        try:
            import h5py
        except ImportError as e:
            raise ImportError("reading an HDF5 register needs h5py") from e
        flat = {}
        with h5py.File(str(path), "r") as f:
            f.visititems(lambda name, obj: flat.__setitem__(name, obj[()]) if isinstance(obj, h5py.Dataset) else None)
    else:
        with np.load(str(path) if os.path.exists(str(path)) else str(path) + ".npz", allow_pickle=False) as z:
            flat = {k.replace("__", "/"): z[k] for k in z.files}
    return validate(_unflatten(flat))


def model_arguments(reg, **overrides):
    """What montecosmo/model.py:518-553 takes from a register, as keyword arguments for this package:
       forward   -> FieldLevelForward(**forward)  (geometry, oversampling, sky, painting, tabulated linear power)
       density   -> FieldLevelLogDensity(fwd, count_obs=density['count_mesh'], ..., selec_mesh=, mask_mesh=)
       loc       -> fiducial values of the latents: Omega_m, sigma8 from cosmo_fid, and ngbars = n_tracers / (observed cells
                    x cell_length^3) (model.py:546-549)
       white_mesh-> the registered initial conditions (or None).
    `overrides` replace forward arguments (evolution, nbody_n_steps, ...)."""
    validate(reg)
    count = np.asarray(reg["count_mesh"], dtype=np.float64)
    mask = None if reg.get("mask_mesh") is None else np.asarray(reg["mask_mesh"], dtype=bool)
    fwd = dict(final_shape=tuple(int(s) for s in count.shape), cell_length=float(reg["cell_length"]),
               box_center=tuple(float(v) for v in np.ravel(reg["box_center"])), box_rotvec=tuple(float(v) for v in np.ravel(reg["box_rotvec"])),
               init_oversamp=float(reg["init_oversamp"]), paint_oversamp=float(reg["paint_oversamp"]))
    for k in ("a_obs", "curved_sky", "paint_order", "interlace_order", "paint_deconv"):
        if k in reg:
            fwd[k] = reg[k]
    if reg.get("kernel_type", "rectangular") != "rectangular":
        raise NotImplementedError("FieldLevelForward paints with kernel_type='rectangular' (nbody.paint itself takes 'kaiser_bessel')")
    if reg.get("lin_kpow") is not None:
        lk = np.asarray(reg["lin_kpow"], dtype=np.float64)
        fwd["lin_kpow"] = (lk[0], lk[1])
    fwd.update(overrides)
    n_cells = int(mask.sum()) if mask is not None else count.size
    n_tracers = float(reg.get("n_tracers", count[mask].sum() if mask is not None else count.sum()))
    ngbar = n_tracers / (n_cells * float(reg["cell_length"]) ** 3)
    sel = reg.get("selec_mesh")
    sel = None if (sel is None or np.ndim(sel) == 0) else np.asarray(sel, dtype=np.float64)
    white = reg.get("white_mesh", reg.get("white_fake"))
    return dict(forward=fwd, density=dict(count_mesh=count, selec_mesh=sel, mask_mesh=mask),
                loc=dict(Omega_m=float(reg["cosmo_fid"]["Omega_m"]), sigma8=float(reg["cosmo_fid"]["sigma8"]), ngbars=ngbar),
                white_mesh=None if white is None else np.asarray(white))
