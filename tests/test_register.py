"""Register schema (reference: run/register.py:8-18, model.py:518-553): write / read round trip and the model arguments derived
from it.  CPU only; the HDF5 container is exercised where h5py is installed."""
import numpy as np
import pytest

from montecosmo_amd import register


def _reg(rng, cut_sky):
    shape = (8, 6, 10)
    count = rng.poisson(3.0, shape).astype(np.float64)
    reg = dict(cell_length=25., box_center=np.array([10., -20., 1500.]), box_rotvec=np.array([0.1, 0., -0.2]), init_oversamp=1.5,
               paint_oversamp=1.75, cosmo_fid=dict(Omega_m=0.3137721, sigma8=0.8076354), count_mesh=count, n_tracers=float(count.sum()),
               paint_order=2, interlace_order=2, paint_deconv=True, kernel_type="rectangular", cell_budget=480, padding=0.2,
               lin_kpow=np.stack([np.logspace(-4, 1, 32), np.linspace(1e4, 1., 32)]),
               white_fake=(rng.standard_normal((12, 10, 9)) + 1j * rng.standard_normal((12, 10, 9))).astype(np.complex64))
    if cut_sky:
        mask = rng.uniform(size=shape) < 0.7
        reg.update(mask_mesh=mask, selec_mesh=rng.uniform(0.2, 1., (14, 10, 18)), n_randoms=1e6, a_obs=None, curved_sky=True,
                   n_tracers=float(count[mask].sum()))
    else:
        reg.update(a_obs=0.6, curved_sky=False)
    return reg


@pytest.mark.parametrize("cut_sky", [False, True])
@pytest.mark.parametrize("suffix", [".npz", ".h5"])
def test_register_round_trip_and_model_arguments(tmp_path, cut_sky, suffix):
    if suffix == ".h5":
        pytest.importorskip("h5py")
    rng = np.random.default_rng(3)
    reg = _reg(rng, cut_sky)
    path = register.save_register(str(tmp_path / ("register_test" + suffix)), reg)
    back = register.load_register(path)
    assert set(back) == {k for k, v in reg.items() if v is not None}            # None is "absent", as in the reference's h5save
    for k, v in reg.items():
        if v is None:
            continue
        if isinstance(v, dict):
            assert back[k] == v
        elif isinstance(v, str):
            assert back[k] == v and isinstance(back[k], str)
        else:
            assert np.array_equal(np.asarray(back[k]), np.asarray(v))
    args = register.model_arguments(back, evolution="lpt")
    f = args["forward"]
    assert f["final_shape"] == (8, 6, 10) and f["cell_length"] == 25. and f["init_oversamp"] == 1.5 and f["evolution"] == "lpt"
    assert f["curved_sky"] is cut_sky and ("a_obs" not in f if cut_sky else f["a_obs"] == 0.6)
    assert np.array_equal(f["lin_kpow"][0], reg["lin_kpow"][0])
    n_cells = reg["mask_mesh"].sum() if cut_sky else reg["count_mesh"].size
    assert np.isclose(args["loc"]["ngbars"], reg["n_tracers"] / (n_cells * 25. ** 3))         # model.py:546-549
    assert args["loc"]["Omega_m"] == 0.3137721 and args["white_mesh"].shape == (12, 10, 9)
    assert (args["density"]["mask_mesh"] is None) != cut_sky


def test_register_schema_is_enforced(tmp_path):
    rng = np.random.default_rng(4)
    reg = _reg(rng, False)
    bad = dict(reg)
    del bad["cell_length"]
    with pytest.raises(KeyError, match="cell_length"):
        register.save_register(str(tmp_path / "a.npz"), bad)
    with pytest.raises(KeyError, match="outside the schema"):
        register.save_register(str(tmp_path / "b.npz"), dict(reg, surprise=1))
    with pytest.raises(KeyError, match="sigma8"):
        register.save_register(str(tmp_path / "c.npz"), dict(reg, cosmo_fid=dict(Omega_m=0.3)))
